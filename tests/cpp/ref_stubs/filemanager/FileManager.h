#pragma once
#include <memory>
#include <string>
namespace milvus { class FileManager { public: virtual ~FileManager() = default; }; using FileManagerPtr = std::shared_ptr<FileManager>; }
