#pragma once
#include <iostream>
#include <sstream>
namespace google { struct NullStream { template <class T> NullStream& operator<<(const T&) { return *this; } }; }
#define LOG(sev) ::google::NullStream()
#define VLOG(n) ::google::NullStream()
#define DLOG(sev) ::google::NullStream()
#define CHECK(x) ::google::NullStream()
