#pragma once
// compile-only: an old single-header nlohmann::json that happens to be on this image (the reference pins 3.11.3)
#include "/opt/conda/include/json.hpp"
namespace nlohmann {
inline std::string to_string(const json& j) { return j.dump(); }  // present in 3.11, absent in 3.1
}
