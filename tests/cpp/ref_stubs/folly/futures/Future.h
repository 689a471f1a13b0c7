#pragma once
#include <stdexcept>
namespace folly {
struct FutureCancellation : std::runtime_error {
    FutureCancellation() : std::runtime_error("cancelled") {}
};
}  // namespace folly
