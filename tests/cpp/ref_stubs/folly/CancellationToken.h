#pragma once
// compile-only stand-in (folly is absent here)
namespace folly {
class CancellationToken {
 public:
    bool isCancellationRequested() const { return false; }
};
}  // namespace folly
