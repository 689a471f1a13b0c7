#pragma once
// compile-only stand-in for milvus-common's OpContext
#include <folly/CancellationToken.h>
namespace milvus {
struct OpContext {
    folly::CancellationToken cancellation_token;
};
}  // namespace milvus
