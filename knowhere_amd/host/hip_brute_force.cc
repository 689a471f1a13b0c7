// knowhere_amd/host/hip_brute_force.cc -- knowhere::BruteForce::Search / SearchWithBuf / RangeSearch <fp32> routed to the
// GPU_HIP_BRUTE_FORCE node.
//
// The reference's BruteForce::Search (src/common/comp/brute_force.cc:70-230) scans the base dataset on the CPU thread
// pool; tests/ut/test_gpu_search.cc:84 and :233 use it as the ground truth of the GPU indexes.  Inside a Knowhere tree
// that function stays what it is (this file is not part of the plugin there); in the standalone build the same
// signature is served by the HIP flat scan, so the re-run of the reference's test flow exercises the device path for
// both sides of the comparison and the C++ tests need no CPU scan of their own.
#include "hip_index_node.h"

#if !defined(KNHIP_WITH_KNOWHERE_HEADERS)
namespace knowhere {

template <>
expected<DataSetPtr>
BruteForce::Search<fp32>(const DataSetPtr base_dataset, const DataSetPtr query_dataset, const Json& config,
                         const BitsetView& bitset, milvus::OpContext* op_context) {
    auto r = IndexFactory::Instance().Create<fp32>(IndexEnum::INDEX_HIP_BRUTEFORCE,
                                                   Version::GetCurrentVersion().VersionNumber());
    if (!r.has_value()) return expected<DataSetPtr>::Err(r.error(), r.what());
    auto idx = r.value();
    Status s = idx.Build(base_dataset, config);
    if (s != Status::success) return expected<DataSetPtr>::Err(s, "brute force: base dataset rejected");
    return idx.Search(query_dataset, config, bitset, op_context);
}

// BruteForce::SearchWithBuf (src/common/comp/brute_force.cc:395-560): the same search into the caller's buffers
// [nq][k] -- what Milvus calls on growing segments
template <>
Status
BruteForce::SearchWithBuf<fp32>(const DataSetPtr base_dataset, const DataSetPtr query_dataset, int64_t* ids, float* dis,
                                const Json& config, const BitsetView& bitset, milvus::OpContext* op_context) {
    if (ids == nullptr || dis == nullptr) return Status::invalid_args;
    auto r = Search<fp32>(base_dataset, query_dataset, config, bitset, op_context);
    if (!r.has_value()) return r.error();
    const int64_t nq = query_dataset->GetRows(), k = r.value()->GetDim();
    std::copy(r.value()->GetIds(), r.value()->GetIds() + nq * k, ids);
    std::copy(r.value()->GetDistance(), r.value()->GetDistance() + nq * k, dis);
    return Status::success;
}

// BruteForce::RangeSearch (src/common/comp/brute_force.cc:562-770): radius / range_filter over the base rows, lims + ids +
// distances per query -- the node's RangeSearch on a GPU_HIP_BRUTE_FORCE index (range.hip)
template <>
expected<DataSetPtr>
BruteForce::RangeSearch<fp32>(const DataSetPtr base_dataset, const DataSetPtr query_dataset, const Json& config,
                              const BitsetView& bitset, milvus::OpContext* op_context) {
    auto r = IndexFactory::Instance().Create<fp32>(IndexEnum::INDEX_HIP_BRUTEFORCE,
                                                   Version::GetCurrentVersion().VersionNumber());
    if (!r.has_value()) return expected<DataSetPtr>::Err(r.error(), r.what());
    auto idx = r.value();
    Status s = idx.Build(base_dataset, config);
    if (s != Status::success) return expected<DataSetPtr>::Err(s, "brute force: base dataset rejected");
    return idx.RangeSearch(query_dataset, config, bitset, op_context);
}

}  // namespace knowhere
#endif
