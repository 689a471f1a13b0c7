/* include/knhip_shards.h -- C ABI of the list-sharded multi-GPU search.
 *
 * Replaces what faiss does host-side for sharded indexes -- IndexShards::search + merge_knn_results
 * (/root/reference/thirdparty/faiss/faiss/IndexShards.cpp:247-256, utils/Heap.h:636) -- with one step per query batch:
 *   the coarse quantizer on a slice of the queries per GPU + an all-gather of the assignment; every GPU scans the inverted
 *   lists it owns for a CANONICAL top-(k + 1) (knhip_search_canonical_device on its own index: centroids, codebooks / SQ
 *   ranges replicated, the lists of the other GPUs empty), ONE all-gather of the per-GPU (nq, k + 1) partials over RCCL /
 *   xGMI (packed 12 bytes per entry: distance + id), one merge kernel per GPU (knhip_merge_topk_device); the queries whose
 *   k-th and (k + 1)-th entries tie are resolved over ALL shards' candidates (knhip_tie_flag_device, per-shard
 *   knhip_tie_arrivals_device, one more small all-gather, knhip_tie_resolve_device: the steps of include/knhip.h); rank
 *   0's copy is returned.  Candidates of different lists are disjoint and the boundary rule is applied once, after the
 *   merge, so the result is bit-identical to the single-GPU search of the whole index, including WHICH of several rows
 *   tied at the k-th distance is returned (tests/test_gpu_shards.py: world 2 / 3 / 4, no licence).  With a refine store
 *   (knhip_shard_group_set_raw / _set_raw_rows) the shards exchange per-candidate distances and run one selection.
 * The host side is C++ (knowhere_amd/host/shard_group.cc): one worker thread per GPU inside one process,
 * ncclCommInitAll over device_ids[] -- the shape a Knowhere node owning several devices would use.  transport:
 *   KNHIP_SHARDS_RCCL    ncclAllGather on the workers' streams (needs distinct devices)
 *   KNHIP_SHARDS_STAGED  the same protocol with the all-gather done by device-to-device copies through rank 0's
 *                        buffer: any device list, also several ranks on ONE device (how the protocol is tested on a
 *                        single-GPU box). */
#ifndef KNHIP_SHARDS_H
#define KNHIP_SHARDS_H
#include "knhip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct knhip_shard_group knhip_shard_group;
enum { KNHIP_SHARDS_RCCL = 0, KNHIP_SHARDS_STAGED = 1 };

/* n_devices ranks; rank r works on device_ids[r].  RCCL transport: the communicators are created here. */
int knhip_shard_group_create(int32_t n_devices, const int32_t* device_ids, int32_t transport, knhip_shard_group** out);
void knhip_shard_group_destroy(knhip_shard_group* g);
/* rank r's index: created on device_ids[r] by the caller, holding the lists rank r owns (not owned by the group) */
int knhip_shard_group_set_index(knhip_shard_group* g, int32_t rank, const knhip_index* idx);
/* one Search() of the whole (sharded) index: host queries [nq][dim] in, host ids / distances [nq][k] out.
 * stage_ms (may be NULL): [n_devices][4] = per rank {search, all-gather + merge, -, total} milliseconds of this call. */
int knhip_shard_group_search(knhip_shard_group* g, const float* queries, int64_t nq, int32_t k, int32_t nprobe,
                             const uint8_t* bitset, int64_t bitset_nbits, int64_t* out_ids, float* out_dist,
                             float* stage_ms);
/* Refine stage (Knowhere's `refine` / `refine_k`, IndexRefine::search over IndexShards): rank r holds the raw fp32 rows
 * of the vector ids [id_base, id_base + nrows) in its HBM (d_rows: device pointer on device_ids[r]; the split of the raw
 * rows need not follow the split of the lists).  search_refine: k_base candidates per rank -> all-gather + merge (the same
 * merged candidates on every rank) -> every rank re-ranks exactly the candidates whose rows it holds -> all-gather + merge
 * of the (nq, k) partials.  A rank with nrows = 0 holds no raw rows and contributes an empty partial.  A failure on any
 * rank before a collective makes EVERY rank skip it (agreement under a host barrier); a failure of the collective's own
 * enqueue marks the group unusable (later calls fail at once; destroy aborts the communicators).  stage_ms: [n_devices][7] = {search, gather + merge, -, refine, gather + merge, -, total}. */
int knhip_shard_group_set_raw(knhip_shard_group* g, int32_t rank, const float* d_rows, int64_t nrows, int64_t id_base);
/* the same with a quantised refine store (knhip_rows: refine_type fp16 / bf16 / sq8): row r of `rows` = vector id
 * id_base + r; every rank's store must carry the same sq8 ranges.  NULL clears it.  A rank uses this store if set, else
 * its fp32 rows. */
int knhip_shard_group_set_raw_rows(knhip_shard_group* g, int32_t rank, const knhip_rows* rows, int64_t id_base);
int knhip_shard_group_search_refine(knhip_shard_group* g, const float* queries, int64_t nq, int32_t k, int32_t k_base,
                                    int32_t nprobe, const uint8_t* bitset, int64_t bitset_nbits, int64_t* out_ids,
                                    float* out_dist, float* stage_ms);
int32_t knhip_shard_group_size(const knhip_shard_group* g);
/* message of the last failed knhip_shard_group_* call on this thread (the per-rank text: "rank r: ...") */
const char* knhip_shard_group_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
