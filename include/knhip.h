/* include/knhip.h -- the drop-in boundary: C ABI between the C++ host IndexNode and the
 * hand-written HIP (gfx950) Search() kernels.
 *
 * Plain pointers and sizes only; no C++/torch types; never throws; every entry point
 * returns 0 on success or a negative knhip_status, with the message available from
 * knhip_last_error() (thread-local).  One knhip_index lives on ONE device (one process per
 * GPU); multi-GPU list sharding is done by giving each rank's index only the inverted lists
 * it owns (all other lists empty) and merging the per-rank partial top-k with
 * knhip_merge_topk_device after an RCCL all-gather.
 *
 * What each entry point replaces in the reference (/root/reference):
 *   knhip_index_create / set_* / add_*   the state GpuCuvsIndexNode::Train builds on device
 *                                        (src/index/gpu_cuvs/gpu_cuvs.h:88-119) and the faiss
 *                                        objects IvfIndexNode::Train/Add fill
 *                                        (src/index/ivf/ivf.cc:547-844): IndexFlat centroids,
 *                                        ProductQuantizer::centroids, ScalarQuantizer::trained,
 *                                        ArrayInvertedLists codes/ids
 *                                        (thirdparty/faiss/faiss/invlists/InvertedLists.h:264-266)
 *   knhip_search                         IvfIndexNode::Search (src/index/ivf/ivf.cc:889-1168),
 *                                        FlatIndexNode::Search (src/index/flat/flat.cc:76-148),
 *                                        BruteForce::Search (src/common/comp/brute_force.cc:258-392),
 *                                        cuvs_knowhere_index::search
 *                                        (src/common/cuvs/integration/cuvs_knowhere_index.cuh:507-633)
 *   knhip_search_device                  same, inputs/outputs already resident in HBM
 *   knhip_coarse_search_device           quantizer->search (thirdparty/faiss/faiss/IndexIVF.cpp:336-342)
 *   knhip_merge_topk_*                   merge_knn_results (thirdparty/faiss/faiss/utils/Heap.h:636),
 *                                        IndexShards (thirdparty/faiss/faiss/IndexShards.cpp:247-256)
 *   knhip_fvec_* / knhip_int8_*          the src/simd hook table (src/simd/hook.h:33-139),
 *                                        scalar semantics of src/simd/distances_ref.cc
 *
 * Numeric contract ("exact" mode, the default): every distance is computed with the
 * reference's scalar operation order, one IEEE rounding per operation (no FMA contraction),
 * so returned distances are bit-equal to the scalar reference; returned ids are the k best
 * in canonical order (L2: distance asc, id asc; IP: distance desc, id desc -- the order
 * heap_reorder produces, thirdparty/faiss/faiss/utils/Heap.h:427-457).
 * Candidates TIED at the k-th distance: the reference's heap admits first-come (strict improve,
 * impl/ResultHandler.h:258-279) and evicts by id (heap_replace_top / cmp2, utils/Heap.h:113-151), so which of them
 * is returned depends on the scan order (probe rank, then storage position).  The library returns the same ones
 * (knhip_search*, knhip_search_preassigned_device, knhip_search_refine, knhip_refine_device): the search runs for
 * k + 1 canonical results, a query whose (k + 1)-th ties with its k-th has its probed lists scanned once more in
 * scan order and the rule applied (one 4-byte read-back per batch tells whether there is such a query;
 * KNHIP_TIES=canonical skips it and returns the canonical k: graph capture, or callers that do not care).
 * Conditions: the ids of every list ascend in storage order -- what IvfIndexNode::Add produces (ids are the running
 * row numbers); knhip_index_add_lists re-sorts a list that does not, and ties inside it are then taken in id order.
 * Not covered (canonical answer): k = 1024; BRUTE_FORCE with k >= 100 (the reference switches to a reservoir,
 * impl/ResultHandler.h:719-728).  A list-sharded index returns the same answer as one index: the rule is applied once, after
 * the merge, over all shards' candidates (knhip_search_canonical_device .. knhip_tie_resolve_device below,
 * knhip_shard_group_*); only a caller that merges tie-resolved per-shard results itself (knhip_merge_topk_*) gets the
 * canonical choice among the shards' survivors.
 */
#ifndef KNHIP_H
#define KNHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4: knhip_train_params gained spherical/reserved, knhip_stage_times its prefilter counters.  5: tie_queries.  6: the quantised refine store (knhip_rows_*, knhip_search_refine_rows), knhip_range_search_ranked.
 * 7: sixteen profiling stages (sample / tables / refine / ties itemised), tie_anomalies; the tie rule for list-sharded
 * indexes (knhip_search_canonical_device, knhip_tie_*, knhip_refine_distances / _combine / _select).
 * 8: the refine stores sq6 / int8 / sq4u (KNHIP_ROWS_SQ6 / _INT8 / _SQ4U) and knhip_rows_train_uniform.
 * 9: knhip_ties_rule_applies (one place decides whether a search follows the reference's boundary rule: single index, shard
 * group and the torch.distributed host agree); value 3 of the profile field pq_filter_form: the decode form of the IVF-PQ prefilter;
 * pq_nbits 1 .. 8 (host-side list codes in the reference's bit-string form).
 * Callers compare knhip_abi_version() with the header they were built against. */
#define KNHIP_ABI_VERSION 9

typedef struct knhip_index knhip_index;

typedef enum knhip_kind {
    KNHIP_BRUTE_FORCE = 0, /* FLAT / BruteForce */
    KNHIP_IVF_FLAT = 1,
    KNHIP_IVF_PQ = 2,
    KNHIP_IVF_SQ8 = 3
} knhip_kind;

typedef enum knhip_metric { KNHIP_L2 = 0, KNHIP_IP = 1 } knhip_metric;
/* COSINE is handled above the ABI exactly as the reference does: base normalised at
 * train/add, query copied+normalised per search, metric mapped to IP
 * (src/index/ivf/ivf.cc:559-565, 1068-1071; src/common/metric.h:30). */

typedef enum knhip_status {
    KNHIP_OK = 0,
    KNHIP_ERR_INVALID_ARGS = -1,    /* Status::invalid_args */
    KNHIP_ERR_NOT_TRAINED = -2,     /* Status::index_not_trained */
    KNHIP_ERR_EMPTY_INDEX = -3,     /* Status::empty_index */
    KNHIP_ERR_NOT_IMPLEMENTED = -4, /* Status::not_implemented */
    KNHIP_ERR_HIP_RUNTIME = -5,     /* Status::cuda_runtime_error (reused for HIP) */
    KNHIP_ERR_OUT_OF_MEMORY = -6    /* Status::malloc_error */
} knhip_status;

typedef struct knhip_desc {
    int32_t kind;   /* knhip_kind */
    int32_t metric; /* knhip_metric */
    int32_t dim;
    int32_t device;   /* HIP device ordinal */
    int64_t nlist;    /* IVF kinds */
    int32_t pq_m;     /* IVF_PQ: sub-quantizers (must divide dim) */
    int32_t pq_nbits; /* IVF_PQ: 1 .. 8 (ABI 9; 8 before).  Codebooks are [pq_m][2^nbits][dim / pq_m].  HOST-side list codes
                       * (knhip_index_add_lists / knhip_index_get_lists) are the reference's code bytes: pq_m indices of nbits bits
                       * as a little-endian bit string of (pq_m nbits + 7) / 8 bytes (faiss ProductQuantizer.cpp:69,
                       * PQEncoderGeneric); DEVICE-side codes (knhip_index_encode_device, knhip_index_set_lists_device) are one
                       * byte per sub-quantizer for every width */
    /* IVF_PQ + L2: precomputed term-2 table limit in bytes; 0 = the reference default 2 GiB
     * (thirdparty/faiss/faiss/IndexIVFPQ.cpp:375).  Above it the residual-table form is used,
     * as the reference does (:441-456). */
    int64_t precomputed_table_max_bytes;
} knhip_desc;

/* ---- library ---- */
int knhip_abi_version(void);
int knhip_device_count(void);
const char* knhip_last_error(void);
/* free / total HBM of a device: what cuvs_knowhere_index::deserialize looks at to place a loaded index on the device
 * with the most free memory (src/common/cuvs/integration/cuvs_knowhere_index.cuh:678-690) */
int knhip_device_memory(int32_t device, int64_t* free_bytes, int64_t* total_bytes);

/* ---- index lifetime and contents (all pointers below are HOST pointers) ---- */
int knhip_index_create(const knhip_desc* desc, knhip_index** out);
void knhip_index_destroy(knhip_index* idx);

/* coarse centroids [nlist][dim] fp32 (IndexFlat xb of the quantizer) */
int knhip_index_set_coarse(knhip_index* idx, const float* centroids);
/* PQ codebooks [M][ksub][dsub] fp32 (ProductQuantizer::centroids) */
int knhip_index_set_pq(knhip_index* idx, const float* codebooks);
/* SQ8 trained params: vmin[dim], vdiff[dim] (ScalarQuantizer::trained) */
int knhip_index_set_sq(knhip_index* idx, const float* vmin, const float* vdiff);
/* COSINE with stored norms, as the CPU nodes keep it for FLAT and IVF_FLAT: raw rows + one float per row, applied to the
 * finished inner product of every scanned row (Search and RangeSearch):
 *   mode 1: dis = <q, y> / scale      IVFFlatScanner with code norms (cppcontrib/knowhere/IndexIVFFlat.cpp:199-210),
 *                                      scale = the row's L2 norm as knowhere::NormalizeVecs returns it
 *   mode 2: dis = clamp(<q, y> * scale, -1, 1)   IndexFlatCosine -> exhaustive_cosine_seq_impl
 *                                      (cppcontrib/knowhere/utils/distances.cpp:367-409), scale = inverse L2 norm
 * scale: one float per stored entry in the canonical order of knhip_index_get_lists (BRUTE_FORCE: row order); NULL or
 * mode 0 switches it off.  Inner-product BRUTE_FORCE / IVF_FLAT indexes only; the values belong to the current
 * contents: set them again after an Add.  Queries are normalised by the caller (the node), as in the reference. */
int knhip_index_set_row_scale(knhip_index* idx, const float* scale, int32_t mode);
/* inverted lists in faiss ArrayInvertedLists layout: list_sizes[nlist],
 * codes[l] -> uint8[len][code_size] (IVF_FLAT: the fp32 rows), ids[l] -> int64[len].
 * Replaces any previous content.  codes[l]/ids[l] may be NULL when list_sizes[l] == 0. */
int knhip_index_add_lists(knhip_index* idx, const int64_t* list_sizes, const uint8_t* const* codes,
                          const int64_t* const* ids);
/* BRUTE_FORCE base vectors [n][dim] fp32; ids NULL => 0..n-1 (+ id_offset, cf. tensor begin id
 * include/knowhere/dataset.h:412) */
int knhip_index_add_vectors(knhip_index* idx, int64_t n, const float* x, const int64_t* ids,
                            int64_t id_offset);

/* Device-resident bulk variants (GPU build path; pointers are DEVICE pointers on idx's
 * device, consumed before the call returns): entries sorted by list, ids ascending inside
 * each list; list_offsets is a HOST array [nlist+1]. */
int knhip_index_set_coarse_device(knhip_index* idx, const float* d_centroids);
int knhip_index_set_lists_device(knhip_index* idx, const int64_t* list_offsets,
                                 const uint8_t* d_codes, const int64_t* d_ids);
int knhip_index_add_vectors_device(knhip_index* idx, int64_t n, const float* d_x,
                                   const int64_t* d_ids, int64_t id_offset);

/* ---- GPU build: Train / Add on the device ------------------------------------------------------------
 * Replaces the arithmetic of IvfIndexNode::Train / Add (reference src/index/ivf/ivf.cc:547-844):
 *   knhip_index_train*   IndexIVF::train (thirdparty/faiss/faiss/IndexIVF.cpp:1175-1270) = Level1Quantizer::train_q1
 *                        (:55-121, Clustering::train with the index's own exact search as the assigner,
 *                        Clustering.cpp:95-380, impl/ClusteringHelpers.cpp:36-240) + train_encoder:
 *                        ProductQuantizer::train (impl/ProductQuantizer.cpp:130-215, one 256-centroid k-means per
 *                        sub-space on residuals) / ScalarQuantizer::train (QT_8bit, RS_minmax).  Same sub-sampling
 *                        draws (rand_perm with the reference's seeds), same initial centroids, same update and
 *                        empty-cluster split arithmetic, spherical k-means (Clustering::post_process_centroids ->
 *                        fvec_renorm_L2, Clustering.cpp:35-38) for the inner product as IndexIVF's constructor sets
 *                        it (IndexIVF.cpp:178-181), 10 iterations for the level-1 quantizer (IndexIVF.cpp:44) and 25
 *                        for the PQ codebooks; the assignment is the exact sequential search (the reference switches
 *                        to a BLAS expansion above its batch threshold: near-ties may differ).  Pinned:
 *                        tests/test_oracle.py::test_train_add_restatement_equals_reference (oracle.c == the
 *                        reference's own IndexIVF::train + add, bit for bit, L2 and IP, default parameters) and
 *                        tests/test_gpu_build.py (this library == oracle.c and == reference-generated goldens).
 *                        Coarse centroids already set (knhip_index_set_coarse*) are kept.
 *   knhip_index_add*     IndexIVF::add_core (IndexIVF.cpp:212-287): quantizer->assign, encode_vectors
 *                        (compute_residual + ProductQuantizer::compute_code / SQ8 encode_vector), append to the
 *                        inverted lists.  May be called repeatedly; ids NULL => running numbers continuing the
 *                        current count (what Knowhere passes), otherwise ascending ids larger than the stored ones.
 *                        Codes and assignments are bit-equal to the scalar reference (oracle.c orc_pq_encode ...).
 *   knhip_kmeans_device  the Clustering restatement alone (k-means of device rows).
 * params NULL or zero fields => the reference defaults: 256 points per centroid, seed 1234; iterations 10 for the
 * level-1 quantizer of knhip_index_train* (Level1Quantizer, IndexIVF.cpp:44) and 25 for knhip_kmeans_device
 * (ClusteringParameters); spherical 0 = what the reference does (on for an inner-product index's level-1 quantizer,
 * off for knhip_kmeans_device), 1 = on, 2 = off.  niter applies to the level-1 quantizer only: the PQ codebooks always
 * train with the ClusteringParameters defaults, as ProductQuantizer::train does. */
typedef struct knhip_train_params {
    int32_t niter;
    int32_t max_points_per_centroid;
    int64_t seed;
    int32_t spherical;
    int32_t reserved;
} knhip_train_params;
int knhip_kmeans_device(int32_t metric, int32_t dim, int64_t n, const float* d_x, int64_t k,
                        const knhip_train_params* params, float* d_centroids, int32_t device);
int knhip_index_train(knhip_index* idx, int64_t n, const float* x, const knhip_train_params* params);
int knhip_index_train_device(knhip_index* idx, int64_t n, const float* d_x, const knhip_train_params* params);
int knhip_index_add(knhip_index* idx, int64_t n, const float* x, const int64_t* ids);
int knhip_index_add_device(knhip_index* idx, int64_t n, const float* d_x, const int64_t* d_ids);
/* IVF_FLAT: rows x_store appended to the lists their companions x_assign are assigned to -- IndexIVFFlatCosine::
 * add_with_ids (cppcontrib/knowhere/IndexIVFFlat.cpp:516-524) assigns by the normalised row and stores the raw one */
int knhip_index_add_assigned_by(knhip_index* idx, int64_t n, const float* x_store, const float* x_assign,
                                const int64_t* ids);
/* quantizer->assign of n HOST rows (IndexIVF::add_core's first step, IndexIVF.cpp:236): assign [n] int64, the list every
 * row would be appended to (first maximum for the inner product, as IndexFlat::assign).  What a node that deals its
 * inverted lists over several devices needs to route a row to the device owning its list. */
int knhip_index_assign(const knhip_index* idx, int64_t n, const float* x, int64_t* assign);
/* assignment + codes of n device rows without adding them: d_assign [n] int64, d_codes [n][code_size] */
int knhip_index_encode_device(const knhip_index* idx, int64_t n, const float* d_x, int64_t* d_assign, uint8_t* d_codes,
                              void* stream);
/* read the trained state / the inverted lists back (host buffers): Serialize needs the faiss objects
 * (src/index/ivf/ivf.cc:1717-1744).  get_lists: codes [count][code_size] and ids [count], list after list
 * (get_list_sizes gives the split); BRUTE_FORCE: codes = the raw fp32 rows, ids unused */
int knhip_index_get_coarse(const knhip_index* idx, float* centroids);
int knhip_index_get_pq(const knhip_index* idx, float* codebooks);
int knhip_index_get_sq(const knhip_index* idx, float* vmin, float* vdiff);
int knhip_index_get_list_sizes(const knhip_index* idx, int64_t* sizes);
int knhip_index_get_lists(const knhip_index* idx, uint8_t* codes, int64_t* ids);
/* BRUTE_FORCE: device pointer to the resident raw rows [count][dim] (valid until the next Add / destroy) */
int knhip_index_get_vectors_device(const knhip_index* idx, const float** d_rows);

int knhip_index_get_desc(const knhip_index* idx, knhip_desc* out);  /* the descriptor the index was created with */
int64_t knhip_index_count(const knhip_index* idx);         /* stored vectors */
int64_t knhip_index_device_bytes(const knhip_index* idx);  /* HBM held by the index */
int knhip_index_uses_precomputed_table(const knhip_index* idx);

/* Range search: every vector strictly inside the radius (L2: dist < radius, IP: dist > radius) among the
 * lists visited in coarse order, with the reference's early stop after `max_empty_result_buckets`
 * consecutive lists without a hit (0 = visit every list).  Replaces IvfIndexNode::RangeSearch (reference
 * src/index/ivf/ivf.cc:1231-1420: nprobe = nlist, IVFSearchParameters::max_empty_result_buckets, default 2)
 * -> IndexIVF::range_search_preassigned (thirdparty/faiss/faiss/IndexIVF.cpp:812-990, parallel_mode 0) and,
 * for KNHIP_BRUTE_FORCE, IndexFlat::range_search (thirdparty/faiss/faiss/utils/distances.cpp
 * range_search_L2sqr / range_search_inner_product; no early stop).  The caller applies `range_filter`
 * (reference src/common/range_util.cc:27-48).
 * Host pointers.  lims[nq + 1] receives the per-query offsets; *out_ids / *out_dist receive malloc'ed arrays
 * of lims[nq] entries (release with knhip_free) in the reference's emission order: list by list in coarse
 * order, storage order inside a list.  Distances are bit-equal to the scalar reference.
 * Supported: KNHIP_BRUTE_FORCE, KNHIP_IVF_FLAT, KNHIP_IVF_SQ8, KNHIP_IVF_PQ (any m x 8 bit); nlist <= 65536. */
int knhip_range_search(const knhip_index* idx, const float* queries, int64_t nq, float radius,
                       int32_t max_empty_result_buckets, const uint8_t* bitset, int64_t bitset_nbits, int64_t* lims,
                       int64_t** out_ids, float** out_dist);
/* The same search over EVERY list (no early stop) that also returns how many hits each coarse rank contributed:
 * (*out_rank_counts)[q * nlist + r] = hits of query q in its r-th nearest list (malloc'ed, nq * nlist entries, release with
 * knhip_free).  What a list-sharded deployment needs to apply the reference's early stop ACROSS shards: every shard holds
 * all centroids (so it ranks the lists like every other shard) but only the entries of the lists it owns; the host sums
 * the counts over the shards per (query, rank), walks the ranks with the rule of IndexIVF::range_search_preassigned
 * (IndexIVF.cpp:917-933: stop after max_empty_result_buckets consecutive lists without a hit) and takes each rank's hits
 * from the shard that owns the list (knowhere_amd/host/hip_index_node.cc, RangeSearch).  IVF kinds only. */
int knhip_range_search_ranked(const knhip_index* idx, const float* queries, int64_t nq, float radius, const uint8_t* bitset,
                              int64_t bitset_nbits, int64_t* lims, int64_t** out_ids, float** out_dist,
                              int32_t** out_rank_counts);
/* Coarse ranks the last knhip_range_search on this index scanned per query (its last batch): the IVF kinds probe in
 * waves of ranks (64, 128, 256, ...) and stop once every query has met the reference's early stop, so the cost follows
 * max_empty_result_buckets instead of nlist.  nlist when every list was scanned (max_empty = 0, or nlist <= 128). */
int64_t knhip_index_last_range_ranks(const knhip_index* idx);
void knhip_free(void* p);

/* ---- search ---- */
/* Host boundary (what the IndexNode calls): queries [nq][dim] host fp32; bitset host bytes
 * LSB-first, bit set => id filtered OUT (include/knowhere/bitsetview_idselector.h:20-31),
 * NULL => none; out_ids[nq*k] int64 / out_dist[nq*k] fp32 host, best-first; missing results
 * id = -1, dist = +FLT_MAX (L2) / -FLT_MAX (IP) (thirdparty/faiss/faiss/utils/Heap.h:338-341).
 * nprobe is clamped to nlist (thirdparty/faiss/faiss/IndexIVF.cpp:321-322); ignored for
 * BRUTE_FORCE.  Thread-safe for concurrent calls on one index (every call takes its own scratch and stream).
 * The *_device entry points below share one scratch per stream: calls on the SAME stream are serialised inside
 * the library while they enqueue. */
int knhip_search(const knhip_index* idx, const float* queries, int64_t nq, int32_t k,
                 int32_t nprobe, const uint8_t* bitset, int64_t bitset_nbits, int64_t* out_ids,
                 float* out_dist);
/* Search k_base candidates, re-rank them exactly against the raw fp32 rows held by `raw` (a BRUTE_FORCE knhip_index on
 * the same device whose row r is vector id r + id_offset) and return the k best: faiss::IndexRefine::search
 * (thirdparty/faiss/faiss/IndexRefine.cpp:61-140), Knowhere's build-time `refine` + search-time `refine_k`
 * (src/index/ivf/ivf.cc:673-700, 1073-1103).  Queries go up once, candidates never leave the device. */
int knhip_search_refine(const knhip_index* idx, const knhip_index* raw, const float* queries, int64_t nq, int32_t k,
                        int32_t k_base, int32_t nprobe, const uint8_t* bitset, int64_t bitset_nbits, int64_t* out_ids,
                        float* out_dist);
/* ---- quantised refine store: Knowhere's `refine_type` = fp16 / bf16 / sq8 (src/index/ivf/ivf_config.h:113-135,
 * src/index/refine/refine_utils.cc:99-160: the refine index is a faiss::IndexScalarQuantizer(d, QT_fp16 / QT_bf16 /
 * QT_8bit, metric) instead of IndexFlat).  knhip_rows keeps the encoded rows in HBM (row r = vector id r):
 *   fp16: the reference's scalar encode_fp16 (utils/fp16-inl.h:32-86: 11 mantissa bits kept, then round half UP -- IEEE
 *         round-to-nearest except that exact ties go up; builds with F16C round ties to even there);
 *   bf16: (bits + 0x8000) >> 16 (utils/bf16.h:27-32);
 *   sq8:  per-dimension ranges trained over ALL training rows (ScalarQuantizer::train, RS_minmax, rangestat_arg 0),
 *         code_i = (int)(255 * clamp((x_i - vmin_i) / vdiff_i, 0, 1)), x_i = vmin_i + vdiff_i * ((code_i + 0.5) / 255)
 *         (impl/scalar_quantizer/quantizers.h:108-146, codecs.h:26-41);
 *   sq6:  (round 5) the same ranges, 6-bit codes, four per three bytes: code_i = (int)((double)xi * 63.0),
 *         x_i = vmin_i + vdiff_i * ((code_i + 0.5) / 63) (Codec6bit, codecs.h:63-118); code_size = (6 d + 7) / 8;
 *   int8: (round 5) QT_8bit_direct_signed, Knowhere's int8 data format: code_i = (uint8_t)(x_i + 128) for values in
 *         [-128, 127], x_i = code_i - 128 (quantizers.h:350-379); nothing to train.  Distances through the float-domain
 *         distance computer (SIMDLevel::NONE; AVX2 / AVX-512 builds of the reference switch to an integer-domain computer
 *         that truncates the QUERY to bytes, sq-dispatch.h:543-560 -- not restated).
 *   sq4u: (round 5) QT_4bit_uniform: ONE range for all dimensions (train_Uniform over the n * d values,
 *         impl/scalar_quantizer/training.cpp:209-332): RS_minmax by default, RS_quantiles with argument 0.01 where Knowhere
 *         sets it (L2: refine_utils.cc:176-180) -- knhip_rows_train_uniform(rangestat, arg); code_i = (int)((double)xi * 15.0),
 *         two codes per byte (Codec4bit, codecs.h:43-59), x_i = vmin + vdiff * ((code_i + 0.5) / 15); trained = {vmin, vdiff}:
 *         set_trained / get_trained move ONE float each for this type.
 * knhip_search_refine_rows = knhip_search_refine with the second stage reading this store through the scalar quantizer's
 * distance computer (sequential: decode x_i, then (q_i - x_i)^2 / q_i * x_i added in order): bit-equal to the reference.
 * get_codes / add_codes move the faiss code bytes (Serialize / Deserialize: "IxSQ" inside "IxRF"). */
typedef struct knhip_rows knhip_rows;
enum { KNHIP_ROWS_FP16 = 1, KNHIP_ROWS_BF16 = 2, KNHIP_ROWS_SQ8 = 3, KNHIP_ROWS_SQ6 = 4, KNHIP_ROWS_INT8 = 5, KNHIP_ROWS_SQ4U = 6 };
int knhip_rows_create(int32_t device, int32_t dim, int32_t row_type, knhip_rows** out);
void knhip_rows_destroy(knhip_rows* rows);
int knhip_rows_train(knhip_rows* rows, int64_t n, const float* x);                 /* sq8 / sq6 / sq4u ranges; a no-op otherwise */
/* sq4u: the one range from all n * dim values; rangestat 0 = RS_minmax (widened by arg), 2 = RS_quantiles (the arg-quantile
 * and its mirror: ScalarQuantizer::RangeStat).  knhip_rows_train on an sq4u store = (0, 0), the ScalarQuantizer default */
int knhip_rows_train_uniform(knhip_rows* rows, int64_t n, const float* x, int32_t rangestat, float rangestat_arg);
int knhip_rows_set_trained(knhip_rows* rows, const float* vmin, const float* vdiff);
int knhip_rows_get_trained(const knhip_rows* rows, float* vmin, float* vdiff);
int knhip_rows_add(knhip_rows* rows, int64_t n, const float* x);                   /* encode + append (host fp32 rows) */
int knhip_rows_add_codes(knhip_rows* rows, int64_t n, const uint8_t* codes);       /* append already encoded rows */
int knhip_rows_get_codes(const knhip_rows* rows, uint8_t* out);                    /* [count][code_size] host */
int64_t knhip_rows_count(const knhip_rows* rows);
int64_t knhip_rows_code_size(const knhip_rows* rows);
int64_t knhip_rows_device_bytes(const knhip_rows* rows);
int knhip_search_refine_rows(const knhip_index* idx, const knhip_rows* rows, const float* queries, int64_t nq, int32_t k,
                             int32_t k_base, int32_t nprobe, const uint8_t* bitset, int64_t bitset_nbits, int64_t* out_ids,
                             float* out_dist);
/* the second stage alone, device pointers (knhip_refine_device over a quantised store): row r of `rows` = vector id
 * id_base + r; candidates outside the store are skipped slots -- what a rank of a sharded deployment runs on the part
 * of the store it holds (include/knhip_shards.h) */
int knhip_refine_rows_device(int32_t metric, const knhip_rows* rows, int64_t id_base, const float* d_queries, int64_t nq,
                             const int64_t* d_cand_ids, int32_t k_base, int32_t k, float* d_out_dist, int64_t* d_out_ids,
                             void* stream);
/* rows by id (IndexNode::GetVectorByIds); out [n][dim] host.  BRUTE_FORCE: row = id - id_offset.  IVF_FLAT: through a
 * direct map built on first use from the index's own ids (16 bytes per vector in HBM; the reference's
 * make_direct_map / reconstruct, thirdparty/faiss/faiss/IndexIVF.cpp); an id that is not stored is an error. */
int knhip_index_get_vectors(const knhip_index* idx, int64_t n, const int64_t* ids, float* out);
/* the tolerant form for an index that holds only part of the id space (one shard of a list-sharded IVF_FLAT, one row
 * range of a sharded FLAT): found[i] = 1 and out row i written where id i is stored here, found[i] = 0 and the row left
 * untouched otherwise; never an error for an absent id. */
int knhip_index_find_vectors(const knhip_index* idx, int64_t n, const int64_t* ids, float* out, uint8_t* found);
/* Same with every buffer already in HBM; enqueued on `stream` (hipStream_t, NULL = default
 * stream) and NOT synchronised at the end.  One exception inside: an IVF_PQ m = 32 batch that takes the matrix-core
 * prefilter (pq_filter.hip) waits once on `stream` in the middle of the batch the FIRST time a (k, nprobe) pair is seen
 * on this index, and every 64th time after: the selectivity guard reads two counters back to choose between the integer
 * form, the half-precision form and the exact kernels.  All other batches take the decision from the previous batch's
 * counters (pinned buffer + event, polled, never waited for).  The decision only picks kernels -- results do not depend
 * on it.  KNHIP_PQF_GUARD=0 or KNHIP_PQF=0 removes the wait altogether (graph capture). */
int knhip_search_device(const knhip_index* idx, const float* d_queries, int64_t nq, int32_t k,
                        int32_t nprobe, const uint8_t* d_bitset, int64_t bitset_nbits,
                        int64_t* d_out_ids, float* d_out_dist, void* stream);
/* Search with a GIVEN coarse assignment: d_keys / d_coarse_dis [nq][nprobe] as knhip_coarse_search_device
 * returns them (best-first; key < 0 = no list).  Replaces IndexIVF::search_preassigned
 * (thirdparty/faiss/faiss/IndexIVF.cpp:401-768), which is what IndexIVF::search calls after its coarse
 * quantizer step (:336-350).  Lets a list-sharded deployment run the coarse quantizer once per query instead
 * of once per rank: every rank assigns its slice of the batch, the assignments are all-gathered, every rank
 * scans the lists it owns.  nprobe must be <= nlist.  Same results as knhip_search_device. */
int knhip_search_preassigned_device(const knhip_index* idx, const float* d_queries, int64_t nq, int32_t k,
                                    int32_t nprobe, const int64_t* d_keys, const float* d_coarse_dis,
                                    const uint8_t* d_bitset, int64_t bitset_nbits, int64_t* d_out_ids,
                                    float* d_out_dist, void* stream);
/* Coarse quantizer only: top-nprobe centroids per query, best-first. */
int knhip_coarse_search_device(const knhip_index* idx, const float* d_queries, int64_t nq,
                               int32_t nprobe, int64_t* d_out_keys, float* d_out_dist,
                               void* stream);

/* ---- refine: exact re-rank of candidate ids against raw fp32 vectors resident in HBM ----
 * Replaces the second stage of faiss::IndexRefine::search
 * (thirdparty/faiss/faiss/IndexRefine.cpp:104-140; Knowhere's `refine` / `refine_k`,
 * src/index/ivf/ivf.cc:1073-1103).  d_base is row-major [nbase][dim], row r holds id id_base + r.
 * cand ids [nq][k_base] as returned by knhip_search_device with k = k_base (a -1 ends a row). */
int knhip_refine_device(int32_t metric, int32_t dim, const float* d_base, int64_t nbase,
                        int64_t id_base, const float* d_queries, int64_t nq, const int64_t* d_cand_ids,
                        int32_t k_base, int32_t k, float* d_out_dist, int64_t* d_out_ids, void* stream);

/* ---- multi-GPU: the reference's answer from a list-sharded index, ties at the k-th distance included ----
 * The reference's own sharding contract is ids-equal (tests/ut/test_bruteforce.cc:128-181; faiss IndexShards merges
 * per-shard heaps, IndexShards.cpp:247-256 -- whose tie behaviour is that of each shard's heap).  Here every shard resolves
 * nothing on its own: the rule of the header comment is applied ONCE, after the merge, over all shards' candidates:
 *   1. every shard: knhip_search_canonical_device for k + 1 results (no tie rule: the canonical order only);
 *   2. exchange + knhip_merge_topk_device of the (nq, k + 1) partials: the global canonical top-(k + 1) on every shard;
 *   3. knhip_tie_flag_device: the first k of every row -> the result; queries whose (k + 1)-th entry ties with the k-th are
 *      flagged (ascending list, identical on every shard; one 4-byte read-back);
 *   4. only if any is flagged: every shard knhip_tie_arrivals_device (its first k arrivals at or below the k-th distance, each
 *      with its place in the global scan order), exchange, knhip_tie_resolve_device writes the rule's answer over the rows.
 * include/knhip_shards.h does exactly this (C++ host); knowhere_amd/sharded.py does it over torch.distributed.
 * BRUTE_FORCE with k >= 100 and k = 1024 stay canonical, as on one index. */
/* 1 when a search of this kind and k follows the reference's boundary rule (k + 1 canonical results, flag, arrivals, resolve),
 * 0 when it returns the canonical k: KNHIP_TIES=canonical, k + 1 > 1024, BRUTE_FORCE with k >= 100 (the reference's reservoir).
 * The hosts of a list-sharded index ask this instead of restating the conditions. */
int knhip_ties_rule_applies(int32_t kind, int32_t k);
/* canonical top-k, no tie rule.  d_keys / d_coarse_dis: the coarse assignment [nq][nprobe] (IndexIVF::search_preassigned;
 * both NULL: assigned inside; BRUTE_FORCE: NULL) */
int knhip_search_canonical_device(const knhip_index* idx, const float* d_queries, int64_t nq, int32_t k, int32_t nprobe,
                                  const int64_t* d_keys, const float* d_coarse_dis, const uint8_t* d_bitset,
                                  int64_t bitset_nbits, int64_t* d_out_ids, float* d_out_dist, void* stream);
/* rows of k + 1 canonical results -> the first k to (d_out_dist, d_out_ids) [nq][k]; d_flagged: int32 [2 nq + 1] scratch whose
 * first *nflag_out entries are the flagged queries in ascending order.  Synchronises the stream (reads the count back). */
int knhip_tie_flag_device(const float* d_can_dist, const int64_t* d_can_ids, int64_t nq, int32_t k, float* d_out_dist,
                          int64_t* d_out_ids, int32_t* d_flagged, int32_t* nflag_out, void* stream);
/* this index's first k arrivals with distance <= the query's k-th distance (>= for IP) in scan order, for the flagged
 * queries: d_arr_dist / d_arr_ids / d_arr_key [nflag][k], d_arr_n [nflag] (arrivals found; min(k, .) are stored).
 * d_queries, d_keys, d_coarse_dis, d_can_dist ([.][k + 1]) are the whole batch's arrays (rows = d_flagged entries).
 * key: (probe rank << 40 | position in the list) for the IVF kinds; key_base + row for BRUTE_FORCE (key_base = the number
 * of rows held by the shards in front of this one). */
int knhip_tie_arrivals_device(const knhip_index* idx, const float* d_queries, const int32_t* d_flagged, int32_t nflag,
                              const float* d_can_dist, int32_t k, int32_t nprobe, const int64_t* d_keys,
                              const float* d_coarse_dis, const uint8_t* d_bitset, int64_t bitset_nbits, int64_t key_base,
                              float* d_arr_dist, int64_t* d_arr_ids, int64_t* d_arr_key, int64_t* d_arr_n, void* stream);
/* arrivals of all shards [nshards][nflag][k] (+ [nshards][nflag]) -> the rule's answer over the flagged rows of
 * (d_out_dist, d_out_ids) [nq][k] */
int knhip_tie_resolve_device(int32_t metric, int32_t nshards, const int32_t* d_flagged, int32_t nflag, int32_t k,
                             const float* d_can_dist, const int64_t* d_can_ids, const float* d_arr_dist,
                             const int64_t* d_arr_ids, const int64_t* d_arr_key, const int64_t* d_arr_n, float* d_out_dist,
                             int64_t* d_out_ids, void* stream);
/* host forms (results merged on the CPU: knhip_merge_topk_host's companions): flagged [nq] 0 / 1; the arrival arrays are
 * [nshards][nq][k] / [nshards][nq], rows of unflagged queries are not read */
int knhip_tie_flag_host(const float* can_dist, const int64_t* can_ids, int64_t nq, int32_t k, float* out_dist,
                        int64_t* out_ids, uint8_t* flagged_out);
int knhip_tie_resolve_host(int32_t metric, int32_t nshards, int64_t nq, int32_t k, const uint8_t* flagged,
                           const float* can_dist, const int64_t* can_ids, const float* arr_dist, const int64_t* arr_ids,
                           const int64_t* arr_key, const int64_t* arr_n, float* out_dist, int64_t* out_ids);
/* Sharded refine (the rows cut into one id range per shard).  IndexRefine pushes the re-scored candidates through its heap in
 * CANDIDATE order whoever holds their rows, so the selection needs every candidate's distance: every shard computes the
 * distances of the candidates it holds ([nq][k_base]; the all-ones pattern elsewhere), the arrays are exchanged and combined
 * (every candidate is held once), and ONE selection -- the single index's, tie rule included -- runs on the result. */
int knhip_refine_distances_device(int32_t metric, int32_t dim, const float* d_base, int64_t nbase, int64_t id_base,
                                  const float* d_queries, int64_t nq, const int64_t* d_cand_ids, int32_t k_base,
                                  float* d_out_dist, void* stream);
int knhip_refine_rows_distances_device(int32_t metric, const knhip_rows* rows, int64_t id_base, const float* d_queries,
                                       int64_t nq, const int64_t* d_cand_ids, int32_t k_base, float* d_out_dist,
                                       void* stream);
int knhip_refine_combine_device(int32_t nshards, int64_t n, const float* d_parts /*[nshards][n]*/, float* d_out, void* stream);
int knhip_refine_select_device(int32_t metric, int64_t nq, const int64_t* d_cand_ids, const float* d_dist, int32_t k_base,
                               int32_t k, float* d_out_dist, int64_t* d_out_ids, void* stream);
/* host form of the selection (distances already combined) */
int knhip_refine_select_host(int32_t metric, int64_t nq, const int64_t* cand_ids, const float* dist, int32_t k_base, int32_t k,
                             float* out_dist, int64_t* out_ids);

/* ---- multi-GPU: merge of per-shard partial top-k ----
 * parts laid out [nshard][nq][k]; ids < 0 are empty slots. */
int knhip_merge_topk_device(int32_t metric, int64_t nq, int32_t k, int32_t nshard,
                            const float* d_dist_parts, const int64_t* d_ids_parts, float* d_out_dist,
                            int64_t* d_out_ids, void* stream);
int knhip_merge_topk_host(int32_t metric, int64_t nq, int32_t k, int32_t nshard,
                          const float* dist_parts, const int64_t* ids_parts, float* out_dist,
                          int64_t* out_ids);

/* ---- src/simd distance primitives, HIP equivalents (device pointers, exact scalar order) ---- */
/* dis[i] = ||x - y_i||^2, y row-major [ny][d]          (fvec_L2sqr_ny,        hook.h:60) */
int knhip_fvec_L2sqr_ny(float* d_dis, const float* d_x, const float* d_y, int64_t d, int64_t ny,
                        void* stream);
/* ip[i] = <x, y_i>                                     (fvec_inner_products_ny, hook.h:63) */
int knhip_fvec_inner_products_ny(float* d_ip, const float* d_x, const float* d_y, int64_t d,
                                 int64_t ny, void* stream);
/* out[i] = ||x_i||^2 for n rows                        (fvec_norm_L2sqr,       hook.h:39) */
int knhip_fvec_norms_L2sqr(float* d_out, const float* d_x, int64_t d, int64_t n, void* stream);
/* c = a + bf * b                                       (fvec_madd,             hook.h:66) */
int knhip_fvec_madd(int64_t n, const float* d_a, float bf, const float* d_b, float* d_c,
                    void* stream);
/* int8 rows: int32 accumulate then cast                (int8_vec_L2sqr / _inner_product,
 *                                                       distances_ref.cc:386-404)       */
int knhip_int8_vec_L2sqr_ny(float* d_dis, const int8_t* d_x, const int8_t* d_y, int64_t d,
                            int64_t ny, void* stream);
int knhip_int8_vec_inner_products_ny(float* d_ip, const int8_t* d_x, const int8_t* d_y, int64_t d,
                                     int64_t ny, void* stream);

/* ---- the rest of the hook table (src/simd/hook.h:33-123; scalar definitions src/simd/distances_ref.cc) ----
 * Same conventions: device pointers, the reference's scalar operation order, results bit-equal to the *_ref functions.
 * Scalar-valued hooks (one x against one y) are the ny = 1 case of the row entries. */
/* dis[i] = sum_j |x_j - y_ij|                          (fvec_L1,   hook.h:42; distances_ref.cc:39-46) */
int knhip_fvec_L1_ny(float* d_dis, const float* d_x, const float* d_y, int64_t d, int64_t ny, void* stream);
/* dis[i] = max_j |x_j - y_ij|                          (fvec_Linf, hook.h:45; distances_ref.cc:48-55) */
int knhip_fvec_Linf_ny(float* d_dis, const float* d_x, const float* d_y, int64_t d, int64_t ny, void* stream);
/* out[i] = ||x_i||^2, float products summed in a double (fvec_norm_L2sqr_ref, distances_ref.cc:57-64; the entry above,
 * knhip_fvec_norms_L2sqr, is the float-accumulator form the FAISS tables use) */
int knhip_fvec_norms_L2sqr_ref(float* d_out, const float* d_x, int64_t d, int64_t n, void* stream);
/* y transposed: vector i is column i of y[d][d_offset]; dis[i] = ||x||^2 + y_sqlen[i] - 2 <x, y_i>
 *                                                      (fvec_L2sqr_ny_transposed, hook.h:66; distances_ref.cc:84-101) */
int knhip_fvec_L2sqr_ny_transposed(float* d_dis, const float* d_x, const float* d_y, const float* d_y_sqlen, int64_t d,
                                   int64_t d_offset, int64_t ny, void* stream);
/* distances into d_dis_tmp[ny] and *d_nearest = first index of the minimum (0 if ny == 0 or nothing below +inf)
 *                                                      (fvec_L2sqr_ny_nearest, hook.h:72; distances_ref.cc:106-121) */
int knhip_fvec_L2sqr_ny_nearest(float* d_dis_tmp, const float* d_x, const float* d_y, int64_t d, int64_t ny,
                                int64_t* d_nearest, void* stream);
/*                                                      (fvec_L2sqr_ny_nearest_y_transposed, hook.h:80; :128-145) */
int knhip_fvec_L2sqr_ny_nearest_y_transposed(float* d_dis_tmp, const float* d_x, const float* d_y,
                                             const float* d_y_sqlen, int64_t d, int64_t d_offset, int64_t ny,
                                             int64_t* d_nearest, void* stream);
/* c = a + bf * b and *d_imin = first index of the minimum of c below 1e20, -1 if none
 *                                                      (fvec_madd_and_argmin, hook.h:84; distances_ref.cc:154-168) */
int knhip_fvec_madd_and_argmin(int64_t n, const float* d_a, float bf, const float* d_b, float* d_c, int64_t* d_imin,
                               void* stream);
/* four rows sharing x, d_out4[r] = dist(x, y_r)        (fvec_inner_product_batch_4 / fvec_L2sqr_batch_4, hook.h:89-97) */
int knhip_fvec_batch_4(int32_t metric, const float* d_x, const float* d_y0, const float* d_y1, const float* d_y2,
                       const float* d_y3, int64_t d, float* d_out4, void* stream);
/* typed operands (include/knowhere/operands.h): fp16 / bf16 as 16-bit patterns, int8.
 * op 0: *_vec_L2sqr, 1: *_vec_inner_product, 2: *_vec_norm_L2sqr (x unused), one x against ny rows
 *                                                      (hook.h:104-123; distances_ref.cc:236-262, 311-337, 386-412) */
enum { KNHIP_DT_FP16 = 0, KNHIP_DT_BF16 = 1, KNHIP_DT_INT8 = 2 };
int knhip_typed_vec_ny(int32_t dtype, int32_t op, float* d_out, const void* d_x, const void* d_y, int64_t d,
                       int64_t ny, void* stream);
/*                                                      (*_vec_{inner_product,L2sqr}_batch_4) */
int knhip_typed_vec_batch_4(int32_t dtype, int32_t metric, const void* d_x, const void* d_y0, const void* d_y1,
                            const void* d_y2, const void* d_y3, int64_t d, float* d_out4, void* stream);
/* int32 results                                        (ivec_inner_product / ivec_L2sqr, hook.h:100-101) */
int knhip_ivec_ny(int32_t metric, int32_t* d_out, const int8_t* d_x, const int8_t* d_y, int64_t d, int64_t ny,
                  void* stream);

/* ---- profiling hooks (bench.py / rocprof cross-check) ---- */
#define KNHIP_NSTAGE 16
typedef struct knhip_stage_times {
    /* accumulated HIP-event milliseconds per stage since the last reset, and launch counts */
    float ms[KNHIP_NSTAGE];
    int64_t launches[KNHIP_NSTAGE];
    /* algorithmic bytes / flops of the last search (SURVEY.md section 8d definitions) */
    double scan_bytes;    /* sum over (query, probe) of len(list) * code_size */
    double coarse_flops;  /* 2 * nq * nlist * dim */
    int64_t scan_items;   /* work items launched by the scan kernel */
    int64_t coarse_fallback_queries; /* queries whose MFMA-prefilter certificate failed (exact redo) */
    double scan_bytes_rank0; /* part of scan_bytes handled by the rank-0 (dump + select / exact) phase */
    /* prefilter paths (mfma_scan.hip, pq_filter.hip): queries finished from their candidate lists, queries whose
       candidate list overflowed (redone by the exact kernels), candidates the filter passed */
    int64_t mscan_queries;
    int64_t mscan_overflow_queries;
    int64_t mscan_candidates;
    double mscan_stream_bytes; /* bytes the prefilter streams: sum over its units of len(list) * code_size */
    int64_t mscan_recomputed;  /* candidates that got an exact distance (IVF_PQ: after the finish kernel's pruning) */
    int64_t pq_filter_form;    /* IVF_PQ prefilter of the last search: 0 none (exact kernels), 1 half precision, 2 int8, 3 decode form;
                                * BRUTE_FORCE: 10 = the last search ran on the matrix cores (bf16 prefilter + exact re-rank), 0 = row scan */
    int64_t tie_queries;       /* queries with candidates tied at their k-th distance beyond the k-th place, resolved by the
                                  reference's first-come admission rule (scan order) instead of the canonical order */
    int64_t tie_anomalies;     /* ... of those, rows the resolution left at their canonical copy because its dump pass found
                                  fewer entries than the row holds (expected 0: counted so that it cannot go unnoticed) */
} knhip_stage_times;
/* stage indices */
enum {
    KNHIP_STAGE_COARSE = 0,   /* query x centroid distances + top-nprobe */
    KNHIP_STAGE_GROUP = 1,    /* (query,probe) -> per-list work table */
    KNHIP_STAGE_LUT = 2,      /* PQ query tables */
    KNHIP_STAGE_SCAN = 3,     /* per-list code scan (ADC / flat / SQ8) -- the dominant kernel */
    KNHIP_STAGE_MERGE = 4,    /* per-query merge of per-probe partial top-k */
    KNHIP_STAGE_OTHER = 5,
    KNHIP_STAGE_SCAN_RANK0 = 6, /* exact IVF_PQ kernels: rank-0 probes in dump mode + radix select (pq_scan_v2.hip);
                                   prefilter paths: the SAMPLE pass (tau_q from each query's closest lists) */
    KNHIP_STAGE_TABLES = 7,   /* prefilter paths: the queries' tables / operands of the filter + the selectivity guard */
    KNHIP_STAGE_REFINE = 8,   /* exact re-rank of the first stage's candidates (knhip_search_refine*, knhip_refine*_device
                                 when given the index) */
    KNHIP_STAGE_TIES = 9      /* k-th-boundary ties: detection, read-back, dump pass and rule of the flagged queries */
};
int knhip_profile_enable(knhip_index* idx, int on);
int knhip_profile_reset(knhip_index* idx);
int knhip_profile_get(const knhip_index* idx, knhip_stage_times* out);
const char* knhip_stage_kernel_name(int stage, int kind);

#ifdef __cplusplus
}
#endif
#endif /* KNHIP_H */
