/* oracle/oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the reference (zilliztech/knowhere @ /root/reference) Search()
 * arithmetic for BruteForce(FLAT) / IVF-Flat / IVF-PQ / IVF-SQ8.  It is the CHECKER for the
 * HIP path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it.  The product (knowhere_amd/, libknhip.so) never links or calls anything here.
 *
 * Parity status: PINNED.  The reference tests hold no golden id/distance vectors for this
 * path, so the restatement is pinned against outputs of the reference
 * itself: oracle/_ref/libknowhere_ref.so (the reference's own FAISS sources compiled in
 * place, scalar SIMDLevel::NONE build) -- tests/test_oracle_vs_ref.py requires bit-equal
 * distances and equal ids -- and against fixtures generated from it under tests/golden/.
 *
 * "T:" below abbreviates /root/reference/thirdparty/faiss/faiss/.
 */
#ifndef KNOWHERE_AMD_ORACLE_H
#define KNOWHERE_AMD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_FLAT = 0, ORC_IVF_FLAT = 1, ORC_IVF_PQ = 2, ORC_IVF_SQ8 = 3 };
enum { ORC_L2 = 0, ORC_IP = 1 };

/* A trained + populated index as plain arrays (same bytes the HIP side is given). */
typedef struct orc_index {
    int32_t kind;   /* ORC_* */
    int32_t metric; /* ORC_L2 | ORC_IP */
    int32_t d;
    int32_t M;     /* IVF_PQ: sub-quantizers */
    int32_t nbits; /* IVF_PQ: bits per code (8) */
    int32_t use_precomputed_table; /* IVF_PQ L2: 1 = term-2 table, 0 = residual tables */
    int64_t nlist;
    int64_t code_size;             /* bytes per stored vector */
    const float* centroids;        /* [nlist][d] */
    const float* pq_centroids;     /* [M][ksub][dsub] */
    const float* precomputed_table;/* [nlist][M*ksub] or NULL */
    const float* sq_trained;       /* vmin[d] then vdiff[d] */
    const int64_t* list_sizes;     /* [nlist] */
    const uint8_t* const* list_codes; /* [nlist] -> [len][code_size] */
    const int64_t* const* list_ids;   /* [nlist] -> [len] */
    const float* const* list_norms;   /* IVF_FLAT + COSINE: [nlist] -> [len] stored L2 norms (NULL otherwise) */
} orc_index;

/* ---- distance primitives (reference src/simd/distances_ref.cc:21-76) ---- */
float orc_fvec_L2sqr(const float* x, const float* y, size_t d);
float orc_fvec_inner_product(const float* x, const float* y, size_t d);
float orc_fvec_norm_L2sqr(const float* x, size_t d);
void orc_fvec_madd(size_t n, const float* a, float bf, const float* b, float* c);
void orc_fvec_L2sqr_ny(float* dis, const float* x, const float* y, size_t d, size_t ny);
void orc_fvec_inner_products_ny(float* ip, const float* x, const float* y, size_t d, size_t ny);
void orc_fvec_L2sqr_batch_4(const float* x, const float* y0, const float* y1, const float* y2,
                            const float* y3, size_t d, float* d0, float* d1, float* d2, float* d3);
int32_t orc_int8_vec_inner_product(const int8_t* x, const int8_t* y, size_t d);
int32_t orc_int8_vec_L2sqr(const int8_t* x, const int8_t* y, size_t d);

/* ---- rest of the src/simd hook table (src/simd/hook.h:33-123; scalar definitions distances_ref.cc) ----
 * typed operands: type 0 fp16, 1 bf16 (raw uint16 patterns), 2 int8; op 0 L2sqr, 1 inner product, 2 norm_L2sqr */
float orc_simd_fvec_L1(const float* x, const float* y, size_t d);
float orc_simd_fvec_Linf(const float* x, const float* y, size_t d);
float orc_simd_fvec_norm_L2sqr(const float* x, size_t d);
void orc_simd_fvec_L2sqr_ny_transposed(float* dis, const float* x, const float* y, const float* y_sqlen,
                                       size_t d, size_t d_offset, size_t ny);
size_t orc_simd_fvec_L2sqr_ny_nearest(float* tmp, const float* x, const float* y, size_t d, size_t ny);
size_t orc_simd_fvec_L2sqr_ny_nearest_y_transposed(float* tmp, const float* x, const float* y,
                                                   const float* y_sqlen, size_t d, size_t d_offset, size_t ny);
int orc_simd_fvec_madd_and_argmin(size_t n, const float* a, float bf, const float* b, float* c);
void orc_simd_fvec_batch_4(int is_l2, const float* x, const float* y0, const float* y1, const float* y2,
                           const float* y3, size_t d, float* out4);
float orc_simd_typed(int type, int op, const void* x, const void* y, size_t d);
void orc_simd_typed_batch_4(int type, int is_l2, const void* x, const void* y0, const void* y1, const void* y2,
                            const void* y3, size_t d, float* out4);
int32_t orc_simd_ivec(int is_l2, const int8_t* x, const int8_t* y, size_t d);

/* ---- COSINE (a12).  Queries passed to the cosine searches are already normalised (orc_normalize_vecs), as the
 * nodes do with CopyAndNormalizeVecs (src/index/flat/flat.cc:113-117, src/index/ivf/ivf.cc:946-950). ---- */
/* knowhere::NormalizeVecs, src/common/utils.cc:60-93: rows normalised in place, norms[i] = the divisor (1 if skipped) */
void orc_normalize_vecs(float* x, int64_t n, int d, float* norms);
/* L2NormsStorage::add, cppcontrib/knowhere/IndexCosine.cpp:236-250: inv[i] = 1 / sqrt(||x_i||^2), 1 for a zero row */
void orc_inverse_l2_norms(const float* x, int64_t n, int d, float* inv);
/* IndexFlatCosine::search -> knn_cosine (cppcontrib/knowhere/utils/distances.cpp:367-409): clamp(<q, y_j> * inv_j) */
int orc_flat_cosine_search(int d, int64_t nb, const float* xb, const float* inv_norms, int64_t nq, const float* xq,
                           int64_t k, const uint8_t* bitset, int64_t nbits, float* D, int64_t* I);

/* ---- top-k heap (T:utils/Heap.h) ---- */
void orc_heap_heapify(int is_max, size_t k, float* val, int64_t* ids);
void orc_heap_replace_top(int is_max, size_t k, float* val, int64_t* ids, float v, int64_t id);
size_t orc_heap_reorder(int is_max, size_t k, float* val, int64_t* ids);

/* ---- PQ tables ---- */
void orc_pq_inner_prod_table(int d, int M, int nbits, const float* pq_centroids, const float* x,
                             float* table /* [M][ksub] */);
void orc_pq_distance_table(int d, int M, int nbits, const float* pq_centroids, const float* x,
                           float* table);
void orc_pq_precompute_table(int d, int M, int nbits, int64_t nlist, const float* centroids,
                             const float* pq_centroids, float* out /* [nlist][M*ksub] */);
/* the per-(query,list) LUT and dis0 the ADC scan uses */
float orc_ivfpq_list_table(const orc_index* idx, const float* q, int64_t list_no, float coarse_dis,
                           const float* sim_table_2 /* query table or NULL */,
                           float* sim_table /* [M*ksub] out */);

/* ---- searches; bitset: bit set => row filtered OUT, LSB-first; NULL = no filter ---- */
int orc_flat_search(int metric, int d, int64_t nb, const float* xb, int64_t nq, const float* xq,
                    int64_t k, const uint8_t* bitset, int64_t nbits, float* D, int64_t* I);
int orc_coarse_search(const orc_index* idx, int64_t nq, const float* xq, int64_t nprobe, float* D,
                      int64_t* I);
int orc_ivf_search(const orc_index* idx, int64_t nq, const float* xq, int64_t k, int64_t nprobe,
                   const uint8_t* bitset, int64_t nbits, float* D, int64_t* I);
/* same, with the probe assignment given (T:IndexIVF.cpp:401 search_preassigned) */
int orc_ivf_search_preassigned(const orc_index* idx, int64_t nq, const float* xq, int64_t k,
                               int64_t nprobe, const int64_t* keys, const float* coarse_dis,
                               const uint8_t* bitset, int64_t nbits, float* D, int64_t* I);

/* Refine (second stage of IndexRefine::search, T:IndexRefine.cpp:104-140): re-score candidate
 * labels (stop at the first -1) with the exact metric on the raw vectors, keep the k best.
 * base row r holds id id_base + r. */
/* range search (results malloc'ed, release with orc_free) */
int orc_ivf_range_search(const orc_index* idx, int64_t nq, const float* xq, float radius,
                         int64_t max_empty_result_buckets, const uint8_t* bitset, int64_t nbits, int64_t* lims,
                         int64_t** out_ids, float** out_dis);
int orc_flat_range_search(int metric, int d, int64_t nb, const float* xb, int64_t nq, const float* xq, float radius,
                          const uint8_t* bitset, int64_t nbits, int64_t* lims, int64_t** out_ids, float** out_dis);
void orc_free(void* p);

int orc_refine(int metric, int d, const float* base, int64_t nbase, int64_t id_base, int64_t nq,
               const float* xq, int64_t k_base, const int64_t* cand_ids, int64_t k, float* D, int64_t* I);
/* quantised refine store (row_type 1 fp16, 2 bf16, 3 sq8): IndexRefine over faiss::IndexScalarQuantizer */
int64_t orc_rows_code_size(int row_type, int d);
void orc_rows_train(int d, int64_t n, const float* x, float* trained);
void orc_rows_train_uniform(int rangestat, float rs_arg, int64_t n_values, const float* x, float* trained);
void orc_rows_encode(int row_type, int d, int64_t n, const float* x, const float* trained, uint8_t* codes);
void orc_rows_decode(int row_type, int d, int64_t n, const uint8_t* codes, const float* trained, float* x);
int orc_refine_rows(int metric, int d, int row_type, const uint8_t* codes, const float* trained, int64_t nbase, int64_t nq,
                    const float* xq, int64_t k_base, const int64_t* cand_ids, int64_t k, float* D, int64_t* I);

/* host-side merge of per-shard partial results (T:utils/Heap.h:636 merge_knn_results semantics:
 * the k best of the union in canonical order) -- checker for the multi-GPU merge */
int orc_merge_topk(int metric, int64_t nq, int64_t k, int nshard, const float* D_parts,
                   const int64_t* I_parts, float* D, int64_t* I);

/* ---- build-side helpers used to make test indexes (restated add path) ---- */
/* nearest centroid by the metric (IndexFlat::assign) */
void orc_assign(int metric, int d, int64_t nlist, const float* centroids, int64_t n, const float* x,
                int64_t* out);
void orc_assign_dis(int metric, int d, int64_t nlist, const float* centroids, int64_t n, const float* x,
                    int64_t* out, float* out_dis);
/* PQ encode one (residual) vector: T:impl/ProductQuantizer.cpp:282-299 compute_code */
void orc_pq_compute_code(int d, int M, int nbits, const float* pq_centroids, const float* x,
                         uint8_t* code);
/* SQ8 encode / decode: T:impl/scalar_quantizer/quantizers.h:108-146, codecs.h:26-41 */
void orc_sq8_encode(int d, const float* trained, const float* x, uint8_t* code);
void orc_sq8_decode(int d, const float* trained, const uint8_t* code, float* x);

/* training restatements (Clustering / IndexIVF::train), see oracle.c */
void orc_rand_perm(int64_t* perm, int64_t n, int64_t seed);
void orc_renorm_L2(int d, int64_t n, float* x);
void orc_kmeans(int metric, int d, int64_t n, const float* x, int64_t ld, int off, int64_t k, int niter,
                int max_points, int64_t seed, int spherical, float* centroids);
/* niter <= 0: the level-1 quantizer's own default (10, T:IndexIVF.cpp:44); the PQ codebooks always train with the
 * ClusteringParameters default (25).  The coarse k-means is spherical for the inner product (T:IndexIVF.cpp:178-181) */
void orc_train_ivf(int kind, int metric, int d, int64_t nlist, int M, int64_t n, const float* x, int niter,
                   int max_points, int64_t seed, int coarse_given, float* centroids, float* pq_centroids,
                   float* sq_trained);

#ifdef __cplusplus
}
#endif
#endif
