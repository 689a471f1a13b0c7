/* oracle/oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Plain-C restatement of the arithmetic the reference executes on its CPU Search() path,
 * in the scalar (SIMDLevel::NONE / *_ref) form the reference's own tests use as the
 * known-answer (reference tests/ut/test_simd.cc:259-568).  Every floating-point operation
 * keeps the reference's order and is rounded once (build: -ffp-contract=off, no fast-math),
 * so a result can be compared BIT-FOR-BIT with oracle/_ref and with the HIP kernels.
 *
 * "T:" abbreviates /root/reference/thirdparty/faiss/faiss/.
 */
#include "oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Distance primitives.
 * reference src/simd/distances_ref.cc:21-37 (fvec_inner_product_ref, fvec_L2sqr_ref),
 * T:utils/simd_impl/distances_autovec-inl.h:43-66 (baseline scalar loops: same order).
 * ---------------------------------------------------------------------------------------- */
float orc_fvec_L2sqr(const float* x, const float* y, size_t d) {
    float res = 0;
    for (size_t i = 0; i < d; i++) {
        const float tmp = x[i] - y[i];
        res += tmp * tmp;
    }
    return res;
}

float orc_fvec_inner_product(const float* x, const float* y, size_t d) {
    float res = 0;
    for (size_t i = 0; i < d; i++) {
        res += x[i] * y[i];
    }
    return res;
}

/* T:utils/simd_impl/distances_autovec-inl.h:28-39 (float accumulator; the double in
 * distances_ref.cc:58-65 is noted there as a suspected typo; IVFPQ uses the faiss one). */
float orc_fvec_norm_L2sqr(const float* x, size_t d) {
    float res = 0;
    for (size_t i = 0; i < d; i++) {
        res += x[i] * x[i];
    }
    return res;
}

/* T:utils/distances_simd.cpp:33-43  c = a + bf * b */
void orc_fvec_madd(size_t n, const float* a, float bf, const float* b, float* c) {
    for (size_t i = 0; i < n; i++) {
        c[i] = a[i] + bf * b[i];
    }
}

/* reference src/simd/distances_ref.cc:67-81 */
void orc_fvec_L2sqr_ny(float* dis, const float* x, const float* y, size_t d, size_t ny) {
    for (size_t i = 0; i < ny; i++) {
        dis[i] = orc_fvec_L2sqr(x, y, d);
        y += d;
    }
}

void orc_fvec_inner_products_ny(float* ip, const float* x, const float* y, size_t d, size_t ny) {
    for (size_t i = 0; i < ny; i++) {
        ip[i] = orc_fvec_inner_product(x, y, d);
        y += d;
    }
}

/* reference src/simd/distances_ref.cc:159-186 (batch_4: four independent sequential sums) */
void orc_fvec_L2sqr_batch_4(const float* x, const float* y0, const float* y1, const float* y2,
                            const float* y3, size_t d, float* d0, float* d1, float* d2, float* d3) {
    *d0 = orc_fvec_L2sqr(x, y0, d);
    *d1 = orc_fvec_L2sqr(x, y1, d);
    *d2 = orc_fvec_L2sqr(x, y2, d);
    *d3 = orc_fvec_L2sqr(x, y3, d);
}

/* reference src/simd/distances_ref.cc:386-404: int32 accumulate, cast at the end */
int32_t orc_int8_vec_inner_product(const int8_t* x, const int8_t* y, size_t d) {
    int32_t res = 0;
    for (size_t i = 0; i < d; i++) {
        res += (int32_t)x[i] * (int32_t)y[i];
    }
    return res;
}

int32_t orc_int8_vec_L2sqr(const int8_t* x, const int8_t* y, size_t d) {
    int32_t res = 0;
    for (size_t i = 0; i < d; i++) {
        const int32_t tmp = (int32_t)x[i] - (int32_t)y[i];
        res += tmp * tmp;
    }
    return res;
}

/* ------------------------------------------------------------------------------------------
 * The rest of the src/simd hook table (reference src/simd/hook.h:33-123), restated from the scalar
 * definitions in src/simd/distances_ref.cc.  Pinned against the reference's own *_ref functions
 * (oracle/ref_simd.cpp over distances_ref.cc compiled where it lies) in tests/test_oracle.py.
 * fp16 / bf16 operands are raw 16-bit patterns (include/knowhere/operands.h:53-160).
 * ---------------------------------------------------------------------------------------- */
#include <math.h>

/* distances_ref.cc:39-55 */
float orc_simd_fvec_L1(const float* x, const float* y, size_t d) {
    float res = 0;
    for (size_t i = 0; i < d; i++) {
        res += fabsf(x[i] - y[i]);
    }
    return res;
}
float orc_simd_fvec_Linf(const float* x, const float* y, size_t d) {
    float res = 0;
    for (size_t i = 0; i < d; i++) {
        res = fmaxf(res, fabsf(x[i] - y[i]));
    }
    return res;
}
/* distances_ref.cc:57-64: float products summed in a DOUBLE, rounded once on return */
float orc_simd_fvec_norm_L2sqr(const float* x, size_t d) {
    double res = 0;
    for (size_t i = 0; i < d; i++) {
        const float p = x[i] * x[i];
        res += (double)p;
    }
    return (float)res;
}
/* distances_ref.cc:84-101: y is [d][d_offset] (column i = vector i); expanded form */
void orc_simd_fvec_L2sqr_ny_transposed(float* dis, const float* x, const float* y, const float* y_sqlen,
                                       size_t d, size_t d_offset, size_t ny) {
    float x_sqlen = 0;
    for (size_t j = 0; j < d; j++) {
        x_sqlen += x[j] * x[j];
    }
    for (size_t i = 0; i < ny; i++) {
        float dp = 0;
        for (size_t j = 0; j < d; j++) {
            dp += x[j] * y[i + j * d_offset];
        }
        const float s = x_sqlen + y_sqlen[i];
        const float t = 2 * dp;
        dis[i] = s - t;
    }
}
/* distances_ref.cc:106-121 / :128-145: first strict minimum below +inf, 0 if none */
static size_t orc_first_min(const float* dis, size_t ny) {
    size_t best = 0;
    float vmin = HUGE_VALF;
    for (size_t i = 0; i < ny; i++) {
        if (dis[i] < vmin) {
            vmin = dis[i];
            best = i;
        }
    }
    return best;
}
size_t orc_simd_fvec_L2sqr_ny_nearest(float* tmp, const float* x, const float* y, size_t d, size_t ny) {
    orc_fvec_L2sqr_ny(tmp, x, y, d, ny);
    return orc_first_min(tmp, ny);
}
size_t orc_simd_fvec_L2sqr_ny_nearest_y_transposed(float* tmp, const float* x, const float* y,
                                                   const float* y_sqlen, size_t d, size_t d_offset, size_t ny) {
    orc_simd_fvec_L2sqr_ny_transposed(tmp, x, y, y_sqlen, d, d_offset, ny);
    return orc_first_min(tmp, ny);
}
/* distances_ref.cc:154-168: c = a + bf * b and the first index whose c is below 1e20 and minimal (-1 if none) */
int orc_simd_fvec_madd_and_argmin(size_t n, const float* a, float bf, const float* b, float* c) {
    float vmin = 1e20f;
    int imin = -1;
    for (size_t i = 0; i < n; i++) {
        c[i] = a[i] + bf * b[i];
        if (c[i] < vmin) {
            vmin = c[i];
            imin = (int)i;
        }
    }
    return imin;
}
/* distances_ref.cc:170-210: four independent sequential sums sharing x */
void orc_simd_fvec_batch_4(int is_l2, const float* x, const float* y0, const float* y1, const float* y2,
                           const float* y3, size_t d, float* out4) {
    const float* ys[4] = {y0, y1, y2, y3};
    for (int r = 0; r < 4; r++) {
        out4[r] = is_l2 ? orc_fvec_L2sqr(x, ys[r], d) : orc_fvec_inner_product(x, ys[r], d);
    }
}

/* IEEE binary16 -> binary32 (exact); operands.h:73-100 */
static float orc_half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: renormalise */
            int e = -1;
            do {
                man <<= 1;
                e++;
            } while (!(man & 0x400u));
            bits = sign | ((uint32_t)(112 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 112) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}
/* bfloat16 -> binary32: the pattern is the high half (operands.h:141-147) */
static float orc_bf16_to_float(uint16_t h) {
    const uint32_t bits = (uint32_t)h << 16;
    float f;
    memcpy(&f, &bits, 4);
    return f;
}
static float orc_typed_at(int type, const void* p, size_t i) {
    if (type == 0) return orc_half_to_float(((const uint16_t*)p)[i]);
    if (type == 1) return orc_bf16_to_float(((const uint16_t*)p)[i]);
    return (float)((const int8_t*)p)[i];
}
/* type 0 fp16, 1 bf16, 2 int8; op 0 L2sqr, 1 inner product, 2 norm_L2sqr.
 * fp16 / bf16: convert each element to float, float arithmetic, sequential (distances_ref.cc:236-262, :311-337;
 * the norm sums float products in a double); int8: int32 accumulate then cast (distances_ref.cc:386-412). */
float orc_simd_typed(int type, int op, const void* x, const void* y, size_t d) {
    if (type == 2) {
        int32_t res = 0;
        const int8_t* a = (const int8_t*)x;
        const int8_t* b = (const int8_t*)y;
        for (size_t i = 0; i < d; i++) {
            if (op == 0) {
                const int32_t t = (int32_t)a[i] - (int32_t)b[i];
                res += t * t;
            } else if (op == 1) {
                res += (int32_t)a[i] * (int32_t)b[i];
            } else {
                res += (int32_t)a[i] * (int32_t)a[i];
            }
        }
        return (float)res;
    }
    if (op == 2) {
        double res = 0;
        for (size_t i = 0; i < d; i++) {
            const float v = orc_typed_at(type, x, i);
            const float p = v * v;
            res += (double)p;
        }
        return (float)res;
    }
    float res = 0;
    for (size_t i = 0; i < d; i++) {
        const float a = orc_typed_at(type, x, i), b = orc_typed_at(type, y, i);
        if (op == 0) {
            const float t = a - b;
            res += t * t;
        } else {
            res += a * b;
        }
    }
    return res;
}
void orc_simd_typed_batch_4(int type, int is_l2, const void* x, const void* y0, const void* y1, const void* y2,
                            const void* y3, size_t d, float* o) {
    const void* ys[4] = {y0, y1, y2, y3};
    for (int r = 0; r < 4; r++) {
        o[r] = orc_simd_typed(type, is_l2 ? 0 : 1, x, ys[r], d);
    }
}
/* distances_ref.cc:217-233 (ivec_*: the obsolete hnsw-sq entries, int32 results) */
int32_t orc_simd_ivec(int is_l2, const int8_t* x, const int8_t* y, size_t d) {
    return is_l2 ? orc_int8_vec_L2sqr(x, y, d) : orc_int8_vec_inner_product(x, y, d);
}

/* ------------------------------------------------------------------------------------------
 * Top-k heap: T:utils/Heap.h:112-151 (heap_replace_top), :45-75 (heap_pop), :318-343
 * (heap_heapify), :427-457 (heap_reorder); comparators T:utils/ordered_key_value.h:43-83.
 * is_max=1 is CMax (keeps the k SMALLEST, L2); is_max=0 is CMin (keeps the k LARGEST, IP).
 * ---------------------------------------------------------------------------------------- */
static inline int cmp2(int is_max, float a1, float b1, int64_t a2, int64_t b2) {
    if (is_max) {
        return (a1 > b1) || ((a1 == b1) && (a2 > b2));
    }
    return (a1 < b1) || ((a1 == b1) && (a2 < b2));
}

static inline int cmp1(int is_max, float a, float b) {
    return is_max ? (a > b) : (a < b);
}

static inline float neutral(int is_max) {
    return is_max ? FLT_MAX : -FLT_MAX;
}

void orc_heap_heapify(int is_max, size_t k, float* val, int64_t* ids) {
    for (size_t i = 0; i < k; i++) {
        val[i] = neutral(is_max);
        ids[i] = -1;
    }
}

/* sift (v,id) down from the root of the 1-based heap of size k */
static void sift_from_root(int is_max, size_t k, float* val, int64_t* ids, float v, int64_t id) {
    float* hv = val - 1;
    int64_t* hi = ids - 1;
    size_t i = 1;
    for (;;) {
        size_t i1 = i << 1, i2 = i1 + 1;
        if (i1 > k) {
            break;
        }
        size_t c;
        if (i2 == k + 1 || cmp2(is_max, hv[i1], hv[i2], hi[i1], hi[i2])) {
            c = i1;
        } else {
            c = i2;
        }
        if (cmp2(is_max, v, hv[c], id, hi[c])) {
            break;
        }
        hv[i] = hv[c];
        hi[i] = hi[c];
        i = c;
    }
    hv[i] = v;
    hi[i] = id;
}

void orc_heap_replace_top(int is_max, size_t k, float* val, int64_t* ids, float v, int64_t id) {
    sift_from_root(is_max, k, val, ids, v, id);
}

static void heap_pop(int is_max, size_t k, float* val, int64_t* ids) {
    /* the last element is re-inserted from the root over the remaining k-1 slots */
    float v = val[k - 1];
    int64_t id = ids[k - 1];
    sift_from_root(is_max, k - 1 ? k - 1 : 0, val, ids, v, id);
    /* note: T:utils/Heap.h:45-75 sifts over size k with the i1>k guard reading slot k,
     * which holds (v,id) itself; restricting to k-1 gives the same placement. */
}

size_t orc_heap_reorder(int is_max, size_t k, float* val, int64_t* ids) {
    size_t i, ii;
    for (i = 0, ii = 0; i < k; i++) {
        float v = val[0];
        int64_t id = ids[0];
        if (k - i > 1) {
            heap_pop(is_max, k - i, val, ids);
        }
        val[k - ii - 1] = v;
        ids[k - ii - 1] = id;
        if (id != -1) {
            ii++;
        }
    }
    size_t nel = ii;
    memmove(val, val + k - ii, ii * sizeof(*val));
    memmove(ids, ids + k - ii, ii * sizeof(*ids));
    for (; ii < k; ii++) {
        val[ii] = neutral(is_max);
        ids[ii] = -1;
    }
    return nel;
}

/* HeapResultHandler::add_result, T:impl/ResultHandler.h:271-278: strict improvement only */
static inline void heap_add(int is_max, size_t k, float* val, int64_t* ids, float dis, int64_t id) {
    if (cmp1(is_max, val[0], dis)) {
        orc_heap_replace_top(is_max, k, val, ids, dis, id);
    }
}

/* Result sink of a list scan: top-k heap (HeapResultHandler) or range collector
 * (RangeQueryResult::add, T:impl/AuxIndexStructures.h; admission C::cmp(radius, dis), i.e. strictly
 * inside the radius: L2 dis < radius, IP dis > radius -- T:IndexIVFFlat.cpp scan_codes_range,
 * IVFPQScanner_impl.h scan_codes_range, utils/distances.cpp range_search_L2sqr). */
typedef struct orc_sink {
    int is_max;
    /* top-k */
    size_t k;
    float* val;
    int64_t* ids;
    /* range (k == 0) */
    float radius;
    size_t n, cap;
    float* rdis;
    int64_t* rids;
} orc_sink;

static inline void sink_add(orc_sink* s, float dis, int64_t id) {
    if (s->k > 0) {
        if (cmp1(s->is_max, s->val[0], dis)) {
            orc_heap_replace_top(s->is_max, s->k, s->val, s->ids, dis, id);
        }
        return;
    }
    if (cmp1(s->is_max, s->radius, dis)) {
        if (s->n == s->cap) {
            s->cap = s->cap ? s->cap * 2 : 256;
            s->rdis = (float*)realloc(s->rdis, sizeof(float) * s->cap);
            s->rids = (int64_t*)realloc(s->rids, sizeof(int64_t) * s->cap);
        }
        s->rdis[s->n] = dis;
        s->rids[s->n] = id;
        s->n++;
    }
}

/* BitsetViewIDSelector::is_member, reference include/knowhere/bitsetview_idselector.h:20-31 */
static inline int filtered_out(const uint8_t* bitset, int64_t nbits, int64_t id) {
    if (!bitset || id < 0 || id >= nbits) {
        return 0;
    }
    return (bitset[id >> 3] >> (id & 7)) & 1;
}

/* ------------------------------------------------------------------------------------------
 * FLAT / BruteForce: reference src/index/flat/flat.cc:98-118 -> IndexFlat::search(1,..) ->
 * T:utils/distances.cpp:326-362 exhaustive_L2sqr_seq / :283-322 exhaustive_inner_product_seq
 * with a HeapResultHandler (k < 100).  [k >= 100 uses a reservoir in the reference; it returns
 * the same set except for exact distance ties at the k-th boundary -- not restated.]
 * ---------------------------------------------------------------------------------------- */
int orc_flat_search(int metric, int d, int64_t nb, const float* xb, int64_t nq, const float* xq,
                    int64_t k, const uint8_t* bitset, int64_t nbits, float* D, int64_t* I) {
    const int is_max = (metric == ORC_L2);
    for (int64_t i = 0; i < nq; i++) {
        const float* x = xq + i * (int64_t)d;
        float* simi = D + i * k;
        int64_t* idxi = I + i * k;
        orc_heap_heapify(is_max, (size_t)k, simi, idxi);
        for (int64_t j = 0; j < nb; j++) {
            if (filtered_out(bitset, nbits, j)) {
                continue;
            }
            const float* y = xb + j * (int64_t)d;
            float dis = is_max ? orc_fvec_L2sqr(x, y, (size_t)d)
                               : orc_fvec_inner_product(x, y, (size_t)d);
            heap_add(is_max, (size_t)k, simi, idxi, dis, j);
        }
        orc_heap_reorder(is_max, (size_t)k, simi, idxi);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * COSINE.  reference src/common/utils.cc:60-93 NormalizeVec: the squared norm comes from the fvec_norm_L2sqr hook
 * (scalar level = distances_ref.cc:57-64, float products summed in a double); a row is divided by sqrt(norm^2)
 * only when norm^2 > 0 and |1 - norm^2| > 1e-5; the returned norm is that divisor, else 1.
 * ---------------------------------------------------------------------------------------- */
void orc_normalize_vecs(float* x, int64_t n, int d, float* norms) {
    for (int64_t i = 0; i < n; i++) {
        float* v = x + i * (int64_t)d;
        const float ns = orc_simd_fvec_norm_L2sqr(v, (size_t)d);
        float norm = 1.0f;
        if (ns > 0 && fabsf(1.0f - ns) > 0.00001f) {
            norm = sqrtf(ns);
            for (int j = 0; j < d; j++) {
                v[j] = v[j] / norm;
            }
        }
        if (norms) {
            norms[i] = norm;
        }
    }
}

/* T:cppcontrib/knowhere/IndexCosine.cpp:236-250 */
void orc_inverse_l2_norms(const float* x, int64_t n, int d, float* inv) {
    for (int64_t i = 0; i < n; i++) {
        const float ns = orc_simd_fvec_norm_L2sqr(x + i * (int64_t)d, (size_t)d);
        inv[i] = (ns == 0.0f) ? 1.0f : (1.0f / sqrtf(ns));
    }
}

/* FLAT + COSINE: FlatIndexNode::Search (reference src/index/flat/flat.cc:98-122) -> IndexFlatCosine::search
 * (IndexCosine.cpp:303-312) -> knn_cosine -> exhaustive_cosine_seq_impl (cppcontrib/knowhere/utils/distances.cpp
 * :367-409): similarity = clamp(<q, y_j> * inv_norm_j, -1, 1), larger is better, heap for k < 100 */
int orc_flat_cosine_search(int d, int64_t nb, const float* xb, const float* inv_norms, int64_t nq, const float* xq,
                           int64_t k, const uint8_t* bitset, int64_t nbits, float* D, int64_t* I) {
    for (int64_t i = 0; i < nq; i++) {
        const float* x = xq + i * (int64_t)d;
        float* simi = D + i * k;
        int64_t* idxi = I + i * k;
        orc_heap_heapify(0, (size_t)k, simi, idxi);
        for (int64_t j = 0; j < nb; j++) {
            if (filtered_out(bitset, nbits, j)) {
                continue;
            }
            float dis = orc_fvec_inner_product(x, xb + j * (int64_t)d, (size_t)d) * inv_norms[j];
            dis = dis < -1.0f ? -1.0f : (dis > 1.0f ? 1.0f : dis);
            heap_add(0, (size_t)k, simi, idxi, dis, j);
        }
        orc_heap_reorder(0, (size_t)k, simi, idxi);
    }
    return 0;
}

/* Coarse quantizer: T:IndexIVF.cpp:336-342 quantizer->search(1, q, nprobe) == the FLAT search
 * above over the centroids (never filtered: quantizer_params is null). */
int orc_coarse_search(const orc_index* idx, int64_t nq, const float* xq, int64_t nprobe, float* D,
                      int64_t* I) {
    return orc_flat_search(idx->metric, idx->d, idx->nlist, idx->centroids, nq, xq, nprobe, NULL,
                           0, D, I);
}

/* ------------------------------------------------------------------------------------------
 * Product quantizer tables.
 * T:impl/ProductQuantizer.cpp:443-485: one fvec_*_ny per sub-quantizer.
 * ---------------------------------------------------------------------------------------- */
void orc_pq_inner_prod_table(int d, int M, int nbits, const float* cb, const float* x,
                             float* table) {
    const size_t ksub = (size_t)1 << nbits, dsub = (size_t)d / (size_t)M;
    for (int m = 0; m < M; m++) {
        orc_fvec_inner_products_ny(table + m * ksub, x + m * dsub, cb + m * ksub * dsub, dsub,
                                   ksub);
    }
}

void orc_pq_distance_table(int d, int M, int nbits, const float* cb, const float* x, float* table) {
    const size_t ksub = (size_t)1 << nbits, dsub = (size_t)d / (size_t)M;
    for (int m = 0; m < M; m++) {
        orc_fvec_L2sqr_ny(table + m * ksub, x + m * dsub, cb + m * ksub * dsub, dsub, ksub);
    }
}

/* T:IndexIVFPQ.cpp:465-484 (use_precomputed_table == 1):
 *   r_norms[m][j] = ||cb[m][j]||^2 ; tab = <c_list,m , cb[m][j]> ; tab = r_norms + 2.0 * tab */
void orc_pq_precompute_table(int d, int M, int nbits, int64_t nlist, const float* centroids,
                             const float* cb, float* out) {
    const size_t ksub = (size_t)1 << nbits, dsub = (size_t)d / (size_t)M;
    const size_t m_ksub = (size_t)M * ksub;
    float* r_norms = (float*)malloc(sizeof(float) * m_ksub);
    for (size_t m = 0; m < (size_t)M; m++) {
        for (size_t j = 0; j < ksub; j++) {
            r_norms[m * ksub + j] = orc_fvec_norm_L2sqr(cb + (m * ksub + j) * dsub, dsub);
        }
    }
    for (int64_t i = 0; i < nlist; i++) {
        float* tab = out + (size_t)i * m_ksub;
        orc_pq_inner_prod_table(d, M, nbits, cb, centroids + i * (int64_t)d, tab);
        orc_fvec_madd(m_ksub, r_norms, 2.0f, tab, tab);
    }
    free(r_norms);
}

/* Per-(query, list) table + dis0: T:impl/pq_code_distance/IVFPQ_QueryTables.cpp:110-145.
 *  IP            : dis0 = <q, c_list>, table = query table (sim_table == inner-prod table)
 *  L2, precomp=1 : dis0 = coarse_dis,  table = precomp[list] + (-2) * sim_table_2
 *  L2, precomp=0 : dis0 = 0,           table = ||(q - c_list)_m - cb[m][j]||^2           */
float orc_ivfpq_list_table(const orc_index* idx, const float* q, int64_t list_no, float coarse_dis,
                           const float* sim_table_2, float* sim_table) {
    const size_t ksub = (size_t)1 << idx->nbits;
    const size_t m_ksub = (size_t)idx->M * ksub;
    const float* c = idx->centroids + list_no * (int64_t)idx->d;
    if (idx->metric == ORC_IP) {
        /* the scan uses the query table directly (sim_table set by init_query_IP) */
        if (sim_table_2 && sim_table != sim_table_2) {
            memcpy(sim_table, sim_table_2, sizeof(float) * m_ksub);
        }
        return orc_fvec_inner_product(q, c, (size_t)idx->d);
    }
    if (idx->use_precomputed_table == 1) {
        orc_fvec_madd(m_ksub, idx->precomputed_table + (size_t)list_no * m_ksub, -2.0f, sim_table_2,
                      sim_table);
        return coarse_dis;
    }
    /* residual tables: quantizer->compute_residual (x - centroid) then compute_distance_table */
    float* r = (float*)malloc(sizeof(float) * (size_t)idx->d);
    for (int i = 0; i < idx->d; i++) {
        r[i] = q[i] - c[i];
    }
    orc_pq_distance_table(idx->d, idx->M, idx->nbits, idx->pq_centroids, r, sim_table);
    free(r);
    return 0.0f;
}

/* ADC: T:impl/pq_code_distance/pq_code_distance-inl.h:73-90: result = 0; result += tab[m][c_m]
 * in m order; the caller adds dis0 in front: dis = dis0 + result
 * (T:impl/pq_code_distance/IVFPQScanner_impl.h:147-150). */
/* Code widths other than 8 bits: T:impl/ProductQuantizer-inl.h:67-101 PQDecoderGeneric::decode / T:impl/ProductQuantizer.h:195
 * PQEncoderGeneric -- the M indices are a little-endian bit string, index m in bits [m nbits, (m + 1) nbits), the code
 * takes (M nbits + 7) / 8 bytes (T:impl/ProductQuantizer.cpp:69).  The scanner of such an index is the same template with
 * the generic decoder (T:impl/pq_code_distance/pq_code_distance-inl.h:73-90): the same sequential sum. */
static inline uint32_t pq_code_get(const uint8_t* code, int nbits, int m) {
    const size_t bit = (size_t)m * (size_t)nbits;
    size_t byte = bit >> 3;
    int off = (int)(bit & 7), got = 0;
    uint32_t v = 0;
    while (got < nbits) {
        const int take = (8 - off) < (nbits - got) ? (8 - off) : (nbits - got);
        v |= (((uint32_t)code[byte] >> off) & ((1u << take) - 1u)) << got;
        got += take;
        off = 0;
        byte++;
    }
    return v;
}

int64_t orc_pq_code_size(int M, int nbits) {
    return ((int64_t)M * nbits + 7) / 8;
}

/* M byte-wide indices -> the reference's code bytes, and back (nbits <= 8) */
void orc_pq_pack(int M, int nbits, const uint8_t* idx, uint8_t* code) {
    memset(code, 0, (size_t)orc_pq_code_size(M, nbits));
    for (int m = 0; m < M; m++) {
        const size_t bit = (size_t)m * (size_t)nbits;
        const uint32_t v = (uint32_t)idx[m] << (bit & 7); /* (at most 15 bits) */
        code[bit >> 3] |= (uint8_t)v;
        if ((bit & 7) + (size_t)nbits > 8) {
            code[(bit >> 3) + 1] |= (uint8_t)(v >> 8);
        }
    }
}

void orc_pq_unpack(int M, int nbits, const uint8_t* code, uint8_t* idx) {
    for (int m = 0; m < M; m++) {
        idx[m] = (uint8_t)pq_code_get(code, nbits, m);
    }
}

static inline float adc_distance(int M, size_t ksub, const float* sim_table, const uint8_t* code) {
    const float* tab = sim_table;
    float result = 0;
    if (ksub == 256) {
        for (int m = 0; m < M; m++) {
            result += tab[code[m]];
            tab += ksub;
        }
        return result;
    }
    int nbits = 0;
    while (((size_t)1 << nbits) < ksub) {
        nbits++;
    }
    for (int m = 0; m < M; m++) {
        result += tab[pq_code_get(code, nbits, m)];
        tab += ksub;
    }
    return result;
}

/* SQ8: T:impl/scalar_quantizer/codecs.h:37-41 decode_component = (code + 0.5f) / 255.0f,
 *      T:impl/scalar_quantizer/quantizers.h:139-145 x = vmin[i] + xi * vdiff[i],
 *      T:impl/scalar_quantizer/similarities.h:46-49 (L2: tmp = y - x; accu += tmp*tmp),
 *      :80-82 (IP: accu += y * x) */
static inline float sq8_component(const float* trained, int d, const uint8_t* code, int i) {
    const float xi = ((float)code[i] + 0.5f) / 255.0f;
    return trained[i] + xi * trained[d + i];
}

static float sq8_distance(int metric, int d, const float* trained, const float* y,
                          const uint8_t* code) {
    float accu = 0;
    if (metric == ORC_L2) {
        for (int i = 0; i < d; i++) {
            const float x = sq8_component(trained, d, code, i);
            const float tmp = y[i] - x;
            accu += tmp * tmp;
        }
    } else {
        for (int i = 0; i < d; i++) {
            const float x = sq8_component(trained, d, code, i);
            accu += y[i] * x;
        }
    }
    return accu;
}

/* ------------------------------------------------------------------------------------------
 * IVF search: T:IndexIVF.cpp:401-768 search_preassigned, parallel_mode 0, max_codes = 0:
 * per query: heapify; for each probe in coarse-rank order: skip key<0 / empty lists,
 * set_list, scan_codes in storage order through a HeapResultHandler; heap_reorder.
 * ---------------------------------------------------------------------------------------- */
typedef struct orc_scan_state {
    float* sim_table;
    float* sim_table_2;
    float* resid;
} orc_scan_state;

static void scan_state_init(const orc_index* idx, orc_scan_state* st) {
    const size_t ksub = (size_t)1 << (idx->kind == ORC_IVF_PQ ? idx->nbits : 0);
    const size_t m_ksub = (size_t)idx->M * ksub;
    st->sim_table = st->sim_table_2 = NULL;
    st->resid = (float*)malloc(sizeof(float) * (size_t)idx->d);
    if (idx->kind == ORC_IVF_PQ) {
        st->sim_table = (float*)malloc(sizeof(float) * m_ksub);
        st->sim_table_2 = (float*)malloc(sizeof(float) * m_ksub);
    }
}

static void scan_state_free(orc_scan_state* st) {
    free(st->sim_table);
    free(st->sim_table_2);
    free(st->resid);
}

/* set_query: T:impl/pq_code_distance/IVFPQ_QueryTables.cpp:44-67 */
static void scan_set_query(const orc_index* idx, orc_scan_state* st, const float* q) {
    if (idx->kind == ORC_IVF_PQ) {
        if (idx->metric == ORC_IP) {
            orc_pq_inner_prod_table(idx->d, idx->M, idx->nbits, idx->pq_centroids, q, st->sim_table);
        } else if (idx->use_precomputed_table == 1) {
            orc_pq_inner_prod_table(idx->d, idx->M, idx->nbits, idx->pq_centroids, q, st->sim_table_2);
        }
    }
}

/* set_list + scan_codes of one inverted list in storage order into the sink */
static void scan_one_list(const orc_index* idx, orc_scan_state* st, const float* q, int64_t key, float cdis,
                          const uint8_t* bitset, int64_t nbits, orc_sink* sk) {
    const int is_max = (idx->metric == ORC_L2);
    const int d = idx->d;
    const size_t ksub = (size_t)1 << (idx->kind == ORC_IVF_PQ ? idx->nbits : 0);
    const int64_t len = idx->list_sizes[key];
    const uint8_t* codes = idx->list_codes[key];
    const int64_t* ids = idx->list_ids[key];
    if (idx->kind == ORC_IVF_FLAT) {
        /* T:IndexIVFFlat.cpp IVFFlatScanner::scan_codes: dis = metric(q, row) */
        for (int64_t j = 0; j < len; j++) {
            if (filtered_out(bitset, nbits, ids[j])) {
                continue;
            }
            const float* y = (const float*)(codes + j * idx->code_size);
            float dis = is_max ? orc_fvec_L2sqr(q, y, (size_t)d) : orc_fvec_inner_product(q, y, (size_t)d);
            if (idx->list_norms != NULL && idx->list_norms[key] != NULL) {
                /* IndexIVFFlatCosine: T:cppcontrib/knowhere/IndexIVFFlat.cpp:199-213 dis = ip(q, raw row) / norm[j] */
                dis = dis / idx->list_norms[key][j];
            }
            sink_add(sk, dis, ids[j]);
        }
    } else if (idx->kind == ORC_IVF_PQ) {
        const float dis0 = orc_ivfpq_list_table(idx, q, key, cdis,
                                                idx->metric == ORC_IP ? st->sim_table : st->sim_table_2,
                                                st->sim_table);
        for (int64_t j = 0; j < len; j++) {
            if (filtered_out(bitset, nbits, ids[j])) {
                continue;
            }
            const float dis = dis0 + adc_distance(idx->M, ksub, st->sim_table, codes + j * idx->code_size);
            sink_add(sk, dis, ids[j]);
        }
    } else if (idx->kind == ORC_IVF_SQ8) {
        /* reference thirdparty/faiss/faiss/cppcontrib/knowhere/IndexScalarQuantizer.cpp
         * :196-290 (IP: accu0 = coarse_dis, dis = accu0 + <q, x>),
         * :292-400 (L2: query residual q - c_list, dis = ||r - x||^2); by_residual */
        const float* y = q;
        float accu0 = 0;
        if (idx->metric == ORC_IP) {
            accu0 = cdis;
        } else {
            const float* c = idx->centroids + key * (int64_t)d;
            for (int t = 0; t < d; t++) {
                st->resid[t] = q[t] - c[t];
            }
            y = st->resid;
        }
        for (int64_t j = 0; j < len; j++) {
            if (filtered_out(bitset, nbits, ids[j])) {
                continue;
            }
            float dis = sq8_distance(idx->metric, d, idx->sq_trained, y, codes + j * idx->code_size);
            if (idx->metric == ORC_IP) {
                dis = accu0 + dis;
            }
            sink_add(sk, dis, ids[j]);
        }
    }
}

int orc_ivf_search_preassigned(const orc_index* idx, int64_t nq, const float* xq, int64_t k,
                               int64_t nprobe, const int64_t* keys, const float* coarse_dis,
                               const uint8_t* bitset, int64_t nbits, float* D, int64_t* I) {
    const int is_max = (idx->metric == ORC_L2);
    const int d = idx->d;
    orc_scan_state st;
    scan_state_init(idx, &st);
    for (int64_t i = 0; i < nq; i++) {
        const float* q = xq + i * (int64_t)d;
        orc_sink sk;
        memset(&sk, 0, sizeof(sk));
        sk.is_max = is_max;
        sk.k = (size_t)k;
        sk.val = D + i * k;
        sk.ids = I + i * k;
        orc_heap_heapify(is_max, (size_t)k, sk.val, sk.ids);
        scan_set_query(idx, &st, q);
        for (int64_t ik = 0; ik < nprobe; ik++) {
            const int64_t key = keys[i * nprobe + ik];
            if (key < 0 || idx->list_sizes[key] == 0) {
                continue;
            }
            scan_one_list(idx, &st, q, key, coarse_dis[i * nprobe + ik], bitset, nbits, &sk);
        }
        orc_heap_reorder(is_max, (size_t)k, sk.val, sk.ids);
    }
    scan_state_free(&st);
    return 0;
}

int orc_ivf_search(const orc_index* idx, int64_t nq, const float* xq, int64_t k, int64_t nprobe,
                   const uint8_t* bitset, int64_t nbits, float* D, int64_t* I) {
    if (nprobe > idx->nlist) {
        nprobe = idx->nlist; /* T:IndexIVF.cpp:321-322 */
    }
    float* cd = (float*)malloc(sizeof(float) * (size_t)(nq * nprobe));
    int64_t* keys = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nq * nprobe));
    orc_coarse_search(idx, nq, xq, nprobe, cd, keys);
    int rc = orc_ivf_search_preassigned(idx, nq, xq, k, nprobe, keys, cd, bitset, nbits, D, I);
    free(cd);
    free(keys);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Range search.
 *   IVF: IvfIndexNode::RangeSearch (reference src/index/ivf/ivf.cc:1231-1420) drives
 *   index->range_search(1, q, radius, res, params{nprobe = nlist, max_empty_result_buckets, sel}) ->
 *   T:IndexIVF.cpp:774-990 range_search_preassigned, parallel_mode 0: lists in coarse order, each
 *   scanned in storage order; after a list the counter of consecutive lists that added nothing is
 *   bumped or reset and the loop stops when it reaches max_empty_result_buckets (> 0); empty lists
 *   and key < 0 count as "added nothing".
 *   FLAT: IndexFlat::range_search -> T:utils/distances.cpp range_search_L2sqr / _inner_product:
 *   every unfiltered row strictly inside the radius, in storage order.
 * Results: lims[nq + 1], and malloc'ed ids / distances (release with orc_free).
 * ---------------------------------------------------------------------------------------- */
static void range_flush(orc_sink* sk, int64_t i, int64_t* lims, int64_t** out_ids, float** out_dis, size_t* total,
                        size_t* cap) {
    if (*total + sk->n > *cap) {
        *cap = (*total + sk->n) * 2 + 256;
        *out_ids = (int64_t*)realloc(*out_ids, sizeof(int64_t) * *cap);
        *out_dis = (float*)realloc(*out_dis, sizeof(float) * *cap);
    }
    if (sk->n) {
        memcpy(*out_ids + *total, sk->rids, sizeof(int64_t) * sk->n);
        memcpy(*out_dis + *total, sk->rdis, sizeof(float) * sk->n);
    }
    *total += sk->n;
    lims[i + 1] = (int64_t)*total;
    free(sk->rids);
    free(sk->rdis);
}

int orc_ivf_range_search(const orc_index* idx, int64_t nq, const float* xq, float radius,
                         int64_t max_empty_result_buckets, const uint8_t* bitset, int64_t nbits, int64_t* lims,
                         int64_t** out_ids, float** out_dis) {
    const int is_max = (idx->metric == ORC_L2);
    const int d = idx->d;
    const int64_t nprobe = idx->nlist; /* ivf.cc:1290 / :1391: every list is a candidate */
    float* cd = (float*)malloc(sizeof(float) * (size_t)nprobe);
    int64_t* keys = (int64_t*)malloc(sizeof(int64_t) * (size_t)nprobe);
    orc_scan_state st;
    scan_state_init(idx, &st);
    size_t total = 0, cap = 0;
    *out_ids = NULL;
    *out_dis = NULL;
    lims[0] = 0;
    for (int64_t i = 0; i < nq; i++) {
        const float* q = xq + i * (int64_t)d;
        orc_coarse_search(idx, 1, q, nprobe, cd, keys);
        orc_sink sk;
        memset(&sk, 0, sizeof(sk));
        sk.is_max = is_max;
        sk.radius = radius;
        scan_set_query(idx, &st, q);
        size_t prev = 0;
        int64_t ndup = 0;
        for (int64_t ik = 0; ik < nprobe; ik++) {
            const int64_t key = keys[ik];
            if (key >= 0 && idx->list_sizes[key] != 0) {
                scan_one_list(idx, &st, q, key, cd[ik], bitset, nbits, &sk);
            }
            if (max_empty_result_buckets > 0) {
                ndup = (sk.n == prev) ? ndup + 1 : 0;
                if (ndup >= max_empty_result_buckets) {
                    break;
                }
                prev = sk.n;
            }
        }
        range_flush(&sk, i, lims, out_ids, out_dis, &total, &cap);
    }
    scan_state_free(&st);
    free(cd);
    free(keys);
    return 0;
}

int orc_flat_range_search(int metric, int d, int64_t nb, const float* xb, int64_t nq, const float* xq, float radius,
                          const uint8_t* bitset, int64_t nbits, int64_t* lims, int64_t** out_ids, float** out_dis) {
    const int is_max = (metric == ORC_L2);
    size_t total = 0, cap = 0;
    *out_ids = NULL;
    *out_dis = NULL;
    lims[0] = 0;
    for (int64_t i = 0; i < nq; i++) {
        const float* x = xq + i * (int64_t)d;
        orc_sink sk;
        memset(&sk, 0, sizeof(sk));
        sk.is_max = is_max;
        sk.radius = radius;
        for (int64_t j = 0; j < nb; j++) {
            if (filtered_out(bitset, nbits, j)) {
                continue;
            }
            const float* y = xb + j * (int64_t)d;
            sink_add(&sk, is_max ? orc_fvec_L2sqr(x, y, (size_t)d) : orc_fvec_inner_product(x, y, (size_t)d), j);
        }
        range_flush(&sk, i, lims, out_ids, out_dis, &total, &cap);
    }
    return 0;
}

void orc_free(void* p) {
    free(p);
}

/* ------------------------------------------------------------------------------------------
 * Refine: T:IndexRefine.cpp:108-140.  dc(idx) of an IndexFlat is fvec_L2sqr / fvec_inner_product
 * (T:IndexFlat.cpp FlatL2Dis / FlatIPDis); reorder_2_heaps = heapify(k) + heap_addn(all k_base,
 * labels -1 included: their stale distances are never admitted because... they keep the
 * base-stage value, so restate literally) + heap_reorder  (T:utils/Heap.h reorder_2_heaps).
 * ---------------------------------------------------------------------------------------- */
int orc_refine(int metric, int d, const float* base, int64_t nbase, int64_t id_base, int64_t nq,
               const float* xq, int64_t k_base, const int64_t* cand_ids, int64_t k, float* D, int64_t* I) {
    const int is_max = (metric == ORC_L2);
    for (int64_t i = 0; i < nq; i++) {
        const float* q = xq + i * (int64_t)d;
        float* simi = D + i * k;
        int64_t* idxi = I + i * k;
        orc_heap_heapify(is_max, (size_t)k, simi, idxi);
        for (int64_t j = 0; j < k_base; j++) {
            const int64_t id = cand_ids[i * k_base + j];
            if (id == -1) {
                break; /* entries after the first -1 are all -1 (sentinel tail): nothing to add */
            }
            if (id - id_base < 0 || id - id_base >= nbase) {
                continue; /* (also any other negative id: a candidate another shard re-ranks) */
            }
            const float* y = base + (id - id_base) * (int64_t)d;
            const float dis = is_max ? orc_fvec_L2sqr(q, y, (size_t)d)
                                     : orc_fvec_inner_product(q, y, (size_t)d);
            heap_add(is_max, (size_t)k, simi, idxi, dis, id);
        }
        orc_heap_reorder(is_max, (size_t)k, simi, idxi);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Quantised refine store: Knowhere's refine_type = fp16 / bf16 / sq8 makes the refine index a
 * faiss::IndexScalarQuantizer (R:src/index/refine/refine_utils.cc:150-185).  Row types:
 * 1 QT_fp16, 2 QT_bf16, 3 QT_8bit.  Restated from the SIMDLevel::NONE classes:
 *   train   T:impl/scalar_quantizer/training.cpp train_NonUniform, RS_minmax, rangestat_arg 0
 *           (vmin = column minimum, vdiff = column maximum - vmin, over all rows)
 *   encode  T:impl/scalar_quantizer/quantizers.h:108-127 + codecs.h Codec8bit::encode_component
 *           (int)(255 * x); QuantizerFP16 -> encode_fp16 (fp16-inl.h: ties round UP, see orc_encode_fp16),
 *           QuantizerBF16 -> encode_bf16 = (bits + 0x8000) >> 16 (T:utils/bf16.h:28-33)
 *   decode  reconstruct_component: vmin[i] + ((code + 0.5f) / 255.0f) * vdiff[i]
 *   dis     DCTemplate<Quantizer, Similarity, NONE>::compute_distance: one accumulator, i ascending,
 *           L2: tmp = q[i] - x_i, accu += tmp * tmp; IP: accu += q[i] * x_i
 *           (T:impl/scalar_quantizer/similarities.h, distance_computers.h)
 * ---------------------------------------------------------------------------------------- */
static uint16_t orc_encode_fp16(float f) {
    /* T:utils/fp16-inl.h:32-86 (the build without F16C, which is what SIMDLevel::NONE code sees): keep 11 mantissa bits,
     * rescale by 2^-112 in fp32 (half subnormals come out as fp32 subnormals), add half a unit, take bits 13.. -- i.e.
     * round HALF UP after truncating to 11 bits, NOT round-to-nearest-even: exact ties go up.  Restated literally. */
    uint32_t fint, sign;
    memcpy(&fint, &f, 4);
    sign = fint & 0x80000000u;
    fint ^= sign;
    const uint32_t f32infty = 255u << 23, round_mask = ~0xfffu, magic = 15u << 23, capb = (31u << 23) - 0x1000u;
    int32_t o = (fint > f32infty) ? 0x7e00 : 0x7c00;
    const uint32_t tb = fint & round_mask;
    float t, mg, cap;
    memcpy(&t, &tb, 4);
    memcpy(&mg, &magic, 4);
    memcpy(&cap, &capb, 4);
    volatile float fscale = t * mg; /* (volatile: one rounding to fp32, subnormals kept) */
    float fs = fscale;
    if (cap < fs) {
        fs = cap;
    }
    uint32_t fb;
    memcpy(&fb, &fs, 4);
    const int32_t fint2 = (int32_t)(fb - round_mask);
    if (fint < f32infty) {
        o = fint2 >> 13;
    }
    return (uint16_t)(o | (sign >> 16));
}

static float orc_decode_fp16(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) {
            x = sign;
        } else {
            const float v = (float)m * 5.9604644775390625e-08f; /* m * 2^-24, exact */
            memcpy(&x, &v, 4);
            x |= sign;
        }
    } else if (e == 31) {
        x = sign | 0x7f800000u | (m << 13);
    } else {
        x = sign | ((e + 112u) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &x, 4);
    return f;
}

/* row types: 1 fp16, 2 bf16, 3 QT_8bit, 4 QT_6bit, 5 QT_8bit_direct_signed, 6 QT_4bit_uniform (ScalarQuantizer::set_derived_sizes,
 * T:impl/ScalarQuantizer.cpp: code_size = d for the 8-bit types, (d * 6 + 7) / 8 for QT_6bit, 2 d for the 16-bit ones) */
int64_t orc_rows_code_size(int row_type, int d) {
    if (row_type == 4) {
        return ((int64_t)d * 6 + 7) / 8;
    }
    if (row_type == 6) { /* QT_4bit_uniform: two codes per byte */
        return ((int64_t)d * 4 + 7) / 8;
    }
    return (row_type == 3 || row_type == 5) ? d : 2 * (int64_t)d;
}

void orc_rows_train(int d, int64_t n, const float* x, float* trained) {
    for (int j = 0; j < d; j++) {
        float lo = HUGE_VALF, hi = -HUGE_VALF;
        for (int64_t i = 0; i < n; i++) {
            const float v = x[i * d + j];
            if (v < lo) lo = v;
            if (v > hi) hi = v;
        }
        const float vexp = (hi - lo) * 0.0f; /* rangestat_arg */
        lo -= vexp;
        hi += vexp;
        trained[j] = lo;
        trained[d + j] = hi - lo;
    }
}

static int cmp_f32(const void* a, const void* b) {
    const float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

/* QT_4bit_uniform (row type 6): ONE range for all dimensions, train_Uniform over the n * d values
 * (T:impl/scalar_quantizer/training.cpp:209-332).  rangestat 0 = RS_minmax with rangestat_arg (the ScalarQuantizer default,
 * what Knowhere keeps for the inner product), 2 = RS_quantiles (Knowhere sets it with arg 0.01 for L2,
 * src/index/refine/refine_utils.cc:176-180): o = (idx_t)(rs_arg * n) -- a FLOAT product of the float argument and the count
 * --, vmin = the o-th smallest value, vmax = the (n - 1 - o)-th.  trained = {vmin, vmax - vmin}. */
void orc_rows_train_uniform(int rangestat, float rs_arg, int64_t n_values, const float* x, float* trained) {
    float vmin, vmax;
    if (rangestat == 2) {
        float* c = (float*)malloc((size_t)n_values * sizeof(float));
        memcpy(c, x, (size_t)n_values * sizeof(float));
        qsort(c, (size_t)n_values, sizeof(float), cmp_f32);
        int64_t o = (int64_t)(rs_arg * n_values);
        if (o < 0) o = 0;
        if (o > n_values - o) o = n_values / 2;
        vmin = c[o];
        vmax = c[n_values - 1 - o];
        free(c);
    } else {
        vmin = HUGE_VALF;
        vmax = -HUGE_VALF;
        for (int64_t i = 0; i < n_values; i++) {
            if (x[i] < vmin) vmin = x[i];
            if (x[i] > vmax) vmax = x[i];
        }
        const float vexp = (vmax - vmin) * rs_arg;
        vmin -= vexp;
        vmax += vexp;
    }
    vmax -= vmin;
    trained[0] = vmin;
    trained[1] = vmax;
}

void orc_rows_encode(int row_type, int d, int64_t n, const float* x, const float* trained, uint8_t* codes) {
    for (int64_t r = 0; r < n; r++) {
        const float* xr = x + r * d;
        if (row_type == 3) {
            uint8_t* c = codes + r * d;
            for (int i = 0; i < d; i++) {
                float xi = 0;
                if (trained[d + i] != 0) {
                    xi = (xr[i] - trained[i]) / trained[d + i];
                    if (xi < 0) xi = 0;
                    if (xi > 1.0) xi = 1.0;
                }
                c[i] = (uint8_t)(int)(255 * xi);
            }
        } else if (row_type == 4) {
            /* QuantizerTemplate<Codec6bit, NON_UNIFORM>::encode_vector (T:impl/scalar_quantizer/quantizers.h:124-137) over
             * Codec6bit::encode_component (codecs.h:66-90): the codes are OR-ed into a zeroed row, four 6-bit values per
             * three bytes; `x * 63.0` is a DOUBLE product (exact for a float in [0, 1]) truncated to int */
            const int64_t cs = orc_rows_code_size(4, d);
            uint8_t* c = codes + r * cs;
            memset(c, 0, (size_t)cs);
            for (int i = 0; i < d; i++) {
                float xi = 0;
                if (trained[d + i] != 0) {
                    xi = (xr[i] - trained[i]) / trained[d + i];
                    if (xi < 0) xi = 0;
                    if (xi > 1.0) xi = 1.0;
                }
                const int bits = (int)(xi * 63.0);
                uint8_t* g = c + (i >> 2) * 3;
                switch (i & 3) {
                    case 0: g[0] |= (uint8_t)bits; break;
                    case 1: g[0] |= (uint8_t)(bits << 6); g[1] |= (uint8_t)(bits >> 2); break;
                    case 2: g[1] |= (uint8_t)(bits << 4); g[2] |= (uint8_t)(bits >> 4); break;
                    default: g[2] |= (uint8_t)(bits << 2); break;
                }
            }
        } else if (row_type == 6) {
            /* QuantizerTemplate<Codec4bit, UNIFORM>::encode_vector (quantizers.h:76-90) over Codec4bit::encode_component
             * (codecs.h:47-52): code[i / 2] |= (int)(xi * 15.0) << ((i & 1) << 2), a DOUBLE product, into a zeroed row */
            const int64_t cs = orc_rows_code_size(6, d);
            uint8_t* c = codes + r * cs;
            memset(c, 0, (size_t)cs);
            const float vmin = trained[0], vdiff = trained[1];
            for (int i = 0; i < d; i++) {
                float xi = 0;
                if (vdiff != 0) {
                    xi = (xr[i] - vmin) / vdiff;
                    if (xi < 0) xi = 0;
                    if (xi > 1.0) xi = 1.0;
                }
                c[i / 2] |= (uint8_t)((int)(xi * 15.0) << ((i & 1) << 2));
            }
        } else if (row_type == 5) {
            /* Quantizer8bitDirectSigned::encode_vector (quantizers.h:362-366): code = (uint8_t)(x + 128) -- defined for
             * values in [-128, 127] (Knowhere's int8 data format); the conversion truncates */
            uint8_t* c = codes + r * d;
            for (int i = 0; i < d; i++) {
                c[i] = (uint8_t)(int)(xr[i] + 128);
            }
        } else {
            uint16_t* c = (uint16_t*)(codes + r * 2 * (int64_t)d);
            for (int i = 0; i < d; i++) {
                if (row_type == 1) {
                    c[i] = orc_encode_fp16(xr[i]);
                } else {
                    uint32_t b;
                    memcpy(&b, &xr[i], 4);
                    c[i] = (uint16_t)((b + 0x8000u) >> 16);
                }
            }
        }
    }
}

static float rows_component(int row_type, int d, const uint8_t* code, const float* trained, int i) {
    if (row_type == 3) {
        const float xi = (code[i] + 0.5f) / 255.0f;
        return trained[i] + xi * trained[d + i];
    }
    if (row_type == 4) { /* Codec6bit::decode_component (codecs.h:92-118) inside QuantizerTemplate::reconstruct_component */
        const uint8_t* g = code + (i >> 2) * 3;
        uint8_t bits;
        switch (i & 3) {
            case 0: bits = g[0] & 0x3f; break;
            case 1: bits = (uint8_t)(g[0] >> 6); bits |= (uint8_t)((g[1] & 0xf) << 2); break;
            case 2: bits = (uint8_t)(g[1] >> 4); bits |= (uint8_t)((g[2] & 3) << 4); break;
            default: bits = (uint8_t)(g[2] >> 2); break;
        }
        const float xi = (bits + 0.5f) / 63.0f;
        return trained[i] + xi * trained[d + i];
    }
    if (row_type == 6) { /* Codec4bit::decode_component (codecs.h:54-58) inside the UNIFORM reconstruct_component */
        const float xi = (((code[i / 2] >> ((i & 1) << 2)) & 0xf) + 0.5f) / 15.0f;
        return trained[0] + xi * trained[1];
    }
    if (row_type == 5) { /* Quantizer8bitDirectSigned::reconstruct_component (quantizers.h:374-378) */
        return (float)(code[i] - 128);
    }
    const uint16_t v = ((const uint16_t*)code)[i];
    if (row_type == 1) {
        return orc_decode_fp16(v);
    }
    const uint32_t b = (uint32_t)v << 16;
    float f;
    memcpy(&f, &b, 4);
    return f;
}

void orc_rows_decode(int row_type, int d, int64_t n, const uint8_t* codes, const float* trained, float* x) {
    const int64_t cs = orc_rows_code_size(row_type, d);
    for (int64_t r = 0; r < n; r++) {
        for (int i = 0; i < d; i++) {
            x[r * d + i] = rows_component(row_type, d, codes + r * cs, trained, i);
        }
    }
}

/* IndexRefine::search over an IndexScalarQuantizer refine index (candidates as orc_refine) */
int orc_refine_rows(int metric, int d, int row_type, const uint8_t* codes, const float* trained, int64_t nbase, int64_t nq,
                    const float* xq, int64_t k_base, const int64_t* cand_ids, int64_t k, float* D, int64_t* I) {
    const int is_max = (metric == ORC_L2);
    const int64_t cs = orc_rows_code_size(row_type, d);
    for (int64_t i = 0; i < nq; i++) {
        const float* q = xq + i * (int64_t)d;
        float* simi = D + i * k;
        int64_t* idxi = I + i * k;
        orc_heap_heapify(is_max, (size_t)k, simi, idxi);
        for (int64_t j = 0; j < k_base; j++) {
            const int64_t id = cand_ids[i * k_base + j];
            if (id == -1) {
                break;
            }
            if (id < 0 || id >= nbase) {
                continue;
            }
            const uint8_t* code = codes + id * cs;
            float accu = 0;
            for (int c = 0; c < d; c++) {
                const float xi = rows_component(row_type, d, code, trained, c);
                if (is_max) {
                    const float tmp = q[c] - xi;
                    accu += tmp * tmp;
                } else {
                    accu += q[c] * xi;
                }
            }
            heap_add(is_max, (size_t)k, simi, idxi, accu, id);
        }
        orc_heap_reorder(is_max, (size_t)k, simi, idxi);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Shard merge.  The k best of the union of per-shard results; the reference proves the
 * property in tests/ut/test_bruteforce.cc:128-181 (heaps.addn_with_ids over partitions) and
 * T:utils/Heap.h:636 merge_knn_results.  Restated with the same heap.
 * ---------------------------------------------------------------------------------------- */
int orc_merge_topk(int metric, int64_t nq, int64_t k, int nshard, const float* D_parts,
                   const int64_t* I_parts, float* D, int64_t* I) {
    const int is_max = (metric == ORC_L2);
    for (int64_t i = 0; i < nq; i++) {
        float* simi = D + i * k;
        int64_t* idxi = I + i * k;
        orc_heap_heapify(is_max, (size_t)k, simi, idxi);
        for (int s = 0; s < nshard; s++) {
            const float* dp = D_parts + ((int64_t)s * nq + i) * k;
            const int64_t* ip = I_parts + ((int64_t)s * nq + i) * k;
            for (int64_t j = 0; j < k; j++) {
                if (ip[j] < 0) {
                    continue;
                }
                heap_add(is_max, (size_t)k, simi, idxi, dp[j], ip[j]);
            }
        }
        orc_heap_reorder(is_max, (size_t)k, simi, idxi);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Build-side helpers (restated add path), used to make test indexes on boxes without _ref.
 * ---------------------------------------------------------------------------------------- */
/* IndexFlat::assign == search with k=1: first strict improvement wins */
void orc_assign_dis(int metric, int d, int64_t nlist, const float* centroids, int64_t n, const float* x,
                    int64_t* out, float* out_dis) {
    for (int64_t i = 0; i < n; i++) {
        const float* xi = x + i * (int64_t)d;
        int64_t best = -1;
        float bd = (metric == ORC_L2) ? FLT_MAX : -FLT_MAX;
        for (int64_t j = 0; j < nlist; j++) {
            const float* c = centroids + j * (int64_t)d;
            if (metric == ORC_L2) {
                float dis = orc_fvec_L2sqr(xi, c, (size_t)d);
                if (bd > dis) {
                    bd = dis;
                    best = j;
                }
            } else {
                float dis = orc_fvec_inner_product(xi, c, (size_t)d);
                if (bd < dis) {
                    bd = dis;
                    best = j;
                }
            }
        }
        out[i] = best;
        if (out_dis) {
            out_dis[i] = bd;
        }
    }
}

void orc_assign(int metric, int d, int64_t nlist, const float* centroids, int64_t n, const float* x,
                int64_t* out) {
    orc_assign_dis(metric, d, nlist, centroids, n, x, out, NULL);
}

/* T:impl/ProductQuantizer.cpp:250-280 compute_1_code (nbits = 8): per sub-vector the nearest
 * codeword, first minimum wins (reference src/simd/distances_ref.cc:100-117). */
void orc_pq_compute_code(int d, int M, int nbits, const float* cb, const float* x, uint8_t* code) {
    const size_t ksub = (size_t)1 << nbits, dsub = (size_t)d / (size_t)M;
    for (int m = 0; m < M; m++) {
        const float* xs = x + m * dsub;
        size_t best = 0;
        float bd = HUGE_VALF;
        for (size_t j = 0; j < ksub; j++) {
            float dis = orc_fvec_L2sqr(xs, cb + (m * ksub + j) * dsub, dsub);
            if (dis < bd) {
                bd = dis;
                best = j;
            }
        }
        code[m] = (uint8_t)best;
    }
    if (nbits != 8) { /* (the caller's buffer holds max(M, code size) bytes) */
        uint8_t tmp[4096];
        if (M <= 4096) {
            memcpy(tmp, code, (size_t)M);
            orc_pq_pack(M, nbits, tmp, code);
        }
    }
}

/* T:impl/scalar_quantizer/quantizers.h:118-133 + codecs.h:29-35 */
void orc_sq8_encode(int d, const float* trained, const float* x, uint8_t* code) {
    const float* vmin = trained;
    const float* vdiff = trained + d;
    for (int i = 0; i < d; i++) {
        float xi = 0;
        if (vdiff[i] != 0) {
            xi = (x[i] - vmin[i]) / vdiff[i];
            if (xi < 0) {
                xi = 0;
            }
            if (xi > 1.0) {
                xi = 1.0;
            }
        }
        code[i] = (uint8_t)(int)(255 * xi);
    }
}

void orc_sq8_decode(int d, const float* trained, const uint8_t* code, float* x) {
    for (int i = 0; i < d; i++) {
        x[i] = sq8_component(trained, d, code, i);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Training restatements (pinned against knhip_index_train* / knhip_kmeans_device on the GPU)
 * ---------------------------------------------------------------------------------------------- */

/* std::mt19937 (the generator behind faiss::RandomGenerator, T:utils/random.cpp:35-55) */
typedef struct {
    uint32_t s[624];
    int i;
} orc_mt;
static void mt_seed(orc_mt* m, uint32_t seed) {
    m->s[0] = seed;
    for (int i = 1; i < 624; i++) {
        m->s[i] = 1812433253u * (m->s[i - 1] ^ (m->s[i - 1] >> 30)) + (uint32_t)i;
    }
    m->i = 624;
}
static uint32_t mt_next(orc_mt* m) {
    if (m->i >= 624) {
        for (int k = 0; k < 624; k++) {
            const uint32_t y = (m->s[k] & 0x80000000u) | (m->s[(k + 1) % 624] & 0x7fffffffu);
            m->s[k] = m->s[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        m->i = 0;
    }
    uint32_t y = m->s[m->i++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* T:utils/random.cpp:188-199 rand_perm */
void orc_rand_perm(int64_t* perm, int64_t n, int64_t seed) {
    for (int64_t i = 0; i < n; i++) {
        perm[i] = i;
    }
    orc_mt mt;
    mt_seed(&mt, (uint32_t)seed);
    for (int64_t i = 0; i + 1 < n; i++) {
        const int64_t i2 = i + (int64_t)(mt_next(&mt) % (uint32_t)(n - i));
        const int64_t t = perm[i];
        perm[i] = perm[i2];
        perm[i2] = t;
    }
}

/* T:Clustering.cpp:95-380 Clustering::train_encoded (nredo 1, RANDOM init, no weights, no early stop) with an exact
 * k = 1 search as the assigner; T:impl/ClusteringHelpers.cpp:36-240 subsample_training_set / compute_centroids /
 * split_clusters.  x: n rows of leading dimension ld, the clustered vector = columns [off, off + d). */
/* T:utils/distances.cpp:238-275 fvec_renorm_L2: rows of non-zero norm scaled by (float)(1.0 / sqrtf(norm2)) */
void orc_renorm_L2(int d, int64_t n, float* x) {
    for (int64_t i = 0; i < n; i++) {
        float* xi = x + i * d;
        const float nr = orc_fvec_norm_L2sqr(xi, (size_t)d);
        if (nr > 0) {
            const float inv_nr = (float)(1.0 / sqrtf(nr));
            for (int j = 0; j < d; j++) {
                xi[j] *= inv_nr;
            }
        }
    }
}

/* spherical != 0: Clustering::post_process_centroids (T:Clustering.cpp:35-38) renormalises the centroids after the
 * initialisation (:251) and after every update (:347) -- what IndexIVF switches on for the inner product
 * (T:IndexIVF.cpp:178-181).  The iteration loop ends early when the objective (sum of the assignment distances, float,
 * in row order) repeats bit for bit (early_stop_threshold 0: T:Clustering.cpp:362-377, src/index/clustering_config.h:34). */
void orc_kmeans(int metric, int d, int64_t n, const float* x, int64_t ld, int off, int64_t k, int niter,
                int max_points, int64_t seed, int spherical, float* centroids) {
    int64_t nx = n;
    int64_t* perm = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    if (n > k * (int64_t)max_points) {
        orc_rand_perm(perm, n, seed);
        nx = k * (int64_t)max_points;
    } else {
        for (int64_t i = 0; i < n; i++) {
            perm[i] = i;
        }
    }
    float* xs = (float*)malloc(sizeof(float) * (size_t)nx * (size_t)d);
    for (int64_t i = 0; i < nx; i++) {
        memcpy(xs + i * d, x + perm[i] * ld + off, sizeof(float) * (size_t)d);
    }
    if (nx == k) {
        memcpy(centroids, xs, sizeof(float) * (size_t)k * (size_t)d);
        free(xs);
        free(perm);
        return;
    }
    int64_t* p2 = (int64_t*)malloc(sizeof(int64_t) * (size_t)nx);
    orc_rand_perm(p2, nx, seed + 1);
    for (int64_t i = 0; i < k; i++) {
        memcpy(centroids + i * d, xs + p2[i] * d, sizeof(float) * (size_t)d);
    }
    free(p2);
    if (spherical) {
        orc_renorm_L2(d, k, centroids);
    }
    int64_t* assign = (int64_t*)malloc(sizeof(int64_t) * (size_t)nx);
    float* adis = (float*)malloc(sizeof(float) * (size_t)nx);
    float* hassign = (float*)malloc(sizeof(float) * (size_t)k);
    float prev_obj = 0;
    for (int it = 0; it < niter; it++) {
        orc_assign_dis(metric, d, k, centroids, nx, xs, assign, adis);
        float obj = 0;
        for (int64_t i = 0; i < nx; i++) {
            obj += adis[i];
        }
        /* compute_centroids: members summed in index order, scaled by 1 / count */
        memset(centroids, 0, sizeof(float) * (size_t)k * (size_t)d);
        memset(hassign, 0, sizeof(float) * (size_t)k);
        for (int64_t i = 0; i < nx; i++) {
            float* c = centroids + assign[i] * d;
            hassign[assign[i]] += 1.0f;
            for (int j = 0; j < d; j++) {
                c[j] += xs[i * d + j];
            }
        }
        int any_empty = 0;
        for (int64_t ci = 0; ci < k; ci++) {
            if (hassign[ci] == 0) {
                any_empty = 1;
                continue;
            }
            const float norm = 1 / hassign[ci];
            for (int j = 0; j < d; j++) {
                centroids[ci * d + j] *= norm;
            }
        }
        if (any_empty) { /* split_clusters */
            const float EPS = 1.f / 1024.f;
            orc_mt mt;
            mt_seed(&mt, 1234u);
            for (int64_t ci = 0; ci < k; ci++) {
                if (hassign[ci] != 0) {
                    continue;
                }
                int64_t cj = 0, n_tries = 0;
                const int64_t max_tries = 10 * k;
                int found = 0;
                for (cj = 0; n_tries < max_tries; cj = (cj + 1) % k) {
                    const float p = (float)((hassign[cj] - 1.0) / (float)(nx - k));
                    const float r = mt_next(&mt) / (float)4294967295u;
                    if (r < p) {
                        found = 1;
                        break;
                    }
                    n_tries++;
                }
                if (!found) {
                    cj = 0;
                    for (int64_t j = 1; j < k; j++) {
                        if (hassign[j] > hassign[cj]) {
                            cj = j;
                        }
                    }
                }
                memcpy(centroids + ci * d, centroids + cj * d, sizeof(float) * (size_t)d);
                for (int j = 0; j < d; j++) {
                    if (j % 2 == 0) {
                        centroids[ci * d + j] *= 1 + EPS;
                        centroids[cj * d + j] *= 1 - EPS;
                    } else {
                        centroids[ci * d + j] *= 1 - EPS;
                        centroids[cj * d + j] *= 1 + EPS;
                    }
                }
                hassign[ci] = hassign[cj] / 2;
                hassign[cj] -= hassign[ci];
            }
        }
        if (spherical) {
            orc_renorm_L2(d, k, centroids);
        }
        if (it > 0) {
            const double change = (prev_obj == 0) ? DBL_MAX : fabs((double)(prev_obj - obj)) / fabs((double)prev_obj);
            if (change <= 0.0) {
                break;
            }
        }
        prev_obj = obj;
    }
    free(hassign);
    free(adis);
    free(assign);
    free(xs);
    free(perm);
}

/* T:IndexIVF.cpp:1175-1270 IndexIVF::train for IVF_PQ / IVF_SQ8 / IVF_FLAT with by_residual: coarse k-means, encoder
 * sub-sample (first rows of rand_perm(n, 1234)), residuals, then T:impl/ProductQuantizer.cpp:130-215 (one k-means per
 * sub-space, 25 iterations, seed 1234) or RS_minmax ranges (trained = vmin[d], vdiff[d]).
 * kind: 1 IVF_FLAT, 2 IVF_PQ, 3 IVF_SQ8.  coarse_given != 0: centroids are an input. */
void orc_train_ivf_nbits(int kind, int metric, int d, int64_t nlist, int M, int nbits, int64_t n, const float* x, int niter,
                         int max_points, int64_t seed, int coarse_given, float* centroids, float* pq_centroids,
                         float* sq_trained);

void orc_train_ivf(int kind, int metric, int d, int64_t nlist, int M, int64_t n, const float* x, int niter,
                   int max_points, int64_t seed, int coarse_given, float* centroids, float* pq_centroids,
                   float* sq_trained) {
    orc_train_ivf_nbits(kind, metric, d, nlist, M, 8, n, x, niter, max_points, seed, coarse_given, centroids, pq_centroids,
                        sq_trained);
}

/* nbits: the PQ code width (ksub = 2^nbits codebook entries per sub-quantizer; the encoder trains on at most 256 ksub
 * points: T:IndexIVFPQ.cpp:97-99 train_encoder_num_vectors = max_points_per_centroid * ksub) */
void orc_train_ivf_nbits(int kind, int metric, int d, int64_t nlist, int M, int nbits, int64_t n, const float* x, int niter,
                         int max_points, int64_t seed, int coarse_given, float* centroids, float* pq_centroids,
                         float* sq_trained) {
    const int64_t ksub = (int64_t)1 << nbits;
    if (!coarse_given) {
        /* level-1 quantizer: cp.niter = 10 unless the caller overrides it (T:IndexIVF.cpp:44), spherical for the inner
         * product (T:IndexIVF.cpp:178-181) */
        orc_kmeans(metric, d, n, x, d, 0, nlist, niter > 0 ? niter : 10, max_points, seed, metric == ORC_IP, centroids);
    }
    if (kind == 1) {
        return;
    }
    const int64_t max_nt = kind == 2 ? 256 * ksub : 100000;
    int64_t nt = n;
    int64_t* perm = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    if (n > max_nt) {
        orc_rand_perm(perm, n, 1234);
        nt = max_nt;
    } else {
        for (int64_t i = 0; i < n; i++) {
            perm[i] = i;
        }
    }
    float* xt = (float*)malloc(sizeof(float) * (size_t)nt * (size_t)d);
    for (int64_t i = 0; i < nt; i++) {
        memcpy(xt + i * d, x + perm[i] * d, sizeof(float) * (size_t)d);
    }
    int64_t* assign = (int64_t*)malloc(sizeof(int64_t) * (size_t)nt);
    orc_assign(metric, d, nlist, centroids, nt, xt, assign);
    for (int64_t i = 0; i < nt; i++) {
        for (int j = 0; j < d; j++) {
            xt[i * d + j] = xt[i * d + j] - centroids[assign[i] * d + j];
        }
    }
    if (kind == 2) {
        const int dsub = d / M;
        for (int m = 0; m < M; m++) {
            orc_kmeans(ORC_L2, dsub, nt, xt, d, m * dsub, ksub, 25, 256, 1234, 0, pq_centroids + (size_t)m * (size_t)ksub * dsub);
        }
    } else {
        for (int j = 0; j < d; j++) {
            float lo = HUGE_VALF, hi = -HUGE_VALF;
            for (int64_t i = 0; i < nt; i++) {
                const float v = xt[i * d + j];
                if (v < lo) lo = v;
                if (v > hi) hi = v;
            }
            sq_trained[j] = lo;
            sq_trained[d + j] = hi - lo;
        }
    }
    free(assign);
    free(xt);
    free(perm);
}
