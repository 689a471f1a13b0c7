// knowhere_amd/csrc/pq_scan_any.hip -- exact IVF-PQ ADC scan for ANY number of 8-bit sub-quantizers.
//
// The reference's IVF_PQ accepts every m that divides dim (src/index/ivf/ivf_config.h:118, faiss::ProductQuantizer);
// the fast kernels of this backend (pq_filter.hip, pq_scan_q4.hip, pq_scan_v2.hip, pq_scan.hip) are built for
// m in {8, 16, 32, 64}.  This kernel serves every other width (m = 1 .. 128: 4, 12, 24, 48, 96, ...) with the same
// results as the reference:
//   * tables   precompute_list_tables_L2 / _IP and the residual tables (IVFPQ_QueryTables.cpp:110-230): the same
//              arithmetic as pq_scan.hip's LUT build and range.hip::pq_adc_dump_kernel
//   * distance PQCodeDistanceScalar::distance_single_code (pq_code_distance-inl.h:69-90): the m table values summed from 0
//              in m order, then dis0 + sum (IVFPQScanner_impl.h:109-181)
//   * top-k    canonical (distance, id) partial lists, merged by topk.hip::merge_partials -- boundary ties are then
//              resolved like for every other kernel (knhip_api.hip::search_batch_ties).  k <= 64: one list per wave built
//              by sorting (pq_scan_any_kernel, partial slot = 4 * probe rank + wave); k > 64: one list per workgroup by
//              block-wide selection (pq_scan_any_block_kernel)
// One workgroup per (query, probed list): the (query, list) table [m][256] fp32 in LDS (m KB; 160 KB LDS holds m = 128),
// one thread per stored vector reading its m code bytes from the list-sorted AoS codes.  Not a tuned kernel: LDS
// gathers with random bank conflicts, a table build per (query, list); it is the completeness path, the headline
// shapes never reach it.  (Round 4, 10M x 96, m = 24, nprobe 64, batch 10k: sequential top-k insertion took 109 ms at
// k = 101, of which tables + distances are 22; now 38 ms there and 30 ms at k = 11: DESIGN.md section 7.)
#include "common.h"
#include "kernels.h"

namespace knhip {

constexpr int PA_THREADS = 256;
constexpr int PA_WAVES = PA_THREADS / KN_WAVE;

int pq_scan_any_supports(int M, int d) {
    // LDS: the table (M KB here; 256 * dsub floats in the encoder, build.hip::launch_pq_encode)
    return M >= 1 && M <= 128 && d % M == 0 && (size_t)256 * (d / M) * sizeof(float) <= 144 * 1024;
}


// ---- wave-wide helpers of the k <= 64 path -----------------------------------------------------------------------------
// Sequential WaveTopK insertion costs a scalar round trip per candidate (readlane -> compare -> DPP shift): measured on
// this kernel 143 of its 165 ms per 10k x 64-probe batch.  So the list is built by SORTING instead: the lanes' minima
// bound the wave's k-th best, the few rows within the bound are gathered into the free lanes by shuffles and the 64 lanes
// are sorted once (bitonic network on __shfl_xor, no LDS); later super-chunks meet a full list and a tight k-th distance.
template <bool IS_L2>
__device__ __forceinline__ bool pa_pair_better(float ad, int64_t ai, float bd, int64_t bi) {
    if (ai < 0) {
        return false; // (an empty slot is never better)
    }
    if (bi < 0) {
        return true;
    }
    return better<IS_L2>(ad, ai, bd, bi);
}

// best-first over the 64 lanes, canonical (distance, id) order, empty slots last
template <bool IS_L2>
__device__ __forceinline__ void pa_wave_sort_pairs(float& d, int64_t& i) {
    const int lane = lane_id();
#pragma unroll
    for (int size = 2; size <= KN_WAVE; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const float od = __shfl_xor(d, stride, KN_WAVE);
            const int64_t oi = __shfl_xor(i, stride, KN_WAVE);
            const bool keep_better = ((lane & size) == 0) == ((lane & stride) == 0);
            const bool take = keep_better ? pa_pair_better<IS_L2>(od, oi, d, i) : pa_pair_better<IS_L2>(d, i, od, oi);
            if (take) {
                d = od;
                i = oi;
            }
        }
    }
}

// the same network on bare values (the lane minima): best-first
template <bool IS_L2>
__device__ __forceinline__ void pa_wave_sort_vals(float& v) {
    const int lane = lane_id();
#pragma unroll
    for (int size = 2; size <= KN_WAVE; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const float o = __shfl_xor(v, stride, KN_WAVE);
            const bool keep_better = ((lane & size) == 0) == ((lane & stride) == 0);
            const bool o_better = IS_L2 ? (o < v) : (o > v);
            const bool v_better = IS_L2 ? (v < o) : (v > o);
            if (keep_better ? o_better : v_better) {
                v = o;
            }
        }
    }
}

// position of the n-th (0-based) set bit of m; n < popcount(m)
__device__ __forceinline__ int pa_nth_set_bit(unsigned long long m, int n) {
    int pos = 0;
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) {
        const int c = __popcll(m & ((1ull << w) - 1ull));
        if (n >= c) {
            n -= c;
            m >>= w;
            pos += w;
        }
    }
    return pos;
}

constexpr int PA_S = 4; // rows per lane per super-chunk (k <= 64 path)

// ---- k > 64: block-wide selection ---------------------------------------------------------------------------------------
// A sorted list of 100 .. 1024 entries per wave costs ~3 k insertions per wave of 610 rows (measured: 109 ms for a 10k x
// 64-probe batch at k = 101, 27 k vector instructions per wave).  Here the workgroup selects instead: the rows of a tile
// (PB_T) become order-preserving 64-bit keys (distance key << 32 | position: the canonical (distance, id) order -- lists
// are stored in ascending id order), kept in registers; the k-th smallest key of the tile is found by bisection on the
// distance key (32 rounds of ballot counts; a second bisection on the position part only when the boundary is tied), the
// rows up to it are compacted into LDS next to the list carried over from the earlier tiles, and one bitonic sort in LDS
// leaves the new list.  ONE partial list per (query, probe).
constexpr int PB_T = 4096;                    // rows per tile
constexpr int PB_RPT = PB_T / PA_THREADS;     // ... per thread
constexpr unsigned long long PB_NONE = ~0ull; // an absent row / list entry

template <bool IS_L2>
__device__ __forceinline__ uint32_t pb_key(float f) {
    const uint32_t b = __float_as_uint(f);
    const uint32_t asc = (b & 0x80000000u) ? ~b : (b | 0x80000000u); // ascending in f
    return IS_L2 ? asc : ~asc;
}
template <bool IS_L2>
__device__ __forceinline__ float pb_unkey(uint32_t key) {
    const uint32_t asc = IS_L2 ? key : ~key;
    const uint32_t b = (asc & 0x80000000u) ? (asc & 0x7fffffffu) : ~asc;
    return __uint_as_float(b);
}

size_t pq_scan_any_block_smem(int M, int k) {
    int P = 2;
    while (P < 2 * k) {
        P <<= 1;
    }
    return (size_t)M * 256 * sizeof(float) + (size_t)P * 8 + 64;
}

// number of the workgroup's values v (PB_RPT per thread, PB_NONE = none) with pred(v), known to every thread.
// s_cnt: 2 x PA_WAVES counters in LDS, used alternately (`phase`) so that one barrier per call suffices.
template <class Pred>
__device__ __forceinline__ int pb_block_count(const unsigned long long (&v)[PB_RPT], Pred pred, int* s_cnt, int& phase) {
    int c = 0;
#pragma unroll
    for (int u = 0; u < PB_RPT; u++) {
        c += __popcll(__ballot(v[u] != PB_NONE && pred(v[u])));
    }
    int* slot = s_cnt + (phase & 1) * PA_WAVES;
    if (lane_id() == 0) {
        slot[threadIdx.x / KN_WAVE] = c;
    }
    __syncthreads();
    int tot = 0;
#pragma unroll
    for (int w = 0; w < PA_WAVES; w++) {
        tot += slot[w];
    }
    phase++;
    return tot;
}

template <bool IS_L2>
__global__ __launch_bounds__(PA_THREADS) void pq_scan_any_block_kernel(PqAnyArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* lut = reinterpret_cast<float*>(smem);                                             // [M][256]
    const int M = a.M, dsub = a.d / a.M, k = a.k;
    int P = 2; // entries sorted per tile: the carried list (<= k) + the tile's candidates (<= k), a power of two
    while (P < 2 * k) {
        P <<= 1;
    }
    unsigned long long* buf = reinterpret_cast<unsigned long long*>(smem + (size_t)M * 256 * sizeof(float)); // [P]
    int* s_cnt = reinterpret_cast<int*>(buf + P);                                            // [2][PA_WAVES] + cursor
    int* s_n = s_cnt + 2 * PA_WAVES;
    const int tid = threadIdx.x;
    const int64_t q = blockIdx.x / a.nprobe;
    const int slot = (int)(blockIdx.x % a.nprobe);
    float* pd = a.partial_d + (q * a.nprobe + slot) * (int64_t)k;
    int64_t* pi = a.partial_i + (q * a.nprobe + slot) * (int64_t)k;
    const int64_t list = a.keys[q * a.nprobe + slot];
    const int64_t len = (list >= 0 && list < a.nlist) ? a.list_len[list] : 0;
    if (len <= 0) {
        for (int e = tid; e < k; e += PA_THREADS) {
            pd[e] = worst_dist<IS_L2>();
            pi[e] = -1;
        }
        return;
    }
    if (a.lut_mode == PQ_LUT_RESIDUAL) { // ||(q - c_list)_m - cb[m][c]||^2
        for (int e = tid; e < M * 256; e += PA_THREADS) {
            const int m = e >> 8, c = e & 255;
            const float* y = a.cb + ((int64_t)m * 256 + c) * dsub;
            const float* x = a.queries + q * a.d + m * dsub;
            const float* cl = a.centroids + list * a.d + m * dsub;
            float t = 0.f;
            for (int i = 0; i < dsub; i++) {
                t = l2_step(t, fsub_x(x[i], cl[i]), y[i]);
            }
            lut[e] = t;
        }
    } else {
        const float* tq = a.t2t + q * 256 * M;
        const float* tp = a.lut_mode == PQ_LUT_PRECOMP ? a.precomp_t + list * 256 * M : nullptr;
        int c = tid / M, m = tid % M;
        const int dc = PA_THREADS / M, dm = PA_THREADS % M;
        for (int e = tid; e < M * 256; e += PA_THREADS) {
            float t = tq[e];
            if (tp != nullptr) {
                t = fadd_x(tp[e], fmul_x(-2.0f, t));
            }
            lut[m * 256 + c] = t;
            c += dc;
            m += dm;
            if (m >= M) {
                m -= M;
                c++;
            }
        }
    }
    for (int e = tid; e < P; e += PA_THREADS) {
        buf[e] = PB_NONE;
    }
    __syncthreads();
    const float dis0 = a.lut_mode == PQ_LUT_RESIDUAL ? 0.f : a.coarse_dis[q * a.nprobe + slot];
    const int64_t row_off = a.list_row_off[list];
    const bool words = (M & 3) == 0;
    int lcnt = 0;   // entries of the carried list: buf[0 .. lcnt), sorted
    int phase = 0;
    for (int64_t t0 = 0; t0 < len; t0 += PB_T) {
        // ---- the tile's rows -> keys in registers ----
        unsigned long long v[PB_RPT];
#pragma unroll
        for (int u = 0; u < PB_RPT; u++) {
            const int64_t pos = t0 + (int64_t)u * PA_THREADS + tid;
            v[u] = PB_NONE;
            if (pos < len) {
                const int64_t id = a.ids[row_off + pos];
                if (!bitset_filtered(a.bitset, a.bitset_nbits, id)) {
                    const uint8_t* code = a.codes + (row_off + pos) * M;
                    float acc = 0.f;
                    if (words) {
                        const uint32_t* cw = reinterpret_cast<const uint32_t*>(code);
                        for (int m = 0; m < M; m += 4) {
                            const uint32_t w = cw[m >> 2];
                            acc = fadd_x(acc, lut[(m + 0) * 256 + (w & 0xffu)]);
                            acc = fadd_x(acc, lut[(m + 1) * 256 + ((w >> 8) & 0xffu)]);
                            acc = fadd_x(acc, lut[(m + 2) * 256 + ((w >> 16) & 0xffu)]);
                            acc = fadd_x(acc, lut[(m + 3) * 256 + (w >> 24)]);
                        }
                    } else {
                        for (int m = 0; m < M; m++) {
                            acc = fadd_x(acc, lut[m * 256 + code[m]]);
                        }
                    }
                    const float dis = fadd_x(dis0, acc);
                    // (a distance equal to the neutral value is never admitted: ResultHandler.h:271-278)
                    if (IS_L2 ? dis < worst_dist<IS_L2>() : dis > worst_dist<IS_L2>()) {
                        const uint32_t tie = IS_L2 ? (uint32_t)pos : ~(uint32_t)pos;
                        v[u] = ((unsigned long long)pb_key<IS_L2>(dis) << 32) | tie;
                    }
                }
            }
        }
        // rows that cannot displace the carried list are out at once
        const unsigned long long kth_carried = lcnt == k ? buf[k - 1] : PB_NONE;
#pragma unroll
        for (int u = 0; u < PB_RPT; u++) {
            if (v[u] >= kth_carried) {
                v[u] = PB_NONE;
            }
        }
        int nvalid = pb_block_count(v, [](unsigned long long) { return true; }, s_cnt, phase);
        if (nvalid == 0) {
            continue;
        }
        // ---- the k-th smallest key of the tile (all of them when there are at most k) ----
        unsigned long long theta = PB_NONE; // rows <= theta are the tile's candidates
        if (nvalid > k) {
            uint32_t lo = 0u, hi = 0xffffffffu;
            while (lo < hi) { // smallest distance key with at least k rows at or below it
                const uint32_t mid = lo + ((hi - lo) >> 1);
                const int c = pb_block_count(v, [mid](unsigned long long x) { return (uint32_t)(x >> 32) <= mid; }, s_cnt, phase);
                if (c >= k) {
                    hi = mid;
                } else {
                    lo = mid + 1u;
                }
            }
            const uint32_t dk = lo;
            const int below = pb_block_count(v, [dk](unsigned long long x) { return (uint32_t)(x >> 32) < dk; }, s_cnt, phase);
            const int at = pb_block_count(v, [dk](unsigned long long x) { return (uint32_t)(x >> 32) == dk; }, s_cnt, phase);
            uint32_t tk = 0xffffffffu;
            if (below + at > k) { // the boundary is tied: the (k - below) first of the tied rows in canonical order
                const int need = k - below;
                uint32_t l2 = 0u, h2 = 0xffffffffu;
                while (l2 < h2) {
                    const uint32_t mid = l2 + ((h2 - l2) >> 1);
                    const int c = pb_block_count(
                            v, [dk, mid](unsigned long long x) { return (uint32_t)(x >> 32) == dk && (uint32_t)x <= mid; }, s_cnt,
                            phase);
                    if (c >= need) {
                        h2 = mid;
                    } else {
                        l2 = mid + 1u;
                    }
                }
                tk = l2;
            }
            theta = ((unsigned long long)dk << 32) | tk;
        }
        // ---- candidates -> buf[lcnt ..), then one sort of the P entries ----
        if (tid == 0) {
            *s_n = lcnt;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PB_RPT; u++) {
            if (v[u] <= theta) { // (PB_NONE <= theta only when theta == PB_NONE: excluded next)
                if (v[u] != PB_NONE) {
                    const int at = atomicAdd(s_n, 1);
                    buf[at] = v[u]; // (at < lcnt + k <= P)
                }
            }
        }
        __syncthreads();
        const int n = *s_n;
        for (int size = 2; size <= P; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int t = tid; t < P / 2; t += PA_THREADS) {
                    const int lo = (t / stride) * stride * 2 + (t % stride);
                    const int hi = lo + stride;
                    const bool up = (lo & size) == 0;
                    const unsigned long long x = buf[lo], y = buf[hi];
                    if ((x > y) == up) {
                        buf[lo] = y;
                        buf[hi] = x;
                    }
                }
                __syncthreads();
            }
        }
        lcnt = min(k, n);
        for (int e = lcnt + tid; e < P; e += PA_THREADS) { // what fell off the list
            buf[e] = PB_NONE;
        }
        __syncthreads();
    }
    for (int e = tid; e < k; e += PA_THREADS) {
        float od = worst_dist<IS_L2>();
        int64_t oi = -1;
        if (e < lcnt) {
            const unsigned long long x = buf[e];
            const uint32_t tie = (uint32_t)x;
            od = pb_unkey<IS_L2>((uint32_t)(x >> 32));
            oi = a.ids[row_off + (int64_t)(IS_L2 ? tie : ~tie)];
        }
        pd[e] = od;
        pi[e] = oi;
    }
}

template <bool IS_L2>
__global__ __launch_bounds__(PA_THREADS) void pq_scan_any_kernel(PqAnyArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* lut = reinterpret_cast<float*>(smem); // [M][256]
    const int lane = lane_id();
    const int wave = threadIdx.x / KN_WAVE;
    const int64_t q = blockIdx.x / a.nprobe;
    const int slot = (int)(blockIdx.x % a.nprobe);
    float* pd = a.partial_d + ((q * a.nprobe + slot) * PA_WAVES + wave) * (int64_t)a.k;
    int64_t* pi = a.partial_i + ((q * a.nprobe + slot) * PA_WAVES + wave) * (int64_t)a.k;
    WaveTopK<IS_L2, 1> top; // k <= 64: one list entry per lane
    top.init(a.k);
    const int64_t list = a.keys[q * a.nprobe + slot];
    const int64_t len = (list >= 0 && list < a.nlist) ? a.list_len[list] : 0;
    if (len <= 0) { // (uniform over the workgroup) an empty partial list
        top.store(pd, pi);
        return;
    }
    const int M = a.M, dsub = a.d / a.M;
    if (a.lut_mode == PQ_LUT_RESIDUAL) { // ||(q - c_list)_m - cb[m][c]||^2
        for (int e = threadIdx.x; e < M * 256; e += PA_THREADS) {
            const int m = e >> 8, c = e & 255;
            const float* y = a.cb + ((int64_t)m * 256 + c) * dsub;
            const float* x = a.queries + q * a.d + m * dsub;
            const float* cl = a.centroids + list * a.d + m * dsub;
            float t = 0.f;
            for (int i = 0; i < dsub; i++) {
                t = l2_step(t, fsub_x(x[i], cl[i]), y[i]);
            }
            lut[e] = t;
        }
    } else {
        // the source tables are [c][m]: read them in their own order (coalesced), scatter into the [m][256] LDS table
        const float* tq = a.t2t + q * 256 * M;                                              // <q_m, cb[m][c]>
        const float* tp = a.lut_mode == PQ_LUT_PRECOMP ? a.precomp_t + list * 256 * M : nullptr;
        int c = threadIdx.x / M, m = threadIdx.x % M;
        const int dc = PA_THREADS / M, dm = PA_THREADS % M; // (c, m) of entry e + 256 from those of e, without a division
        for (int e = threadIdx.x; e < M * 256; e += PA_THREADS) {
            float t = tq[e];
            if (tp != nullptr) {
                t = fadd_x(tp[e], fmul_x(-2.0f, t));
            }
            lut[m * 256 + c] = t;
            c += dc;
            m += dm;
            if (m >= M) {
                m -= M;
                c++;
            }
        }
    }
    __syncthreads();
    const float dis0 = a.lut_mode == PQ_LUT_RESIDUAL ? 0.f : a.coarse_dis[q * a.nprobe + slot];
    const int64_t row_off = a.list_row_off[list];
    float kd = worst_dist<IS_L2>();
    int64_t ki = -1;
    // (no shared per-query bound here -- common.h's gthr --: with one workgroup per (query, probe) the 4 x nprobe waves of a
    // query publish into one cache line at about the same time, and the same-line atomics serialise: measured +55 ms on the
    // 109 ms of a 10k x 64-probe batch)
    const bool words = (M & 3) == 0; // (rows of M bytes start 4-byte aligned)
    // distance of the stored vector at `pos` of the list (false: past the end or filtered)
    auto row_distance = [&](int64_t pos, float& dis, int64_t& id) -> bool {
        dis = 0.f;
        id = -1;
        if (pos >= len) {
            return false;
        }
        id = a.ids[row_off + pos];
        if (bitset_filtered(a.bitset, a.bitset_nbits, id)) {
            return false;
        }
        const uint8_t* code = a.codes + (row_off + pos) * M;
        float acc = 0.f;
        if (words) {
            const uint32_t* cw = reinterpret_cast<const uint32_t*>(code);
            for (int m = 0; m < M; m += 4) {
                const uint32_t w = cw[m >> 2];
                acc = fadd_x(acc, lut[(m + 0) * 256 + (w & 0xffu)]);
                acc = fadd_x(acc, lut[(m + 1) * 256 + ((w >> 8) & 0xffu)]);
                acc = fadd_x(acc, lut[(m + 2) * 256 + ((w >> 16) & 0xffu)]);
                acc = fadd_x(acc, lut[(m + 3) * 256 + (w >> 24)]);
            }
        } else {
            for (int m = 0; m < M; m++) {
                acc = fadd_x(acc, lut[m * 256 + code[m]]);
            }
        }
        dis = fadd_x(dis0, acc);
        return true;
    };
    // candidates of one row per lane through the sequential insertion (the stragglers of the sorted path)
    auto insert_passing = [&](bool pass, float dis, int64_t id) {
        unsigned long long mk = __ballot(pass && top.admits(dis, id, kd, ki));
        while (mk) {
            const int l = __ffsll((long long)mk) - 1;
            mk &= mk - 1;
            const float cd = readlane_f(dis, l);
            const int64_t ci = readlane_i64(id, l);
            if (top.admits(cd, ci, kd, ki)) {
                top.insert(cd, ci);
                kd = top.kth_dist();
                ki = top.kth_idx();
            }
        }
    };
    {
        // ---- one list entry per lane, built by sorting (see the helpers above) ----
        const int k = a.k;
        int cnt = 0; // valid entries of the list (wave-uniform)
        for (int64_t b0 = (int64_t)wave * KN_WAVE * PA_S; b0 < len; b0 += (int64_t)PA_THREADS * PA_S) {
            float dis[PA_S];
            int64_t id[PA_S];
            bool ok[PA_S];
#pragma unroll
            for (int u = 0; u < PA_S; u++) {
                ok[u] = row_distance(b0 + u * KN_WAVE + lane, dis[u], id[u]);
            }
            float bound = worst_dist<IS_L2>();
            if (cnt < k) {
                // the k-th best of the lanes' minima bounds the k-th best row of this super-chunk: k lanes hold a row at
                // least that good (fewer than k lanes with a row: the worst value, no pruning)
                float lm = worst_dist<IS_L2>();
#pragma unroll
                for (int u = 0; u < PA_S; u++) {
                    if (ok[u]) {
                        lm = tighter<IS_L2>(lm, dis[u]);
                    }
                }
                pa_wave_sort_vals<IS_L2>(lm);
                bound = tighter<IS_L2>(bound, readlane_f(lm, k - 1));
            }
            unsigned long long msk[PA_S];
            int C = 0;
#pragma unroll
            for (int u = 0; u < PA_S; u++) {
                msk[u] = __ballot(ok[u] && within_gthr<IS_L2>(dis[u], bound) && top.admits(dis[u], id[u], kd, ki));
                C += __popcll(msk[u]);
            }
            if (C == 0) {
                continue;
            }
            if (cnt + C <= KN_WAVE && (cnt < k || C > 2)) {
                // lanes [cnt, cnt + C) fetch the candidates (numbered row by row, lane by lane), then one sort
                const int j = lane - cnt;
                const bool sel = j >= 0 && j < C;
                int su = 0, nu = 0, base = 0;
                bool found = false;
#pragma unroll
                for (int u = 0; u < PA_S; u++) {
                    const int c = __popcll(msk[u]);
                    if (sel && !found && j < base + c) {
                        su = u;
                        nu = j - base;
                        found = true;
                    }
                    base += c;
                }
                unsigned long long mj = msk[0];
#pragma unroll
                for (int u = 1; u < PA_S; u++) {
                    if (su == u) {
                        mj = msk[u];
                    }
                }
                const int src = sel ? pa_nth_set_bit(mj, nu) : lane;
#pragma unroll
                for (int u = 0; u < PA_S; u++) {
                    const float fd = __shfl(dis[u], src, KN_WAVE);
                    const int64_t fi = __shfl(id[u], src, KN_WAVE);
                    if (sel && su == u) {
                        top.d[0] = fd;
                        top.i[0] = fi;
                    }
                }
                pa_wave_sort_pairs<IS_L2>(top.d[0], top.i[0]);
                cnt = min(k, cnt + C);
                if (lane >= k) {
                    top.d[0] = worst_dist<IS_L2>();
                    top.i[0] = -1;
                }
                kd = top.kth_dist();
                ki = top.kth_idx();
            } else {
#pragma unroll
                for (int u = 0; u < PA_S; u++) {
                    insert_passing((msk[u] >> lane) & 1ull, dis[u], id[u]);
                }
                cnt = __popcll(__ballot(top.i[0] >= 0));
            }
        }
    }
    top.store(pd, pi);
}

int pq_scan_any_parts(int k) {
    return k > KN_WAVE ? 1 : PA_WAVES; // partial lists per (query, probe): block selection / one list per wave
}

hipError_t launch_pq_scan_any(const PqAnyArgs& a, int64_t nq, bool is_l2, hipStream_t s) {
    if (nq <= 0 || a.nprobe <= 0) {
        return hipSuccess;
    }
    if (!pq_scan_any_supports(a.M, a.d) || a.k <= 0 || a.k > KN_MAX_K) {
        return hipErrorInvalidValue;
    }
    const unsigned grid = (unsigned)(nq * a.nprobe);
    if (a.k > KN_WAVE) {
        const size_t sm = pq_scan_any_block_smem(a.M, a.k);
        auto kern = is_l2 ? pq_scan_any_block_kernel<true> : pq_scan_any_block_kernel<false>;
        if (sm > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)sm);
            if (e != hipSuccess) {
                return e;
            }
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(PA_THREADS), sm, s, a);
        return hipGetLastError();
    }
    const size_t sm = (size_t)a.M * 256 * sizeof(float);
    auto kern = is_l2 ? pq_scan_any_kernel<true> : pq_scan_any_kernel<false>;
    if (sm > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)sm);
        if (e != hipSuccess) {
            return e;
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(PA_THREADS), sm, s, a);
    return hipGetLastError();
}

} // namespace knhip
