// knowhere_amd/csrc/topk.hip -- k-selection kernels for gfx950 (wave64).
//
//   merge_partials : per query, the k best of nslot sorted partial lists
//                    == the effect of the reference's single per-query heap visited probe by
//                    probe (thirdparty/faiss/faiss/IndexIVF.cpp:499-508, 642-665) and of
//                    merge_knn_results over shards (utils/Heap.h:636).
//   row_select     : per row, the k best of n values with the column index as id -- the
//                    coarse quantizer's top-nprobe (IndexFlat::search over the centroids,
//                    thirdparty/faiss/faiss/utils/distances.cpp:834-875).
// Both return the canonical order of heap_reorder (L2: dist asc, id asc; IP: dist desc, id desc)
// and pad missing results with id -1 / the neutral distance (utils/Heap.h:338-341).
#include "common.h"
#include "kernels.h"

namespace knhip {

// ---------------------------------------------------------------------------------------------
// merge_partials: one wave per query
// ---------------------------------------------------------------------------------------------
template <bool IS_L2, int R>
__global__ __launch_bounds__(256) void merge_partials_kernel(const float* __restrict__ pd,
                                                             const int64_t* __restrict__ pi,
                                                             int64_t nq, int nslot, int k,
                                                             int64_t q_stride, int64_t slot_stride,
                                                             float* __restrict__ out_d,
                                                             int64_t* __restrict__ out_i,
                                                             const int32_t* __restrict__ q_only) {
    const int lane = lane_id();
    const int64_t q = (int64_t)blockIdx.x * (blockDim.x / KN_WAVE) + threadIdx.x / KN_WAVE;
    if (q >= nq) {
        return;
    }
    if (q_only != nullptr && (q_only[nq] == 0 || q_only[q] == 0)) {
        return; // (mfma_scan.hip fallback: only the flagged queries are merged)
    }
    WaveTopK<IS_L2, R> top;
    top.init(k);
    float kd = worst_dist<IS_L2>();
    int64_t ki = -1;
    const float* qd = pd + q * q_stride;
    const int64_t* qi = pi + q * q_stride;
    // Slot lists are sorted best-first and SENTINEL-TERMINATED: nothing behind the first id < 0 of a slot is
    // defined (the scan kernels stop writing there).
    // Slot 0 (the closest list / first shard) usually supplies a large share of the result: load it straight
    // into the wave-resident list (element e -> lane e % 64, register e / 64) so that the rank loop below can
    // stop early.
    int first_end = k;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int e = r * KN_WAVE + lane;
        int64_t id0 = -1;
        float d0 = worst_dist<IS_L2>();
        if (e < k && e < first_end) {
            id0 = qi[e];
            if (id0 >= 0) {
                d0 = qd[e];
            }
        }
        const unsigned long long endm = __ballot(e < k && e < first_end && id0 < 0);
        if (endm) {
            first_end = min(first_end, r * KN_WAVE + __ffsll((long long)endm) - 1);
        }
        const bool ok = e < first_end && id0 >= 0;
        top.d[r] = ok ? d0 : worst_dist<IS_L2>();
        top.i[r] = ok ? id0 : -1;
    }
    kd = top.kth_dist();
    ki = top.kth_idx();
    // per lane: bit g set = slot g * 64 + lane is exhausted (sentinel seen, or an entry that could not enter:
    // every later one of that slot is worse and the bound only tightens)
    unsigned long long done = 0;
    for (int r = 0; r < k; r++) {
        bool any = false;
        int g = 0;
        for (int s0 = 0; s0 < nslot; s0 += KN_WAVE, g++) {
            const int s = s0 + lane;
            float cd = worst_dist<IS_L2>();
            int64_t ci = -1;
            const bool live = s < nslot && s > 0 && !((done >> g) & 1ull);
            if (live) {
                ci = qi[(int64_t)s * slot_stride + r];
                if (ci >= 0) {
                    cd = qd[(int64_t)s * slot_stride + r];
                }
            }
            const bool pass = (ci >= 0) && top.admits(cd, ci, kd, ki);
            if (!pass) {
                done |= 1ull << g;
            }
            unsigned long long m = __ballot(pass);
            any |= (m != 0);
            while (m) {
                const int l = __ffsll((long long)m) - 1;
                m &= m - 1;
                const float xd = __shfl(cd, l, KN_WAVE);
                const int64_t xi = shfl_i64(ci, l);
                if (top.admits(xd, xi, kd, ki)) {
                    top.insert(xd, xi);
                    kd = top.kth_dist();
                    ki = top.kth_idx();
                }
            }
        }
        if (!any) {
            break; // every slot is sorted best-first: deeper ranks cannot enter either
        }
    }
    top.store(out_d + q * k, out_i + q * k);
}

// merge_partials for more than 4096 slots (nprobe above 4096): the kernel above tracks its exhausted slots in one
// 64-bit word per lane (bit g = slot 64 g + lane), which holds 64 x 64 slots; this copy keeps the bits in LDS.  (A
// separate kernel so that the one every search runs stays exactly as validated.)
constexpr int MP_BIG_GROUPS = 1024; // 65536 slots
template <bool IS_L2, int R>
__global__ __launch_bounds__(256) void merge_partials_big_kernel(const float* __restrict__ pd,
                                                             const int64_t* __restrict__ pi,
                                                             int64_t nq, int nslot, int k,
                                                             int64_t q_stride, int64_t slot_stride,
                                                             float* __restrict__ out_d,
                                                             int64_t* __restrict__ out_i,
                                                             const int32_t* __restrict__ q_only) {
    const int lane = lane_id();
    const int64_t q = (int64_t)blockIdx.x * (blockDim.x / KN_WAVE) + threadIdx.x / KN_WAVE;
    if (q >= nq) {
        return;
    }
    if (q_only != nullptr && (q_only[nq] == 0 || q_only[q] == 0)) {
        return; // (mfma_scan.hip fallback: only the flagged queries are merged)
    }
    // exhausted-slot bits of this wave, word g = the 64 lanes' bits of slots [64 g, 64 g + 64): wave-private (written
    // by lane 0 and read by the whole wave in program order)
    __shared__ unsigned long long s_done_all[4 * MP_BIG_GROUPS];
    volatile unsigned long long* s_done = s_done_all + (threadIdx.x / KN_WAVE) * MP_BIG_GROUPS;
    for (int g = lane; g < MP_BIG_GROUPS; g += KN_WAVE) {
        s_done[g] = 0ull;
    }
    WaveTopK<IS_L2, R> top;
    top.init(k);
    float kd = worst_dist<IS_L2>();
    int64_t ki = -1;
    const float* qd = pd + q * q_stride;
    const int64_t* qi = pi + q * q_stride;
    // Slot lists are sorted best-first and SENTINEL-TERMINATED: nothing behind the first id < 0 of a slot is
    // defined (the scan kernels stop writing there).
    // Slot 0 (the closest list / first shard) usually supplies a large share of the result: load it straight
    // into the wave-resident list (element e -> lane e % 64, register e / 64) so that the rank loop below can
    // stop early.
    int first_end = k;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int e = r * KN_WAVE + lane;
        int64_t id0 = -1;
        float d0 = worst_dist<IS_L2>();
        if (e < k && e < first_end) {
            id0 = qi[e];
            if (id0 >= 0) {
                d0 = qd[e];
            }
        }
        const unsigned long long endm = __ballot(e < k && e < first_end && id0 < 0);
        if (endm) {
            first_end = min(first_end, r * KN_WAVE + __ffsll((long long)endm) - 1);
        }
        const bool ok = e < first_end && id0 >= 0;
        top.d[r] = ok ? d0 : worst_dist<IS_L2>();
        top.i[r] = ok ? id0 : -1;
    }
    kd = top.kth_dist();
    ki = top.kth_idx();
    // a slot is exhausted once its sentinel was seen or an entry could not enter (every later one of that slot is
    // worse and the bound only tightens)
    for (int r = 0; r < k; r++) {
        bool any = false;
        int g = 0;
        for (int s0 = 0; s0 < nslot; s0 += KN_WAVE, g++) {
            const int s = s0 + lane;
            float cd = worst_dist<IS_L2>();
            int64_t ci = -1;
            const bool live = s < nslot && s > 0 && !((s_done[g] >> lane) & 1ull);
            if (live) {
                ci = qi[(int64_t)s * slot_stride + r];
                if (ci >= 0) {
                    cd = qd[(int64_t)s * slot_stride + r];
                }
            }
            const bool pass = (ci >= 0) && top.admits(cd, ci, kd, ki);
            unsigned long long m = __ballot(pass);
            if (lane == 0) {
                s_done[g] = s_done[g] | ~m;
            }
            any |= (m != 0);
            while (m) {
                const int l = __ffsll((long long)m) - 1;
                m &= m - 1;
                const float xd = __shfl(cd, l, KN_WAVE);
                const int64_t xi = shfl_i64(ci, l);
                if (top.admits(xd, xi, kd, ki)) {
                    top.insert(xd, xi);
                    kd = top.kth_dist();
                    ki = top.kth_idx();
                }
            }
        }
        if (!any) {
            break; // every slot is sorted best-first: deeper ranks cannot enter either
        }
    }
    top.store(out_d + q * k, out_i + q * k);
}

hipError_t launch_merge_partials(const float* partial_d, const int64_t* partial_i, int64_t nq,
                                 int nslot, int k, int64_t q_stride, int64_t slot_stride, bool is_l2,
                                 float* out_d, int64_t* out_i, hipStream_t s, const int32_t* q_only) {
    if (nq <= 0) {
        return hipSuccess;
    }
    const unsigned grid = (unsigned)((nq + 3) / 4);
    if (nslot > KN_WAVE * MP_BIG_GROUPS) {
        return hipErrorInvalidValue;
    }
    if (nslot > KN_WAVE * KN_WAVE) {
        KN_DISPATCH_R(k, {
            if (is_l2) {
                hipLaunchKernelGGL((merge_partials_big_kernel<true, R_>), dim3(grid), dim3(256), 0, s, partial_d, partial_i,
                                   nq, nslot, k, q_stride, slot_stride, out_d, out_i, q_only);
            } else {
                hipLaunchKernelGGL((merge_partials_big_kernel<false, R_>), dim3(grid), dim3(256), 0, s, partial_d, partial_i,
                                   nq, nslot, k, q_stride, slot_stride, out_d, out_i, q_only);
            }
        });
        return hipGetLastError();
    }
    KN_DISPATCH_R(k, {
        if (is_l2) {
            hipLaunchKernelGGL((merge_partials_kernel<true, R_>), dim3(grid), dim3(256), 0, s,
                               partial_d, partial_i, nq, nslot, k, q_stride, slot_stride, out_d, out_i, q_only);
        } else {
            hipLaunchKernelGGL((merge_partials_kernel<false, R_>), dim3(grid), dim3(256), 0, s,
                               partial_d, partial_i, nq, nslot, k, q_stride, slot_stride, out_d, out_i, q_only);
        }
    });
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// row_select: one 256-thread workgroup per row; radix select (4 x 8-bit digits, LDS histogram)
// for the k-th key, gather, bitonic sort of the survivors in LDS (GSORT: in a global scratch row, for
// k above what the LDS holds -- nprobe / range search on indexes with up to 65536 lists).
// ---------------------------------------------------------------------------------------------
constexpr int RS_THREADS = 256;
constexpr int RS_LDS_MAX_K = 16384; // the selected keys are sorted in LDS: 8 bytes each, 128 KB of the CU's 160
constexpr int RS_MAX_K = 65536;     // above RS_LDS_MAX_K the sort runs in global memory (slow path, rare shapes)

// order-preserving map float -> uint32 such that "better" == smaller key
template <bool IS_L2>
__device__ __forceinline__ uint32_t rs_key(float f) {
    uint32_t b = __float_as_uint(f);
    uint32_t asc = (b & 0x80000000u) ? ~b : (b | 0x80000000u); // ascending in f
    return IS_L2 ? asc : ~asc;
}
template <bool IS_L2>
__device__ __forceinline__ float rs_unkey(uint32_t key) {
    uint32_t asc = IS_L2 ? key : ~key;
    uint32_t b = (asc & 0x80000000u) ? (asc & 0x7fffffffu) : ~asc;
    return __uint_as_float(b);
}

template <bool IS_L2, bool GSORT = false>
__global__ __launch_bounds__(RS_THREADS) void row_select_kernel(const float* __restrict__ vals,
                                                                int64_t n_fixed, int k, int kp,
                                                                int64_t* __restrict__ out_keys,
                                                                float* __restrict__ out_d,
                                                                const int32_t* __restrict__ row_flags,
                                                                int64_t stride,
                                                                const int64_t* __restrict__ var_keys,
                                                                int key_stride,
                                                                const int64_t* __restrict__ var_len,
                                                                int64_t n_cap, const int32_t* __restrict__ n_row,
                                                                unsigned long long* __restrict__ sort_scratch) {
    extern __shared__ __align__(16) unsigned char smem[];
    int64_t n = n_fixed;
    if (row_flags != nullptr && row_flags[blockIdx.x] == 0) {
        return; // only rows flagged by the coarse certificate are re-selected
    }
    if (var_keys != nullptr) { // variable-length rows (rank-0 list dumps)
        const int64_t key = var_keys[(int64_t)blockIdx.x * key_stride];
        n = key >= 0 ? var_len[key] : 0;
        if (n_cap > 0) {
            n = min(n, n_cap); // (only the first n_cap values of a row were written: mfma_scan.hip sample)
        }
    }
    if (n_row != nullptr) {
        n = n_row[blockIdx.x]; // row lengths given directly
    }
    unsigned long long* cand = GSORT ? sort_scratch + (int64_t)blockIdx.x * kp
                                     : reinterpret_cast<unsigned long long*>(smem); // [kp]
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_prefix, s_need, s_count, s_wave_tot[RS_THREADS / KN_WAVE], s_taken;
    const int tid = threadIdx.x;
    const float* row = vals + (int64_t)blockIdx.x * (stride > 0 ? stride : n);
    const int keff = (int)min((int64_t)k, n);

    if (keff == 0) { // empty row: all sentinels
        for (int e = tid; e < k; e += RS_THREADS) {
            out_keys[(int64_t)blockIdx.x * k + e] = -1;
            out_d[(int64_t)blockIdx.x * k + e] = worst_dist<IS_L2>();
        }
        return;
    }
    // ---- radix select: find key T with count(key < T) < keff <= count(key <= T) ----
    uint32_t prefix = 0, prefix_mask = 0;
    uint32_t need = (uint32_t)keff; // rank (1-based) of the wanted key among keys matching prefix
    for (int shift = 24; shift >= 0; shift -= 8) {
        hist[tid] = 0;
        __syncthreads();
        if (shift == 24) {
            // top digit = sign + exponent bits: a whole row falls into two or three bins, so the counts are
            // combined per wave (one atomic per distinct digit) instead of 64 atomics on the same address
            for (int64_t i0 = 0; i0 < n; i0 += RS_THREADS) {
                const int64_t i = i0 + tid;
                const bool act = i < n;
                const uint32_t dg = act ? (rs_key<IS_L2>(row[i]) >> 24) : 0xffffffffu;
                unsigned long long rem = __ballot(act);
                while (rem) {
                    const int l = __ffsll((long long)rem) - 1;
                    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)dg, l);
                    const unsigned long long m = __ballot(act && dg == d0);
                    if ((tid & (KN_WAVE - 1)) == l) {
                        atomicAdd(&hist[d0], (uint32_t)__popcll(m));
                    }
                    rem &= ~m;
                }
            }
        } else {
            for (int64_t i = tid; i < n; i += RS_THREADS) {
                const uint32_t key = rs_key<IS_L2>(row[i]);
                if ((key & prefix_mask) == prefix) {
                    atomicAdd(&hist[(key >> shift) & 0xff], 1u);
                }
            }
        }
        __syncthreads();
        {   // parallel scan of the 256 bins (one per thread): the bin where the running count reaches `need`
            const uint32_t h = hist[tid];
            uint32_t c = h;
#pragma unroll
            for (int dlt = 1; dlt < KN_WAVE; dlt <<= 1) {
                const uint32_t up = __shfl_up(c, dlt, KN_WAVE);
                c += (tid & (KN_WAVE - 1)) >= dlt ? up : 0u;
            }
            if ((tid & (KN_WAVE - 1)) == KN_WAVE - 1) {
                s_wave_tot[tid / KN_WAVE] = c;
            }
            __syncthreads();
            for (int w = 0; w < tid / KN_WAVE; w++) {
                c += s_wave_tot[w];
            }
            if (c >= need && c - h < need) {
                s_prefix = prefix | ((uint32_t)tid << shift);
                s_need = need - (c - h);
            }
        }
        __syncthreads();
        prefix = s_prefix;
        need = s_need;
        prefix_mask |= 0xffu << shift;
        __syncthreads();
    }
    const uint32_t T = prefix; // the keff-th best key; `need` of the elements equal to T are wanted

    // ---- gather: everything better than T, then the `need` lowest/highest-index ties at T ----
    if (tid == 0) {
        s_count = 0;
        s_taken = 0;
    }
    for (int i = tid; i < kp; i += RS_THREADS) {
        cand[i] = ~0ull;
    }
    __syncthreads();
    for (int64_t i = tid; i < n; i += RS_THREADS) {
        const uint32_t key = rs_key<IS_L2>(row[i]);
        if (key < T) {
            const uint32_t pos = atomicAdd(&s_count, 1u);
            const uint32_t tie = IS_L2 ? (uint32_t)i : ~(uint32_t)i;
            cand[pos] = ((unsigned long long)key << 32) | tie;
        }
    }
    __syncthreads();
    const uint32_t base = s_count;
    // ties at T in canonical index order (L2: ascending index, IP: descending index)
    const int lane = tid & (KN_WAVE - 1), wave = tid / KN_WAVE;
    const int64_t ntile = (n + RS_THREADS - 1) / RS_THREADS;
    for (int64_t tile = 0; tile < ntile; tile++) {
        if (s_taken >= need) {
            break;
        }
        const int64_t pos_in_order = tile * RS_THREADS + tid;
        const int64_t i = IS_L2 ? pos_in_order : (n - 1 - pos_in_order);
        bool flag = false;
        if (pos_in_order < n) {
            flag = rs_key<IS_L2>(row[i]) == T;
        }
        const unsigned long long bal = __ballot(flag);
        const uint32_t before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) {
            s_wave_tot[wave] = __popcll(bal);
        }
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (int w = 0; w < RS_THREADS / KN_WAVE; w++) {
            if (w < wave) {
                woff += s_wave_tot[w];
            }
            tot += s_wave_tot[w];
        }
        const uint32_t taken = s_taken;
        if (flag) {
            const uint32_t ord = taken + woff + before;
            if (ord < need) {
                const uint32_t tie = IS_L2 ? (uint32_t)i : ~(uint32_t)i;
                cand[base + ord] = ((unsigned long long)T << 32) | tie;
            }
        }
        __syncthreads();
        if (tid == 0) {
            s_taken = taken + tot;
        }
        __syncthreads();
    }
    __syncthreads();

    // ---- bitonic sort of cand[0..kp) ascending (composite key: better first) ----
    for (int size = 2; size <= kp; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < kp / 2; t += RS_THREADS) {
                const int lo = (t / stride) * stride * 2 + (t % stride);
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const unsigned long long a = cand[lo], b = cand[hi];
                if ((a > b) == up) {
                    cand[lo] = b;
                    cand[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    // ---- emit ----
    int64_t* ok = out_keys + (int64_t)blockIdx.x * k;
    float* od = out_d + (int64_t)blockIdx.x * k;
    for (int e = tid; e < k; e += RS_THREADS) {
        if (e < keff) {
            const unsigned long long c = cand[e];
            const uint32_t key = (uint32_t)(c >> 32);
            const uint32_t tie = (uint32_t)c;
            ok[e] = IS_L2 ? (int64_t)tie : (int64_t)(~tie);
            od[e] = rs_unkey<IS_L2>(key);
        } else {
            ok[e] = -1;
            od[e] = worst_dist<IS_L2>();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// row_select_thr: the same selection (the k best of a row of n values, canonical order) for k << n, in two passes
// over the row instead of the six of the radix select, and without its contended LDS histogram (distance rows fall
// into two or three bins of the upper digits).  The row is cut into G >= 2 k disjoint groups (group g = the elements
// i with i mod G == g); the k-th smallest of the G group minima is an upper bound of the row's k-th smallest value --
// k groups hold an element at least that good -- so every element <= that bound is a candidate (expected
// ~ k ln(1 / (1 - k / G)) n / (n / G ...) ~ 1.2 .. 1.6 k of them); they are compacted into LDS (wave-aggregated) and
// sorted canonically.  Rows whose candidates do not fit RT_CAP (masses of equal values) are flagged for the radix
// kernel.  Round 3: the coarse stage spent 0.93 of its 1.68 ms (C3) / 5.4 of 17.8 ms (C5) in the radix select.
// ---------------------------------------------------------------------------------------------
constexpr int RT_THREADS = 256;
constexpr int RT_CAP = 2048;    // candidates per row (LDS: 16 KB)
constexpr int RT_GMAX = 2048;   // group minima per row (LDS: 8 KB)

template <bool IS_L2>
__global__ __launch_bounds__(RT_THREADS) void row_select_thr_kernel(const float* __restrict__ vals, int64_t n, int k,
                                                                    int kp, int G, int64_t* __restrict__ out_keys,
                                                                    float* __restrict__ out_d,
                                                                    int32_t* __restrict__ ovf_flags) {
    __shared__ unsigned long long cand[RT_CAP];
    __shared__ uint32_t gmin[RT_GMAX];
    __shared__ uint32_t s_count;
    const int tid = threadIdx.x;
    const float* row = vals + (int64_t)blockIdx.x * n;
    // ---- pass 1: group minima (thread t owns the groups g with g mod RT_THREADS == t: all its elements) ----
    const int gper = G / RT_THREADS; // groups per thread (G is a multiple of RT_THREADS)
    uint32_t lm[RT_GMAX / RT_THREADS];
#pragma unroll
    for (int u = 0; u < RT_GMAX / RT_THREADS; u++) {
        lm[u] = 0xffffffffu;
    }
    // 16-byte loads, four in flight: thread t takes the float4 pieces t, t + 256, ...; piece number `it` of a thread goes
    // to its group it mod gper (any disjoint cut of the row into G groups serves the bound)
    const int64_t n4 = n >> 2; // (rows are 16-byte aligned when n is a multiple of 4; the tail goes element by element)
    const bool vec = (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(row) & 15) == 0);
    if (vec) {
        const float4* row4 = reinterpret_cast<const float4*>(row);
        int it = 0;
#pragma unroll 8
        for (int64_t j = tid; j < n4; j += RT_THREADS, it++) {
            const float4 v = row4[j];
            const uint32_t key = min(min(rs_key<IS_L2>(v.x), rs_key<IS_L2>(v.y)), min(rs_key<IS_L2>(v.z), rs_key<IS_L2>(v.w)));
            const int u = it % gper;
#pragma unroll
            for (int uu = 0; uu < RT_GMAX / RT_THREADS; uu++) {
                if (uu == u) {
                    lm[uu] = min(lm[uu], key);
                }
            }
        }
    } else {
        int it = 0;
        for (int64_t i = tid; i < n; i += RT_THREADS, it++) {
            const uint32_t key = rs_key<IS_L2>(row[i]);
            const int u = it % gper;
#pragma unroll
            for (int uu = 0; uu < RT_GMAX / RT_THREADS; uu++) {
                if (uu == u) {
                    lm[uu] = min(lm[uu], key);
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < RT_GMAX / RT_THREADS; u++) {
        if (u < gper) {
            gmin[u * RT_THREADS + tid] = lm[u];
        }
    }
    if (tid == 0) {
        s_count = 0;
    }
    __syncthreads();
    // ---- the k-th smallest group minimum: bitonic sort of G keys ----
    for (int size = 2; size <= G; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < G / 2; t += RT_THREADS) {
                const int lo = (t / stride) * stride * 2 + (t % stride);
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const uint32_t a = gmin[lo], b = gmin[hi];
                if ((a > b) == up) {
                    gmin[lo] = b;
                    gmin[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    const uint32_t theta = gmin[k - 1];
    __syncthreads();
    // ---- pass 2: everything <= theta (the row is L2-resident from pass 1) ----
    const int lane = tid & (KN_WAVE - 1);
    auto take = [&](bool hit, uint32_t key, int64_t i) {
        const unsigned long long bal = __ballot(hit);
        if (bal != 0ull) {
            uint32_t base = 0;
            if (lane == 0) {
                base = atomicAdd(&s_count, (uint32_t)__popcll(bal));
            }
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (hit) {
                const uint32_t pos = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                if (pos < RT_CAP) {
                    const uint32_t tie = IS_L2 ? (uint32_t)i : ~(uint32_t)i;
                    cand[pos] = ((unsigned long long)key << 32) | tie;
                }
            }
        }
    };
    if (vec) {
        const float4* row4 = reinterpret_cast<const float4*>(row);
#pragma unroll 8
        for (int64_t j0 = 0; j0 < n4; j0 += RT_THREADS) {
            const int64_t j = j0 + tid;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const bool in = j < n4;
            if (in) {
                v = row4[j];
            }
            const uint32_t k0 = rs_key<IS_L2>(v.x), k1 = rs_key<IS_L2>(v.y), k2 = rs_key<IS_L2>(v.z), k3 = rs_key<IS_L2>(v.w);
            const bool any = in && (min(min(k0, k1), min(k2, k3)) <= theta);
            if (__ballot(any) != 0ull) { // (most pieces hold no candidate: one ballot instead of four)
                take(in && k0 <= theta, k0, 4 * j);
                take(in && k1 <= theta, k1, 4 * j + 1);
                take(in && k2 <= theta, k2, 4 * j + 2);
                take(in && k3 <= theta, k3, 4 * j + 3);
            }
        }
    } else {
        for (int64_t i0 = 0; i0 < n; i0 += RT_THREADS) {
            const int64_t i = i0 + tid;
            uint32_t key = 0xffffffffu;
            bool hit = false;
            if (i < n) {
                key = rs_key<IS_L2>(row[i]);
                hit = key <= theta;
            }
            take(hit, key, i);
        }
    }
    __syncthreads();
    const uint32_t C = s_count;
    if (C > RT_CAP) { // (masses of equal values: the radix kernel takes the row)
        if (tid == 0) {
            ovf_flags[blockIdx.x] = 1;
        }
        return;
    }
    if (tid == 0) {
        ovf_flags[blockIdx.x] = 0;
    }
    int P = kp;
    while (P < (int)C) {
        P <<= 1;
    }
    for (int i = (int)C + tid; i < P; i += RT_THREADS) {
        cand[i] = ~0ull;
    }
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < P / 2; t += RT_THREADS) {
                const int lo = (t / stride) * stride * 2 + (t % stride);
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const unsigned long long a = cand[lo], b = cand[hi];
                if ((a > b) == up) {
                    cand[lo] = b;
                    cand[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    int64_t* ok = out_keys + (int64_t)blockIdx.x * k;
    float* od = out_d + (int64_t)blockIdx.x * k;
    for (int e = tid; e < k; e += RT_THREADS) {
        const unsigned long long c = cand[e];
        const uint32_t key = (uint32_t)(c >> 32);
        const uint32_t tie = (uint32_t)c;
        ok[e] = IS_L2 ? (int64_t)tie : (int64_t)(~tie);
        od[e] = rs_unkey<IS_L2>(key);
    }
}

// true when the two-pass selection applies: k candidates of n with 2 k groups of at least 8 elements each
bool row_select_thr_supports(int64_t n, int k) {
    int G = RT_THREADS;
    while (G < 2 * k) {
        G <<= 1;
    }
    return k >= 1 && k <= RT_CAP / 2 && G <= RT_GMAX && n >= (int64_t)G * 8 && n <= 0x7fffffffll;
}

// ovf_flags [nrows]: 1 = the row was left to the radix kernel (launch_row_select with these flags afterwards)
hipError_t launch_row_select_thr(const float* vals, int64_t nrows, int64_t n, int k, bool is_l2, int64_t* out_keys,
                                 float* out_d, int32_t* ovf_flags, hipStream_t s) {
    if (nrows <= 0) {
        return hipSuccess;
    }
    if (!row_select_thr_supports(n, k) || ovf_flags == nullptr) {
        return hipErrorInvalidValue;
    }
    int G = RT_THREADS;
    while (G < 2 * k) {
        G <<= 1;
    }
    int kp = 2;
    while (kp < k) {
        kp <<= 1;
    }
    if (is_l2) {
        hipLaunchKernelGGL(row_select_thr_kernel<true>, dim3((unsigned)nrows), dim3(RT_THREADS), 0, s, vals, n, k, kp, G,
                           out_keys, out_d, ovf_flags);
    } else {
        hipLaunchKernelGGL(row_select_thr_kernel<false>, dim3((unsigned)nrows), dim3(RT_THREADS), 0, s, vals, n, k, kp, G,
                           out_keys, out_d, ovf_flags);
    }
    return hipGetLastError();
}

size_t row_select_max_k() {
    return RS_MAX_K;
}

size_t row_select_lds_max_k() {
    return RS_LDS_MAX_K;
}

// more than 48 KB of dynamic LDS (k > 4096) has to be allowed per kernel
static hipError_t rs_allow_smem(size_t sm, bool is_l2) {
    if (sm <= 48 * 1024) {
        return hipSuccess;
    }
    const void* kern = is_l2 ? reinterpret_cast<const void*>(row_select_kernel<true>)
                             : reinterpret_cast<const void*>(row_select_kernel<false>);
    return hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
}

hipError_t launch_row_select(const float* vals, int64_t nrows, int64_t n, int k, bool is_l2,
                             int64_t* out_keys, float* out_d, const int32_t* row_flags, hipStream_t s,
                             unsigned long long* sort_scratch) {
    if (nrows <= 0 || k <= 0) {
        return hipSuccess;
    }
    if (k > RS_MAX_K || n > 0xffffffffll) {
        return hipErrorInvalidValue;
    }
    int kp = 2;
    while (kp < k) {
        kp <<= 1;
    }
    if (k > RS_LDS_MAX_K) { // the selected keys do not fit the LDS: sorted in the caller's scratch ([nrows][kp] u64)
        if (sort_scratch == nullptr) {
            return hipErrorInvalidValue;
        }
        if (is_l2) {
            hipLaunchKernelGGL((row_select_kernel<true, true>), dim3((unsigned)nrows), dim3(RS_THREADS), 0, s, vals, n, k, kp,
                               out_keys, out_d, row_flags, (int64_t)0, nullptr, 0, nullptr, (int64_t)0, nullptr, sort_scratch);
        } else {
            hipLaunchKernelGGL((row_select_kernel<false, true>), dim3((unsigned)nrows), dim3(RS_THREADS), 0, s, vals, n, k, kp,
                               out_keys, out_d, row_flags, (int64_t)0, nullptr, 0, nullptr, (int64_t)0, nullptr, sort_scratch);
        }
        return hipGetLastError();
    }
    const size_t sm = (size_t)kp * 8;
    if (hipError_t e = rs_allow_smem(sm, is_l2); e != hipSuccess) {
        return e;
    }
    if (is_l2) {
        hipLaunchKernelGGL((row_select_kernel<true>), dim3((unsigned)nrows), dim3(RS_THREADS), sm, s,
                           vals, n, k, kp, out_keys, out_d, row_flags, (int64_t)0, nullptr, 0, nullptr, (int64_t)0, nullptr,
                           nullptr);
    } else {
        hipLaunchKernelGGL((row_select_kernel<false>), dim3((unsigned)nrows), dim3(RS_THREADS), sm, s,
                           vals, n, k, kp, out_keys, out_d, row_flags, (int64_t)0, nullptr, 0, nullptr, (int64_t)0, nullptr,
                           nullptr);
    }
    return hipGetLastError();
}

hipError_t launch_row_select_var(const float* vals, int64_t stride, const int64_t* keys, int key_stride,
                                 const int64_t* list_len, int64_t nrows, int k, bool is_l2, int64_t* out_keys,
                                 float* out_d, hipStream_t s, int64_t n_cap, const int32_t* n_row) {
    if (nrows <= 0 || k <= 0) {
        return hipSuccess;
    }
    if (k > RS_LDS_MAX_K) {
        return hipErrorInvalidValue;
    }
    int kp = 2;
    while (kp < k) {
        kp <<= 1;
    }
    const size_t sm = (size_t)kp * 8;
    if (hipError_t e = rs_allow_smem(sm, is_l2); e != hipSuccess) {
        return e;
    }
    if (is_l2) {
        hipLaunchKernelGGL((row_select_kernel<true>), dim3((unsigned)nrows), dim3(RS_THREADS), sm, s, vals,
                           (int64_t)0, k, kp, out_keys, out_d, nullptr, stride, keys, key_stride, list_len, n_cap, n_row,
                           nullptr);
    } else {
        hipLaunchKernelGGL((row_select_kernel<false>), dim3((unsigned)nrows), dim3(RS_THREADS), sm, s, vals,
                           (int64_t)0, k, kp, out_keys, out_d, nullptr, stride, keys, key_stride, list_len, n_cap, n_row,
                           nullptr);
    }
    return hipGetLastError();
}

} // namespace knhip
