// knowhere_amd/csrc/pq_scan_q4.hip -- IVF-PQ ADC scan, M = 32 x 8 bit, dsub = 4: persistent workgroups,
// FOUR (query, probe) pairs per work item, one ds_read_b128 per lookup step.
//
// Same contract and arithmetic as pq_scan_v2.hip / pq_scan.hip: dis = dis0 + (((0 + LUT[0][c0]) + LUT[1][c1]) + ...)
// summed in m order, dis0 added last -- bit-equal to PQCodeDistanceScalar and scan_list_with_table (reference
// thirdparty/faiss/faiss/impl/pq_code_distance/pq_code_distance-inl.h:69-90, IVFPQScanner_impl.h:109-181), with
// LUT = precomp[list] + (-2) * <q_m, cb>  (L2, precomputed table: IVFPQ_QueryTables.cpp:126-192),
//       <q_m, cb>                          (IP:  :110-124),
//       ||(q - c_list)_m - cb||^2          (L2, residual tables: :194-230).
//
// Why this shape (round-1 profile of pq_scan_v2: the ADC scan is an LDS-GATHER problem, not an HBM one):
//  * the LDS delivers 256 B/clk/CU only to conflict-free ds_read_b64 / ds_read_b128.  With the 4 queries of an
//    item interleaved in the table, LUT[code][m][4] (16-byte entries, 128 KB), one ds_read_b128 serves 4 lookups
//    of a lane; its four 16-lane service groups each see 16 different stagger phases (pq_stream_phase,
//    kernels.h) = 16 consecutive m = 16 different bank quads: zero bank conflicts for ANY code values.  (v2:
//    2 queries, 8-byte entries, 16 phases in a 32-lane group = 2-way conflict on every lookup.)
//  * per 256 lookups a wave issues 1 ds_read_b128 + 1 SDWA shift (token -> address) + 2 v_pk_add_f32
//    (+2 under flipped EXEC in the 15 split steps of a 32-step window): VALU ~0.86 of the LDS time, so the
//    loop can run at the LDS rate; v2 needed 1 + 1 + 1..2 per 128 lookups.
//  * 128 KB of LUT = one 16-wave workgroup per CU, so nothing overlaps an item's set-up with another
//    workgroup's scan.  The set-up therefore must not wait for HBM: the kernel is PERSISTENT, every workgroup
//    pulls items in list order from its XCD's counter (lists stay L2-resident while their ~20 items run on
//    that XCD, the dispatcher's in-order locality without the dispatcher), and the LUT is COMPUTED from the
//    L2-resident codebook (128 KB) and the list's precomputed-table row (32 KB) instead of being read from
//    per-query tables (4 x 32 KB per item from HBM; no [nq][256][32] scratch, no query-table kernel).
//  * half the items of v2: half the code-stream traffic, table set-ups, end-of-item merges.
//
// Code layout: the stream16 blocks of pq_scan_v2.hip (16-bit tokens code << 8 | m << 3 = LUT byte address >> 1,
// 8 steps per 16-byte block per lane).
#include "common.h"
#include "kernels.h"

#include <cstdio>

namespace knhip {

constexpr int P4_KSUB = 256;
constexpr int P4_M = 32;
constexpr int P4_DSUB = 4;
constexpr int P4_Q = 4;
constexpr int P4_WAVES = 16;
constexpr int P4_THREADS = P4_WAVES * KN_WAVE;
constexpr int P4_LUT_BYTES = P4_KSUB * P4_M * P4_Q * 4; // 131072
constexpr int P4_CTL_BYTES = 256;                        // next-item mailbox behind the LUT

typedef float p4_f32x2 __attribute__((ext_vector_type(2)));
typedef float p4_f32x4 __attribute__((ext_vector_type(4)));

// ---- codebook [m][c][4] -> c-major float4 cb_t[c][m] ----------------------------------------------------
__global__ void pq_cb_transpose_kernel(const float* __restrict__ cb, int M, int dsub, float4* __restrict__ cb_t) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P4_KSUB * M) {
        return;
    }
    const int c = e / M, m = e % M;
    const float* y = cb + ((int64_t)m * P4_KSUB + c) * dsub;
    cb_t[e] = make_float4(y[0], y[1], y[2], y[3]);
}

hipError_t launch_pq_cb_transpose(const float* cb, int M, int dsub, float4* cb_t, hipStream_t s) {
    if (dsub != P4_DSUB) {
        return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(pq_cb_transpose_kernel, dim3((unsigned)((P4_KSUB * M + 255) / 256)), dim3(256), 0, s, cb, M,
                       dsub, cb_t);
    return hipGetLastError();
}

// ---- flat work records + counter reset --------------------------------------------------------------------
__global__ void p4_prepare_kernel(PqScanArgs a, int64_t nrec) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < 8 * 16) {
        a.q4_ctr[r] = 0;
    }
    if (r >= nrec) {
        return;
    }
    const int64_t item_lo = a.item_lo ? *a.item_lo : 0;
    const int64_t nitems = *a.item_hi - item_lo;
    if (r >= nitems) {
        return;
    }
    P4Rec rec{};
    const KnItem it = a.items[item_lo + r];
    const int npair = it.npair < P4_Q ? it.npair : P4_Q;
    rec.list = it.list;
    rec.npair = npair;
    rec.len = a.list_len[it.list];
    rec.sblk0 = a.list_sblk_off[it.list];
    rec.row_off = a.list_row_off[it.list];
    for (int j = 0; j < P4_Q; j++) {
        const KnPair p = a.pairs[it.pair0 + (j < npair ? j : npair - 1)];
        rec.q[j] = p.q;
        rec.slot[j] = p.slot;
        rec.dis0[j] = (a.lut_mode == PQ_LUT_RESIDUAL) ? 0.f : a.coarse_dis[(int64_t)p.q * a.nslot + p.slot];
    }
    a.recs4[r] = rec;
}

// ---- 4 steps of accumulate -----------------------------------------------------------------------------------
// operands: %0/%1 = new sums (queries 01 / 23), %2/%3 = old sums, %4..%11 = the 4 LUT entries as pairs
// (x01, x23), %12..%15 = EXEC masks of the lanes already on the window's new vector (split steps only).
// EXEC is all-ones again before a block ends; the compiler never sees it changed.
#define P4_SPLIT(MK, X, Y)                     \
    "s_mov_b32 exec_lo, " MK "\n\t"            \
    "s_mov_b32 exec_hi, " MK "\n\t"            \
    "v_pk_add_f32 %0, %0, " X "\n\t"           \
    "v_pk_add_f32 %1, %1, " Y "\n\t"           \
    "s_not_b64 exec, exec\n\t"                 \
    "v_pk_add_f32 %2, %2, " X "\n\t"           \
    "v_pk_add_f32 %3, %3, " Y "\n\t"
#define P4_PLAIN(X, Y)                         \
    "v_pk_add_f32 %0, %0, " X "\n\t"           \
    "v_pk_add_f32 %1, %1, " Y "\n\t"
#define P4_VALS(v)                                                                                         \
    "v"(__builtin_shufflevector(v[0], v[0], 0, 1)), "v"(__builtin_shufflevector(v[0], v[0], 2, 3)),         \
    "v"(__builtin_shufflevector(v[1], v[1], 0, 1)), "v"(__builtin_shufflevector(v[1], v[1], 2, 3))

// U = index of the 2-step unit inside the 32-step window (steps 2U, 2U + 1)
template <int U>
__device__ __forceinline__ void p4_accum2(p4_f32x2& n01, p4_f32x2& n23, p4_f32x2& o01, p4_f32x2& o23,
                                          const p4_f32x4 (&v)[2]) {
    static_assert(PQ_STREAM_PHASES == 16, "window steps 15..31 have every lane on the new vector");
    if constexpr (U < 7) {
        asm volatile(P4_SPLIT("%8", "%4", "%5") P4_SPLIT("%9", "%6", "%7") "s_mov_b64 exec, -1\n\t"
                     : "+v"(n01), "+v"(n23), "+v"(o01), "+v"(o23)
                     : P4_VALS(v), "i"(pq_stream_mask(2 * U)), "i"(pq_stream_mask(2 * U + 1)));
    } else if constexpr (U == 7) {
        asm volatile(P4_SPLIT("%8", "%4", "%5") "s_mov_b64 exec, -1\n\t" P4_PLAIN("%6", "%7")
                     : "+v"(n01), "+v"(n23), "+v"(o01), "+v"(o23)
                     : P4_VALS(v), "i"(pq_stream_mask(14)));
    } else {
        asm volatile(P4_PLAIN("%4", "%5") P4_PLAIN("%6", "%7")
                     : "+v"(n01), "+v"(n23), "+v"(o01), "+v"(o23)
                     : P4_VALS(v));
    }
}

template <bool IS_L2>
__device__ __forceinline__ float p4_prefilter(float kd, float dis0) {
    const float slack = (fabsf(kd) + fabsf(dis0)) * 4.8e-7f + 1e-30f;
    return IS_L2 ? (kd - dis0) + slack : (kd - dis0) - slack;
}

__device__ __forceinline__ int p4_sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

// s_setprio takes an immediate: wave-uniform switch
__device__ __forceinline__ void p4_setprio(int p) {
    if (p == 0) {
        __builtin_amdgcn_s_setprio(0);
    } else if (p == 1) {
        __builtin_amdgcn_s_setprio(1);
    } else if (p == 2) {
        __builtin_amdgcn_s_setprio(2);
    } else {
        __builtin_amdgcn_s_setprio(3);
    }
}

// token (low / high half of a code word) -> LDS byte address of the 16-byte LUT entry: one SDWA shift
__device__ __forceinline__ uint32_t p4_addr_lo(uint32_t w, uint32_t one) {
    uint32_t a;
    asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
        : "=v"(a)
        : "v"(w), "s"(one));
    return a;
}
__device__ __forceinline__ uint32_t p4_addr_hi(uint32_t w, uint32_t one) {
    uint32_t a;
    asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
        : "=v"(a)
        : "v"(w), "s"(one));
    return a;
}

// ---- LUT[c][m][query] of one work item ------------------------------------------------------------------------
// thread (wave, lane): m = lane & 31, c = 16 * wave + 2 * u + (lane >> 5), u = 0..7: every global load is a
// contiguous 1 KiB (codebook) / 256 B (table row) per wave, every LDS store a contiguous 1 KiB.  All 20 loads are
// requested in two batches; the arithmetic is the reference's: inner product / squared distance accumulated
// in dimension order from 0 with one rounding per operation (fvec_inner_product / fvec_L2sqr scalar forms,
// src/simd/distances_ref.cc:21-37), then LUT = precomp + (-2) * ip (fvec_madd, IVFPQ_QueryTables.cpp:140-145).
template <int MODE>
__device__ __forceinline__ void p4_build_lut(const PqScanArgs& a, unsigned char* smem, const int wave, const int lane_i,
                                             const int64_t list, const int32_t (&q_of)[P4_Q]) {
    const int m = lane_i & 31;
    const int c0 = 16 * wave + (lane_i >> 5);
    const float4* cbp = a.cb_t + c0 * P4_M + m;
    float4* l4 = reinterpret_cast<float4*>(smem) + c0 * P4_M + m;
    float4 x[P4_Q];
#pragma unroll
    for (int j = 0; j < P4_Q; j++) {
        x[j] = *reinterpret_cast<const float4*>(a.queries + (int64_t)q_of[j] * a.d + m * P4_DSUB);
    }
    if (MODE == PQ_LUT_RESIDUAL) { // residual tables: ||(q - c_list)_m - cb[m][c]||^2
        const float4 cl = *reinterpret_cast<const float4*>(a.centroids + list * a.d + m * P4_DSUB);
#pragma unroll
        for (int j = 0; j < P4_Q; j++) {
            x[j] = make_float4(fsub_x(x[j].x, cl.x), fsub_x(x[j].y, cl.y), fsub_x(x[j].z, cl.z), fsub_x(x[j].w, cl.w));
        }
    }
    const float* pt = a.precomp_t + list * (int64_t)(P4_KSUB * P4_M) + c0 * P4_M + m;
    // two passes of 4 entries keep the register peak of this phase below the scan loop's
#pragma unroll
    for (int h = 0; h < 2; h++) {
        float4 y[4];
        float p[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            y[u] = cbp[(h * 4 + u) * 2 * P4_M];
            p[u] = MODE == PQ_LUT_PRECOMP ? pt[(h * 4 + u) * 2 * P4_M] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float o[P4_Q];
#pragma unroll
            for (int j = 0; j < P4_Q; j++) {
                float t;
                if (MODE == PQ_LUT_RESIDUAL) {
                    t = l2_step(0.f, x[j].x, y[u].x);
                    t = l2_step(t, x[j].y, y[u].y);
                    t = l2_step(t, x[j].z, y[u].z);
                    t = l2_step(t, x[j].w, y[u].w);
                } else {
                    t = ip_step(0.f, x[j].x, y[u].x);
                    t = ip_step(t, x[j].y, y[u].y);
                    t = ip_step(t, x[j].z, y[u].z);
                    t = ip_step(t, x[j].w, y[u].w);
                    if (MODE == PQ_LUT_PRECOMP) {
                        t = fadd_x(p[u], fmul_x(-2.0f, t));
                    }
                }
                o[j] = t;
            }
            l4[(h * 4 + u) * 2 * P4_M] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// Phase timers (profiling build only: make prof -> libknhip_prof.so, never shipped): wave 0 of every workgroup
// sums the shader cycles it spends per phase of an item; printed per launch by launch_q4_r.
#ifdef KNHIP_PHASE_TIMERS
#define P4_T(i)                                                         \
    do {                                                                \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();     \
        tacc[i] += t_ - tlast;                                          \
        tlast = t_;                                                     \
    } while (0)
#define P4_COUNT(i, n) tacc[i] += (unsigned long long)(n)
__device__ unsigned long long g_p4_prof[16 * 10];
#else
#define P4_T(i)
#define P4_COUNT(i, n)
#endif

template <bool IS_L2, int R>
__global__ __launch_bounds__(P4_THREADS) void pq_scan_q4_kernel(PqScanArgs a) {
    constexpr int QG = P4_Q;
#ifdef KNHIP_PHASE_TIMERS
    unsigned long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
#endif
    extern __shared__ __align__(16) unsigned char smem[];
    // mailbox behind the LUT: [0] = index of the next item, [1] = "some wave holds a candidate" flag of the
    // current item, [8..32) = the next item's 96-byte record
    int* ctl = reinterpret_cast<int*>(smem + P4_LUT_BYTES);
    const int lane = lane_id();
    const int wave = p4_sgpr((int)(threadIdx.x / KN_WAVE)); // wave-uniform: scalar loop control
    if ((uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)smem) != 0u) {
        __builtin_trap(); // the 16-bit tokens assume the LUT at LDS offset 0
    }

    const int64_t item_lo = a.item_lo ? *a.item_lo : 0;
    const int nitems = (int)(*a.item_hi - item_lo);
    const int per = (nitems + 7) / 8;
    // this workgroup's XCD: its counter hands out a contiguous eighth of the (list-sorted) items in order, so a
    // list's codes and table row are fetched from HBM once and then served by that XCD's L2.  Exhausted ->
    // help the next XCD.  (Placement is a speed matter only: any workgroup may process any item.)
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int xcd = (int)(xcc & 7u);
    int fetch_t = 0; // thread 0: counters [xcd, xcd + fetch_t) are known to be exhausted
    auto fetch_slow = [&]() -> int {
        while (fetch_t < 8) {
            const int x = (xcd + fetch_t) & 7;
            const int base = x * per;
            const int cnt = min(per, nitems - base);
            if (cnt > 0) {
                const int i = atomicAdd(a.q4_ctr + x * 16, 1);
                if (i < cnt) {
                    return base + i;
                }
            }
            fetch_t++;
        }
        return -1;
    };
    constexpr int REC_WORDS = (int)(sizeof(P4Rec) / 4);
    if (wave == 0) {
        int first = -1;
        if (lane == 0) {
            first = fetch_slow();
        }
        first = __builtin_amdgcn_readlane(first, 0);
        if (lane < REC_WORDS && first >= 0) {
            ctl[8 + lane] = (int)reinterpret_cast<const uint32_t*>(a.recs4 + first)[lane];
        }
        if (lane == 0) {
            ctl[0] = first;
            ctl[1] = 0;
        }
    }
    __syncthreads();
    int cur = p4_sgpr(ctl[0]);
    const uint32_t one = 1u;
    const int k = a.k;

    while (cur >= 0) {
        P4_T(7);
        int lane_i = lane;
        asm volatile("" : "+v"(lane_i)); // nothing lane-derived is hoisted out of the item loop
        // The next item is fetched without ever waiting: the atomic is issued here and consumed after the LUT
        // barrier; the next record is requested then and parked in the mailbox after this wave's scan loop.
        int f_x = 0, f_i = 0;
        if (wave == 0 && lane_i == 0 && fetch_t < 8) {
            f_x = (xcd + fetch_t) & 7;
            f_i = atomicAdd(a.q4_ctr + f_x * 16, 1);
        }
        // ---- the item's record (mailbox), fields broadcast to SGPRs --------------------------------------------
        const uint32_t rw = lane_i < REC_WORDS ? (uint32_t)ctl[8 + lane_i] : 0u;
        auto rl = [&](int i) { return (uint32_t)__builtin_amdgcn_readlane((int)rw, i); };
        const int64_t list = (int64_t)rl(0);
        const int npair = (int)rl(1);
        int32_t q_of[QG], slot_of[QG];
        float dis0[QG];
#pragma unroll
        for (int j = 0; j < QG; j++) {
            q_of[j] = (int32_t)rl(2 + j);
            slot_of[j] = (int32_t)rl(6 + j);
            dis0[j] = __uint_as_float(rl(10 + j));
        }
        const int64_t len = (int64_t)(((uint64_t)rl(15) << 32) | rl(14));
        const int64_t sblk0 = (int64_t)(((uint64_t)rl(17) << 32) | rl(16));
        const int64_t row_off = (int64_t)(((uint64_t)rl(19) << 32) | rl(18));

        // ---- everything the scan will wait for is requested NOW, so that its latency hides behind the LUT build:
        // the shared thresholds, this wave's first code blocks, the candidate histogram row -----------------------
        float gt[QG];
#pragma unroll
        for (int j = 0; j < QG; j++) {
            gt[j] = gthr_load<IS_L2>(a.gthr + q_of[j]);
        }
        // this wave's groups of 64 vectors: as even as possible (the slowest wave sets the item's time)
        const int ngroups = (int)((len + 63) / 64);
        const int gbase = ngroups / P4_WAVES, grem = ngroups % P4_WAVES;
        const int G0 = wave * gbase + min(wave, grem);
        const int G1 = G0 + gbase + (wave < grem ? 1 : 0);
        const int nwin = G1 > G0 ? (G1 - G0 + 1) : 0; // one extra (half) window drains the stagger
        const uint4* cbase = a.codes_skew + (sblk0 + (int64_t)G0 * 4) * 64 + lane_i; // 4 blocks per window
        auto load_blk = [&](int b) { return cbase[(int64_t)b * 64]; };                // past-the-end blocks exist (slack)
        // candidate histogram of this item's queries: wave j refreshes query j's bound from it
        int32_t h_q = q_of[0];
#pragma unroll
        for (int j = 1; j < QG; j++) {
            h_q = wave == j ? q_of[j] : h_q;
        }
        uint32_t h_lo_w = 0, h_shift_w = KN_HIST_OFF, h_cnt = 0;
        const bool h_on = a.ghist != nullptr;
        if (h_on && wave < npair) {
            const uint2 mt = a.gmeta[h_q];
            h_lo_w = mt.x;
            h_shift_w = mt.y;
            h_cnt = __hip_atomic_load(a.ghist + (int64_t)h_q * KN_HIST_BINS + lane_i, __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
        }

        // ---- LUT[c][m][query] built in LDS from the codebook (+ the list's precomputed-table row) ---------------
        if (a.lut_mode == PQ_LUT_PRECOMP) {
            p4_build_lut<PQ_LUT_PRECOMP>(a, smem, wave, lane_i, list, q_of);
        } else if (a.lut_mode == PQ_LUT_IP) {
            p4_build_lut<PQ_LUT_IP>(a, smem, wave, lane_i, list, q_of);
        } else {
            p4_build_lut<PQ_LUT_RESIDUAL>(a, smem, wave, lane_i, list, q_of);
        }

        // this wave's first code blocks: requested before the top-k / histogram set-up and the LUT barrier
        uint4 U0 = make_uint4(0, 0, 0, 0), U1 = U0, U2 = U0, U3 = U0;
        if (nwin > 0) {
            U0 = load_blk(0);
            U1 = load_blk(1);
            U2 = load_blk(2);
            U3 = load_blk(3);
        }
        WaveTopK<IS_L2, R, int32_t> top[QG];
        float kd[QG], pre[QG];
        int32_t ki[QG];
        int ncand[QG];
#pragma unroll
        for (int j = 0; j < QG; j++) {
            ncand[j] = 0;
            top[j].init(k);
            kd[j] = worst_dist<IS_L2>();
            ki[j] = -1;
        }
        if (h_shift_w != KN_HIST_OFF) {
            // inclusive prefix sum of the 64 bins across the wave; first bin where k vectors are reached
            uint32_t cum = h_cnt;
#pragma unroll
            for (int dlt = 1; dlt < KN_WAVE; dlt <<= 1) {
                const uint32_t up = __shfl_up(cum, dlt, KN_WAVE);
                cum += lane_i >= dlt ? up : 0u;
            }
            const unsigned long long reach = __ballot(cum >= (uint32_t)k);
            const int b = reach ? __ffsll((long long)reach) - 1 : KN_HIST_BINS;
            if (b < KN_HIST_BINS - 1) { // (the last bin also collects everything beyond the range)
                const unsigned long long edge =
                        (unsigned long long)h_lo_w + (((unsigned long long)b + 1ull) << h_shift_w) - 1ull;
                if (edge < 0xffffffffull) {
                    const float bound = dist_key_inv<IS_L2>((uint32_t)edge);
                    if (bound == bound && fabsf(bound) < FLT_MAX) {
                        bool better_bound = false;
#pragma unroll
                        for (int j = 0; j < QG; j++) {
                            if (j == wave) {
                                better_bound = IS_L2 ? bound < gt[j] : bound > gt[j];
                                gt[j] = tighter<IS_L2>(gt[j], bound);
                            }
                        }
                        if (better_bound && lane_i == 0) {
                            gthr_publish<IS_L2>(a.gthr + h_q, bound);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < QG; j++) {
            pre[j] = p4_prefilter<IS_L2>(tighter<IS_L2>(kd[j], gt[j]), dis0[j]);
        }
        P4_T(0); // record + LUT build
        __syncthreads();
        P4_T(1); // wait for the other waves' LUT parts
        int nxt = -1;
        uint32_t rw_next = 0;
        if (wave == 0) {
            if (lane_i == 0) {
                if (fetch_t < 8) {
                    const int base = f_x * per;
                    const int cnt = min(per, nitems - base);
                    if (f_i < cnt) {
                        nxt = base + f_i;
                    } else {
                        fetch_t++;
                        nxt = fetch_slow(); // own range exhausted: help the next XCD (tail only)
                    }
                }
            }
            nxt = __builtin_amdgcn_readlane(nxt, 0);
            if (lane_i < REC_WORDS && nxt >= 0) {
                rw_next = reinterpret_cast<const uint32_t*>(a.recs4 + nxt)[lane_i];
            }
        }

        typedef __attribute__((address_space(3))) const p4_f32x4 lds_f4;
        auto lut_read = [&](uint32_t addr) -> p4_f32x4 { return *reinterpret_cast<lds_f4*>(addr); };
        // 2 lookups from one code word (two tokens)
        auto issue2 = [&](uint32_t w0, p4_f32x4 (&v)[2]) {
            v[0] = lut_read(p4_addr_lo(w0, one));
            v[1] = lut_read(p4_addr_hi(w0, one));
        };

        P4_T(2); // set-up of the scan
        if (nwin > 0) {
            p4_f32x2 n01 = {0.f, 0.f}, n23 = {0.f, 0.f}, o01 = {0.f, 0.f}, o23 = {0.f, 0.f};
            // LUT reads run FOUR 2-step units (8 lookups per lane) ahead of the accumulate that consumes them: four
            // value buffers, each refilled right after it has been consumed, so a wave keeps 6-8 ds_read_b128 in
            // flight at all times.  Code registers: at the top of window w, U0 = block 4w+4 (the NEXT window's first
            // block: its tokens feed the reads issued in units 12..15), U1..U3 = blocks 4w+1..4w+3; each register is
            // reloaded with block +4 as soon as its last word has been turned into LUT reads (3/4 of a window ahead
            // of its next use).
            p4_f32x4 B0[2], B1[2], B2[2], B3[2];
            issue2(U0.x, B0);
            issue2(U0.y, B1);
            issue2(U0.z, B2);
            issue2(U0.w, B3);
            U0 = load_blk(4);
            const int last_group = ngroups - 1;
            const unsigned long long tail_mask = (len & 63) ? ((1ull << (len & 63)) - 1ull) : ~0ull;

#define P4_UNIT(U, BUF, WORD)                                      \
    __builtin_amdgcn_sched_barrier(0);                             \
    p4_accum2<U>(n01, n23, o01, o23, BUF);                         \
    __builtin_amdgcn_sched_barrier(0);                             \
    issue2(WORD, BUF);

            for (int w = 0;; w++) {
                // The SIMD arbitrates its 4 waves by priority, then age: with equal priorities the oldest wave of a
                // SIMD runs ~1.8x faster than the youngest and idles at the item's barrier.  Rotating the priority
                // every window (waves w, w + 4, w + 8, w + 12 share a SIMD) gives all four the same average speed.
                p4_setprio((w + (wave >> 2)) & 3);
                // thresholds published by other workgroups meanwhile (consumed in the middle of this window)
                float gnext[QG];
#pragma unroll
                for (int qi = 0; qi < QG; qi++) {
                    gnext[qi] = gthr_load<IS_L2>(a.gthr + q_of[qi]);
                }
                P4_UNIT(0, B0, U1.x)
                P4_UNIT(1, B1, U1.y)
                P4_UNIT(2, B2, U1.z)
                P4_UNIT(3, B3, U1.w)
                U1 = load_blk(4 * w + 5);
                P4_UNIT(4, B0, U2.x)
                P4_UNIT(5, B1, U2.y)
                P4_UNIT(6, B2, U2.z)
                P4_UNIT(7, B3, U2.w)
                U2 = load_blk(4 * w + 6);
                __builtin_amdgcn_sched_barrier(0);

                // ---- step 15 passed: o01 / o23 hold the FINISHED sums of group G0 + w - 1 in every lane ----------
                if (w > 0) {
                    const float Y[QG] = {o01.x, o01.y, o23.x, o23.y};
                    // fast path: one compare per query against the (possibly stale, i.e. looser) prefilter and
                    // one "did a shared threshold move" test; everything else only when one of them fires
                    const unsigned long long vmask = (G0 + w - 1 == last_group) ? tail_mask : ~0ull;
                    unsigned long long any = 0;
                    bool moved = false;
#pragma unroll
                    for (int qi = 0; qi < QG; qi++) {
                        any |= __ballot(IS_L2 ? (Y[qi] <= pre[qi]) : (Y[qi] >= pre[qi]));
                        moved |= IS_L2 ? (gnext[qi] < gt[qi]) : (gnext[qi] > gt[qi]);
                    }
                    any &= vmask;
                    if (any != 0 || __ballot(moved) != 0) {
                        P4_COUNT(9, 1);
                        const int32_t vbase = (G0 + w - 1) * 64;
#pragma unroll
                        for (int qi = 0; qi < QG; qi++) {
                            gt[qi] = tighter<IS_L2>(gt[qi], gnext[qi]);
                            pre[qi] = p4_prefilter<IS_L2>(tighter<IS_L2>(kd[qi], gt[qi]), dis0[qi]);
                            const float o = Y[qi];
                            unsigned long long mm = __ballot(IS_L2 ? (o <= pre[qi]) : (o >= pre[qi])) & vmask;
                            if (mm != 0 && qi < npair) {
                                bool tightened = false;
                                while (mm) {
                                    const int l = __ffsll((long long)mm) - 1;
                                    mm &= mm - 1;
                                    const int32_t v = vbase + l;
                                    const float dis = fadd_x(dis0[qi], readlane_f(o, l));
                                    if (!within_gthr<IS_L2>(dis, gt[qi]) || !top[qi].admits(dis, v, kd[qi], ki[qi])) {
                                        continue;
                                    }
                                    if (a.bitset != nullptr &&
                                        bitset_filtered(a.bitset, a.bitset_nbits, a.ids[row_off + v])) {
                                        continue;
                                    }
                                    top[qi].insert(dis, v);
                                    ncand[qi]++;
                                    kd[qi] = top[qi].kth_dist();
                                    ki[qi] = top[qi].kth_idx();
                                    tightened = true;
                                    if (h_on && lane_i == 0) { // one more vector at this distance
                                        const uint2 mt = a.gmeta[q_of[qi]];
                                        if (mt.y != KN_HIST_OFF) {
                                            atomicAdd(a.ghist + (int64_t)q_of[qi] * KN_HIST_BINS +
                                                              hist_bin(dist_key<IS_L2>(dis), mt.x, mt.y),
                                                      1u);
                                        }
                                    }
                                }
                                if (tightened && ki[qi] >= 0 && lane_i == 0) {
                                    gthr_publish<IS_L2>(a.gthr + q_of[qi], kd[qi]);
                                }
                                pre[qi] = p4_prefilter<IS_L2>(tighter<IS_L2>(kd[qi], gt[qi]), dis0[qi]);
                            }
                        }
                    }
                }
                if (w == nwin - 1) {
                    break; // the drain window ends here: every lane's last vector is finished
                }
                P4_UNIT(8, B0, U3.x)
                P4_UNIT(9, B1, U3.y)
                P4_UNIT(10, B2, U3.z)
                P4_UNIT(11, B3, U3.w)
                U3 = load_blk(4 * w + 7);
                P4_UNIT(12, B0, U0.x)
                P4_UNIT(13, B1, U0.y)
                P4_UNIT(14, B2, U0.z)
                P4_UNIT(15, B3, U0.w)
                U0 = load_blk(4 * w + 8);
                __builtin_amdgcn_sched_barrier(0);
                o01 = n01;
                o23 = n23;
                n01 = p4_f32x2{0.f, 0.f};
                n23 = p4_f32x2{0.f, 0.f};
            }
#undef P4_UNIT
            __builtin_amdgcn_s_setprio(0);
        }
        P4_T(3); // window loop
        P4_COUNT(6, nwin);

        if (wave == 0) { // park the next item
            if (lane_i < REC_WORDS) {
                ctl[8 + lane_i] = (int)rw_next;
            }
            if (lane_i == 0) {
                ctl[0] = nxt;
            }
        }
        const int nc_any = ncand[0] | ncand[1] | ncand[2] | ncand[3];
        if (nc_any != 0 && lane_i == 0) {
            ctl[1] = 1; // (same value from every writer)
        }
        // ---- merge the waves' lists; wave qi finishes query qi ----------------------------------------------------
        // Partial lists are written SENTINEL-TERMINATED: entries [0, n) and, if n < k, one id = -1 behind them
        // (merge_partials never reads past the first sentinel of a slot).  In the bulk phase most (query, list)
        // pairs contribute nothing: such an item ends with ONE barrier and an 8-byte store per query.
        __syncthreads(); // LUT is dead, every wave's flag write has landed
        P4_T(4);         // wait for the slowest wave's scan
        const bool any_cand = ctl[1] != 0;
        if (!any_cand) {
            if (wave < npair && lane_i == 0) {
#pragma unroll
                for (int qi = 0; qi < QG; qi++) {
                    if (wave == qi) {
                        const int64_t po = ((int64_t)q_of[qi] * a.nslot + slot_of[qi]) * k;
                        a.partial_d[po] = worst_dist<IS_L2>();
                        a.partial_i[po] = -1;
                    }
                }
            }
        } else {
            int* s_cnt = reinterpret_cast<int*>(smem); // [QG][P4_WAVES]
            float* md = reinterpret_cast<float*>(smem + 256);
            int32_t* mi = reinterpret_cast<int32_t*>(smem + 256 + (size_t)QG * P4_WAVES * k * 4);
#pragma unroll
            for (int qi = 0; qi < QG; qi++) {
                const int n = ncand[qi] < k ? ncand[qi] : k;
                if (lane_i == 0) {
                    s_cnt[qi * P4_WAVES + wave] = n;
                }
                if (n > 0) {
                    top[qi].store(md + (qi * P4_WAVES + wave) * k, mi + (qi * P4_WAVES + wave) * k);
                }
            }
            __syncthreads();
            if (wave == 0 && lane_i == 0) {
                ctl[1] = 0; // (read by everyone before the barrier above)
            }
#pragma unroll
            for (int qi = 0; qi < QG; qi++) {
                if (qi < npair && wave == qi) {
                    // the other waves' list lengths in one read; only non-empty lists are walked
                    const int on_l = lane_i < P4_WAVES ? s_cnt[qi * P4_WAVES + lane_i] : 0;
                    unsigned long long nz = __ballot(on_l > 0 && lane_i != wave);
                    while (nz) {
                        const int ow = __ffsll((long long)nz) - 1;
                        nz &= nz - 1;
                        const int on = __builtin_amdgcn_readlane(on_l, ow);
                        const float* od = md + (qi * P4_WAVES + ow) * k;
                        const int32_t* oi = mi + (qi * P4_WAVES + ow) * k;
                        for (int e = 0; e < on; e++) {
                            const float cd = od[e];
                            const int32_t ci = oi[e];
                            if (ci < 0 || !top[qi].admits(cd, ci, kd[qi], ki[qi])) {
                                break;
                            }
                            top[qi].insert(cd, ci);
                            kd[qi] = top[qi].kth_dist();
                            ki[qi] = top[qi].kth_idx();
                        }
                    }
                    if (ki[qi] >= 0 && lane_i == 0) {
                        gthr_publish<IS_L2>(a.gthr + q_of[qi], kd[qi]);
                    }
                    int nvalid = 0; // the list is sorted with its empty entries at the tail
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        nvalid += __popcll(__ballot(r * KN_WAVE + lane_i < k && top[qi].i[r] >= 0));
                    }
                    float* pd = a.partial_d + ((int64_t)q_of[qi] * a.nslot + slot_of[qi]) * k;
                    int64_t* pi = a.partial_i + ((int64_t)q_of[qi] * a.nslot + slot_of[qi]) * k;
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int e = r * KN_WAVE + lane_i;
                        if (e < nvalid) {
                            pd[e] = top[qi].d[r];
                            pi[e] = a.ids[row_off + (int64_t)top[qi].i[r]];
                        } else if (e == nvalid && e < k) {
                            pd[e] = worst_dist<IS_L2>();
                            pi[e] = -1;
                        }
                    }
                }
            }
            __syncthreads(); // the merge scratch aliases the next item's LUT
        }
        P4_T(5); // merge + result write
        P4_COUNT(8, 1);
        cur = p4_sgpr(ctl[0]);
    }
#ifdef KNHIP_PHASE_TIMERS
    if (lane == 0) {
        for (int i = 0; i < 10; i++) {
            atomicAdd(&g_p4_prof[wave * 10 + i], tacc[i]);
        }
    }
#endif
}

template <bool IS_L2, int R>
static hipError_t launch_q4_r(const PqScanArgs& a, int64_t items_bound, hipStream_t s) {
    const size_t merge_bytes = 256 + (size_t)P4_Q * P4_WAVES * a.k * 8;
    const size_t sm = std::max<size_t>(P4_LUT_BYTES, merge_bytes) + P4_CTL_BYTES;
    auto kern = pq_scan_q4_kernel<IS_L2, R>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)sm);
    if (e != hipSuccess) {
        return e;
    }
    hipLaunchKernelGGL(p4_prepare_kernel, dim3((unsigned)((std::max<int64_t>(items_bound, 128) + 255) / 256)), dim3(256),
                       0, s, a, items_bound);
    int dev = 0, ncu = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (ncu <= 0) {
        ncu = 256;
    }
    // one resident workgroup per CU (128 KB of LDS each); never more workgroups than items
    const int64_t wgs = std::max<int64_t>(1, std::min<int64_t>(ncu, items_bound));
#ifdef KNHIP_PHASE_TIMERS
    static unsigned long long zero[160] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_p4_prof), zero, sizeof(zero));
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(P4_THREADS), sm, s, a);
#ifdef KNHIP_PHASE_TIMERS
    (void)hipStreamSynchronize(s);
    unsigned long long h[160];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_p4_prof), sizeof(h));
    fprintf(stderr, "[p4 timers] ticks per item: wave | lut wait1 setup windows wait2 merge top | windows/item slow-ends/item\n");
    for (int w = 0; w < 16; w++) {
        const unsigned long long* r = h + w * 10;
        const double n = r[8] ? (double)r[8] : 1.0;
        fprintf(stderr, "[p4 timers] %2d | %6.0f %6.0f %6.0f %6.0f %6.0f %6.0f %5.0f | %.2f %.3f   (items %llu)\n", w, r[0] / n,
                r[1] / n, r[2] / n, r[3] / n, r[4] / n, r[5] / n, r[7] / n, r[6] / n, r[9] / n, r[8]);
    }
#endif
    return hipGetLastError();
}

// k <= 192 (R = 1, 2, 3).  R = 3 exists for ONE case that matters: the k-th-boundary tie rule searches for k + 1 results
// (knhip_api.hip, search_batch_ties), and a caller's k = 128 must not fall off this kernel onto the systolic one and its
// second copy of the codes.
bool pq_scan_q4_supports(int M, int d, int k) {
    return M == P4_M && d == P4_M * P4_DSUB && k <= 192;
}

hipError_t launch_pq_scan_q4(const PqScanArgs& a, bool is_l2, int64_t items_bound, hipStream_t s) {
    if (items_bound <= 0) {
        return hipSuccess;
    }
    if (a.k <= 64) {
        return is_l2 ? launch_q4_r<true, 1>(a, items_bound, s) : launch_q4_r<false, 1>(a, items_bound, s);
    }
    if (a.k <= 128) {
        return is_l2 ? launch_q4_r<true, 2>(a, items_bound, s) : launch_q4_r<false, 2>(a, items_bound, s);
    }
    return is_l2 ? launch_q4_r<true, 3>(a, items_bound, s) : launch_q4_r<false, 3>(a, items_bound, s);
}

} // namespace knhip
