// knowhere_amd/csrc/pq_scan_v2.hip -- IVF-PQ ADC scan, M = 32, "lane-stationary staggered" form.
//
// Same contract and arithmetic as pq_scan.hip (dis = dis0 + (((0 + LUT[0][c0]) + LUT[1][c1]) ...) in m
// order, bit-equal to PQCodeDistanceScalar, reference
// thirdparty/faiss/faiss/impl/pq_code_distance/pq_code_distance-inl.h:69-90 and
// IVFPQScanner_impl.h:147-150) but re-shaped around what the gfx950 issue rates actually are
// (tools/ubench/valu_rates.hip, measured on MI355X): a VOP2 v_add_f32 issues every ~2 cycles per SIMD,
// v_pk_add_f32 every ~4 (two adds), while the systolic kernel's DPP v_fmac (3.9), v_perm_b32 (3.6) and
// v_cmp_e64 (3.6) made it VALU-issue bound at ~26 cycles per 128 lookups (rocprof: SQ_INSTS_VALU 6.5 per
// step, VALU ~88 % busy).
//
// Here every lane OWNS one vector at a time and walks its 32 sub-quantizers in order m = 0..31, one per
// step, accumulating privately -- nothing crosses lanes.  Lane l of a 32-lane half starts its vector l
// steps late, so at any step the 32 lanes sit on 32 different m: with the LUT stored LUT[code][m][query]
// (8-byte entries) they hit 32 different bank pairs -> still zero LDS bank conflicts.  In a 32-step window
// lanes l <= j are already on the window's new vector, lanes l > j still finish the previous one; two
// accumulator pairs (new / old) and an EXEC mask per step select which one receives the LUT pair:
//     s_mov exec, {l <= j}   ; v_pk_add_f32 acc_new, acc_new, lut      (both queries in one instruction)
//     s_not exec, exec       ; v_pk_add_f32 acc_old, acc_old, lut
// At the end of a window acc_old holds a FINISHED sum in all 64 lanes (64 vectors x 2 queries): one
// compare per query per window, candidates handled wave-parallel.  The LDS address of each lookup
// (code << 8 | m << 3) depends only on (lane, step), so it is baked into the code stream at index build
// time as a 16-bit value: the per-step address "computation" is one v_and / v_lshr.
//
// HBM layout of a list ("stream16"): the list is cut into groups of 64 vectors; lane L's stream is
//   S_L[T] = (code[64 * e + L][m] << 8) | (m << 3),  e = (T - l) / 32,  m = (T - l) mod 32,  l = L mod 32
// (zero before T = l and for vectors past the end), stored as blocks of 8 steps [T/8][64 lanes][8 x u16]:
// one fully contiguous 1 KiB global_load_dwordx4 per wave per 8 steps.  A wave that owns groups [G0, G1)
// runs windows G0 .. G1 (one extra window drains the stagger); the first window's "old" sums belong to
// the previous wave and are ignored.
//
// Per-step budget (2 queries, 128 lookups): 1 VOP2 (address) + 2 v_pk_add_f32 + 1 ds_read_b64 (+3 SALU)
// = ~10 VALU-issue cycles per SIMD and 2 LDS cycles per wave -- both near their limits at 4 waves/SIMD.
#include "common.h"
#include "kernels.h"

#include <cstdlib>

namespace knhip {

constexpr int P2_KSUB = 256;
constexpr int P2_M = 32;
constexpr int P2_WAVES = 8;
constexpr int P2_THREADS = P2_WAVES * KN_WAVE;

typedef float p2_f32x2 __attribute__((ext_vector_type(2)));

// ---- AoS codes [len][32] -> stream16 blocks ------------------------------------------------------
// blocks of 8 steps: uint4 out[blk][lane]; block count per list = stream_blocks(len)
__global__ void pq_stream16_kernel(const uint8_t* __restrict__ codes, const int64_t* __restrict__ list_row_off,
                                   const int64_t* __restrict__ list_len,
                                   const int64_t* __restrict__ list_sblk_off, int64_t nlist,
                                   uint4* __restrict__ out) {
    const int64_t l = blockIdx.y + (int64_t)blockIdx.z * gridDim.y;
    if (l >= nlist) {
        return;
    }
    const int64_t len = list_len[l];
    const int64_t nblk = list_sblk_off[l + 1] - list_sblk_off[l];
    const int64_t row_off = list_row_off[l];
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nblk * 64;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t blk = t / 64;
        const int L = (int)(t % 64);
        const int lo = pq_stream_phase(L);
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const int64_t T = blk * 8 + s;
            uint32_t val = 0;
            if (T >= lo) {
                const int64_t e = (T - lo) / 32;
                const int m = (int)((T - lo) % 32);
                const int64_t v = e * 64 + L;
                uint32_t code = 0;
                if (v < len) {
                    code = codes[(row_off + v) * P2_M + m];
                }
                val = (code << 8) | ((uint32_t)m << 3);
            }
            w[s >> 1] |= val << (16 * (s & 1));
        }
        out[(list_sblk_off[l] + blk) * 64 + L] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

int64_t pq_stream16_blocks(int64_t len) {
    // 32 steps per group of 64 vectors + 31 stagger steps, in blocks of 8 steps, + slack so that every
    // wave's drain window and its code prefetch (two windows + one block ahead) stay inside the list
    const int64_t ngroups = (len + 63) / 64;
    return (ngroups * 32 + 32) / 8 + 16;
}

// ---- 8 steps of accumulate ---------------------------------------------------------------------------
// Split steps (window steps 0..14: lanes of the later phases still finish the previous vector) carry the EXEC
// mask of the lanes already on the new vector (pq_stream_mask, kernels.h) as an immediate.  EXEC is restored
// to all-ones before a block ends; the compiler never sees it changed.
// operands: %0 = new sums, %1 = old sums, %2..%9 = the 8 LUT pairs, %10..%17 = the 8 masks
#define P2_SPLIT(MK, LUT)                                                 \
    "s_mov_b32 exec_lo, " MK "\n\t"                                      \
    "s_mov_b32 exec_hi, " MK "\n\t"                                      \
    "v_pk_add_f32 %0, %0, " LUT "\n\t"                                   \
    "s_not_b64 exec, exec\n\t"                                           \
    "v_pk_add_f32 %1, %1, " LUT "\n\t"
#define P2_PLAIN(LUT) "v_pk_add_f32 %0, %0, " LUT "\n\t"

template <int Q>
__device__ __forceinline__ void p2_accum8(p2_f32x2& an, p2_f32x2& ao, const p2_f32x2 (&v)[8]);

template <>
__device__ __forceinline__ void p2_accum8<0>(p2_f32x2& an, p2_f32x2& ao, const p2_f32x2 (&v)[8]) {
    asm volatile(P2_SPLIT("%10", "%2") P2_SPLIT("%11", "%3") P2_SPLIT("%12", "%4") P2_SPLIT("%13", "%5")
                 P2_SPLIT("%14", "%6") P2_SPLIT("%15", "%7") P2_SPLIT("%16", "%8") P2_SPLIT("%17", "%9")
                 "s_mov_b64 exec, -1\n\t"
                 : "+v"(an), "+v"(ao)
                 : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
                   "i"(pq_stream_mask(0)), "i"(pq_stream_mask(1)), "i"(pq_stream_mask(2)), "i"(pq_stream_mask(3)),
                   "i"(pq_stream_mask(4)), "i"(pq_stream_mask(5)), "i"(pq_stream_mask(6)), "i"(pq_stream_mask(7)));
}
template <>
__device__ __forceinline__ void p2_accum8<1>(p2_f32x2& an, p2_f32x2& ao, const p2_f32x2 (&v)[8]) {
    static_assert(PQ_STREAM_PHASES == 16, "steps 15.. of a window have every lane on the new vector");
    asm volatile(P2_SPLIT("%10", "%2") P2_SPLIT("%11", "%3") P2_SPLIT("%12", "%4") P2_SPLIT("%13", "%5")
                 P2_SPLIT("%14", "%6") P2_SPLIT("%15", "%7") P2_SPLIT("%16", "%8")
                 "s_mov_b64 exec, -1\n\t" P2_PLAIN("%9")
                 : "+v"(an), "+v"(ao)
                 : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
                   "i"(pq_stream_mask(8)), "i"(pq_stream_mask(9)), "i"(pq_stream_mask(10)), "i"(pq_stream_mask(11)),
                   "i"(pq_stream_mask(12)), "i"(pq_stream_mask(13)), "i"(pq_stream_mask(14)), "i"(0));
}
template <>
__device__ __forceinline__ void p2_accum8<2>(p2_f32x2& an, p2_f32x2& ao, const p2_f32x2 (&v)[8]) {
    asm volatile(P2_PLAIN("%2") P2_PLAIN("%3") P2_PLAIN("%4") P2_PLAIN("%5") P2_PLAIN("%6") P2_PLAIN("%7")
                 P2_PLAIN("%8") P2_PLAIN("%9")
                 : "+v"(an), "+v"(ao)
                 : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
}
template <>
__device__ __forceinline__ void p2_accum8<3>(p2_f32x2& an, p2_f32x2& ao, const p2_f32x2 (&v)[8]) {
    p2_accum8<2>(an, ao, v);
}

template <bool IS_L2>
__device__ __forceinline__ float p2_prefilter(float kd, float dis0) {
    const float slack = (fabsf(kd) + fabsf(dis0)) * 4.8e-7f + 1e-30f;
    return IS_L2 ? (kd - dis0) + slack : (kd - dis0) - slack;
}

// DUMP = true: rank-0 phase.  No top-k is kept: every finished distance of the list is written to
// a.dump (row = query, column = offset inside the list) and a radix select picks the k best afterwards
// (launch_rank0_select) -- selecting k of a whole list by sorted insertion is what made k = 100 cost
// twice k = 10 (tools ablation).  The selection also seeds the shared threshold before
// the bulk scan of the remaining probes starts.
// One work item = (list, up to two (query, probe) pairs): LUT build, staggered scan, end-of-item merge.
template <bool IS_L2, int R, bool DUMP>
__device__ __forceinline__ void p2_process_item(const PqScanArgs& a, unsigned char* smem, const int lane, const int wave,
                                                const int tid, const int npair, const int64_t list, const int64_t len,
                                                const int64_t sblk0, const int64_t row_off, const int32_t (&q_of)[2],
                                                const int32_t (&slot_of)[2], const float (&dis0)[2]) {
    constexpr int QG = 2;
    float* lut = reinterpret_cast<float*>(smem); // [256][32][2]

    // candidate histogram of this item's queries: wave j refreshes query j's bound from it; the row is
    // requested here so that its latency hides behind the table loads of the LUT build
    uint32_t h_lo[QG], h_shift[QG];
#pragma unroll
    for (int j = 0; j < QG; j++) {
        h_lo[j] = 0;
        h_shift[j] = KN_HIST_OFF;
        if (!DUMP && a.ghist != nullptr) {
            const uint2 mt = a.gmeta[q_of[j]];
            h_lo[j] = mt.x;
            h_shift[j] = mt.y;
        }
    }
    uint32_t h_cnt = 0;
    const bool h_one = wave == 1; // (selects instead of register-array indexing)
    const int32_t h_q = h_one ? q_of[1] : q_of[0];
    const uint32_t h_lo_w = h_one ? h_lo[1] : h_lo[0], h_shift_w = h_one ? h_shift[1] : h_shift[0];
    const bool h_mine = !DUMP && wave < npair && h_shift_w != KN_HIST_OFF;
    if (h_mine) {
        h_cnt = __hip_atomic_load(a.ghist + (int64_t)h_q * KN_HIST_BINS + lane, __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- LUT[code][m][query] in LDS -------------------------------------------------------------
    if (a.lut_mode == PQ_LUT_RESIDUAL) {
        const int dsub = a.d / P2_M;
        const float* cl = a.centroids + list * a.d;
        for (int e = tid; e < P2_KSUB * P2_M; e += P2_THREADS) {
            const int c = e / P2_M, m = e % P2_M;
            const float* y = a.cb + ((int64_t)m * P2_KSUB + c) * dsub;
#pragma unroll
            for (int j = 0; j < QG; j++) {
                const float* x = a.queries + (int64_t)q_of[j] * a.d + m * dsub;
                float res = 0.f;
                for (int i = 0; i < dsub; i++) {
                    res = l2_step(res, fsub_x(x[i], cl[m * dsub + i]), y[i]);
                }
                lut[e * QG + j] = res;
            }
        }
    } else {
        // 2048 float4 slots, 512 threads: 4 per thread, all 12 loads issued before the first use
        const bool pre = a.lut_mode == PQ_LUT_PRECOMP;
        const float4* pt = reinterpret_cast<const float4*>(a.precomp_t + (pre ? list * (int64_t)(P2_KSUB * P2_M) : 0));
        const float4* ta = reinterpret_cast<const float4*>(a.t2t + (int64_t)q_of[0] * (P2_KSUB * P2_M));
        const float4* tb = reinterpret_cast<const float4*>(a.t2t + (int64_t)q_of[1] * (P2_KSUB * P2_M));
        constexpr int NSLOT4 = P2_KSUB * P2_M / 4;
        constexpr int NU = (NSLOT4 + P2_THREADS - 1) / P2_THREADS;
        float4 xa[NU], xb[NU], pp[NU];
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const int e4 = tid + u * P2_THREADS;
            const bool in = (NSLOT4 % P2_THREADS == 0) || e4 < NSLOT4;
            xa[u] = in ? ta[e4] : make_float4(0.f, 0.f, 0.f, 0.f);
            xb[u] = in ? tb[e4] : make_float4(0.f, 0.f, 0.f, 0.f);
            pp[u] = (pre && in) ? pt[e4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4* l4 = reinterpret_cast<float4*>(lut);
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const int e4 = tid + u * P2_THREADS;
            float4 A = xa[u], B = xb[u];
            if (pre) {
                const float4 p = pp[u];
                A.x = fadd_x(p.x, fmul_x(-2.0f, A.x)); B.x = fadd_x(p.x, fmul_x(-2.0f, B.x));
                A.y = fadd_x(p.y, fmul_x(-2.0f, A.y)); B.y = fadd_x(p.y, fmul_x(-2.0f, B.y));
                A.z = fadd_x(p.z, fmul_x(-2.0f, A.z)); B.z = fadd_x(p.z, fmul_x(-2.0f, B.z));
                A.w = fadd_x(p.w, fmul_x(-2.0f, A.w)); B.w = fadd_x(p.w, fmul_x(-2.0f, B.w));
            }
            if ((NSLOT4 % P2_THREADS == 0) || e4 < NSLOT4) {
                l4[e4 * 2 + 0] = make_float4(A.x, B.x, A.y, B.y);
                l4[e4 * 2 + 1] = make_float4(A.z, B.z, A.w, B.w);
            }
        }
    }
    __syncthreads();
    if ((uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)smem) != 0u) {
        __builtin_trap(); // the baked 16-bit addresses assume the LUT at LDS offset 0
    }

    // ---- this wave's groups --------------------------------------------------------------------------
    const int64_t ngroups = (len + 63) / 64;
    const int64_t gpw = (ngroups + P2_WAVES - 1) / P2_WAVES;
    const int64_t G0 = (int64_t)wave * gpw;
    const int64_t G1 = min(G0 + gpw, ngroups);
    const int64_t nwin = G1 > G0 ? (G1 - G0 + 1) : 0; // one extra window drains the stagger

    WaveTopK<IS_L2, R, int32_t> top[QG];
    float kd[QG], pre[QG], gt[QG];
    int32_t ki[QG];
    int ncand[QG] = {0, 0}; // insertions into this wave's lists (an empty list is the common case in the bulk phase)
#pragma unroll
    for (int j = 0; j < QG; j++) {
        top[j].init(a.k);
        kd[j] = worst_dist<IS_L2>();
        ki[j] = -1;
        gt[j] = gthr_load<IS_L2>(a.gthr + q_of[j]);
        pre[j] = p2_prefilter<IS_L2>(tighter<IS_L2>(kd[j], gt[j]), dis0[j]);
    }

    if (h_mine) {
        // inclusive prefix sum of the 64 bins across the wave; first bin where k vectors are reached
        uint32_t cum = h_cnt;
#pragma unroll
        for (int dlt = 1; dlt < KN_WAVE; dlt <<= 1) {
            const uint32_t up = __shfl_up(cum, dlt, KN_WAVE);
            cum += lane >= dlt ? up : 0u;
        }
        const unsigned long long reach = __ballot(cum >= (uint32_t)a.k);
        const int b = reach ? __ffsll((long long)reach) - 1 : KN_HIST_BINS;
        if (b < KN_HIST_BINS - 1) { // (the last bin also collects everything beyond the range)
            const unsigned long long edge = (unsigned long long)h_lo_w + (((unsigned long long)b + 1ull) << h_shift_w) - 1ull;
            if (edge < 0xffffffffull) {
                const float bound = dist_key_inv<IS_L2>((uint32_t)edge);
                if (bound == bound && fabsf(bound) < FLT_MAX) {
#pragma unroll
                    for (int j = 0; j < QG; j++) {
                        if (j == (h_one ? 1 : 0)) {
                            gt[j] = tighter<IS_L2>(gt[j], bound);
                            pre[j] = p2_prefilter<IS_L2>(tighter<IS_L2>(kd[j], gt[j]), dis0[j]);
                        }
                    }
                    if (lane == 0) {
                        gthr_publish<IS_L2>(a.gthr + h_q, bound);
                    }
                }
            }
        }
    }

    typedef __attribute__((address_space(3))) const p2_f32x2 lds_f2;
    auto lut_read = [&](uint32_t word, int half) -> p2_f32x2 {
        const uint32_t addr = half ? (word >> 16) : (word & 0xffffu);
        return *reinterpret_cast<lds_f2*>(addr);
    };
    // 8 lookups of one code block (uint4 = 8 x u16 addresses)
    auto issue8 = [&](const uint4 w, p2_f32x2 (&v)[8]) {
        v[0] = lut_read(w.x, 0); v[1] = lut_read(w.x, 1);
        v[2] = lut_read(w.y, 0); v[3] = lut_read(w.y, 1);
        v[4] = lut_read(w.z, 0); v[5] = lut_read(w.z, 1);
        v[6] = lut_read(w.w, 0); v[7] = lut_read(w.w, 1);
    };

    if (nwin > 0) {
        const uint4* cbase = a.codes_skew + (sblk0 + G0 * 4) * 64 + lane; // 4 blocks of 8 steps per window
        auto load_blk = [&](int64_t b) { return cbase[b * 64]; };      // past-the-end blocks exist (slack)
        p2_f32x2 an = {0.f, 0.f}, ao = {0.f, 0.f};
        // LUT reads run one block (8 steps) ahead of the accumulate that consumes them (two value
        // buffers); the code stream runs TWO WINDOWS ahead of its first use (three register sets of
        // four blocks, T_w = blocks 4w+1 .. 4w+4 = what window w turns into LUT reads), so that a block
        // that misses L2 has ~64 steps to arrive from HBM.
        p2_f32x2 va[8], vb[8];
        const uint4 b0 = load_blk(0);
        uint4 A0 = load_blk(1), A1 = load_blk(2), A2 = load_blk(3), A3 = load_blk(4);
        uint4 B0 = load_blk(5), B1 = load_blk(6), B2 = load_blk(7), B3 = load_blk(8);
        uint4 C0, C1, C2, C3;
        issue8(b0, va);
        const int64_t last_group = (len + 63) / 64 - 1;
        const unsigned long long tail_mask = (len & 63) ? ((1ull << (len & 63)) - 1ull) : ~0ull;

        // One 32-step window: an accumulates the window's new vectors, ao finishes the previous window's
        // (ao = finished sums of group G0 + w - 1 in every lane at the end).  u0..u3 = T_w, l0..l3
        // receive T_{w+2}.  Called three times per loop trip with the register sets rotated, so no code
        // word is ever copied.
        auto window = [&](const int64_t w, const uint4& u0, const uint4& u1, const uint4& u2, const uint4& u3,
                          uint4& l0, uint4& l1, uint4& l2, uint4& l3) {
            // thresholds published by other waves meanwhile (consumed at the end of this window)
            float gnext[QG];
#pragma unroll
            for (int qi = 0; qi < QG; qi++) {
                gnext[qi] = gthr_load<IS_L2>(a.gthr + q_of[qi]);
            }
            l0 = load_blk(4 * w + 9);
            l1 = load_blk(4 * w + 10);
            l2 = load_blk(4 * w + 11);
            l3 = load_blk(4 * w + 12);
            issue8(u0, vb);
            __builtin_amdgcn_sched_barrier(0);
            p2_accum8<0>(an, ao, va);
            __builtin_amdgcn_sched_barrier(0);
            issue8(u1, va);
            __builtin_amdgcn_sched_barrier(0);
            p2_accum8<1>(an, ao, vb);
            __builtin_amdgcn_sched_barrier(0);
            issue8(u2, vb);
            __builtin_amdgcn_sched_barrier(0);
            p2_accum8<2>(an, ao, va);
            __builtin_amdgcn_sched_barrier(0);
            issue8(u3, va); // block 0 of the next window
            __builtin_amdgcn_sched_barrier(0);
            p2_accum8<3>(an, ao, vb);
            __builtin_amdgcn_sched_barrier(0);
            p2_f32x2& Y = ao;
            // ---- window end --------------------------------------------------------------------------
            if (DUMP) {
                if (w > 0) {
                    const int64_t vbase = (G0 + w - 1) * 64;
                    if (vbase + lane < len) {
                        const bool filt = a.bitset != nullptr &&
                                bitset_filtered(a.bitset, a.bitset_nbits, a.ids[row_off + vbase + lane]);
#pragma unroll
                        for (int qi = 0; qi < QG; qi++) {
                            if (qi < npair) {
                                const float o = qi == 0 ? Y.x : Y.y;
                                a.dump[(int64_t)q_of[qi] * a.dump_stride + (a.dump_by_row ? row_off : 0) + vbase + lane] =
                                        filt ? worst_dist<IS_L2>() : fadd_x(dis0[qi], o);
                            }
                        }
                    }
                }
            } else if (w > 0) {
                // fast path: two compares against the (possibly stale, i.e. looser) prefilter and one
                // "did a shared threshold move" test; everything else only when one of them fires
                const unsigned long long vmask = (G0 + w - 1 == last_group) ? tail_mask : ~0ull;
                const unsigned long long f0 = __ballot(IS_L2 ? (Y.x <= pre[0]) : (Y.x >= pre[0])) & vmask;
                const unsigned long long f1 = __ballot(IS_L2 ? (Y.y <= pre[1]) : (Y.y >= pre[1])) & vmask;
                const unsigned long long moved =
                        __ballot(IS_L2 ? (gnext[0] < gt[0] || gnext[1] < gt[1]) : (gnext[0] > gt[0] || gnext[1] > gt[1]));
                if ((f0 | f1 | moved) != 0) {
                    const int64_t vbase = (G0 + w - 1) * 64;
#pragma unroll
                    for (int qi = 0; qi < QG; qi++) {
                        gt[qi] = tighter<IS_L2>(gt[qi], gnext[qi]);
                        pre[qi] = p2_prefilter<IS_L2>(tighter<IS_L2>(kd[qi], gt[qi]), dis0[qi]);
                        const float o = qi == 0 ? Y.x : Y.y;
                        unsigned long long mm = __ballot(IS_L2 ? (o <= pre[qi]) : (o >= pre[qi])) & vmask;
                        if (mm != 0 && qi < npair) {
                            bool tightened = false;
                            while (mm) {
                                const int l = __ffsll((long long)mm) - 1;
                                mm &= mm - 1;
                                const int32_t v = (int32_t)(vbase + l);
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
                                if (h_shift[qi] != KN_HIST_OFF && lane == 0) { // one more vector at this distance
                                    atomicAdd(a.ghist + (int64_t)q_of[qi] * KN_HIST_BINS +
                                                      hist_bin(dist_key<IS_L2>(dis), h_lo[qi], h_shift[qi]), 1u);
                                }
                            }
                            if (tightened && ki[qi] >= 0 && lane == 0) {
                                gthr_publish<IS_L2>(a.gthr + q_of[qi], kd[qi]);
                            }
                            pre[qi] = p2_prefilter<IS_L2>(tighter<IS_L2>(kd[qi], gt[qi]), dis0[qi]);
                        }
                    }
                }
            }
            ao = an;
            an = p2_f32x2{0.f, 0.f};
        };

        for (int64_t w = 0; w < nwin; w += 3) {
            window(w, A0, A1, A2, A3, C0, C1, C2, C3);
            if (w + 1 >= nwin) {
                break;
            }
            window(w + 1, B0, B1, B2, B3, A0, A1, A2, A3);
            if (w + 2 >= nwin) {
                break;
            }
            window(w + 2, C0, C1, C2, C3, B0, B1, B2, B3);
        }
    }

    if (DUMP) {
        return;
    }
    // ---- merge the waves' lists; wave qi finishes query qi ---------------------------------------------
    // Partial lists are written SENTINEL-TERMINATED: entries [0, n) and, if n < k, one id = -1 behind them
    // (merge_partials never reads past the first sentinel of a slot).  In the bulk phase most (query, list)
    // pairs contribute nothing: that case costs one barrier pair, eight LDS counters and one 8-byte store.
    __syncthreads(); // LUT is dead
    const int k = a.k;
    int* s_cnt = reinterpret_cast<int*>(smem); // [QG][P2_WAVES]
    float* md = reinterpret_cast<float*>(smem + 64);
    int64_t* mi = reinterpret_cast<int64_t*>(smem + 64 + (((size_t)QG * P2_WAVES * k * 4 + 7) & ~(size_t)7));
#pragma unroll
    for (int qi = 0; qi < QG; qi++) {
        const int n = ncand[qi] < k ? ncand[qi] : k;
        if (lane == 0) {
            s_cnt[qi * P2_WAVES + wave] = n;
        }
        if (n > 0) {
            top[qi].store(md + (qi * P2_WAVES + wave) * k, mi + (qi * P2_WAVES + wave) * k);
        }
    }
    __syncthreads();
#pragma unroll
    for (int qi = 0; qi < QG; qi++) {
        if (qi < npair && wave == qi) {
            for (int w = 1; w < P2_WAVES; w++) {
                const int ow = (wave + w) % P2_WAVES;
                const int on = s_cnt[qi * P2_WAVES + ow];
                const float* od = md + (qi * P2_WAVES + ow) * k;
                const int64_t* oi = mi + (qi * P2_WAVES + ow) * k;
                for (int e = 0; e < on; e++) {
                    const float cd = od[e];
                    const int32_t ci = (int32_t)oi[e];
                    if (ci < 0 || !top[qi].admits(cd, ci, kd[qi], ki[qi])) {
                        break;
                    }
                    top[qi].insert(cd, ci);
                    kd[qi] = top[qi].kth_dist();
                    ki[qi] = top[qi].kth_idx();
                }
            }
            if (ki[qi] >= 0 && lane == 0) {
                gthr_publish<IS_L2>(a.gthr + q_of[qi], kd[qi]);
            }
            int nvalid = 0; // the list is sorted with its empty entries at the tail
#pragma unroll
            for (int r = 0; r < R; r++) {
                nvalid += __popcll(__ballot(r * KN_WAVE + lane < k && top[qi].i[r] >= 0));
            }
            float* pd = a.partial_d + ((int64_t)q_of[qi] * a.nslot + slot_of[qi]) * k;
            int64_t* pi = a.partial_i + ((int64_t)q_of[qi] * a.nslot + slot_of[qi]) * k;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int e = r * KN_WAVE + lane;
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
}

// ---- one workgroup per work item (grid = item bound), XCD-aware block -> item mapping ---------------------
template <bool IS_L2, int R, bool DUMP>
__global__ __launch_bounds__(P2_THREADS, 4) void pq_scan_v2_kernel(PqScanArgs a) {
    constexpr int QG = 2;
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / KN_WAVE); // wave-uniform: scalar loop control

    const int64_t item_lo = a.item_lo ? *a.item_lo : 0;
    const int64_t nitems = *a.item_hi - item_lo;
    if ((int64_t)blockIdx.x >= ((nitems + 7) / 8) * 8) {
        return;
    }
    const int64_t item_rel = xcd_item(blockIdx.x, nitems);
    if (item_rel >= nitems) {
        return;
    }
    const KnItem it = a.items[item_lo + item_rel];
    const int npair = it.npair < QG ? it.npair : QG;
    const int64_t list = it.list;
    const int64_t len = a.list_len[list];
    const int64_t sblk0 = a.list_sblk_off[list];
    const int64_t row_off = a.list_row_off[list];
    int32_t q_of[QG], slot_of[QG];
    float dis0[QG];
#pragma unroll
    for (int j = 0; j < QG; j++) {
        const KnPair p = a.pairs[it.pair0 + (j < npair ? j : npair - 1)];
        q_of[j] = p.q;
        slot_of[j] = p.slot;
        dis0[j] = (a.lut_mode == PQ_LUT_RESIDUAL) ? 0.f : a.coarse_dis[(int64_t)p.q * a.nslot + p.slot];
    }
    p2_process_item<IS_L2, R, DUMP>(a, smem, lane, wave, (int)threadIdx.x, npair, list, len, sblk0, row_off, q_of, slot_of,
                                    dis0);
}

template <bool IS_L2, int R, bool DUMP>
static hipError_t launch_v2_r(const PqScanArgs& a, int64_t grid, hipStream_t s) {
    const size_t lut_bytes = (size_t)P2_KSUB * 256;
    const size_t merge_bytes = 64 + (((size_t)2 * P2_WAVES * a.k * 4 + 7) & ~(size_t)7) + (size_t)2 * P2_WAVES * a.k * 8;
    const size_t sm = DUMP ? lut_bytes : std::max(lut_bytes, merge_bytes);
    auto kern = pq_scan_v2_kernel<IS_L2, R, DUMP>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)sm);
    if (e != hipSuccess) {
        return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(P2_THREADS), sm, s, a);
    return hipGetLastError();
}

// k <= 192 (R = 1, 2, 3; R = 3 for the tie rule's k + 1 at a caller's k = 128: see pq_scan_q4_supports); larger k stays on
// the systolic kernel and its layout
bool pq_scan_v2_supports(int M, int k) {
    return M == P2_M && k <= 192;
}

hipError_t launch_pq_scan_v2(const PqScanArgs& a, bool is_l2, bool dump, int64_t grid, hipStream_t s) {
    if (grid <= 0) {
        return hipSuccess;
    }
    grid = (grid + 7) / 8 * 8; // (xcd_item: blocks [0, round_up(nitems, 8)), see launch_pq_scan)
    if (dump) {
        return is_l2 ? launch_v2_r<true, 1, true>(a, grid, s) : launch_v2_r<false, 1, true>(a, grid, s);
    }
    if (a.k <= 64) {
        return is_l2 ? launch_v2_r<true, 1, false>(a, grid, s) : launch_v2_r<false, 1, false>(a, grid, s);
    }
    if (a.k <= 128) {
        return is_l2 ? launch_v2_r<true, 2, false>(a, grid, s) : launch_v2_r<false, 2, false>(a, grid, s);
    }
    return is_l2 ? launch_v2_r<true, 3, false>(a, grid, s) : launch_v2_r<false, 3, false>(a, grid, s);
}

// ---- rank-0 epilogue: list offsets -> ids, partial slot 0, shared threshold -------------------------------
template <bool IS_L2>
__global__ void rank0_finalize_kernel(const int64_t* __restrict__ sel_off, const float* __restrict__ sel_d,
                                      const int64_t* __restrict__ keys, int nprobe,
                                      const int64_t* __restrict__ list_row_off, const int64_t* __restrict__ ids,
                                      int64_t nq, int k, float* __restrict__ partial_d,
                                      int64_t* __restrict__ partial_i, float* __restrict__ gthr) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nq * k) {
        return;
    }
    const int64_t q = t / k;
    const int e = (int)(t % k);
    const int64_t key = keys[q * nprobe];
    int64_t off = sel_off[t];
    float d = sel_d[t];
    // a dumped sentinel (filtered vector) must not surface as a result
    if (off >= 0 && (IS_L2 ? !(d < worst_dist<IS_L2>()) : !(d > worst_dist<IS_L2>()))) {
        off = -1;
    }
    const int64_t id = (off >= 0 && key >= 0) ? ids[list_row_off[key] + off] : -1;
    partial_d[(q * nprobe + 0) * k + e] = id >= 0 ? d : worst_dist<IS_L2>();
    partial_i[(q * nprobe + 0) * k + e] = id;
    if (e == k - 1 && id >= 0) {
        gthr[q] = d; // k real candidates: their k-th distance bounds the final k-th (before the bulk scan starts)
    }
}

// ---- rank-0 epilogue 2: seed the per-query candidate histogram ---------------------------------------------
// One wave per query.  The k selected distances of the closest list span [best, k-th]; 64 bins over that key
// range (bin = (key - key(best)) >> shift).  A later reader that finds cum(bins <= b) >= k knows that k
// vectors are at least as good as the upper edge of bin b: a valid, ever tightening bound on the final k-th
// distance long before any single workgroup has k candidates of its own.
template <bool IS_L2>
__global__ __launch_bounds__(64) void rank0_hist_kernel(const int64_t* __restrict__ sel_off,
                                                        const float* __restrict__ sel_d, int64_t nq, int k,
                                                        uint32_t* __restrict__ ghist, uint2* __restrict__ gmeta) {
    __shared__ uint32_t h[KN_HIST_BINS];
    const int64_t q = blockIdx.x;
    const int lane = threadIdx.x;
    h[lane] = 0;
    __syncthreads();
    auto valid = [&](int e) {
        const float d = sel_d[q * k + e];
        return sel_off[q * k + e] >= 0 && (IS_L2 ? (d < worst_dist<IS_L2>()) : (d > worst_dist<IS_L2>()));
    };
    uint32_t lo = 0, shift = KN_HIST_OFF;
    if (valid(k - 1)) { // the list is sorted best-first: k real candidates
        lo = dist_key<IS_L2>(sel_d[q * k]);
        const uint32_t range = dist_key<IS_L2>(sel_d[q * k + k - 1]) - lo;
        shift = 0;
        while ((range >> shift) >= (uint32_t)(KN_HIST_BINS - 1)) {
            shift++;
        }
        for (int e = lane; e < k; e += 64) {
            atomicAdd(&h[hist_bin(dist_key<IS_L2>(sel_d[q * k + e]), lo, shift)], 1u);
        }
    }
    __syncthreads();
    ghist[q * KN_HIST_BINS + lane] = h[lane];
    if (lane == 0) {
        gmeta[q] = make_uint2(lo, shift);
    }
}

hipError_t launch_rank0_select(const float* dump, int64_t dump_stride, const int64_t* keys, int nprobe,
                               const int64_t* list_len, const int64_t* list_row_off, const int64_t* ids,
                               int64_t nq, int k, bool is_l2, float* partial_d, int64_t* partial_i, float* gthr,
                               int64_t* tmp_keys, float* tmp_d, uint32_t* ghist, uint2* gmeta, hipStream_t s) {
    if (nq <= 0) {
        return hipSuccess;
    }
    hipError_t e = launch_row_select_var(dump, dump_stride, keys, nprobe, list_len, nq, k, is_l2, tmp_keys, tmp_d, s);
    if (e != hipSuccess) {
        return e;
    }
    const unsigned grid = (unsigned)((nq * k + 255) / 256);
    if (is_l2) {
        hipLaunchKernelGGL((rank0_finalize_kernel<true>), dim3(grid), dim3(256), 0, s, tmp_keys, tmp_d, keys, nprobe,
                           list_row_off, ids, nq, k, partial_d, partial_i, gthr);
    } else {
        hipLaunchKernelGGL((rank0_finalize_kernel<false>), dim3(grid), dim3(256), 0, s, tmp_keys, tmp_d, keys, nprobe,
                           list_row_off, ids, nq, k, partial_d, partial_i, gthr);
    }
    if (ghist != nullptr) {
        if (is_l2) {
            hipLaunchKernelGGL((rank0_hist_kernel<true>), dim3((unsigned)nq), dim3(64), 0, s, tmp_keys, tmp_d, nq, k,
                               ghist, gmeta);
        } else {
            hipLaunchKernelGGL((rank0_hist_kernel<false>), dim3((unsigned)nq), dim3(64), 0, s, tmp_keys, tmp_d, nq, k,
                               ghist, gmeta);
        }
    }
    return hipGetLastError();
}

hipError_t launch_pq_stream16(const uint8_t* codes, const int64_t* list_row_off, const int64_t* list_len,
                              const int64_t* list_sblk_off, int64_t nlist, uint4* out, hipStream_t s) {
    if (nlist <= 0) {
        return hipSuccess;
    }
    const unsigned gy = (unsigned)std::min<int64_t>(nlist, 32768);
    const unsigned gz = (unsigned)((nlist + gy - 1) / gy);
    hipLaunchKernelGGL(pq_stream16_kernel, dim3(8, gy, gz), dim3(256), 0, s, codes, list_row_off, list_len,
                       list_sblk_off, nlist, out);
    return hipGetLastError();
}

} // namespace knhip
