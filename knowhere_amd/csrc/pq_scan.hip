// knowhere_amd/csrc/pq_scan.hip -- IVF-PQ query tables + ADC list scan for gfx950.
//
// Replaces, on the device:
//   * ProductQuantizer::compute_inner_prod_table      (thirdparty/faiss/faiss/impl/ProductQuantizer.cpp:471-485)
//   * initialize_IVFPQ_precomputed_table (mode 1)     (thirdparty/faiss/faiss/IndexIVFPQ.cpp:465-484)
//   * QueryTables::precompute_list_tables_L2 / _IP    (.../impl/pq_code_distance/IVFPQ_QueryTables.cpp:110-145)
//   * IVFPQScannerT::scan_list_with_table + PQCodeDistanceScalar
//                                                     (.../IVFPQScanner_impl.h:109-181, pq_code_distance-inl.h:69-90)
//   * HeapResultHandler admission / ordering          (.../impl/ResultHandler.h:258-279, utils/Heap.h:112-151)
//
// Arithmetic contract (bit-equal to the scalar reference):
//   LUT[m][c]   = precomp[list][m][c] + (-2) * <q_m, cb[m][c]>        (L2, precomputed table)
//               = <q_m, cb[m][c]>                                      (IP)
//               = || (q - c_list)_m - cb[m][c] ||^2                    (L2, residual tables)
//   dis         = dis0 + ( ... ((0 + LUT[0][c0]) + LUT[1][c1]) ... + LUT[M-1][c_{M-1}] )
// i.e. the M table values are summed SEQUENTIALLY in m order starting from zero and dis0 is
// added last -- the order of PQCodeDistanceScalar::distance_single_code followed by
// `dis0 + distance` in scan_list_with_table.
//
// Design: a wave-wide SYSTOLIC ADC pipeline.
//   A classic GPU ADC gives each lane one code and lets it gather LUT[m][code[m]] for m = 0..M-1:
//   64 random LDS addresses per instruction, i.e. ~3.5-way bank conflicts on every lookup.  Here
//   lane l instead OWNS sub-quantizer m = l % M for the whole scan: at step t lane m looks up the
//   code byte of vector (t - m), adds it to the partial sum handed over from lane m-1 and hands the
//   result to lane m+1 with a DPP shift.  The LUT is stored transposed, LUT[code][slot], with
//   slot = lane % 32, so the 32 lanes of an LDS access group always hit 32 different bank pairs:
//   ZERO bank conflicts regardless of the code values.  With M = 32 a wave64 is exactly two
//   32-stage pipelines; each retires one finished distance per step, in reference summation order
//   (the partial sum visits m = 0, 1, ... in order), so the result is bit-equal to the CPU sum.
//   The inner step is   v_perm (LDS address) / ds_read_b64 / 2 x v_fmac_f32_dpp / 2 x v_cmp
//   for TWO queries at once: the b64 read fetches LUT entries of two queries that probe the same
//   list (entries interleaved), doubling the LDS bytes per cycle (256 B/clk vs 128 for b32).
//     acc_new = fma(shift(acc), mask, lut)  with mask = 0 in lane 0 of a pipe, else 1:
//     fma(x, 1, v) rounds once == x + v;  fma(x, 0, v) == v restarts the sum exactly.
//
// HBM layout of a list's codes ("skewed"): stored row r, column m holds code[r - m][m] (0 outside
// the list), in blocks of 16 stored rows laid out [M][16 bytes]: lane m fetches its next 16 steps
// with one coalesced 16-byte load.  A pipe that owns vectors [V0, V1) consumes stored rows
// [V0, V1 + M - 1); the first M-1 outputs of a run are discarded (they belong to the previous
// pipe), so no per-pipe duplication of storage is needed.
//
// Work item = (list, 1 or 2 (query, probe-rank) pairs).  Items are sorted by list and mapped
// XCD-aware, so a list's codes are read from HBM about once per XCD and then from its L2; the
// scan is LDS/VALU-issue bound, not HBM bound (DESIGN.md section 4 has the cycle budget).
#include "common.h"
#include "kernels.h"
#include "knhip_env.h"

#include <cstdlib>

namespace knhip {

constexpr int PQ_KSUB = 256;

#ifndef KN_PQ_M
// ---- query tables: T2T[q][c][m] = <q_m, cb[m][c]>  (sequential over dsub) ----------------------
__global__ __launch_bounds__(256) void pq_query_table_kernel(const float* __restrict__ queries,
                                                             const float* __restrict__ cb, int d,
                                                             int M, float* __restrict__ t2t) {
    // codebook reads in its own [m][c][dsub] order (a wave covers whole rows), results staged in LDS as
    // [c][M + 1] (padded: conflict-free) and written out [c][m] fully coalesced
    extern __shared__ float sqv[]; // [d] query, then [256][M + 1] staging
    float* stage = sqv + d;
    const int64_t q = blockIdx.x;
    const int dsub = d / M;
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
        sqv[i] = queries[q * d + i];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < PQ_KSUB * M; e += blockDim.x) {
        const int m = e / PQ_KSUB, c = e % PQ_KSUB;
        const float* y = cb + ((int64_t)m * PQ_KSUB + c) * dsub;
        const float* x = sqv + m * dsub;
        float res = 0.f;
        for (int i = 0; i < dsub; i++) {
            res = ip_step(res, x[i], y[i]);
        }
        stage[c * (M + 1) + m] = res;
    }
    __syncthreads();
    float* out = t2t + q * (int64_t)(PQ_KSUB * M);
    for (int e = threadIdx.x; e < PQ_KSUB * M; e += blockDim.x) {
        out[e] = stage[(e / M) * (M + 1) + (e % M)];
    }
}

// ---- precomputed term-2 table, transposed: PT[list][c][m] = ||cb[m][c]||^2 + 2 * <c_list,m , cb[m][c]>
__global__ __launch_bounds__(256) void pq_precomp_table_kernel(const float* __restrict__ centroids,
                                                               const float* __restrict__ cb, int d,
                                                               int M, float* __restrict__ pt) {
    extern __shared__ float scv[];
    const int64_t l = blockIdx.x;
    const int dsub = d / M;
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
        scv[i] = centroids[l * d + i];
    }
    __syncthreads();
    float* out = pt + l * (int64_t)(PQ_KSUB * M);
    for (int e = threadIdx.x; e < PQ_KSUB * M; e += blockDim.x) {
        const int c = e / M, m = e % M;
        const float* y = cb + ((int64_t)m * PQ_KSUB + c) * dsub;
        const float* x = scv + m * dsub;
        float nrm = 0.f, ip = 0.f;
        for (int i = 0; i < dsub; i++) {
            nrm = ip_step(nrm, y[i], y[i]);  // fvec_norm_L2sqr: res += x*x
        }
        for (int i = 0; i < dsub; i++) {
            ip = ip_step(ip, x[i], y[i]);
        }
        out[e] = fadd_x(nrm, fmul_x(2.0f, ip)); // fvec_madd(r_norms, 2.0, tab, tab)
    }
}

// ---- AoS codes [len][M] -> skewed blocks --------------------------------------------------------
// One thread per (block, lane m): builds the 16 bytes lane m consumes for stored rows 16*blk..+15.
__global__ void pq_skew_codes_kernel(const uint8_t* __restrict__ codes, const int64_t* __restrict__ list_row_off,
                                     const int64_t* __restrict__ list_len,
                                     const int64_t* __restrict__ list_sblk_off, int64_t nlist, int M,
                                     uint4* __restrict__ out) {
    const int64_t l = blockIdx.y + (int64_t)blockIdx.z * gridDim.y;
    if (l >= nlist) {
        return;
    }
    const int64_t len = list_len[l];
    const int64_t nsblk = list_sblk_off[l + 1] - list_sblk_off[l];
    const int64_t row_off = list_row_off[l];
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nsblk * M;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t blk = t / M;
        const int m = (int)(t % M);
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int64_t v = blk * 16 + j - m;
            uint32_t byte = 0;
            if (v >= 0 && v < len) {
                byte = codes[(row_off + v) * M + m];
            }
            w[j >> 2] |= byte << (8 * (j & 3));
        }
        out[(list_sblk_off[l] + blk) * M + m] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

#endif // !KN_PQ_M

// ---- the systolic step ---------------------------------------------------------------------------
// vA/vB come in holding the LUT values of this step and leave holding the new partial sums.
// Hazard bookkeeping (the compiler pads nothing inside an asm string): a DPP source VGPR written
// by a VALU needs 2 wait states before the DPP read.  The chain acc(t) -> acc(t+1) is the only
// such dependence; each block therefore opens with an s_nop sized so that, even if two dependent
// blocks end up back to back, {other fmac of the previous block, s_nop} give the 2 wait states.
// The asm is NOT volatile: it is a pure function of its operands, so the scheduler is free to
// hoist the independent ds_reads of later steps above it (a volatile asm pins every LDS read
// behind an lgkmcnt(0) wait: one read in flight, latency-bound).
template <int M>
__device__ __forceinline__ void systolic_step2(float& vA, float& vB, float accA, float accB,
                                               float mask) {
    if (M >= 32) {
        asm("s_nop 0\n\t"
                "v_fmac_f32_dpp %0, %2, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_fmac_f32_dpp %1, %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                : "+v"(vA), "+v"(vB)
                : "v"(accA), "v"(accB), "v"(mask));
    } else {
        asm("s_nop 0\n\t"
                "v_fmac_f32_dpp %0, %2, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_fmac_f32_dpp %1, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                : "+v"(vA), "+v"(vB)
                : "v"(accA), "v"(accB), "v"(mask));
    }
}
template <int M>
__device__ __forceinline__ void systolic_step1(float& vA, float accA, float mask) {
    if (M >= 32) {
        asm("s_nop 1\n\t"
                "v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                : "+v"(vA)
                : "v"(accA), "v"(mask));
    } else {
        asm("s_nop 1\n\t"
                "v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                : "+v"(vA)
                : "v"(accA), "v"(mask));
    }
}


template <bool IS_L2>
__device__ __forceinline__ float prefilter_bound(float kd, float dis0) {
    // acc passes the exact test (dis0 + acc better-or-equal kd) ==> acc passes `acc <= bound`
    // (L2) / `acc >= bound` (IP).  fl(dis0 + acc) is monotone in acc; two ulps of the larger
    // magnitude cover the rounding of both the subtraction here and the addition there.
    const float slack = (fabsf(kd) + fabsf(dis0)) * 4.8e-7f + 1e-30f;
    return IS_L2 ? (kd - dis0) + slack : (kd - dis0) - slack;
}

// PQ_WAVES waves share one LUT.  The LUT (64 KB) caps a CU at two workgroups, so the waves per
// SIMD -- what hides the latency of the dependent fmac chain -- are set by the workgroup size:
// 4 waves -> 2 per SIMD, 8 -> 4, 16 -> 8.  More waves also means more pipes per list, i.e. more
// run-up steps (M-1 per pipe): measured trade-off in DESIGN.md.
// second launch-bounds argument = waves per SIMD the register allocator must leave room for:
// two workgroups per CU (the LDS limit) = 2 * PQ_WAVES / 4 waves per SIMD, capped at 4 (128 VGPRs)
template <bool IS_L2, int M, int QG, int R, int PQ_WAVES>
__global__ __launch_bounds__(PQ_WAVES * 64, (PQ_WAVES >= 8 ? 4 : 2)) void pq_scan_kernel(PqScanArgs a) {
    constexpr int PQ_THREADS = PQ_WAVES * KN_WAVE;
    static_assert(64 % M == 0, "M must divide the wave");
    static_assert(QG == 1 || QG == 2, "");
    static_assert(!(M == 64 && QG == 2), "M=64 tables take the whole LUT budget");
    constexpr int P = 64 / M;                    // pipes per wave
    constexpr int NP = P * PQ_WAVES;             // pipes per workgroup
    constexpr int NSLOT = (M == 64) ? 64 : 32;   // LUT columns (bank pairs)
    constexpr int ENTRY_B = QG * 4;              // bytes per LUT entry
    static_assert(NSLOT * ENTRY_B == 256, "LDS address = code << 8 | slot * ENTRY_B");
    constexpr int RUNUP = M - 1;

    extern __shared__ __align__(16) unsigned char smem[];
    float* lut = reinterpret_cast<float*>(smem); // [256][NSLOT][QG]

    const int lane = lane_id();
    const int wave = threadIdx.x / KN_WAVE;

    const int64_t nitems = *a.nitems_dev;
    if ((int64_t)blockIdx.x >= ((nitems + 7) / 8) * 8) {
        return;
    }
    const int64_t item = xcd_item(blockIdx.x, nitems);
    if (item >= nitems) {
        return;
    }
    const KnItem it = a.items[item];
    const int npair = it.npair < QG ? it.npair : QG;
    const int64_t list = it.list;
    const int64_t len = a.list_len[list];
    const int64_t sblk0 = a.list_sblk_off[list];
    const int64_t nsblk = a.list_sblk_off[list + 1] - sblk0;
    const int64_t row_off = a.list_row_off[list];

    int32_t q_of[QG], slot_of[QG];
    float dis0[QG];
#pragma unroll
    for (int j = 0; j < QG; j++) {
        const KnPair p = a.pairs[it.pair0 + (j < npair ? j : npair - 1)];
        q_of[j] = p.q;
        slot_of[j] = p.slot;
        dis0[j] = (a.lut_mode == PQ_LUT_RESIDUAL) ? 0.f
                                                  : a.coarse_dis[(int64_t)p.q * a.nslot + p.slot];
    }

    // ---- build the per-(query, list) LUT in LDS, transposed + replicated to 32 slots ----------
    if (a.lut_mode == PQ_LUT_RESIDUAL) {
        const int dsub = a.d / M;
        const float* cl = a.centroids + list * a.d;
        for (int e = threadIdx.x; e < PQ_KSUB * NSLOT; e += PQ_THREADS) {
            const int c = e / NSLOT, s = e % NSLOT, m = s % M;
            const float* y = a.cb + ((int64_t)m * PQ_KSUB + c) * dsub;
#pragma unroll
            for (int j = 0; j < QG; j++) {
                const float* x = a.queries + (int64_t)q_of[j] * a.d + m * dsub;
                float res = 0.f;
                for (int i = 0; i < dsub; i++) {
                    const float r = fsub_x(x[i], cl[m * dsub + i]); // compute_residual
                    res = l2_step(res, r, y[i]);
                }
                lut[e * QG + j] = res;
            }
        }
    } else if (M == NSLOT && QG == 2) {
        // fast path: source tables are already [c][m] == [c][slot]; 4 entries per thread-iteration
        const float4* pt = (a.lut_mode == PQ_LUT_PRECOMP)
                ? reinterpret_cast<const float4*>(a.precomp_t + list * (int64_t)(PQ_KSUB * M))
                : nullptr;
        const float4* ta = reinterpret_cast<const float4*>(a.t2t + (int64_t)q_of[0] * (PQ_KSUB * M));
        const float4* tb = reinterpret_cast<const float4*>(a.t2t + (int64_t)q_of[QG - 1] * (PQ_KSUB * M));
        float4* l4 = reinterpret_cast<float4*>(lut);
        for (int e4 = threadIdx.x; e4 < PQ_KSUB * M / 4; e4 += PQ_THREADS) {
            float4 xa = ta[e4], xb = tb[e4];
            if (a.lut_mode == PQ_LUT_PRECOMP) {
                const float4 p = pt[e4];
                xa.x = fadd_x(p.x, fmul_x(-2.0f, xa.x)); xb.x = fadd_x(p.x, fmul_x(-2.0f, xb.x));
                xa.y = fadd_x(p.y, fmul_x(-2.0f, xa.y)); xb.y = fadd_x(p.y, fmul_x(-2.0f, xb.y));
                xa.z = fadd_x(p.z, fmul_x(-2.0f, xa.z)); xb.z = fadd_x(p.z, fmul_x(-2.0f, xb.z));
                xa.w = fadd_x(p.w, fmul_x(-2.0f, xa.w)); xb.w = fadd_x(p.w, fmul_x(-2.0f, xb.w));
            }
            l4[e4 * 2 + 0] = make_float4(xa.x, xb.x, xa.y, xb.y);
            l4[e4 * 2 + 1] = make_float4(xa.z, xb.z, xa.w, xb.w);
        }
    } else {
        for (int e = threadIdx.x; e < PQ_KSUB * NSLOT; e += PQ_THREADS) {
            const int c = e / NSLOT, s = e % NSLOT, m = s % M;
#pragma unroll
            for (int j = 0; j < QG; j++) {
                float v = a.t2t[(int64_t)q_of[j] * (PQ_KSUB * M) + c * M + m];
                if (a.lut_mode == PQ_LUT_PRECOMP) {
                    v = fadd_x(a.precomp_t[list * (int64_t)(PQ_KSUB * M) + c * M + m],
                               fmul_x(-2.0f, v));
                }
                lut[e * QG + j] = v;
            }
        }
    }
    __syncthreads();

    // ---- per-pipe vector ranges ----------------------------------------------------------------
    const int64_t len16 = (len + 15) / 16;                        // vector blocks in the list
    const int64_t per_pipe_blk = (len16 + NP - 1) / NP;           // blocks per pipe
    const int64_t per_pipe = per_pipe_blk * 16;
    const int pipe_in_wave = lane / M;
    const int m_lane = lane % M;
    const int64_t V0 = ((int64_t)wave * P + pipe_in_wave) * per_pipe; // this lane's pipe
    const int64_t nrun_blk = per_pipe_blk + (RUNUP + 15) / 16;    // blocks each pipe steps through
    const float mask = (m_lane == 0) ? 0.f : 1.f;
    const uint32_t laneoff = (uint32_t)(lane % NSLOT) * ENTRY_B;
    unsigned long long outmask = 0;
#pragma unroll
    for (int p = 0; p < P; p++) {
        outmask |= 1ull << (p * M + M - 1);
    }

    WaveTopK<IS_L2, R, int32_t> top[QG]; // ordered by the offset inside the list (< 2^31)
    float kd[QG], pre[QG], gt[QG];
    int32_t ki[QG];
#pragma unroll
    for (int j = 0; j < QG; j++) {
        top[j].init(a.k);
        kd[j] = worst_dist<IS_L2>();
        ki[j] = -1;
        gt[j] = gthr_load<IS_L2>(a.gthr + q_of[j]);
        pre[j] = prefilter_bound<IS_L2>(tighter<IS_L2>(kd[j], gt[j]), dis0[j]);
    }

    float accA = 0.f, accB = 0.f;
    const uint4* cbase = a.codes_skew + sblk0 * M + m_lane;
    const int64_t blk_first = V0 / 16;
    auto load_blk = [&](int64_t rb) {
        int64_t blk = blk_first + rb;
        blk = blk < nsblk ? blk : nsblk - 1; // past-the-end pipes re-read the zero tail (masked)
        return cbase[blk * M];
    };
    // The LUT sits at LDS offset 0 (this kernel declares no static LDS), so the permuted word
    // (code << 8 | slot * ENTRY_B) IS the LDS address: no per-step base add.
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) const f32x2_t lds_f2;
    typedef __attribute__((address_space(3))) const float lds_f1;
    if ((uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)smem) != 0u) {
        __builtin_trap();
    }
    struct LutVal { // one step's LUT entry: .x query A, .y query B (QG == 1: .y unused)
        float x, y;
    };
    auto lut_read = [&](const uint4 w, int j) {
        const uint32_t ww = (j < 4) ? w.x : (j < 8) ? w.y : (j < 12) ? w.z : w.w;
        // LDS byte address = code << 8 | slot * ENTRY_B   (one v_perm_b32)
        const uint32_t addr =
                __builtin_amdgcn_perm(ww, laneoff, 0x0c0c0000u | ((4u + (uint32_t)(j & 3)) << 8));
        LutVal r;
        if (QG == 2) {
            const f32x2_t v = *reinterpret_cast<lds_f2*>(addr);
            r.x = v.x;
            r.y = v.y;
        } else {
            r.x = *reinterpret_cast<lds_f1*>(addr);
            r.y = 0.f;
        }
        return r;
    };

    // Software pipeline, one block (16 steps) deep: while block rb is consumed (cur[], already
    // in registers or in flight), the LUT reads of block rb+1 are issued one per step into nxt[].
    // sched_barrier pins that interleave -- left alone, the scheduler sinks every ds_read down to
    // its use and the loop becomes LDS-latency bound with one read in flight.
    auto run_block = [&](LutVal (&cur)[16], LutVal (&nxt)[16], const uint4 wnext, const int64_t rb,
                         const bool live) {
        unsigned long long hit = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            nxt[j] = lut_read(wnext, j);
            if (QG == 2) {
                systolic_step2<M>(cur[j].x, cur[j].y, accA, accB, mask);
                accA = cur[j].x;
                accB = cur[j].y;
                hit |= __ballot(IS_L2 ? (accA <= pre[0]) : (accA >= pre[0])) |
                       __ballot(IS_L2 ? (accB <= pre[QG - 1]) : (accB >= pre[QG - 1]));
            } else {
                systolic_step1<M>(cur[j].x, accA, mask);
                accA = cur[j].x;
                hit |= __ballot(IS_L2 ? (accA <= pre[0]) : (accA >= pre[0]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- rare path: some finished sum may enter a top-k ----------------------------------
        if (live && (hit & outmask)) {
#pragma unroll
            for (int qi = 0; qi < QG; qi++) { // pick up thresholds published by other waves
                gt[qi] = gthr_load<IS_L2>(a.gthr + q_of[qi]);
                pre[qi] = prefilter_bound<IS_L2>(tighter<IS_L2>(kd[qi], gt[qi]), dis0[qi]);
            }
#pragma unroll
            for (int j = 0; j < 16; j++) {
#pragma unroll
                for (int qi = 0; qi < QG; qi++) {
                    const float o = (qi == 0) ? cur[j].x : cur[j].y;
                    unsigned long long mm =
                            __ballot(IS_L2 ? (o <= pre[qi]) : (o >= pre[qi])) & outmask;
                    while (mm) {
                        const int l = __ffsll((long long)mm) - 1;
                        mm &= mm - 1;
                        const int p = l / M;
                        const int64_t pv0 = ((int64_t)wave * P + p) * per_pipe;
                        const int64_t t = rb * 16 + j;
                        const int64_t v64 = pv0 + t - RUNUP;
                        const int64_t vend = min(pv0 + per_pipe, len);
                        if (t < RUNUP || v64 >= vend || qi >= npair) {
                            continue;
                        }
                        const int32_t v = (int32_t)v64;
                        const float acc = readlane_f(o, l);
                        const float dis = fadd_x(dis0[qi], acc);
                        if (!within_gthr<IS_L2>(dis, gt[qi]) || !top[qi].admits(dis, v, kd[qi], ki[qi])) {
                            continue;
                        }
                        if (a.bitset != nullptr &&
                            bitset_filtered(a.bitset, a.bitset_nbits, a.ids[row_off + v])) {
                            continue;
                        }
                        top[qi].insert(dis, v);
                        kd[qi] = top[qi].kth_dist();
                        ki[qi] = top[qi].kth_idx();
                        pre[qi] = prefilter_bound<IS_L2>(tighter<IS_L2>(kd[qi], gt[qi]), dis0[qi]);
                        if (ki[qi] >= 0 && lane == 0) {
                            gthr_publish<IS_L2>(a.gthr + q_of[qi], kd[qi]);
                        }
                    }
                }
            }
        }
    };

    // Code words: w1/w2/w3 hold blocks rb+1..rb+3; blocks rb+4, rb+5 are requested at the top of
    // the iteration, i.e. 48 steps before their first LUT read is issued.
    LutVal la[16], lb[16];
    {
        const uint4 w0 = load_blk(0);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            la[j] = lut_read(w0, j);
        }
    }
    uint4 w1 = load_blk(1), w2 = load_blk(2), w3 = load_blk(3);
    for (int64_t rb = 0; rb < nrun_blk; rb += 2) {
        const uint4 w4 = load_blk(rb + 4), w5 = load_blk(rb + 5);
        run_block(la, lb, w1, rb, true);                   // consume rb,   issue rb+1
        run_block(lb, la, w2, rb + 1, rb + 1 < nrun_blk);  // consume rb+1, issue rb+2
        w1 = w3;
        w2 = w4;
        w3 = w5;
    }

    // ---- merge the waves' lists; wave qi finishes query qi ---------------------------------------
    __syncthreads(); // LUT is dead
    const int k = a.k;
    float* md = reinterpret_cast<float*>(smem);                                     // [QG][WAVES][k]
    int64_t* mi = reinterpret_cast<int64_t*>(smem + (((size_t)QG * PQ_WAVES * k * 4 + 7) & ~(size_t)7));
#pragma unroll
    for (int qi = 0; qi < QG; qi++) {
        top[qi].store(md + (qi * PQ_WAVES + wave) * k, mi + (qi * PQ_WAVES + wave) * k);
    }
    __syncthreads();
#pragma unroll
    for (int qi = 0; qi < QG; qi++) {
        if (qi < npair && wave == qi) {
            for (int w = 1; w < PQ_WAVES; w++) {
                const int ow = (wave + w) % PQ_WAVES;
                const float* od = md + (qi * PQ_WAVES + ow) * k;
                const int64_t* oi = mi + (qi * PQ_WAVES + ow) * k;
                for (int e = 0; e < k; e++) {
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
            // the merged list bounds the query's final k-th with the whole list behind it: far
            // tighter than any single wave's slice (matters most for large k)
            if (ki[qi] >= 0 && lane == 0) {
                gthr_publish<IS_L2>(a.gthr + q_of[qi], kd[qi]);
            }
            float* pd = a.partial_d + ((int64_t)q_of[qi] * a.nslot + slot_of[qi]) * k;
            int64_t* pi = a.partial_i + ((int64_t)q_of[qi] * a.nslot + slot_of[qi]) * k;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int e = r * KN_WAVE + lane;
                if (e < k) {
                    const int64_t pos = (int64_t)top[qi].i[r];
                    pd[e] = top[qi].d[r];
                    pi[e] = pos >= 0 ? a.ids[row_off + pos] : -1;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host launchers.  The scan kernel is compiled once per M in its own translation unit
// (-DKN_PQ_M=8|16|32|64, see the Makefile) so the four instantiation sets build in parallel;
// the TU built without KN_PQ_M holds the small kernels and the dispatcher.
// ---------------------------------------------------------------------------------------------
template <bool IS_L2, int M, int QG, int R, int PQ_WAVES>
static hipError_t launch_pq_scan_rw(const PqScanArgs& a, int64_t grid, hipStream_t s) {
    constexpr int PQ_THREADS = PQ_WAVES * KN_WAVE;
    const size_t lut_bytes = (size_t)PQ_KSUB * 256;
    const int k = a.k;
    const size_t merge_bytes = (((size_t)QG * PQ_WAVES * k * 4 + 7) & ~(size_t)7) + (size_t)QG * PQ_WAVES * k * 8;
    const size_t sm = std::max(lut_bytes, merge_bytes);
    auto kern = pq_scan_kernel<IS_L2, M, QG, R, PQ_WAVES>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    if (e != hipSuccess) {
        return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(PQ_THREADS), sm, s, a);
    return hipGetLastError();
}

// Variant choice.  k <= 128 keeps the wave-resident top-k small (R = 1 or 2) and uses 8 waves per
// workgroup (KNHIP_PQ_WAVES=4|8|16 overrides it for M = 32: tuning knob).  Larger k (<= 1024) uses
// R = 16 with 4 waves so the cross-wave merge area (QG * waves * k * 12 B) still fits the LDS.
template <bool IS_L2, int M, int QG>
static hipError_t launch_pq_scan_m(const PqScanArgs& a, int64_t grid, hipStream_t s) {
    static const int waves = knhip_host::env_layout().pq_waves; // (tuning knob, read once per process: knhip_env.h)
    const int k = a.k;
    if (k > 128) {
        return launch_pq_scan_rw<IS_L2, M, QG, 16, 4>(a, grid, s);
    }
    if (M == 32 && waves == 4) {
        return k <= 64 ? launch_pq_scan_rw<IS_L2, M, QG, 1, (M == 32 ? 4 : 8)>(a, grid, s)
                       : launch_pq_scan_rw<IS_L2, M, QG, 2, (M == 32 ? 4 : 8)>(a, grid, s);
    }
    if (M == 32 && waves == 16) {
        return k <= 64 ? launch_pq_scan_rw<IS_L2, M, QG, 1, (M == 32 ? 16 : 8)>(a, grid, s)
                       : launch_pq_scan_rw<IS_L2, M, QG, 2, (M == 32 ? 16 : 8)>(a, grid, s);
    }
    return k <= 64 ? launch_pq_scan_rw<IS_L2, M, QG, 1, 8>(a, grid, s)
                   : launch_pq_scan_rw<IS_L2, M, QG, 2, 8>(a, grid, s);
}

#ifdef KN_PQ_M
#define KN_CAT_(a, b) a##b
#define KN_CAT(a, b) KN_CAT_(a, b)
hipError_t KN_CAT(launch_pq_scan_m, KN_PQ_M)(const PqScanArgs& a, bool is_l2, int64_t grid,
                                              hipStream_t s) {
    constexpr int QG = (KN_PQ_M == 64) ? 1 : 2;
    return is_l2 ? launch_pq_scan_m<true, KN_PQ_M, QG>(a, grid, s)
                 : launch_pq_scan_m<false, KN_PQ_M, QG>(a, grid, s);
}
#else
hipError_t launch_pq_scan_m8(const PqScanArgs& a, bool is_l2, int64_t grid, hipStream_t s);
hipError_t launch_pq_scan_m16(const PqScanArgs& a, bool is_l2, int64_t grid, hipStream_t s);
hipError_t launch_pq_scan_m32(const PqScanArgs& a, bool is_l2, int64_t grid, hipStream_t s);
hipError_t launch_pq_scan_m64(const PqScanArgs& a, bool is_l2, int64_t grid, hipStream_t s);

int pq_scan_supported_m(int M) {
    return M == 8 || M == 16 || M == 32 || M == 64;
}
int pq_scan_qg(int M) {
    return M == 64 ? 1 : 2;
}
int64_t pq_skew_blocks(int64_t len, int M) {
    // vector blocks + run-up tail + one block of slack for past-the-end pipes
    return (len + 15) / 16 + (M - 1 + 15) / 16 + 1;
}

hipError_t launch_pq_scan(const PqScanArgs& a, bool is_l2, int M, int64_t grid, hipStream_t s) {
    if (grid <= 0) {
        return hipSuccess;
    }
    // xcd_item spreads the items over the blocks [0, round_up(nitems, 8)): a grid that is not a multiple of 8 leaves up to
    // seven items of the last stretch without a block (the prefilter's exact fallback launched one block per PAIR: with
    // 258 one-pair items, six partial lists were never written -- found by tests/test_gpu_pqd_fuzz.py, round 6)
    grid = (grid + 7) / 8 * 8;
    switch (M) {
        case 8:
            return launch_pq_scan_m8(a, is_l2, grid, s);
        case 16:
            return launch_pq_scan_m16(a, is_l2, grid, s);
        case 32:
            return launch_pq_scan_m32(a, is_l2, grid, s);
        case 64:
            return launch_pq_scan_m64(a, is_l2, grid, s);
        default:
            return hipErrorInvalidValue;
    }
}

hipError_t launch_pq_query_table(const float* queries, const float* cb, int d, int M, int64_t nq,
                                 float* t2t, hipStream_t s) {
    if (nq <= 0) {
        return hipSuccess;
    }
    const size_t sm = ((size_t)d + (size_t)PQ_KSUB * (M + 1)) * 4;
    if (sm > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pq_query_table_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) {
            return e;
        }
    }
    hipLaunchKernelGGL(pq_query_table_kernel, dim3((unsigned)nq), dim3(256), sm, s, queries, cb, d, M, t2t);
    return hipGetLastError();
}

hipError_t launch_pq_precomp_table(const float* centroids, const float* cb, int d, int M,
                                   int64_t nlist, float* pt, hipStream_t s) {
    if (nlist <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(pq_precomp_table_kernel, dim3((unsigned)nlist), dim3(256), (size_t)d * 4, s,
                       centroids, cb, d, M, pt);
    return hipGetLastError();
}

hipError_t launch_pq_skew_codes(const uint8_t* codes, const int64_t* list_row_off,
                                const int64_t* list_len, const int64_t* list_sblk_off, int64_t nlist,
                                int M, uint4* out, hipStream_t s) {
    if (nlist <= 0) {
        return hipSuccess;
    }
    const unsigned gy = (unsigned)std::min<int64_t>(nlist, 32768);
    const unsigned gz = (unsigned)((nlist + gy - 1) / gy);
    hipLaunchKernelGGL(pq_skew_codes_kernel, dim3(8, gy, gz), dim3(256), 0, s, codes, list_row_off,
                       list_len, list_sblk_off, nlist, M, out);
    return hipGetLastError();
}
#endif // KN_PQ_M

} // namespace knhip
