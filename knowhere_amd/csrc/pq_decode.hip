// knowhere_amd/csrc/pq_decode.hip -- IVF-PQ (M = 32 x 8 bit, dsub = 4) ADC prefilter, DECODE form (gfx950, round 6).
//
// The table forms of pq_filter.hip restate the CPU's algorithm on the matrix cores: a per-query table of 8192 entries, one
// table LOOKUP per (row, query, sub-quantizer), the additions done by a matrix instruction whose other operand is a constant
// selector -- 16 multiply-accumulate slots spent per addition, and a 128 KB table rebuilt in LDS for every 16 queries of a
// list.  On this machine the cheaper formulation is the other one: for L2 with the precomputed term-2 table
//
//     dis(q, v) = dis0(q, list) + psum[v] - 2 <q, y(v)>,      y(v) = the row's DECODED residual (its 32 codebook entries,
//                                                              128 values), psum[v] = sum_m term2[list][m][code_m(v)]
//                                                              (per stored row, built with the layout: pq_filter.hip)
//     dis(q, v) = dis0 + <q, y(v)>                             (inner product)
//
// so the prefilter of a list is a DENSE contraction  [rows of the list] x [queries that probe it] x 128:
//   * the codebook is STATIC: 32 x 256 entries x 4 halves = 64 KB, loaded into LDS once per workgroup (no per-unit table
//     phase at all);
//   * a row is decoded ONCE per (list, <= 128 queries): 16 ds_read_b64 per lane and 32-row tile (lane = (row, half of the
//     sub-quantizers)) ARE the A operand of v_mfma_f32_32x32x16_f16 -- lane (r, h) of step s holds the entries of
//     m = 16 h + 2 s, 16 h + 2 s + 1, i.e. dimensions 64 h + 8 s .. + 8 (the contraction does not care which dimension sits
//     in which k slot as long as both operands agree), which are bytes 2 s, 2 s + 1 of the 16 contiguous code bytes the lane
//     fetched: the canonical AoS codes are read as they lie, 1 KiB per wave and tile, no token stream;
//   * the queries of the unit sit in REGISTERS (lane (n, h): query n's dimensions 64 h .. 64 h + 64 as 32 VGPRs per tile of
//     32 queries; one wave per SIMD, 512 registers): a decoded tile is multiplied against up to four query tiles = 32
//     matrix instructions of 32 cycles; per (row, query) 128 MACs instead of 512 MAC slots, the codes of a list read once
//     per 128 queries instead of once per 16.
// The accumulator starts at -psum[v] * SC / 2 (stored with the index: psum_s) and ends as SC * (<q, y> - psum / 2): one compare per (row, query) against
// SC * (dis0 - tau - eps) / 2, survivors parked in LDS and appended to the candidate lists at the unit's end (ms_emit), the
// exact finish (mfma_scan.hip, KIND 2) recomputes them in the reference's order: no returned value sees half precision.
//
// Error bound (tests/test_pq_decode_bound.py replays it).  Operands: Q = half(sc_q q), Y = half(sc_y y) with powers of two
// sc_q, sc_y fixed per INDEX (pqd_codebook_kernel), SC = sc_q sc_y.  With
// u = 2^-11 (half precision, round to nearest), a = 2^-14 (the bound must hold whether or not the matrix pipe flushes half
// subnormals: an element below the normal range is off by at most the smallest normal), U = 2^-24:
//     |acc / SC - (<q, y> - psum / 2)| <= (2 u + u^2) B_q + a (1 + u) (||y||_1 / sc_q + ||q||_1 / sc_y) + 128 a^2 / SC
//                                          + 136 U ((1 + 3 u) B_q + pabs_max / 2)
//     B_q = sum_m max_c sum_{i in m} |q_i| |cb[m][c][i]|  >=  sum_i |q_i| |y_i(v)| for every row v
// (products of halves are exact in fp32; the accumulation of 128 products onto the start value, in any order and rounding, is
// the last line).  The distance has twice that (L2) plus the reference's own fp32 roundings of its 32-term sums, as in
// pq_filter.hip: 128 U (pabs_max + 2 B_q), and 64 U (|dis0| + |tau|) per pair for the threshold arithmetic.
//
// Reference semantics replaced: IVFPQScannerT::scan_list_with_table (thirdparty/faiss/faiss/impl/pq_code_distance/
// IVFPQScanner_impl.h:109-181) -- only WHICH rows reach the exact finish.
#include "common.h"
#include "kernels.h"
#include "ms_common.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <type_traits>

namespace knhip {

#ifndef PD_EXP
#define PD_EXP 0 // experiment knock-outs (compile time): see pqd_scan
#endif
constexpr int PD_M = 32;
constexpr int PD_KSUB = 256;
constexpr int PD_D = 128;
constexpr int PD_WAVES = 4;
constexpr int PD_THREADS = PD_WAVES * KN_WAVE;
constexpr int PD_CB_BYTES = PD_M * PD_KSUB * 8;     // 65536: [m][c][4 halves]
constexpr int PD_RING = 8;                          // tiles the loads run ahead of the decode (which runs one tile ahead): the
                                                    // loop runs at (memory latency) / PD_RING per tile or at its own pace,
                                                    // whichever is SLOWER -- four slots held it at 1550 cycles per tile (2.6 us
                                                    // of latency under load) against ~820 of matrix work
constexpr int PD_REC_BYTES = 80;                    // a parked lane: {pair, first row, -, -} + its 16 accumulator values
constexpr int PD_REC_CAP = 176;                     // records per wave and unit in the wave's LDS region; beyond: the wave's
                                                    // region in global memory, beyond that the query's overflow route
constexpr int PD_FLAT_CAP = 256;                    // passing rows of a unit, sorted out of the records at its end (more: appended on the spot)
// LDS: the codebook, then one private area per WAVE (a wave runs its own units from start to end: no workgroup barrier
// after the codebook is in place)
constexpr int PD_PA_BYTES = PD_QT * 16;             // a unit's pair arrays: float T[128] (accumulator threshold, scaled), float C[128]
                                                    // (dis0 +- eps), int32 Q[128] (query, -1: none), int32 S[128] (slot)
constexpr int PD_W_PA = 0;                          // two sets: the unit being scanned / flushed and the next one
constexpr int PD_W_REC = PD_W_PA + 2 * PD_PA_BYTES; // [PD_REC_CAP] parked records
constexpr int PD_W_FLAT = PD_W_REC + PD_REC_CAP * PD_REC_BYTES; // uint4 [PD_FLAT_CAP] {pair, row, value bits, rank among the pair's rows}
constexpr int PD_W_PCNT = PD_W_FLAT + PD_FLAT_CAP * 16;         // int32 [128] passing rows per pair, [128] the pair's first slot in its
                                                                // query's candidate list (one reservation per pair and unit)
constexpr int PD_W_CTL = PD_W_PCNT + 2 * PD_QT * 4;             // int32 [16]: 1 = passing rows, 2 = records in global memory
constexpr int PD_W_BYTES = PD_W_CTL + 64;
constexpr int PD_OFF_W = PD_CB_BYTES;
constexpr int PD_SMEM = PD_OFF_W + PD_WAVES * PD_W_BYTES;
static_assert(PD_SMEM <= 160 * 1024, "LDS of one workgroup");
constexpr float PD_U = 5.9604645e-8f;   // 2^-24
constexpr float PD_UH = 4.8828125e-4f;  // 2^-11
constexpr float PD_A = 6.103515625e-5f; // 2^-14: the smallest normal half

#ifdef KNHIP_PHASE_TIMERS
// tools/prof build only: shader-clock ticks per phase, summed per wave over the launch (printed by launch_pqd)
#define PD_T(i)                                                         \
    do {                                                                \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();     \
        tacc[i] += t_ - tlast;                                          \
        tlast = t_;                                                     \
    } while (0)
#define PD_COUNT(i, n) tacc[i] += (unsigned long long)(n)
#define PD_TARG , unsigned long long (&tacc)[8], unsigned long long& tlast
#define PD_TPASS , tacc, tlast
__device__ unsigned long long g_pd_prof[4 * 8 + 8]; // + [32..]: high-water marks of the shared / global record regions and the flat list
#else
#define PD_T(i)
#define PD_COUNT(i, n)
#define PD_TARG
#define PD_TPASS
#endif

typedef _Float16 pd_h8 __attribute__((ext_vector_type(8)));
typedef float pd_f16 __attribute__((ext_vector_type(16)));
typedef uint32_t pd_u4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t pd_rsrc;

size_t pqd_smem() { return PD_SMEM; }

bool pqd_supports(int M, int d) { return M == PD_M && d == PD_D; }

// the power of two that puts `amax` into [2^14, 2^15) (1 for amax = 0; exponent kept within +-60 so that products of
// two such scales and of scaled operands stay far inside the fp32 range)
__host__ __device__ inline float pqd_scale_for(float amax) {
    if (!(amax > 0.f) || !(amax < INFINITY)) {
        return 1.0f;
    }
    int e;
    frexpf(amax, &e); // amax = f 2^e, f in [0.5, 1)
    int s = 15 - e;
    s = s < -60 ? -60 : (s > 60 ? 60 : s);
    return ldexpf(1.0f, s);
}

// ---- per index: the half-precision codebook, the scales, the scaled row constants ----------------------------------------
// cb: FAISS order [m][256][4] fp32.  cb16: [m][256][4] halves of sc_y cb.
// st[0..7] = {sc_y, 1 / sc_y, max |cb|, Ysum = sum_m max_c sum_i |cb[m][c][i]| (>= ||y(v)||_1 of every row),
//             sc_q, 1 / sc_q, SC = sc_q sc_y, 1 / SC}.
// sc_q is fixed per INDEX, not per batch: queries live where the data lives, |x_i| <= |c_i| + |y_i| <= cmax + ymax, and the
// scale puts 4 (cmax + ymax) into [2^14, 2^15): a query up to eight times the data's largest coordinate still fits the half
// range (a larger one gets eps = inf and takes the exact kernels); a smaller one loses nothing (floating point) until its
// coordinates fall 2^-28 below that.  A fixed SC lets the rows' start values -psum SC / 2 be stored with the index.
__global__ __launch_bounds__(PD_KSUB) void pqd_codebook_kernel(const float4* __restrict__ cb, const uint32_t* __restrict__ cmax_bits,
                                                               _Float16* __restrict__ cb16, float* __restrict__ st) {
    __shared__ float s_red[PD_KSUB];
    __shared__ float s_bc[2];
    const int c = threadIdx.x;
    float amax = 0.f;
    for (int m = 0; m < PD_M; m++) {
        const float4 e = cb[m * PD_KSUB + c];
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(e.x), fabsf(e.y)), fmaxf(fabsf(e.z), fabsf(e.w))));
    }
    s_red[c] = amax;
    __syncthreads();
    if (c == 0) {
        float v = 0.f;
        for (int i = 0; i < PD_KSUB; i++) {
            v = fmaxf(v, s_red[i]);
        }
        s_bc[0] = v;
    }
    __syncthreads();
    const float ymax = s_bc[0];
    const float sc = pqd_scale_for(ymax);
    float ysum = 0.f;
    for (int m = 0; m < PD_M; m++) {
        const float4 e = cb[m * PD_KSUB + c];
        _Float16* o = cb16 + ((size_t)m * PD_KSUB + c) * 4;
        o[0] = (_Float16)(e.x * sc);
        o[1] = (_Float16)(e.y * sc);
        o[2] = (_Float16)(e.z * sc);
        o[3] = (_Float16)(e.w * sc);
        __syncthreads();
        s_red[c] = fabsf(e.x) + fabsf(e.y) + fabsf(e.z) + fabsf(e.w);
        __syncthreads();
        if (c == 0) {
            float v = 0.f;
            for (int i = 0; i < PD_KSUB; i++) {
                v = fmaxf(v, s_red[i]);
            }
            ysum += v;
        }
    }
    if (c == 0) {
        const float cmax = __uint_as_float(*cmax_bits);
        const float sq = pqd_scale_for(4.0f * (cmax + ymax));
        st[0] = sc;
        st[1] = 1.0f / sc;
        st[2] = ymax;
        st[3] = ysum * 1.0001f;
        st[4] = sq;
        st[5] = 1.0f / sq;
        st[6] = sq * sc;
        st[7] = 1.0f / (sq * sc);
    }
}

// largest finite |x| of an array (bit pattern of a non-negative float, atomic max; *out zeroed by the caller)
__global__ __launch_bounds__(256) void pqd_absmax_kernel(const float* __restrict__ x, int64_t n, uint32_t* __restrict__ out) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = fabsf(x[i]);
        m = v < INFINITY ? fmaxf(m, v) : m;
    }
#pragma unroll
    for (int dlt = KN_WAVE / 2; dlt > 0; dlt >>= 1) {
        m = fmaxf(m, __shfl_xor(m, dlt, KN_WAVE));
    }
    if (lane_id() == 0 && m > 0.f) {
        atomicMax(out, __float_as_uint(m));
    }
}

// psum_s[i] = -psum[i] SC / 2 (a power-of-two multiple: exact)
__global__ __launch_bounds__(256) void pqd_scale_psum_kernel(const float* __restrict__ psum, int64_t n, const float* __restrict__ st,
                                                             float* __restrict__ psum_s) {
    const float f = -0.5f * st[6];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        psum_s[i] = psum[i] * f;
    }
}

// centroids [ncent] floats; scratch: one uint32; psum / psum_s: npsum floats (L2; null / 0 for the inner product)
hipError_t launch_pqd_index_prep(const float4* cb, const float* centroids, int64_t ncent, void* cb16, float* st,
                                 uint32_t* scratch, const float* psum, int64_t npsum, float* psum_s, hipStream_t s) {
    hipError_t e = hipMemsetAsync(scratch, 0, sizeof(uint32_t), s);
    if (e != hipSuccess) {
        return e;
    }
    if (ncent > 0) {
        const unsigned grid = (unsigned)std::min<int64_t>((ncent + 255) / 256, 1024);
        hipLaunchKernelGGL(pqd_absmax_kernel, dim3(grid), dim3(256), 0, s, centroids, ncent, scratch);
    }
    hipLaunchKernelGGL(pqd_codebook_kernel, dim3(1), dim3(PD_KSUB), 0, s, cb, scratch, static_cast<_Float16*>(cb16), st);
    if (npsum > 0 && psum != nullptr) {
        const unsigned grid = (unsigned)std::min<int64_t>((npsum + 255) / 256, 65535);
        hipLaunchKernelGGL(pqd_scale_psum_kernel, dim3(grid), dim3(256), 0, s, psum, npsum, st, psum_s);
    }
    return hipGetLastError();
}

// ---- per batch: the queries' half rows and error records ---------------------------------------------------------------
// one workgroup per query, thread = code c.  qh16[q][128] = half(sc_q q); qd[q] = {||q||_1, B_q, eps_base, 2 B_q}.
// cst = the index constants above; pabs_max = max_v sum_m |term2| (L2 with the precomputed table; 0 for the inner product)
template <bool IS_L2>
__global__ __launch_bounds__(PD_KSUB) void pqd_query_prep_kernel(const float* __restrict__ queries, const float4* __restrict__ cb,
                                                                 const float* __restrict__ cst, float pabs_max,
                                                                 _Float16* __restrict__ qh16, float* __restrict__ qd) {
    __shared__ float s_q[PD_D];
    __shared__ uint32_t s_max[PD_M];
    const int64_t q = blockIdx.x;
    const int c = threadIdx.x;
    if (c < PD_D) {
        s_q[c] = queries[q * PD_D + c];
    }
    if (c < PD_M) {
        s_max[c] = 0u;
    }
    __syncthreads();
    const float inv_y = cst[1], ysum = cst[3], sc_q = cst[4], inv_q = cst[5], inv_sc = cst[7];
    if (c < PD_D) {
        qh16[q * PD_D + c] = (_Float16)(s_q[c] * sc_q); // (out of the half range: inf here, eps = inf below -> exact path)
    }
    for (int m = 0; m < PD_M; m++) {
        const float4 e = cb[m * PD_KSUB + c];
        float v = fabsf(s_q[4 * m]) * fabsf(e.x) + fabsf(s_q[4 * m + 1]) * fabsf(e.y) + fabsf(s_q[4 * m + 2]) * fabsf(e.z) +
                  fabsf(s_q[4 * m + 3]) * fabsf(e.w);
#pragma unroll
        for (int dlt = KN_WAVE / 2; dlt > 0; dlt >>= 1) {
            v = fmaxf(v, __shfl_xor(v, dlt, KN_WAVE));
        }
        if (lane_id() == 0) {
            atomicMax(&s_max[m], __float_as_uint(v)); // (a NaN query is caught by the finite test below)
        }
    }
    __syncthreads();
    if (c == 0) {
        float B = 0.f, q1 = 0.f;
        bool fits = true;
        for (int m = 0; m < PD_M; m++) {
            B += __uint_as_float(s_max[m]);
        }
        for (int i = 0; i < PD_D; i++) {
            q1 += fabsf(s_q[i]);
            fits = fits && fabsf(s_q[i]) * sc_q < 65504.0f; // (false for NaN / inf too)
        }
        B *= 1.0001f; // (its own fp32 summation)
        q1 *= 1.0001f;
        float eps = INFINITY;
        if (fits && B < INFINITY) {
            const float F = IS_L2 ? 2.0f : 1.0f;
            const float e_prod = (2.0f * PD_UH + PD_UH * PD_UH) * B;
            const float e_sub = PD_A * (1.0f + PD_UH) * (ysum * inv_q + q1 * inv_y) + 128.0f * PD_A * PD_A * inv_sc;
            const float e_acc = 136.0f * PD_U * ((1.0f + 3.0f * PD_UH) * B + 0.5f * pabs_max);
            eps = (F * (e_prod + e_sub + e_acc) + 128.0f * PD_U * (pabs_max + F * B)) * 1.001f;
#if defined(PD_EXP) && (PD_EXP & 32)
            eps *= 16.0f; // (experiment build: what an int8 contraction's bound would let through -- results stay exact)
#endif
#if defined(PD_EXP) && (PD_EXP & 64)
            eps *= 0.25f; // (experiment build: NOT a bound any more -- how much of the filter's time the band tau .. tau + eps is)
#endif
        }
        qd[q * 4 + 0] = q1;
        qd[q * 4 + 1] = B;
        qd[q * 4 + 2] = eps;
        qd[q * 4 + 3] = 2.0f * B;
    }
}

hipError_t launch_pqd_query_prep(const float* queries, const float4* cb, const float* cst, int64_t nq, bool is_l2,
                                 float pabs_max, void* qh16, float* qd, hipStream_t s) {
    if (nq <= 0) {
        return hipSuccess;
    }
    if (is_l2) {
        hipLaunchKernelGGL(pqd_query_prep_kernel<true>, dim3((unsigned)nq), dim3(PD_KSUB), 0, s, queries, cb, cst, pabs_max,
                           static_cast<_Float16*>(qh16), qd);
    } else {
        hipLaunchKernelGGL(pqd_query_prep_kernel<false>, dim3((unsigned)nq), dim3(PD_KSUB), 0, s, queries, cb, cst, 0.f,
                           static_cast<_Float16*>(qh16), qd);
    }
    return hipGetLastError();
}

// ---- the scan ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pd_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// largest of the 16 accumulator values of a lane (v_max3_f32 chains)
__device__ __forceinline__ float pd_max16(const pd_f16& v) {
    const float a = pd_max3(v[0], v[1], v[2]), b = pd_max3(v[3], v[4], v[5]), c = pd_max3(v[6], v[7], v[8]);
    const float d = pd_max3(v[9], v[10], v[11]), e = pd_max3(v[12], v[13], v[14]);
    return fmaxf(pd_max3(a, b, c), pd_max3(d, e, v[15]));
}

// ---- the ring loads: issued and waited for BY HAND ------------------------------------------------------------------------
// The compiler's own wait insertion cannot follow a ring of loads that turns by name through an unrolled loop with exits:
// whatever the structure tried, some body waited with vmcnt(0) or vmcnt(1) -- for the load issued a moment ago, a memory
// round trip per tile -- where 2 (PD_RING - 1) loads may stay in flight.  And a load in inline ISA whose destination the
// compiler manages does not work either: it believes the value is there when the instruction has been issued and moves it
// around (seen: v_accvgpr_read of the destination right behind the load).  So the ring lives in NAMED accumulation registers
// the compiler never sees -- a[216:255], at the far end of a file of which the kernel uses the first ~180 (checked at build
// time: tools/check_pqd_ring.py); every statement
// that touches them lists them as clobbered, which also puts them into the kernel's register count -- the loads are inline
// ISA, and one statement waits (s_waitcnt vmcnt(N): at most N loads still in flight) and copies the slot into ordinary
// registers.  Everything else in the loop that counts in vmcnt (the rare stores of the park path) is YOUNGER than the loads
// waited for and can only make the wait longer, never shorter.  The host pass and the CPU emulation of tests/hipemu keep
// the ring in ordinary variables and take the builtin loads.
typedef int pd_i4 __attribute__((ext_vector_type(4)));
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PD_NO_ASM)
#define PD_RING_CLOBBER "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255", "memory"
struct PdRing {}; // (nothing: the slots are the named registers)
// slot I <- 16 code bytes (+ the start value).  s_nop 4: the scalar offsets are computed right in front of this statement, and
// a vector memory instruction that reads an SGPR needs five wait states behind the scalar instruction that wrote it -- the
// compiler pads that for its own instructions, not for inline ISA (without the padding the loads took the PREVIOUS tile's
// offset now and then: tiles scanned twice, tiles missed)
template <int I, bool WITH_P>
__device__ __forceinline__ void pd_ring_load(PdRing&, pd_i4 rc, int voff_c, int soff_c, pd_i4 rp, int voff_p, int soff_p) {
    static_assert(I >= 0 && I < PD_RING, "eight slots: codes a[216 + 4 I : 219 + 4 I], start value a[248 + I]");
#define PD_LD(WREG, PREG)                                                                                                    \
    if (WITH_P) {                                                                                                            \
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 " WREG ", %0, %1, %2 offen\n\tbuffer_load_dword " PREG ", %3, %4, %5 offen" \
                     :                                                                                                       \
                     : "v"(voff_c), "s"(rc), "s"(soff_c), "v"(voff_p), "s"(rp), "s"(soff_p)                                  \
                     : PD_RING_CLOBBER);                                                                                     \
    } else {                                                                                                                 \
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 " WREG ", %0, %1, %2 offen" : : "v"(voff_c), "s"(rc), "s"(soff_c) : PD_RING_CLOBBER); \
    }
    if (I == 0) {
        PD_LD("a[216:219]", "a248")
    } else if (I == 1) {
        PD_LD("a[220:223]", "a249")
    } else if (I == 2) {
        PD_LD("a[224:227]", "a250")
    } else if (I == 3) {
        PD_LD("a[228:231]", "a251")
    } else if (I == 4) {
        PD_LD("a[232:235]", "a252")
    } else if (I == 5) {
        PD_LD("a[236:239]", "a253")
    } else if (I == 6) {
        PD_LD("a[240:243]", "a254")
    } else {
        PD_LD("a[244:247]", "a255")
    }
#undef PD_LD
}
// wait until at most N loads are in flight, then slot I -> (w, p).  s_nop 7 behind the copies: found by bisection on the
// hardware -- without wait states between these v_accvgpr_read_b32 and the compiler's instructions that follow the statement
// the scan lost rows (tests/test_gpu_pqf.py failed; padding in front of the loads, in front of the copies or behind the
// loads did not help, padding behind the copies did: a hazard the compiler pads for its own instructions and cannot see
// inside inline ISA)
template <int I, int N, bool WITH_P>
__device__ __forceinline__ void pd_ring_take(PdRing&, pd_u4& w, float& p) {
    uint32_t w0, w1, w2, w3;
    float pp = 0.f;
#define PD_TK(W0, W1, W2, W3, PREG)                                                                                          \
    if (WITH_P) {                                                                                                            \
        asm volatile("s_waitcnt vmcnt(%5)\n\tv_accvgpr_read_b32 %0, " W0 "\n\tv_accvgpr_read_b32 %1, " W1                   \
                     "\n\tv_accvgpr_read_b32 %2, " W2 "\n\tv_accvgpr_read_b32 %3, " W3 "\n\tv_accvgpr_read_b32 %4, " PREG "\n\ts_nop 7" \
                     : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(pp)                                                      \
                     : "n"(N)                                                                                                \
                     : PD_RING_CLOBBER);                                                                                     \
    } else {                                                                                                                 \
        asm volatile("s_waitcnt vmcnt(%4)\n\tv_accvgpr_read_b32 %0, " W0 "\n\tv_accvgpr_read_b32 %1, " W1                   \
                     "\n\tv_accvgpr_read_b32 %2, " W2 "\n\tv_accvgpr_read_b32 %3, " W3 "\n\ts_nop 7"                          \
                     : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3)                                                                \
                     : "n"(N)                                                                                                \
                     : PD_RING_CLOBBER);                                                                                     \
    }
    if (I == 0) {
        PD_TK("a216", "a217", "a218", "a219", "a248")
    } else if (I == 1) {
        PD_TK("a220", "a221", "a222", "a223", "a249")
    } else if (I == 2) {
        PD_TK("a224", "a225", "a226", "a227", "a250")
    } else if (I == 3) {
        PD_TK("a228", "a229", "a230", "a231", "a251")
    } else if (I == 4) {
        PD_TK("a232", "a233", "a234", "a235", "a252")
    } else if (I == 5) {
        PD_TK("a236", "a237", "a238", "a239", "a253")
    } else if (I == 6) {
        PD_TK("a240", "a241", "a242", "a243", "a254")
    } else {
        PD_TK("a244", "a245", "a246", "a247", "a255")
    }
#undef PD_TK
    w[0] = w0;
    w[1] = w1;
    w[2] = w2;
    w[3] = w3;
    p = pp;
}
// nothing of the ring is in flight any more (its registers are the compiler's again)
__device__ __forceinline__ void pd_ring_drain(PdRing&) { asm volatile("s_waitcnt vmcnt(0)" : : : PD_RING_CLOBBER); }
#else
struct PdRing {
    pd_u4 w[PD_RING];
    float p[PD_RING];
};
template <int I, bool WITH_P>
__device__ __forceinline__ void pd_ring_load(PdRing& r, pd_i4 rc, int voff_c, int soff_c, pd_i4 rp, int voff_p, int soff_p) {
    const pd_rsrc dc = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<void*>(((uint64_t)(uint32_t)rc[1] << 32) | (uint32_t)rc[0]), 0, rc[2], rc[3]);
    r.w[I] = __builtin_amdgcn_raw_buffer_load_b128(dc, voff_c, soff_c, 0);
    r.p[I] = 0.f;
    if (WITH_P) {
        const pd_rsrc dp = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<void*>(((uint64_t)(uint32_t)rp[1] << 32) | (uint32_t)rp[0]), 0, rp[2], rp[3]);
        r.p[I] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dp, voff_p, soff_p, 0));
    }
}
template <int I, int N, bool WITH_P>
__device__ __forceinline__ void pd_ring_take(PdRing& r, pd_u4& w, float& p) {
    w = r.w[I];
    p = r.p[I];
}
__device__ __forceinline__ void pd_ring_drain(PdRing&) {}
#endif

// LDS written by lanes of a wave and read by OTHER lanes of the same wave (pair arrays, parked records, the flat list): the
// LDS executes a wave's instructions in order, so all it takes is that the compiler keeps the order; the host pass and the
// CPU emulation (one OS thread per lane) need a real rendezvous of the wave's lanes
__device__ __forceinline__ void pd_wave_sync() {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PD_NO_ASM)
    asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
#else
    (void)__ballot(1);
#endif
}

struct PdUnit {
    int64_t len;      // rows of the list
    int64_t row_off;  // first row of the list in the canonical arrays (codes, ids)
    int64_t ps_off;   // first entry of the list in psum_s
    int ntile;        // ceil(len / 32)
};

// One unit with NTQ query tiles: the wave walks the list's 32-row tiles 0, 1, ...  While tile t is multiplied (step
// s = 16 dimensions: NTQ matrix instructions on NTQ different accumulators), step s of tile t + 1 is decoded into the operand
// registers step s just released (its codes arrived three tiles ago).  Returns the number of lanes parked (wave-uniform;
// those beyond PD_REC_CAP sit in the wave's global region, counted in ctl[2]).
template <bool IS_L2, int NTQ>
__device__ __forceinline__ int pqd_scan(const MScanArgs& a, unsigned char* smem, unsigned char* wb, const unsigned char* pa,
                                        int wslot, pd_h8 (&B)[4][8], const PdUnit& un PD_TARG) {
    const int lane = lane_id();
    const int lr = lane & 31, hi = lane >> 5;
    const float* sT = reinterpret_cast<const float*>(pa);
    const int32_t* sPq = reinterpret_cast<const int32_t*>(pa + 2 * PD_QT * 4);
    int32_t* ctl = reinterpret_cast<int32_t*>(wb + PD_W_CTL);
    unsigned char* rec = wb + PD_W_REC;
    const int ntile = un.ntile;
    const int spill_cap = a.pq_spill_cap / PD_WAVES; // the wave's share of its workgroup's global region
    int nrec = 0; // records this wave has parked (wave-uniform)
    float sink = 0.f; // (experiment builds only)
    bool wrote_global = false; // this lane parked a record in global memory

    // the unit's queries (loaded by the caller while the previous unit was flushed: pqd_load_queries): lane (n, h) holds
    // query n's dimensions 64 h .. 64 h + 64 of every tile (8 steps x 8 halves)
    float thr[NTQ];
#pragma unroll
    for (int qt = 0; qt < NTQ; qt++) {
        thr[qt] = sT[qt * 32 + lr];
    }
    // Everything loaded so far is waited for HERE: a value still in flight when the tile loop is entered makes the compiler
    // wait for ALL outstanding loads (vmcnt(0) / lgkmcnt(0)) at its first use in every trip -- the loop-carried prefetches
    // included (the first version stalled a full memory round trip per tile this way)
#pragma unroll
    for (int qt = 0; qt < NTQ; qt++) {
        float th = thr[qt];
        asm volatile("" : "+v"(th));
        thr[qt] = th;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            pd_h8 bq = B[qt][s];
            asm volatile("" : "+v"(bq));
            B[qt][s] = bq;
        }
    }
    // The list's codes and start values through buffer descriptors: the tile's offset rides in the scalar offset, the lane's
    // part (row lr, half hi: 16 of the row's 32 code bytes) is a constant -- no address arithmetic on the vector unit.  A tile
    // past the list's end re-reads the last tile (never used).  The start value of the tile's row `lr` is -psum SC / 2; both
    // halves of the wave load it: the fp32 matrix instruction that spreads the values over the accumulator layout takes
    // k = 0 from the lower half and multiplies the upper half's k = 1 by zero (finite values: no NaN from it).
    // (every input of a descriptor through readfirstlane: the list's offsets came out of memory, and a descriptor the
    // compiler cannot PROVE uniform is loaded through a waterfall loop per load)
    auto make_rsrc = [](const void* p, int bytes) -> pd_i4 {
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        pd_i4 r;
        r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)v);
        r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) & 0xffff; // (stride 0: a raw buffer)
        r[2] = __builtin_amdgcn_readfirstlane(bytes);
        r[3] = 0x00020000;
        return r;
    };
    // (bounds: the whole tiles of the list -- rows past its end inside the last tile belong to the next list or to the
    // allocation's tail and are never emitted; the codes' allocation ends with the last list, whose last tile may reach
    // past it: the record count stops THAT)
    const pd_i4 rc = make_rsrc(a.pq_codes + un.row_off * PD_M, (int)(un.len * PD_M));
    const pd_i4 rp = make_rsrc(IS_L2 ? a.pq_psum_s + un.ps_off : a.pq_sc, IS_L2 ? un.ntile * 32 * 4 : 0);
    const int voff_c = lr * PD_M + hi * 16, voff_p = lr * 4;
    const int t_end = ntile - 1;
    PdRing ring;
    const float one_lo = hi == 0 ? 1.0f : 0.f;
    const uint32_t hbase = (uint32_t)hi * (16u * PD_KSUB * 8u);
    // step s of a tile's operand: the entries of sub-quantizers 16 h + 2 s, + 1 = bytes 2 s, 2 s + 1 of the lane's codes
    auto decode_step = [&](const pd_u4& w, int s, pd_h8& A) {
        const uint32_t ww = w[s >> 1];
        const uint32_t c0 = (ww >> (16 * (s & 1))) & 0xffu, c1 = (ww >> (16 * (s & 1) + 8)) & 0xffu;
        const uint2 e0 = *reinterpret_cast<const uint2*>(smem + (hbase + (uint32_t)(2 * s) * (PD_KSUB * 8u) + c0 * 8u));
        const uint2 e1 = *reinterpret_cast<const uint2*>(smem + (hbase + (uint32_t)(2 * s + 1) * (PD_KSUB * 8u) + c1 * 8u));
        const uint4 both = make_uint4(e0.x, e0.y, e1.x, e1.y);
        A = __builtin_bit_cast(pd_h8, both);
    };
    // experiment knock-outs, COMPILE time (tools/build_pqd_variants.sh builds one library per mask; the results of such a
    // build are wrong, its filter time says what the removed part costs): 1 = nothing ever passes; 2 = no operand refill (no
    // LDS gathers, no address arithmetic); 4 = no loads in the loop; 8 = no compare at all; 16 = no start-value instruction;
    // 32 = the error bound times 16 (pqd_query_prep_kernel: what an int8 contraction would let through; results stay exact)
    constexpr int dbg = PD_EXP;
    // One maximum per lane and query tile, one ballot.  A lane whose maximum passes PARKS its 16 accumulator values as they
    // are (a record of 80 bytes in the wave's own LDS region: no atomic, no round trip -- the count is a wave-uniform
    // register); which of the 16 rows passed is sorted out at the unit's end, one record per thread.  (The first version
    // picked the passing values apart on the spot: 1300 cycles per entry with the matrix pipe idle, 0.6 entries per tile.)
    auto compare = [&](const pd_f16& acc, int qt, int t) {
        if (dbg & 8) { // (one operation that keeps the matrix instructions alive)
            sink += acc[5];
            return;
        }
        const bool p = pd_max16(acc) >= thr[qt];
        const unsigned long long mask = __ballot(p);
        if (dbg & 1) { // (the count keeps the compare alive)
            nrec += __popcll(mask);
            return;
        }
        if (__builtin_expect(mask != 0ull, 0)) {
            const int my = nrec + __popcll(mask & ((1ull << lane) - 1ull));
            nrec += __popcll(mask);
            if (__builtin_expect(nrec <= PD_REC_CAP, 1)) {
                // (wave-uniform: every parked lane of this tile fits the LDS region -- no per-lane region logic)
                if (p) {
                    uint4* r = reinterpret_cast<uint4*>(rec + my * PD_REC_BYTES);
                    r[0] = make_uint4((uint32_t)(qt * 32 + lr), (uint32_t)(t * 32 + 4 * hi), 0u, 0u);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        r[1 + j] = make_uint4(__float_as_uint(acc[4 * j]), __float_as_uint(acc[4 * j + 1]),
                                              __float_as_uint(acc[4 * j + 2]), __float_as_uint(acc[4 * j + 3]));
                    }
                }
            } else if (p) {
                // (LDS and global memory are written through pointers of their OWN address space: a store through a pointer
                // that may be either is a FLAT instruction, and with one of those possibly in flight the compiler waits for
                // ALL outstanding loads and gathers -- vmcnt(0), lgkmcnt(0) -- at every join behind a compare: the first
                // version of this path did that to every tile)
                const uint4 hd = make_uint4((uint32_t)(qt * 32 + lr), (uint32_t)(t * 32 + 4 * hi), 0u, 0u);
                int at2 = -1;
                if (my >= PD_REC_CAP) { // (rare: the wave's LDS region is full -> its global one)
                    at2 = atomicAdd(&ctl[2], 1);
                }
                if (at2 < 0) {
                    uint4* r = reinterpret_cast<uint4*>(rec + my * PD_REC_BYTES);
                    r[0] = hd;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        r[1 + j] = make_uint4(__float_as_uint(acc[4 * j]), __float_as_uint(acc[4 * j + 1]),
                                              __float_as_uint(acc[4 * j + 2]), __float_as_uint(acc[4 * j + 3]));
                    }
                } else if (at2 < spill_cap) {
                    wrote_global = true;
                    uint4* g = reinterpret_cast<uint4*>(a.pq_spill) + ((int64_t)wslot * spill_cap + at2) * (PD_REC_BYTES / 16);
                    g[0] = hd;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        g[1 + j] = make_uint4(__float_as_uint(acc[4 * j]), __float_as_uint(acc[4 * j + 1]),
                                              __float_as_uint(acc[4 * j + 2]), __float_as_uint(acc[4 * j + 3]));
                    }
                } else {
                    // both regions are full (thousands of parked lanes in one unit: a bound that lets a large part of the list
                    // through): the query takes the overflow route of the candidate lists -- retried with the tighter bound
                    // of what it has gathered, else the exact kernels.  Exactness never rests on a capacity.
                    const int32_t q = sPq[qt * 32 + lr];
                    a.overflow[q] = 1;
                    a.overflow[a.nq] = 1;
                }
            }
            PD_COUNT(7, 1); // (no clock reads inside the tile loop: s_memtime answers through lgkmcnt, out of order with the
                            // LDS gathers, and every wait behind it becomes lgkmcnt(0))
        }
    };

    // Loads run PD_RING tiles ahead in a ring of registers (4 code words + one start value per slot).  The ring turns by
    // NAME, never by copying: a register copy waits for its source, i.e. for a load issued a moment ago -- the first
    // version rotated three code registers by assignment and ran at the memory latency per tile (3200 cycles against 770
    // of matrix work).  So the tile loop is unrolled PD_RING times with compile-time slot numbers.
    pd_h8 A[8];
    pd_f16 acc[NTQ], init;
    constexpr int NLD = IS_L2 ? 2 : 1; // loads per tile
    static_assert(PD_RING == 8, "the ring's named registers");
    {
        // tile 0 through slot 0, taken at once; then the ring: slot i <- tile i + 1
        pd_u4 w0;
        float p0;
        pd_ring_load<0, IS_L2>(ring, rc, voff_c, 0, rp, voff_p, 0);
        pd_ring_take<0, 0, IS_L2>(ring, w0, p0);
        pd_ring_load<0, IS_L2>(ring, rc, voff_c, min(1, t_end) * (32 * PD_M), rp, voff_p, min(1, t_end) * (32 * 4));
        pd_ring_load<1, IS_L2>(ring, rc, voff_c, min(2, t_end) * (32 * PD_M), rp, voff_p, min(2, t_end) * (32 * 4));
        pd_ring_load<2, IS_L2>(ring, rc, voff_c, min(3, t_end) * (32 * PD_M), rp, voff_p, min(3, t_end) * (32 * 4));
        pd_ring_load<3, IS_L2>(ring, rc, voff_c, min(4, t_end) * (32 * PD_M), rp, voff_p, min(4, t_end) * (32 * 4));
        pd_ring_load<4, IS_L2>(ring, rc, voff_c, min(5, t_end) * (32 * PD_M), rp, voff_p, min(5, t_end) * (32 * 4));
        pd_ring_load<5, IS_L2>(ring, rc, voff_c, min(6, t_end) * (32 * PD_M), rp, voff_p, min(6, t_end) * (32 * 4));
        pd_ring_load<6, IS_L2>(ring, rc, voff_c, min(7, t_end) * (32 * PD_M), rp, voff_p, min(7, t_end) * (32 * 4));
        pd_ring_load<7, IS_L2>(ring, rc, voff_c, min(8, t_end) * (32 * PD_M), rp, voff_p, min(8, t_end) * (32 * 4));
#pragma unroll
        for (int s = 0; s < 8; s++) {
            decode_step(w0, s, A[s]);
        }
        pd_f16 z;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            z[r] = 0.f;
        }
        init = IS_L2 ? __builtin_amdgcn_mfma_f32_32x32x2f32(p0, one_lo, z, 0, 0, 0) : z;
    }
    // tile t (SLOT = the ring slot that holds tile t + 1's codes and start value): acc <- init + A x B, step by step (NTQ
    // matrix instructions on NTQ different accumulators per step); the operand registers of a step are refilled with tile
    // t + 1's as soon as the step's instructions have been issued; at the end one fp32 matrix instruction spreads the next
    // tile's start values over the accumulator layout (D[r][n] = P[r] * 1) and the slot is refilled with tile t + 5's.  The
    // accumulators of the PREVIOUS tile tp are compared one query tile at a time right before step 0 overwrites them: the
    // compare of query tile qt + 1 runs while step 0 of query tile qt is in the matrix pipe.
    auto tile = [&](auto slot, int t, int tp, bool first) {
        constexpr int SLOT = decltype(slot)::value;
        pd_u4 wn;
        float pn;
        // behind the slot's loads sit those of the other three slots
        if (!(dbg & 4)) {
            pd_ring_take<SLOT, NLD * (PD_RING - 1), IS_L2>(ring, wn, pn);
        } else {
            wn = pd_u4{0x01020304u * (uint32_t)(t & 31), 0x11121314u, 0x21222324u, 0x31323334u};
            pn = 0.f;
        }
#ifdef PD_CHECK_RING
        {   // (debug build: the same tile through the compiler's own loads)
            const pd_rsrc dc = __builtin_amdgcn_make_buffer_rsrc(
                    reinterpret_cast<void*>(((uint64_t)(uint32_t)rc[1] << 32) | (uint32_t)rc[0]), 0, rc[2], rc[3]);
            const pd_rsrc dp = __builtin_amdgcn_make_buffer_rsrc(
                    reinterpret_cast<void*>(((uint64_t)(uint32_t)rp[1] << 32) | (uint32_t)rp[0]), 0, rp[2], rp[3]);
            const int tt = min(t + 1, t_end);
            const pd_u4 cw = __builtin_amdgcn_raw_buffer_load_b128(dc, voff_c, tt * (32 * PD_M), 0);
            const float cp = IS_L2 ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dp, voff_p, tt * (32 * 4), 0)) : 0.f;
            const bool bad = cw[0] != wn[0] || cw[1] != wn[1] || cw[2] != wn[2] || cw[3] != wn[3] ||
                             __float_as_uint(cp) != __float_as_uint(pn);
            if (bad) {
                atomicAdd(&g_pd_prof[39], 1ull);
                if (atomicAdd(&g_pd_prof[38], 0ull) == 0ull) {
                    printf("ring mismatch: slot %d tile %d (ntile %d) lane %d: got %08x %08x %08x %08x | %g, want %08x %08x %08x %08x | %g\n",
                           SLOT, tt, ntile, lane, wn[0], wn[1], wn[2], wn[3], pn, cw[0], cw[1], cw[2], cw[3], cp);
                }
            }
        }
#endif
#pragma unroll
        for (int qt = 0; qt < NTQ; qt++) {
            if (!first) {
                compare(acc[qt], qt, tp);
            }
            acc[qt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], B[qt][0], init, 0, 0, 0);
        }
        if (!(dbg & 2)) {
            decode_step(wn, 0, A[0]);
        }
#pragma unroll
        for (int s = 1; s < 8; s++) {
#pragma unroll
            for (int qt = 0; qt < NTQ; qt++) {
                acc[qt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s], B[qt][s], acc[qt], 0, 0, 0);
            }
            if (!(dbg & 2)) {
                decode_step(wn, s, A[s]);
            }
        }
        if (IS_L2 && !(dbg & 16)) {
            pd_f16 z;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                z[r] = 0.f;
            }
            // (`init` stays in ITS registers over the whole tile: without this the last matrix instruction of step 0 takes
            // them over as its destination, the roles of the register sets rotate from tile to tile and the unrolled loop
            // pays 32 register copies per tile to undo it)
            asm volatile("" : "+v"(init));
            init = __builtin_amdgcn_mfma_f32_32x32x2f32(pn, one_lo, z, 0, 0, 0);
        }
        const int tn = min(t + PD_RING + 1, t_end);
        if (!(dbg & 4)) {
            pd_ring_load<SLOT, IS_L2>(ring, rc, voff_c, tn * (32 * PD_M), rp, voff_p, tn * (32 * 4));
        }
#ifdef PD_DRAIN_EACH
        pd_ring_drain(ring);
#endif
    };
    int t = 0, tl = 0;
    bool first = true;
    PD_T(1);
    // (body i + 1 is reachable ONLY through body i: with every body behind its own `if (t < ntile)` the compiler must assume
    // a path that skips the bodies in front, on which a slot's load has fewer loads behind it, and waits accordingly --
    // vmcnt(1) instead of vmcnt(6))
    // (written out: the optimizer refuses to unroll a loop of eight such bodies, and a slot number that is not a constant
    // of the body cannot name registers)
#define PD_TILE_BODY(I)                                        \
    if (!done) {                                               \
        tile(std::integral_constant<int, I>{}, t, tl, first);  \
        first = false;                                         \
        tl = t;                                                \
        t += 1;                                                \
        PD_COUNT(6, 1);                                        \
        done = t >= ntile;                                     \
    }
    for (;;) {
        bool done = false;
        PD_TILE_BODY(0)
        PD_TILE_BODY(1)
        PD_TILE_BODY(2)
        PD_TILE_BODY(3)
        PD_TILE_BODY(4)
        PD_TILE_BODY(5)
        PD_TILE_BODY(6)
        PD_TILE_BODY(7)
        if (done) {
            break;
        }
    }
#undef PD_TILE_BODY
    pd_ring_drain(ring); // (the loads still in flight -- tiles past the end -- have landed)
#pragma unroll
    for (int qt = 0; qt < NTQ; qt++) {
        compare(acc[qt], qt, tl);
    }
    if ((dbg & 9) && (nrec == 0x7ffffff0 || sink == 1.2345e-30f)) { // (never: the experiment builds' counters stay alive)
        a.overflow[a.nq] = 1;
    }
    if (__ballot(wrote_global) != 0ull) { // (rare) the records in global memory are read back by other lanes of this wave
        __threadfence();
    }
    PD_T(2);
    return (dbg & 9) ? 0 : nrec;
}

// Persistent: one workgroup per CU keeps the codebook in LDS; after that its four waves never meet again -- every WAVE
// pulls its own units (a list x <= 128 of the queries that probe it) in list order from its XCD's counter (the units of one
// list run on one XCD, close in time: the second one finds the codes in that L2) and takes each from its first tile to the
// appends of its passing rows alone, with its own pair arrays, parked records and counters in LDS.  (Until the middle of round 6
// the four waves shared a unit, tiles dealt round-robin: per unit 10 k cycles went to the wave that had parked the most
// lanes, 15 k to barriers around the flush, and every fixed latency of a unit -- the first tile, the pair constants, the
// queries, the appends -- was paid per 48 tiles of a wave instead of per 190.)  LOOP: a fixed grid walks a unit table whose
// size only the device knows (the retry round's one-query units).
template <bool IS_L2, bool LOOP>
__global__ __launch_bounds__(PD_THREADS, 1) void pqd_kernel(MScanArgs a) {
#ifdef KNHIP_PHASE_TIMERS
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
#endif
    extern __shared__ __align__(16) unsigned char smem[];
    const int nunits = (int)*a.nunits_dev;
    if (nunits <= 0) {
        return;
    }
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / KN_WAVE));
    const int wslot = (int)blockIdx.x * PD_WAVES + wave; // the wave's number in the launch (its global record region)
    unsigned char* wb = smem + PD_OFF_W + wave * PD_W_BYTES;
    int32_t* ctl = reinterpret_cast<int32_t*>(wb + PD_W_CTL);
    // unit source
    const int per = (nunits + 7) / 8;
    uint32_t xcc = 0;
    if (!LOOP) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    }
    const int xcd = (int)(xcc & 7u);
    int fetch_t = 0;  // lane 0: counters [xcd, xcd + fetch_t) are known to be exhausted
    int loop_u = wslot;
    auto fetch = [&]() -> int { // (wave-uniform result; lane 0 asks)
        int u = -1;
        if (LOOP) {
            u = loop_u < nunits ? loop_u : -1;
            loop_u += (int)gridDim.x * PD_WAVES;
            return u;
        }
        if (lane == 0) {
            while (fetch_t < 8) {
                const int x = (xcd + fetch_t) & 7;
                const int base = x * per;
                const int cnt = min(per, nunits - base);
                if (cnt > 0) {
                    const int i = atomicAdd(a.pq_ctr + x * 16, 1);
                    if (i < cnt) {
                        u = base + i;
                        break;
                    }
                }
                fetch_t++;
            }
        }
        return __builtin_amdgcn_readfirstlane(u);
    };
    // the codebook: 64 KB, once per workgroup
    {
        const uint4* src = reinterpret_cast<const uint4*>(a.pq_cb16);
        uint4* dst = reinterpret_cast<uint4*>(smem);
#pragma unroll 4
        for (int i = threadIdx.x; i < PD_CB_BYTES / 16; i += PD_THREADS) {
            dst[i] = src[i];
        }
    }
    int cur = fetch();
    int nxt = cur >= 0 ? fetch() : -1;
    __syncthreads(); // (the only workgroup barrier: every wave passes it exactly once, whether or not it got a unit)
    if (cur < 0) {
        return;
    }
    const float SC = a.pq_sc[6], inv_sc = a.pq_sc[7];
    auto pa_of = [&](int par) -> unsigned char* { return wb + PD_W_PA + par * PD_PA_BYTES; };
    // The per-unit work around the scan is latency, not arithmetic -- pair records, thresholds (a chain of two memory round
    // trips), 32 query loads per lane, the appends of the passing rows -- and with one wave per SIMD nothing hides it but
    // the program itself.  So the units are software-pipelined:
    //   * lane j holds pairs j and j + 64 of the NEXT unit in registers (requested while the current unit is flushed and
    //     scanned), and requests those pairs' constants (dis0, tau, the candidate histogram's bound, eps) right behind the
    //     current unit's scan; they arrive while the parked records are sorted out (LDS work);
    //   * the next unit's thresholds go into the OTHER set of pair arrays, and its queries are requested (32 loads per
    //     lane into the B registers, dead since the scan ended) before the passing rows are appended: both sets of global
    //     round trips overlap.
    struct Pairs2 {
        KnPair p[2];
    };
    auto pair_of = [&](int u) -> Pairs2 {
        Pairs2 r;
        r.p[0].q = r.p[1].q = -1;
        r.p[0].slot = r.p[1].slot = 0;
        if (u >= 0) {
            const KnItem it = a.units[u];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (lane + 64 * h < it.npair) {
                    r.p[h] = a.pairs[it.pair0 + lane + 64 * h];
                }
            }
        }
        return r;
    };
    struct PairConst { // what a pair's threshold is made of (requested early, consumed late)
        float dis0[2], tau[2], epsb[2];
    };
    auto pair_request = [&](const Pairs2& pp, PairConst& pc) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const KnPair p = pp.p[h];
            pc.dis0[h] = 0.f;
            pc.tau[h] = worst_dist<IS_L2>();
            pc.epsb[h] = INFINITY;
            if (p.q >= 0) {
                pc.dis0[h] = a.coarse_dis[(int64_t)p.q * a.nslot + p.slot];
                pc.tau[h] = tighter<IS_L2>(a.gthr[p.q], ms_hist_bound_lane<IS_L2>(a, p.q, a.k));
                pc.epsb[h] = a.pq_qd[(int64_t)p.q * 4 + 2];
            }
        }
    };
    auto pair_write = [&](const Pairs2& pp, const PairConst& pc, int par) {
        unsigned char* pa = pa_of(par);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const KnPair p = pp.p[h];
            const int pj = lane + 64 * h;
            float t = INFINITY, c = 0.f;
            if (p.q >= 0) {
                const float eps = pc.epsb[h] + 64.0f * PD_U * (fabsf(pc.dis0[h]) + fabsf(pc.tau[h]));
                if (pc.tau[h] == worst_dist<IS_L2>() || !(eps < INFINITY)) {
                    // no bound (fewer than k unfiltered rows in the sample) or a query the half operands cannot hold: nothing
                    // passes here, the query goes through the exact kernels
                    a.overflow[p.q] = 1;
                    a.overflow[a.nq] = 1;
                } else {
                    // L2: dis0 + psum - 2 dot <= tau + eps  <=>  dot - psum / 2 >= (dis0 - tau - eps) / 2
                    // IP: dis0 + dot >= tau - eps            <=>  dot >= tau - eps - dis0
                    t = IS_L2 ? SC * (((pc.dis0[h] - pc.tau[h]) - eps) * 0.5f) : SC * ((pc.tau[h] - eps) - pc.dis0[h]);
                    c = IS_L2 ? pc.dis0[h] + eps : pc.dis0[h] - eps;
                }
            }
            reinterpret_cast<float*>(pa)[pj] = t;
            reinterpret_cast<float*>(pa + PD_QT * 4)[pj] = c;
            reinterpret_cast<int32_t*>(pa + 2 * PD_QT * 4)[pj] = p.q;
            reinterpret_cast<int32_t*>(pa + 3 * PD_QT * 4)[pj] = p.slot;
        }
    };
    // the unit's queries -> B: lane (n, h) loads query n's dimensions 64 h .. 64 h + 64 of every tile in use
    pd_h8 B[4][8];
    auto load_queries = [&](int par, int ntq) {
        const int lr = lane & 31, hi = lane >> 5;
        const int32_t* sPq = reinterpret_cast<const int32_t*>(pa_of(par) + 2 * PD_QT * 4);
        const uint4* qh = reinterpret_cast<const uint4*>(a.pq_qh16);
#pragma unroll
        for (int qt = 0; qt < 4; qt++) {
            if (qt < ntq) { // (uniform)
                const int32_t q = sPq[qt * 32 + lr];
                const uint4* src = qh + ((int64_t)(q < 0 ? 0 : q) * (PD_D / 8) + hi * 8);
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    const uint4 w = src[s];
                    B[qt][s] = __builtin_bit_cast(pd_h8, w);
                }
            }
        }
    };
    int par = 0;
    {   // the first unit: everything on the spot
        const Pairs2 p0 = pair_of(cur);
        PairConst c0;
        pair_request(p0, c0);
        pair_write(p0, c0, 0);
    }
    Pairs2 pn = pair_of(nxt); // (in flight)
    pd_wave_sync();
    load_queries(0, (a.units[cur].npair + 31) >> 5);
    while (cur >= 0) {
        const KnItem it = a.units[cur];
        const int ntq = (it.npair + 31) >> 5; // query tiles in use (uniform)
        PdUnit un;
        un.len = a.list_len[it.list];
        un.row_off = a.list_row_off[it.list];
        un.ps_off = IS_L2 ? a.pq_sblk_off_r[it.list] * 16 : 0;
        un.ntile = (int)((un.len + 31) >> 5);
        unsigned char* pa = pa_of(par);
        float* sT = reinterpret_cast<float*>(pa);
        float* sC = reinterpret_cast<float*>(pa + PD_QT * 4);
        int32_t* sPq = reinterpret_cast<int32_t*>(pa + 2 * PD_QT * 4);
        int32_t* sPs = reinterpret_cast<int32_t*>(pa + 3 * PD_QT * 4);
        int32_t* pcnt = reinterpret_cast<int32_t*>(wb + PD_W_PCNT);
        int32_t* pbase = pcnt + PD_QT;
        pcnt[lane] = 0;
        pcnt[lane + 64] = 0;
        if (lane == 0) {
            ctl[1] = 0;
            ctl[2] = 0;
        }
        pd_wave_sync();
        const int nn = nxt >= 0 ? fetch() : -1; // the unit after the next
        PD_T(0);
        int nrec = 0;
        if (un.ntile > 0) {
            switch (ntq) {
                case 1: nrec = pqd_scan<IS_L2, 1>(a, smem, wb, pa, wslot, B, un PD_TPASS); break;
                case 2: nrec = pqd_scan<IS_L2, 2>(a, smem, wb, pa, wslot, B, un PD_TPASS); break;
                case 3: nrec = pqd_scan<IS_L2, 3>(a, smem, wb, pa, wslot, B, un PD_TPASS); break;
                default: nrec = pqd_scan<IS_L2, 4>(a, smem, wb, pa, wslot, B, un PD_TPASS); break;
            }
        }
        PD_T(2);
        // the next unit's pair constants: requested now, looked at behind the sorting of this unit's records
        PairConst cn;
        pair_request(pn, cn);
        pd_wave_sync(); // (the last parked records are in LDS)
        PD_T(4);
        // the parked lanes.  Phase A, one record per lane: which of its 16 rows pass (and are not filtered) -> a flat list in
        // LDS, each entry with its rank among the rows of its PAIR (LDS atomics).  Phase B: ONE reservation per pair in its
        // query's candidate list (a global atomic with a returned slot, all pairs' in flight together, beside the loads of
        // the rows' histogram origins), then one passing row per lane: candidate, pessimistic distance, histogram count --
        // nothing that waits.  (One append per row -- ms_emit -- is a global round trip per 64 rows, one after the other: a
        // unit with 460 passing rows spent 8 of them; appending straight from the records, a round trip per row.)
        uint4* flat = reinterpret_cast<uint4*>(wb + PD_W_FLAT);
        auto sort_out = [&](const uint32_t (&rw)[PD_REC_BYTES / 4]) {
            const uint32_t pair = rw[0], row0 = rw[1];
            const float thq = sT[pair];
            uint32_t hm = 0;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const uint32_t pos = row0 + (uint32_t)((r & 3) + 8 * (r >> 2));
                hm |= (__uint_as_float(rw[4 + r]) >= thq && (int64_t)pos < un.len) ? (1u << r) : 0u;
            }
            while (hm != 0u) {
                const int r = __ffs((int)hm) - 1;
                hm &= hm - 1u;
                uint32_t xb = rw[4];
#pragma unroll
                for (int r2 = 1; r2 < 16; r2++) {
                    xb = r == r2 ? rw[4 + r2] : xb;
                }
                const uint32_t pos = row0 + (uint32_t)((r & 3) + 8 * (r >> 2));
                if (a.bitset != nullptr && bitset_filtered(a.bitset, a.bitset_nbits, a.ids[un.row_off + (int64_t)pos])) {
                    continue;
                }
                const int at = atomicAdd(&ctl[1], 1);
                if (at < PD_FLAT_CAP) {
                    const int rank = atomicAdd(&pcnt[pair], 1);
                    flat[at] = make_uint4(pair, pos, xb, (uint32_t)rank);
                } else { // (more passing rows than the list holds: appended on the spot)
                    const float xs = __uint_as_float(xb) * inv_sc, c = sC[pair];
                    ms_emit<IS_L2>(a, sPq[pair], sPs[pair], un.row_off, (int64_t)pos, IS_L2 ? c - 2.0f * xs : c + xs);
                }
            }
        };
        if (nrec > 0) {
            const int n_own = min(nrec, PD_REC_CAP), n_gl = min(ctl[2], a.pq_spill_cap / PD_WAVES);
            for (int i = lane; i < n_own; i += KN_WAVE) {
                uint32_t rw[PD_REC_BYTES / 4];
                const unsigned char* rp = wb + PD_W_REC + i * PD_REC_BYTES;
#pragma unroll
                for (int j = 0; j < PD_REC_BYTES / 16; j++) {
                    const uint4 q4 = reinterpret_cast<const uint4*>(rp)[j];
                    rw[4 * j] = q4.x;
                    rw[4 * j + 1] = q4.y;
                    rw[4 * j + 2] = q4.z;
                    rw[4 * j + 3] = q4.w;
                }
                sort_out(rw);
            }
            // records in global memory were written by lanes of this wave a moment ago: read past the CU's vector cache (a
            // line of this buffer may sit there from an earlier unit)
            const uint32_t* gl = reinterpret_cast<const uint32_t*>(a.pq_spill) +
                                 (int64_t)wslot * (a.pq_spill_cap / PD_WAVES) * (PD_REC_BYTES / 4);
            for (int i = lane; i < n_gl; i += KN_WAVE) {
                uint32_t rw[PD_REC_BYTES / 4];
#pragma unroll
                for (int j = 0; j < PD_REC_BYTES / 4; j++) {
                    rw[j] = __hip_atomic_load(gl + (int64_t)i * (PD_REC_BYTES / 4) + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                sort_out(rw);
            }
        }
        // the next unit's thresholds -> the other set of pair arrays
        pair_write(pn, cn, par ^ 1);
        PD_T(3); // (phase A)
        pd_wave_sync();
#ifdef KNHIP_PHASE_TIMERS
        if (lane == 0) {
            atomicMax(&g_pd_prof[33], (unsigned long long)ctl[2]);
            atomicMax(&g_pd_prof[34], (unsigned long long)ctl[1]);
            atomicAdd(&g_pd_prof[35], (unsigned long long)nrec);
            atomicAdd(&g_pd_prof[36], (unsigned long long)ctl[1]);
            atomicAdd(&g_pd_prof[37], (unsigned long long)(ctl[2] > 0));
        }
#endif
        // the next unit's queries (B is dead since the scan ended) and the pair after it: in flight beside the appends
        if (nxt >= 0) {
            load_queries(par ^ 1, (a.units[nxt].npair + 31) >> 5);
        }
        pn = pair_of(nn);
        const int nflat = min(ctl[1], PD_FLAT_CAP);
        if (nflat > 0) {
            static_assert(PD_FLAT_CAP == 4 * KN_WAVE, "four rows per lane");
            uint4 fe[4];
            uint2 mt[4];
#pragma unroll
            for (int b = 0; b < 4; b++) { // the rows' histogram origins: requested first
                const int i = lane + KN_WAVE * b;
                fe[b] = make_uint4(0u, 0u, 0u, 0u);
                mt[b] = make_uint2(0u, KN_HIST_OFF);
                if (i < nflat) {
                    fe[b] = flat[i];
                    if (a.ghist != nullptr) {
                        mt[b] = a.gmeta[sPq[fe[b].x]];
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < 2; h++) { // one reservation per pair
                const int j = lane + 64 * h;
                const int c = pcnt[j];
                if (c > 0) {
                    const int32_t q = sPq[j];
                    const int base = atomicAdd(a.cand_cnt + q, c);
                    pbase[j] = base;
                    if (base + c > a.cap) {
                        a.overflow[q] = 1;
                        a.overflow[a.nq] = 1; // "some query overflowed": the fallback kernels return at once while this stays 0
                    }
                }
            }
            pd_wave_sync();
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int i = lane + KN_WAVE * b;
                if (i < nflat) {
                    const uint4 h = fe[b];
                    const int32_t q = sPq[h.x];
                    const float x = __uint_as_float(h.z) * inv_sc, c = sC[h.x];
                    const float pess = IS_L2 ? c - 2.0f * x : c + x;
                    const int n = pbase[h.x] + (int)h.w;
                    if (n < a.cap) {
                        a.cand[(int64_t)q * a.cap + n] = ((int64_t)sPs[h.x] << 32) | (int64_t)h.y;
                        if (a.cand_pess != nullptr) {
                            a.cand_pess[(int64_t)q * a.cap + n] = pess;
                        }
                    }
                    if (mt[b].y != KN_HIST_OFF) {
                        atomicAdd(a.ghist + (int64_t)q * KN_HIST_BINS + hist_bin(dist_key<IS_L2>(pess), mt[b].x, mt[b].y), 1u);
                    }
                }
            }
        }
        cur = nxt;
        nxt = nn;
        par ^= 1;
        pd_wave_sync(); // (this unit's flat list and counters have been read)
        PD_T(5);
    }
#ifdef KNHIP_PHASE_TIMERS
    if (lane == 0) {
        for (int i = 0; i < 8; i++) {
            atomicAdd(&g_pd_prof[wave * 8 + i], tacc[i]);
        }
    }
#endif
}

// filter pass only (no sample mode: pq_sample_kernel samples per query); the units must have been cut for PD_QT queries
hipError_t launch_pqd(const MScanArgs& a, bool is_l2, int64_t units_bound, hipStream_t s) {
    if (units_bound <= 0) {
        return hipSuccess;
    }
    if (a.dump != nullptr || a.pq_spill == nullptr || a.pq_spill_cap <= 0 || a.pq_cb16 == nullptr || a.pq_qh16 == nullptr || a.pq_qd == nullptr || a.pq_sc == nullptr ||
        a.pq_codes == nullptr || a.pq_ctr == nullptr || (is_l2 && (a.pq_psum_s == nullptr || a.pq_sblk_off_r == nullptr))) {
        return hipErrorInvalidValue;
    }
    int dev = 0, ncu = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (ncu <= 0) {
        ncu = 256;
    }
    auto kern = a.unit_loop ? (is_l2 ? pqd_kernel<true, true> : pqd_kernel<false, true>)
                            : (is_l2 ? pqd_kernel<true, false> : pqd_kernel<false, false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)PD_SMEM);
    if (e != hipSuccess) {
        return e;
    }
    if (!a.unit_loop) {
        e = hipMemsetAsync(a.pq_ctr, 0, 8 * 16 * sizeof(int32_t), s);
        if (e != hipSuccess) {
            return e;
        }
    }
    // (every wave runs its own units: four per workgroup)
    const int64_t grid = std::min<int64_t>(std::min<int64_t>((units_bound + PD_WAVES - 1) / PD_WAVES, ncu), a.pq_spill_wgs);
#ifdef KNHIP_PHASE_TIMERS
    static unsigned long long zero[40] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pd_prof), zero, sizeof(zero));
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(PD_THREADS), PD_SMEM, s, a);
#ifdef KNHIP_PHASE_TIMERS
    (void)hipStreamSynchronize(s);
    unsigned long long h[40];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pd_prof), sizeof(h));
    {
        int64_t nu = -1;
        (void)hipMemcpy(&nu, a.nunits_dev, sizeof(nu), hipMemcpyDeviceToHost);
        fprintf(stderr, "[pqd timers] ring mismatches %llu\n", h[39]);
        fprintf(stderr, "[pqd timers] launch unit_loop=%d units=%lld | max global %llu max flat %llu | parked lanes %llu "
                        "passing rows %llu | units with global records %llu\n", (int)a.unit_loop, (long long)nu, h[33], h[34],
                h[35], h[36], h[37]);
    }
    if (!a.unit_loop) {
        fprintf(stderr, "[pqd timers] ticks per workgroup (grid %lld): wave | prologue  operands  tiles  flush-A  end-barrier  flush-B+barriers | "
                        "tiles  slow entries\n", (long long)grid);
        for (int w = 0; w < 4; w++) {
            fprintf(stderr, "[pqd timers]   %d |", w);
            for (int i = 0; i < 6; i++) fprintf(stderr, " %10.0f", (double)h[w * 8 + i] / (double)grid);
            fprintf(stderr, " | %8.0f %8.0f\n", (double)h[w * 8 + 6] / (double)grid, (double)h[w * 8 + 7] / (double)grid);
        }
    }
#endif
    return hipGetLastError();
}

} // namespace knhip
