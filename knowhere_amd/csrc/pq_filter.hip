// knowhere_amd/csrc/pq_filter.hip -- IVF-PQ (M = 32 x 8 bit, dsub = 4) ADC scan as a PREFILTER on the matrix cores.
//
// Why: the exact ADC scan (pq_scan_q4.hip) is bound by VALU issue: per 256 lookups of a wave one ds_read_b128, one SDWA
// address shift and 2..4 v_pk_add_f32 (EXEC flips on the staggered steps).  Sums that only FILTER need neither the
// reference's summation order nor its precision, as mfma_scan.hip shows for fp32 rows and SQ8 codes:
//
//   approx(q, v) = dis0 + psum[v] + (1 / sc_q) * sum_m Qh_q[m][code_m(v)]                                  (L2)
//       psum[v]  = sum_m term2[list][m][code_m(v)]     per stored vector, fp32, computed when the layout is built
//                  (term2 = ||cb||^2 + 2 <c_list,m, cb>: IVFPQ_QueryTables.cpp:50-108; no query in it)
//       Qh_q     = half(sc_q * -2 <q_m, cb[m][c]>)     per query; sc_q = the power of two that puts the table's largest
//                  magnitude into [2^14, 2^15): the scaling is exact, every entry keeps 11 significant bits
//   approx(q, v) = dis0 + (1 / sc_q) * sum_m half(sc_q <q_m, cb>)                                           (IP)
//
// The additions run on the MATRIX CORES (round 3; the first version of this file added 8 packed halves per lookup on
// the VALU and had to scale the table to integers <= 2048 to keep those additions exact: eps = 0.0081 A_q).  One LUT
// entry = the 8 queries' halves of one (m, code) = the 16 bytes a lane fetches with one ds_read_b128 = the B operand
// of one v_mfma_f32_16x16x32_f16.  Lane L = (kb = L >> 4, n = L & 15) fetches for vector 16 (kb >> 1) + n of a group
// of 32 vectors, sub-quantizer half kb & 1; the A operand is a constant selector (row i < 8 sums the k blocks 0, 1
// element i, row i >= 8 the k blocks 2, 3 element i - 8), so D[i][n] accumulates in FP32, over the 16 steps of a
// group, query (i & 7)'s 32 table entries of vector 16 (i >> 3) + n: 512 lookups per instruction, no VALU on the data
// path (one SDWA shift per lookup for the address), 240 against 198 lookups/ns/CU in tools/ubench/adc_loop
// (profiles/r03_ubench_adc_loop_mfma.log) -- and, because the accumulator is fp32, the only approximation left is the
// half rounding of the entries themselves:
//
//   |approx - exact| <= eps = 2^-11 A_q (1 + 2^-10) + 2^-20 / sc_q + 128 * 2^-24 * (max_v sum_m |term2| + A_q)
//                              + 64 * 2^-24 * (|dis0| + |tau|),        A_q = sum_m max_c |Qf_q[m][c]|
//   (32 entries rounded to 11 bits: 2^-11 relative each, 2^-25 absolute for the subnormal ones in scaled units; the
//   matrix core's fp32 additions of 32 exact products, in any order and rounding: 64 * 2^-24 of the magnitudes; the
//   fp32 roundings of both sides).  16x tighter than the integer table: the filter passes ~k rows per query instead
//   of ~10 k, on isotropic data a fraction of a percent instead of 10 %.  tests/test_pq_filter_bound.py replays it.
//
// A row whose approx is within eps of the query's bound tau_q goes to the query's candidate list; the finish kernel
// (mfma_scan.hip, KIND 2) recomputes the candidates in the reference's exact order and keeps the canonical top-k --
// the returned values never see half precision.  Bound, sample pass, candidate histogram, retry round and exact
// fallback are the machinery of mfma_scan.hip (MScanArgs).
//
// Token stream `stream16m`: the same 16-bit tokens code << 8 | m << 3 as stream16 (token << 1 = LDS byte address of
// LUT[code][m][8 queries], 16-byte entries), laid out for the lane map above: at step s of a group lane L handles
// m = 16 (kb & 1) + ((pq_stream_phase(L) + s) & 15).  The 16 lanes an LDS gather is serviced together for (kernels.h:
// pq_stream_phase) sit on 16 different m mod 16 = 16 different bank quads: no bank conflict for any code values.
#include "common.h"
#include "kernels.h"
#include "ms_common.h"

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstdlib>

namespace knhip {

constexpr int PF_KSUB = 256;
constexpr int PF_M = 32;
constexpr int PF_DSUB = 4;
constexpr int PF_Q = 8;
constexpr int PF_WAVES = 16;
constexpr int PF_THREADS = PF_WAVES * KN_WAVE;
constexpr int PF_LUT_BYTES = PF_KSUB * PF_M * PF_Q * 2; // 131072
// behind the LUT (ints): [0], [1] unit index of mailbox slot 0 / 1, [8 + 32 s ..) the record of slot s (the current unit
// and the one after it), [128 + 4 j ..) per-pair constants of the current unit
constexpr int PF_CTL_BYTES = 1536; // (the integer form keeps its per-pair constants per unit parity: pqi_kernel)
constexpr int PF_SAMPLE = 8192; // = MS_SAMPLE (mfma_scan.hip): dump columns per query
// Candidate staging behind the control block: a hit costs an LDS atomic and one 16-byte LDS store inside the scan loop;
// the global side of ms_emit (bitset test, atomic on the query's counter with its returned slot, candidate store,
// histogram) waited 1 .. 2 us per hit in the loop -- with ~10^7 hits per batch a fifth of the kernel (round-3 experiment:
// profiles/r03_pqi_experiments.md).  The staged records are written out by all 1024 threads at the end of the unit.
constexpr int PF_STAGE_CAP = 1792;
constexpr int PF_STAGE_BYTES = PF_STAGE_CAP * 16 + 16; // records + two counters (unit parity)
constexpr float PF_U = 5.9604645e-8f;        // 2^-24
constexpr float PF_UH = 4.8828125e-4f;       // 2^-11: unit roundoff of half precision

typedef _Float16 pf_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 pf_h8 __attribute__((ext_vector_type(8))); // one LUT entry: the 8 queries' halves

// ---- AoS codes [len][32] -> token stream of the matrix-core scan ---------------------------------------------------
// uint4 out[blk][lane]: 8 steps per block, 2 blocks per group of 32 vectors (16 vector positions per block)
int64_t pq_stream16r_blocks(int64_t len) {
    // + four groups of slack: every wave's code prefetch runs two window pairs ahead of its last group
    return ((len + 31) / 32) * 2 + 8;
}

// lane L of a wave -> (vector within the group of 32, sub-quantizer at step s)
__host__ __device__ constexpr int pf_lane_vec(int L) { return 16 * (L >> 5) + (L & 15); }
__host__ __device__ constexpr int pf_lane_m(int L, int s) { return 16 * ((L >> 4) & 1) + ((pq_stream_phase(L) + s) & 15); }

__global__ void pq_stream16r_kernel(const uint8_t* __restrict__ codes, const int64_t* __restrict__ list_row_off,
                                    const int64_t* __restrict__ list_len, const int64_t* __restrict__ list_sblk_off,
                                    int64_t nlist, uint4* __restrict__ out) {
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
        const int64_t v = (blk >> 1) * 32 + pf_lane_vec(L);
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const int m = pf_lane_m(L, (int)(blk & 1) * 8 + s);
            uint32_t code = 0;
            if (v < len) {
                code = codes[(row_off + v) * PF_M + m];
            }
            const uint32_t val = (code << 8) | ((uint32_t)m << 3);
            w[s >> 1] |= val << (16 * (s & 1));
        }
        out[(list_sblk_off[l] + blk) * 64 + L] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

hipError_t launch_pq_stream16r(const uint8_t* codes, const int64_t* list_row_off, const int64_t* list_len,
                               const int64_t* list_sblk_off, int64_t nlist, uint4* out, hipStream_t s) {
    if (nlist <= 0) {
        return hipSuccess;
    }
    const unsigned gy = (unsigned)std::min<int64_t>(nlist, 65535);
    const unsigned gz = (unsigned)((nlist + gy - 1) / gy);
    hipLaunchKernelGGL(pq_stream16r_kernel, dim3(8, gy, gz), dim3(256), 0, s, codes, list_row_off, list_len, list_sblk_off,
                       nlist, out);
    return hipGetLastError();
}

// ---- per stored vector: psum = sum_m term2[list][m][code_m], |.| bound -------------------------------------------
// psum[list_sblk_off[l] * 16 + v]: one float per vector position of the token stream (slack positions 0).
// term2 = the entries of the index's precomputed table, i.e. the very values the exact scan adds (an index that
// searches with residual tables -- table over precomputed_table_max_bytes -- does not take this path: without the
// stored entries both sides of the bound would round differently).  pabs_max: max over vectors of sum_m |term2|.
__global__ void pq_psum_kernel(const uint8_t* __restrict__ codes, const int64_t* __restrict__ list_row_off,
                               const int64_t* __restrict__ list_len, const int64_t* __restrict__ list_sblk_off,
                               int64_t nlist, const float* __restrict__ precomp_t, float* __restrict__ psum,
                               uint32_t* __restrict__ pabs_max_bits) {
    const int64_t l = blockIdx.y + (int64_t)blockIdx.z * gridDim.y;
    if (l >= nlist) {
        return;
    }
    const int64_t len = list_len[l];
    const int64_t npos = (list_sblk_off[l + 1] - list_sblk_off[l]) * 16;
    const int64_t base = list_sblk_off[l] * 16;
    const int64_t row_off = list_row_off[l];
    float amax = 0.f;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < npos; v += (int64_t)gridDim.x * blockDim.x) {
        float sum = 0.f, sabs = 0.f;
        if (v < len) {
            const uint8_t* cv = codes + (row_off + v) * PF_M;
            for (int m = 0; m < PF_M; m++) {
                const int c = cv[m];
                const float t2 = precomp_t[(l * PF_KSUB + c) * PF_M + m];
                sum += t2;
                sabs += fabsf(t2);
            }
        }
        psum[base + v] = sum;
        amax = fmaxf(amax, sabs);
    }
    if (amax > 0.f) {
        atomicMax(pabs_max_bits, __float_as_uint(amax)); // (non-negative floats order as their bit patterns)
    }
}

hipError_t launch_pq_psum(const uint8_t* codes, const int64_t* list_row_off, const int64_t* list_len,
                          const int64_t* list_sblk_off, int64_t nlist, const float* precomp_t, float* psum,
                          uint32_t* pabs_max_bits, hipStream_t s) {
    if (nlist <= 0 || precomp_t == nullptr) {
        return hipSuccess;
    }
    const unsigned gy = (unsigned)std::min<int64_t>(nlist, 65535);
    const unsigned gz = (unsigned)((nlist + gy - 1) / gy);
    hipLaunchKernelGGL(pq_psum_kernel, dim3(4, gy, gz), dim3(256), 0, s, codes, list_row_off, list_len, list_sblk_off,
                       nlist, precomp_t, psum, pabs_max_bits);
    return hipGetLastError();
}

// ---- per-query tables: shared pieces ---------------------------------------------------------------------------------
// Thread c of a 256-thread workgroup holds the query's 32 table entries of centroid c.  The codebook is read in FAISS's
// own order cb[m][c][4] (a wave reads 1 KB contiguous per m; the c-major copy the scan kernels use made every lane of
// such a load touch its own cache line: 10 GB of L2 -> L1 traffic per 10^4 queries, round 3 profile).
template <bool IS_L2>
__device__ __forceinline__ void pq_table_values(const float* sq, const float4* __restrict__ cb_m, int c, float (&v)[PF_M]) {
#pragma unroll
    for (int m = 0; m < PF_M; m++) {
        const float4 y = cb_m[m * PF_KSUB + c];
        const float4 x = *reinterpret_cast<const float4*>(sq + m * PF_DSUB);
        float t = ip_step(0.f, x.x, y.x);
        t = ip_step(t, x.y, y.y);
        t = ip_step(t, x.z, y.z);
        t = ip_step(t, x.w, y.w);
        v[m] = IS_L2 ? fmul_x(-2.0f, t) : t;
    }
}

// Reduction of a[m] over the 64 lanes of a wave for all 32 m at once (`a` is consumed): halving exchanges -- 16 + 8 + 4 +
// 2 + 1 + 1 = 32 shuffles instead of 32 x 6.  Returns the reduced value of m = (lane >> 1) & 31.
template <int W, typename Op>
__device__ __forceinline__ void pf_reduce_step(float (&a)[PF_M], Op op, int lane) {
    constexpr int BIT = 2 * W; // partner = lane ^ BIT; lanes with the bit set keep the upper half
    const bool up = (lane & BIT) != 0;
#pragma unroll
    for (int i = 0; i < W; i++) {
        const float send = up ? a[i] : a[i + W];
        const float keep = up ? a[i + W] : a[i];
        a[i] = op(keep, __shfl_xor(send, BIT, KN_WAVE));
    }
}

template <typename Op>
__device__ __forceinline__ float pf_reduce32(float (&a)[PF_M], Op op) {
    const int lane = lane_id();
    pf_reduce_step<16>(a, op, lane);
    pf_reduce_step<8>(a, op, lane);
    pf_reduce_step<4>(a, op, lane);
    pf_reduce_step<2>(a, op, lane);
    pf_reduce_step<1>(a, op, lane);
    return op(a[0], __shfl_xor(a[0], 1, KN_WAVE));
}

// ---- per-query half table ------------------------------------------------------------------------------------------
// One workgroup per query, thread = centroid index c.  qs[q] = {sc, 1 / sc, eps_base, A}.
// Table layout (halves): qh[q][c >> 2][m & 15][c & 3][m >> 4] -- the 16-byte piece thread t = (c >> 2) * 16 + (m & 15)
// of the filter kernel reads holds, for each of its 8 (c, m) cells, this query's entry; the cells of 16 consecutive
// threads at the same piece index are 16 consecutive m of one c: their LUT stores are a contiguous 256 bytes.
template <bool IS_L2>
__global__ __launch_bounds__(PF_KSUB) void pqf_query_table_kernel(const float* __restrict__ queries,
                                                                  const float4* __restrict__ cb_m, int d, float pabs_max,
                                                                  uint32_t* __restrict__ qh, float* __restrict__ qs) {
    __shared__ float sq[PF_M * PF_DSUB];
    __shared__ float smax[PF_M][PF_KSUB / KN_WAVE];
    __shared__ float s_sc;
    const int64_t q = blockIdx.x;
    const int c = threadIdx.x;
    const int wave = c / KN_WAVE;
    if (c < PF_M * PF_DSUB) {
        sq[c] = queries[q * d + c];
    }
    __syncthreads();
    float v[PF_M];
    pq_table_values<IS_L2>(sq, cb_m, c, v);
    {
        float a[PF_M];
#pragma unroll
        for (int m = 0; m < PF_M; m++) {
            const float x = fabsf(v[m]);
            a[m] = (x == x) ? x : INFINITY; // NaN -> "no bound": the query takes the exact kernels
        }
        const float r = pf_reduce32(a, [](float x, float y) { return fmaxf(x, y); });
        if ((lane_id() & 1) == 0) {
            smax[(lane_id() >> 1) & 31][wave] = r;
        }
    }
    __syncthreads();
    if (c == 0) {
        float A = 0.f, gmax = 0.f;
        for (int m = 0; m < PF_M; m++) {
            float a = smax[m][0];
            for (int w = 1; w < PF_KSUB / KN_WAVE; w++) {
                a = fmaxf(a, smax[m][w]);
            }
            A += a;
            gmax = fmaxf(gmax, a);
        }
        // HALF table: entries half(Qf * sc), sc = the power of two that puts the largest magnitude into [2^14, 2^15)
        // (exact scaling, no overflow of the half range); every entry keeps 11 significant bits: relative error 2^-11,
        // absolute 2^-25 (scaled units) for the subnormal ones.  The additions are fp32 (matrix cores).
        float sc = 1.0f, eps = INFINITY;
        if (A < INFINITY) {
            if (gmax > 0.f) {
                int e = 0;
                (void)frexpf(gmax, &e); // gmax = f * 2^e, f in [0.5, 1)
                int p2 = 15 - e;
                p2 = p2 > 126 ? 126 : (p2 < -126 ? -126 : p2);
                sc = ldexpf(1.0f, p2);
            }
            const float isc = 1.0f / sc; // (a power of two in [2^-126, 2^126]: exact)
            eps = PF_UH * A * 1.001f + 9.5367431640625e-7f * isc + 128.0f * PF_U * (pabs_max + A);
        }
        s_sc = sc;
        qs[q * 4 + 0] = sc;
        qs[q * 4 + 1] = 1.0f / sc;
        qs[q * 4 + 2] = eps;
        qs[q * 4 + 3] = A;
    }
    __syncthreads();
    const float sc = s_sc;
    uint32_t* out = qh + q * (PF_KSUB * PF_M / 2);
#pragma unroll
    for (int l16 = 0; l16 < 16; l16++) {
        pf_h2 h;
        h.x = (_Float16)(v[l16] * sc); // (|.| < 2^15: no overflow; round to nearest even)
        h.y = (_Float16)(v[l16 + 16] * sc);
        out[((c >> 2) * 16 + l16) * 4 + (c & 3)] = __builtin_bit_cast(uint32_t, h);
    }
}

// cb_m: the codebook in FAISS order [m][256][4]
hipError_t launch_pqf_query_table(const float* queries, const float4* cb_m, int d, int64_t nq, bool is_l2, float pabs_max,
                                  void* qh, float* qs, hipStream_t s) {
    if (nq <= 0) {
        return hipSuccess;
    }
    if (is_l2) {
        hipLaunchKernelGGL(pqf_query_table_kernel<true>, dim3((unsigned)nq), dim3(PF_KSUB), 0, s, queries, cb_m, d,
                           pabs_max, static_cast<uint32_t*>(qh), qs);
    } else {
        hipLaunchKernelGGL(pqf_query_table_kernel<false>, dim3((unsigned)nq), dim3(PF_KSUB), 0, s, queries, cb_m, d,
                           pabs_max, static_cast<uint32_t*>(qh), qs);
    }
    return hipGetLastError();
}

// ---- flat unit records + counter reset ------------------------------------------------------------------------------
__global__ void pqf_prepare_kernel(MScanArgs a, int64_t nrec) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < 8 * 16) {
        a.pq_ctr[r] = 0;
    }
    if (r >= nrec || r >= *a.nunits_dev) {
        return;
    }
    const bool dump = a.dump != nullptr;
    P8Rec rec{};
    const KnItem it = a.units[r];
    const int npair = it.npair < PF_Q ? it.npair : PF_Q;
    rec.list = it.list;
    rec.npair = npair;
    const int64_t len = a.list_len[it.list];
    rec.len = dump ? (len < PF_SAMPLE ? len : PF_SAMPLE) : len;
    rec.sblk0 = a.pq_sblk_off_r[it.list];
    rec.row_off = a.list_row_off[it.list];
    for (int j = 0; j < PF_Q; j++) {
        const KnPair p = a.pairs[it.pair0 + (j < npair ? j : npair - 1)];
        rec.q[j] = p.q;
        rec.slot[j] = dump ? a.sample_off[(int64_t)p.q * a.nslot + p.slot] : p.slot;
        rec.dis0[j] = a.coarse_dis[(int64_t)p.q * a.nslot + p.slot];
    }
    a.pq_recs[r] = rec;
}

__device__ __forceinline__ int pf_sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float pf_sgpr_f(float v) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v)));
}

__device__ __forceinline__ void pf_setprio(int p) {
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
__device__ __forceinline__ uint32_t pf_addr_lo(uint32_t w, uint32_t one) {
    uint32_t a;
    asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
        : "=v"(a)
        : "v"(w), "s"(one));
    return a;
}
__device__ __forceinline__ uint32_t pf_addr_hi(uint32_t w, uint32_t one) {
    uint32_t a;
    asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
        : "=v"(a)
        : "v"(w), "s"(one));
    return a;
}

// Phase timers (profiling build only: make prof -> libknhip_prof.so, never shipped): wave 0 of every workgroup sums
// the shader cycles it spends per phase of a unit; printed per launch by launch_pqf.
#ifdef KNHIP_PHASE_TIMERS
#define PF_T(i)                                                         \
    do {                                                                \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();     \
        tacc[i] += t_ - tlast;                                          \
        tlast = t_;                                                     \
    } while (0)
#define PF_COUNT(i, n) tacc[i] += (unsigned long long)(n)
__device__ unsigned long long g_pf_prof[16 * 8];
#else
#define PF_T(i)
#define PF_COUNT(i, n)
#endif

// staged candidate records {query, slot, position, pessimistic distance} + flush (see PF_STAGE_CAP)
template <bool IS_L2>
__device__ __forceinline__ void pf_stage_emit(const MScanArgs& a, unsigned char* smem, int par, int32_t q, int32_t slot,
                                              int64_t row_off, int64_t pos, float pess) {
    int* cnt = reinterpret_cast<int*>(smem + PF_LUT_BYTES + PF_CTL_BYTES + PF_STAGE_CAP * 16);
    const int n = atomicAdd(cnt + par, 1);
    if (n < PF_STAGE_CAP) {
        *reinterpret_cast<uint4*>(smem + PF_LUT_BYTES + PF_CTL_BYTES + n * 16) =
                make_uint4((uint32_t)q, (uint32_t)slot, (uint32_t)pos, __float_as_uint(pess));
    } else {
        ms_emit<IS_L2>(a, q, slot, row_off, pos, pess); // (staging full: straight to the candidate list)
    }
}

template <bool IS_L2>
__device__ __forceinline__ void pf_stage_flush(const MScanArgs& a, unsigned char* smem, int par, int64_t row_off) {
    const int* cnt = reinterpret_cast<const int*>(smem + PF_LUT_BYTES + PF_CTL_BYTES + PF_STAGE_CAP * 16);
    const int n = min(cnt[par], PF_STAGE_CAP);
    for (int i = (int)threadIdx.x; i < n; i += PF_THREADS) {
        const uint4 r = *reinterpret_cast<const uint4*>(smem + PF_LUT_BYTES + PF_CTL_BYTES + i * 16);
        ms_emit<IS_L2>(a, (int32_t)r.x, (int32_t)r.y, row_off, (int64_t)r.z, __uint_as_float(r.w));
    }
}

// ---- the filter / sample kernel -----------------------------------------------------------------------------------
// Persistent: one workgroup of 16 waves per CU (128 KB of LUT), units pulled in list order from the XCD's counter as
// pq_scan_q4.hip does.  Per unit: per-pair constants (waves 0..7, one pair each: bound from gthr and the candidate
// histogram), LUT[c][m][8 queries] transposed from the queries' half tables, then every wave scans its share of the
// list's groups of 32 vectors: 16 steps of {SDWA shift, ds_read_b128, v_mfma_f32_16x16x32_f16} per group, one fma +
// compare per (vector, query) at the end of the group.
typedef float pf_f4 __attribute__((ext_vector_type(4)));

template <bool IS_L2, bool DUMP>
__global__ __launch_bounds__(PF_THREADS) void pqf_kernel(MScanArgs a) {
#ifdef KNHIP_PHASE_TIMERS
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
#endif
    extern __shared__ __align__(16) unsigned char smem[];
    int* ctl = reinterpret_cast<int*>(smem + PF_LUT_BYTES);
    float* pc = reinterpret_cast<float*>(smem + PF_LUT_BYTES + 512); // [8][4] = {t, 1 / sc, pess const, -}
    int* pqs = reinterpret_cast<int*>(smem + PF_LUT_BYTES + 640);    // [8][2] = {query, slot / dump column} of the pairs
    const int lane = lane_id();
    const int wave = pf_sgpr((int)(threadIdx.x / KN_WAVE));
    if ((uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)smem) != 0u) {
        __builtin_trap(); // the 16-bit tokens assume the LUT at LDS offset 0
    }
    const int nunits = (int)*a.nunits_dev;
    const int per = (nunits + 7) / 8;
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int xcd = (int)(xcc & 7u);
    int fetch_t = 0; // thread 0: counters [xcd, xcd + fetch_t) are known to be exhausted
    auto fetch_slow = [&]() -> int {
        while (fetch_t < 8) {
            const int x = (xcd + fetch_t) & 7;
            const int base = x * per;
            const int cnt = min(per, nunits - base);
            if (cnt > 0) {
                const int i = atomicAdd(a.pq_ctr + x * 16, 1);
                if (i < cnt) {
                    return base + i;
                }
            }
            fetch_t++;
        }
        return -1;
    };
    constexpr int REC_WORDS = (int)(sizeof(P8Rec) / 4);
    static_assert(REC_WORDS <= KN_WAVE, "one lane per record word");
    // (the word indices read back from the mailbox below)
    static_assert(offsetof(P8Rec, npair) == 4 && offsetof(P8Rec, q) == 8 && offsetof(P8Rec, slot) == 40 &&
                  offsetof(P8Rec, dis0) == 72 && offsetof(P8Rec, len) == 104 && offsetof(P8Rec, sblk0) == 112 &&
                  offsetof(P8Rec, row_off) == 120, "P8Rec layout");
    static_assert(REC_WORDS == 32, "mailbox slots of 32 words");
    // Two-deep mailbox: slot `par` holds the current unit's record, slot `par ^ 1` the next unit's.  Knowing the next
    // unit before the scan lets every wave request ITS pieces of the next unit's 8 query tables (128 KB per unit, mostly
    // from HBM: the tables of a batch exceed the caches) before the scan loop, which hides their latency.
    if (wave == 0) {
        int u0 = -1, u1 = -1;
        if (lane == 0) {
            u0 = fetch_slow();
            u1 = u0 >= 0 ? fetch_slow() : -1;
        }
        u0 = __builtin_amdgcn_readlane(u0, 0);
        u1 = __builtin_amdgcn_readlane(u1, 0);
        if (lane < REC_WORDS && u0 >= 0) {
            ctl[8 + lane] = (int)reinterpret_cast<const uint32_t*>(a.pq_recs + u0)[lane];
        }
        if (lane < REC_WORDS && u1 >= 0) {
            ctl[8 + REC_WORDS + lane] = (int)reinterpret_cast<const uint32_t*>(a.pq_recs + u1)[lane];
        }
        if (lane == 0) {
            ctl[0] = u0;
            ctl[1] = u1;
            int* scnt = reinterpret_cast<int*>(smem + PF_LUT_BYTES + PF_CTL_BYTES + PF_STAGE_CAP * 16);
            scnt[0] = 0;
            scnt[1] = 0;
        }
    }
    __syncthreads();
    int cur = pf_sgpr(ctl[0]);
    int par = 0;
    const uint32_t one = 1u;
    const uint4* qh4 = reinterpret_cast<const uint4*>(a.pq_qh);
    // this thread's 16-byte piece of the 8 query tables of the unit in mailbox slot `slot`
    auto load_tables = [&](int slot, uint4 (&t)[PF_Q], int lane_i) {
        const uint32_t rq = lane_i < REC_WORDS ? (uint32_t)ctl[8 + slot * REC_WORDS + lane_i] : 0u;
#pragma unroll
        for (int j = 0; j < PF_Q; j++) {
            const int64_t q = (int64_t)(int32_t)__builtin_amdgcn_readlane((int)rq, 2 + j);
            t[j] = qh4[q * (PF_KSUB * PF_M / 8) + (wave * KN_WAVE + lane_i)];
        }
    };
    uint4 tp[PF_Q];
#pragma unroll
    for (int j = 0; j < PF_Q; j++) {
        tp[j] = make_uint4(0, 0, 0, 0);
    }
    if (!DUMP && cur >= 0) {
        load_tables(0, tp, lane);
    }

    while (cur >= 0) {
        PF_T(5); // (loop top: the mailbox read of the previous iteration's end)
        int lane_i = lane;
        asm volatile("" : "+v"(lane_i)); // nothing lane-derived is hoisted out of the unit loop
        int f_x = 0, f_i = 0;
        if (wave == 0 && lane_i == 0 && fetch_t < 8) {
            f_x = (xcd + fetch_t) & 7;
            f_i = atomicAdd(a.pq_ctr + f_x * 16, 1);
        }
        // ---- the unit's record (mailbox), fields broadcast to SGPRs ----------------------------------------------
        const uint32_t rw = lane_i < REC_WORDS ? (uint32_t)ctl[8 + par * REC_WORDS + lane_i] : 0u;
        const int nxt_unit = pf_sgpr(ctl[par ^ 1]); // (-1: this is the last unit of the workgroup)
        auto rl = [&](int i) { return (uint32_t)__builtin_amdgcn_readlane((int)rw, i); };
        const int npair = (int)rl(1);
        const int64_t len = (int64_t)(((uint64_t)rl(27) << 32) | rl(26));
        const int64_t sblk0 = (int64_t)(((uint64_t)rl(29) << 32) | rl(28));
        const int64_t row_off = (int64_t)(((uint64_t)rl(31) << 32) | rl(30));

        // (filter mode: the 8 queries' table pieces `tp` were requested one unit ago; they are transposed into the LUT
        // below.  The sample pass -- few, short units -- requests them here instead)
        if (DUMP) {
            load_tables(par, tp, lane_i);
        }
        // this wave's groups of 32 vectors and its first code blocks (2 blocks per group)
        const int ngroups = (int)((len + 31) / 32);
        const int gbase = ngroups / PF_WAVES, grem = ngroups % PF_WAVES;
        const int G0 = wave * gbase + min(wave, grem);
        const int G1 = G0 + gbase + (wave < grem ? 1 : 0);
        const int nwin = G1 - G0;
        const uint4* cbase = a.pq_codes_r + (sblk0 + (int64_t)G0 * 2) * 64 + lane_i;
        auto load_blk = [&](int b) { return cbase[(int64_t)b * 64]; }; // past-the-end blocks exist (slack)
        const float* psb = a.pq_psum + sblk0 * 16 + (int64_t)G0 * 32 + pf_lane_vec(lane_i);
        uint4 U0 = make_uint4(0, 0, 0, 0), U1 = U0, U2 = U0, U3 = U0;
        float psa_next = 0.f, psb_next = 0.f;
        if (nwin > 0) {
            U0 = load_blk(0);
            U1 = load_blk(1);
            U2 = load_blk(2);
            U3 = load_blk(3);
            if (IS_L2) {
                psa_next = psb[0];
                psb_next = psb[32];
            }
        }
        // ---- per-pair constants: wave j < 8 prepares pair j ---------------------------------------------------------
        if (wave < PF_Q) {
            int32_t q = (int32_t)rl(2), slot = (int32_t)rl(10);
            float dis0 = __uint_as_float(rl(18));
#pragma unroll
            for (int j = 1; j < PF_Q; j++) {
                q = wave == j ? (int32_t)rl(2 + j) : q;
                slot = wave == j ? (int32_t)rl(10 + j) : slot;
                dis0 = wave == j ? __uint_as_float(rl(18 + j)) : dis0;
            }
            const float4 s4 = *reinterpret_cast<const float4*>(a.pq_qs + (int64_t)q * 4);
            // pass-nothing defaults (missing pair, no bound, non-finite query)
            float t = IS_L2 ? -INFINITY : INFINITY, pcst = 0.f;
            if (wave < npair) {
                if (DUMP) {
                    const float eps = s4.z + 64.0f * PF_U * fabsf(dis0);
                    pcst = IS_L2 ? dis0 + eps : dis0 - eps; // (a non-finite eps dumps the neutral side: no bound from it)
                } else {
                    float tau = a.gthr[q];
                    tau = tighter<IS_L2>(tau, ms_hist_bound<IS_L2>(a, q, a.k));
                    const float eps = s4.z + 64.0f * PF_U * (fabsf(dis0) + fabsf(tau));
                    if (tau == worst_dist<IS_L2>() || !(eps < INFINITY)) {
                        // no bound (fewer than k unfiltered rows in the sample) or a query the half table cannot hold:
                        // nothing passes here, the query goes through the exact kernels
                        if (lane_i == 0) {
                            a.overflow[q] = 1;
                            a.overflow[a.nq] = 1;
                        }
                    } else {
                        t = IS_L2 ? (tau + eps) - dis0 : (tau - eps) - dis0;
                        pcst = IS_L2 ? dis0 + eps : dis0 - eps;
                    }
                }
            }
            if (lane_i == 0) {
                *reinterpret_cast<float4*>(pc + wave * 4) = make_float4(t, s4.y, pcst, 0.f);
                pqs[wave * 2] = q;
                pqs[wave * 2 + 1] = slot;
            }
        }
        // ---- LUT[c][m][query]: 8 x 8 halves transposed in registers, 8 conflict-free 16-byte stores ------------------
        {
            const int t = wave * KN_WAVE + lane_i;
            const int c4 = t >> 4, l16 = t & 15;
            const uint32_t r[PF_Q][4] = {{tp[0].x, tp[0].y, tp[0].z, tp[0].w}, {tp[1].x, tp[1].y, tp[1].z, tp[1].w},
                                         {tp[2].x, tp[2].y, tp[2].z, tp[2].w}, {tp[3].x, tp[3].y, tp[3].z, tp[3].w},
                                         {tp[4].x, tp[4].y, tp[4].z, tp[4].w}, {tp[5].x, tp[5].y, tp[5].z, tp[5].w},
                                         {tp[6].x, tp[6].y, tp[6].z, tp[6].w}, {tp[7].x, tp[7].y, tp[7].z, tp[7].w}};
            uint4* lut = reinterpret_cast<uint4*>(smem);
#pragma unroll
            for (int e = 0; e < 8; e++) { // cell e of the piece: c = 4 c4 + (e >> 1), m = l16 + 16 (e & 1); word e >> 1
                const int cc = e >> 1, h = e & 1;
                const uint32_t sel = h ? 0x07060302u : 0x05040100u;
                uint4 o;
                o.x = __builtin_amdgcn_perm(r[1][cc], r[0][cc], sel);
                o.y = __builtin_amdgcn_perm(r[3][cc], r[2][cc], sel);
                o.z = __builtin_amdgcn_perm(r[5][cc], r[4][cc], sel);
                o.w = __builtin_amdgcn_perm(r[7][cc], r[6][cc], sel);
                lut[(c4 * 4 + cc) * PF_M + l16 + 16 * h] = o;
            }
        }
        PF_T(0); // record, pair constants, table loads + LUT stores
        __syncthreads();
        PF_T(1); // wait for the other waves' LUT parts
        if (!DUMP && threadIdx.x == 0) { // the NEXT unit's staging counter (its last user's flush ended before this barrier)
            reinterpret_cast<int*>(smem + PF_LUT_BYTES + PF_CTL_BYTES + PF_STAGE_CAP * 16)[par ^ 1] = 0;
        }
        // ---- the unit after the next: index now (the atomic was issued at the top), record requested, parked after the
        // scan in the mailbox slot this unit occupies (every wave has read it) --------------------------------------------
        int nxt = -1;
        uint32_t rw_next = 0;
        if (wave == 0) {
            if (lane_i == 0) {
                if (fetch_t < 8 && nxt_unit >= 0) {
                    const int base = f_x * per;
                    const int cnt = min(per, nunits - base);
                    if (f_i < cnt) {
                        nxt = base + f_i;
                    } else {
                        fetch_t++;
                        nxt = fetch_slow();
                    }
                }
            }
            nxt = __builtin_amdgcn_readlane(nxt, 0);
            if (lane_i < REC_WORDS && nxt >= 0) {
                rw_next = reinterpret_cast<const uint32_t*>(a.pq_recs + nxt)[lane_i];
            }
        }
        // the next unit's table pieces: in flight during the scan
        uint4 tpn[PF_Q];
#pragma unroll
        for (int j = 0; j < PF_Q; j++) {
            tpn[j] = make_uint4(0, 0, 0, 0);
        }
        if (!DUMP && nxt_unit >= 0) {
            load_tables(par ^ 1, tpn, lane_i);
        }
        // this lane's 4 accumulator rows are the queries 4 hb + r of its vector: their constants -> VGPRs
        const int hb = (lane_i >> 4) & 1;
        float thr[4], isc[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float4 c4v = *reinterpret_cast<const float4*>(pc + (4 * hb + r) * 4);
            thr[r] = c4v.x;
            isc[r] = c4v.y;
        }
        // the selector operand: row i = lane & 15 takes element i & 7 of the k blocks 2 (i >> 3), 2 (i >> 3) + 1
        pf_h8 sel;
        {
            const int i = lane_i & 15, kb = lane_i >> 4;
            const bool on = (kb >> 1) == (i >> 3);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                sel[e] = (on && (i & 7) == e) ? (_Float16)1.0f : (_Float16)0.0f;
            }
        }

        typedef __attribute__((address_space(3))) const pf_h8 lds_h8;
        auto lut_read = [&](uint32_t addr) -> pf_h8 { return *reinterpret_cast<lds_h8*>(addr); };
        auto issue2 = [&](uint32_t w0, pf_h8 (&v)[2]) {
            v[0] = lut_read(pf_addr_lo(w0, one));
            v[1] = lut_read(pf_addr_hi(w0, one));
        };

        PF_T(2); // set-up of the scan
        if (nwin > 0) {
            const pf_f4 fz = {0.f, 0.f, 0.f, 0.f};
            // LUT reads run four 2-step units ahead of the matrix instructions that consume them (four value buffers);
            // code registers: at the top of window pair i, U0 = block 4 i + 4, U1..U3 = blocks 4 i + 1..4 i + 3.
            pf_h8 B0[2], B1[2];
            uint4 W0 = U0; // block 4 i at the top of pair i (its first two words are in flight)
            issue2(W0.x, B0);
            issue2(W0.y, B1);
            pf_f4 e0 = fz, e1 = fz, o0 = fz, o1 = fz; // even / odd window of a pair, even / odd step
            const int vec = pf_lane_vec(lane_i);

#define PF_UNIT(A0, A1, C0, C1, BUF, WORD)                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    A0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(sel, BUF[0], C0, 0, 0, 0);                   \
    A1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(sel, BUF[1], C1, 0, 0, 0);                   \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    issue2(WORD, BUF);

            // ---- the 32 vectors of group G are finished: 4 sums per lane (queries 4 hb .. 4 hb + 3 of vector `vec`) ----
            auto finish = [&](const pf_f4& x0, const pf_f4& x1, int G, float ps) {
                const float f[4] = {x0[0] + x1[0], x0[1] + x1[1], x0[2] + x1[2], x0[3] + x1[3]};
                const int64_t pos = (int64_t)G * 32 + vec;
                const bool inside = G < G1 && pos < len;
                if (DUMP) {
                    bool ok = inside;
                    if (ok && a.bitset != nullptr) {
                        ok = !bitset_filtered(a.bitset, a.bitset_nbits, a.ids[row_off + pos]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int j = 4 * hb + r;
                        if (j < npair) {
                            const int32_t off = pqs[j * 2 + 1]; // (sample pass: the pair's first dump column)
                            if (inside && pos < (int64_t)(PF_SAMPLE - off)) {
                                const float pcs = pc[j * 4 + 2];
                                const float v = IS_L2 ? __fmaf_rn(f[r], isc[r], pcs + ps) : __fmaf_rn(f[r], isc[r], pcs);
                                a.dump[(int64_t)pqs[j * 2] * a.dump_stride + off + pos] =
                                        (ok && v == v) ? v : worst_dist<IS_L2>();
                            }
                        }
                    }
                } else {
                    // fast path: 4 fma + 4 compares per lane, the lane masks OR-ed on the scalar unit; whether the row
                    // exists at all (tail of the list, groups of the next wave) is only looked at when something passed
                    unsigned long long mk[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        // L2: ps + f / sc <= t;  IP: f / sc >= t   (t = -inf / +inf: nothing passes)
                        const float v = __fmaf_rn(f[r], isc[r], IS_L2 ? ps : 0.f);
                        mk[r] = __ballot(IS_L2 ? (v <= thr[r]) : (v >= thr[r]));
                    }
                    if (__builtin_expect((mk[0] | mk[1] | mk[2] | mk[3]) != 0ull, 0)) { // (out of line: the hot path falls through)
                        if (inside) {
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                if ((mk[r] >> lane_i) & 1ull) {
                                    const int j = 4 * hb + r;
                                    const float pcs = pc[j * 4 + 2];
                                    const float pess = IS_L2 ? __fmaf_rn(f[r], isc[r], pcs + ps)
                                                             : __fmaf_rn(f[r], isc[r], pcs);
                                    pf_stage_emit<IS_L2>(a, smem, par, pqs[j * 2], pqs[j * 2 + 1], row_off, pos, pess);
                                }
                            }
                        }
                    }
                }
            };

            const int npairs_w = (nwin + 1) >> 1;
            float psb_prev = 0.f;
            for (int i = 0; i < npairs_w; i++) {
                pf_setprio((i + (wave >> 2)) & 3); // (see pq_scan_q4.hip: equal average speed for a SIMD's four waves)
                const float psa = psa_next, psb2 = psb_next;
                if (IS_L2) {
                    psa_next = psb[(int64_t)(2 * i + 2) * 32]; // (the slack groups behind a list exist)
                    psb_next = psb[(int64_t)(2 * i + 3) * 32];
                }
                // even window of the pair: blocks 4 i, 4 i + 1 (in flight at the top: the first two words of block 4 i)
                PF_UNIT(e0, e1, fz, fz, B0, W0.z)
                PF_UNIT(e0, e1, e0, e1, B1, W0.w)
                if (i > 0) { // (the odd window of the previous pair: its last matrix instructions have retired by now)
                    finish(o0, o1, G0 + 2 * i - 1, psb_prev);
                }
                PF_UNIT(e0, e1, e0, e1, B0, U1.x)
                PF_UNIT(e0, e1, e0, e1, B1, U1.y)
                W0 = load_blk(4 * i + 4);
                PF_UNIT(e0, e1, e0, e1, B0, U1.z)
                PF_UNIT(e0, e1, e0, e1, B1, U1.w)
                PF_UNIT(e0, e1, e0, e1, B0, U2.x)
                PF_UNIT(e0, e1, e0, e1, B1, U2.y)
                U1 = load_blk(4 * i + 5);
                // odd window: blocks 4 i + 2, 4 i + 3
                PF_UNIT(o0, o1, fz, fz, B0, U2.z)
                PF_UNIT(o0, o1, o0, o1, B1, U2.w)
                finish(e0, e1, G0 + 2 * i, psa);
                PF_UNIT(o0, o1, o0, o1, B0, U3.x)
                PF_UNIT(o0, o1, o0, o1, B1, U3.y)
                U2 = load_blk(4 * i + 6);
                PF_UNIT(o0, o1, o0, o1, B0, U3.z)
                PF_UNIT(o0, o1, o0, o1, B1, U3.w)
                PF_UNIT(o0, o1, o0, o1, B0, W0.x)
                PF_UNIT(o0, o1, o0, o1, B1, W0.y)
                U3 = load_blk(4 * i + 7);
                __builtin_amdgcn_sched_barrier(0);
                psb_prev = psb2;
            }
            finish(o0, o1, G0 + 2 * npairs_w - 1, psb_prev);
#undef PF_UNIT
            __builtin_amdgcn_s_setprio(0);
        }
        PF_T(3); // window loop
        PF_COUNT(6, nwin);
        if (wave == 0) { // park the unit after the next
            if (lane_i < REC_WORDS) {
                ctl[8 + par * REC_WORDS + lane_i] = (int)rw_next;
            }
            if (lane_i == 0) {
                ctl[par] = nxt;
            }
        }
        __syncthreads(); // the LUT and the pair constants are dead, the mailbox is visible
        PF_T(4);         // wait for the slowest wave's scan
        if (!DUMP) {
            pf_stage_flush<IS_L2>(a, smem, par, row_off);
        }
        PF_COUNT(7, 1);
        cur = nxt_unit;
        par ^= 1;
#pragma unroll
        for (int j = 0; j < PF_Q; j++) {
            tp[j] = tpn[j];
        }
    }
#ifdef KNHIP_PHASE_TIMERS
    if (lane == 0) {
        for (int i = 0; i < 8; i++) {
            atomicAdd(&g_pf_prof[wave * 8 + i], tacc[i]);
        }
    }
#endif
}

// ---- selectivity guard -----------------------------------------------------------------------------------------
// The filter is only as selective as eps against the spread of the distances.  With the fp32-accumulated half table
// eps is 2^-11 of the table magnitude and a few times k rows per query pass on every data set tried; should a data set
// exist where a few percent of the rows pass, the exact finish (32 global gathers per candidate) would cost more than
// the exact scan it replaces.  The sample pass holds an estimate: the share of a query's sample rows whose pessimistic
// distance lies in the band (tau, tau + 2 eps] -- rows that pass only because of eps -- times the rows of its probed
// lists, plus the k rows below tau.  (The band, not everything below tau + 2 eps: the sample is the query's CLOSEST
// list, whose k best rows say nothing about the other lists.)  One wave per query; *poor counts the queries predicted
// to gather more than half their capacity.  The host abandons the prefilter for the batch when more than a quarter of
// its queries are (knhip_api.hip).
template <bool IS_L2>
__global__ __launch_bounds__(256) void pqf_predict_kernel(const float* __restrict__ dump, int64_t stride,
                                                          const int32_t* __restrict__ n_row, const float* __restrict__ gthr,
                                                          const float* __restrict__ qs, const float* __restrict__ qs2,
                                                          const int64_t* __restrict__ keys, int nprobe, int64_t nlist,
                                                          const int64_t* __restrict__ list_len, int64_t nq, int cap, int k,
                                                          int32_t* __restrict__ poor) {
    const int lane = lane_id();
    const int64_t q = (int64_t)blockIdx.x * (blockDim.x / KN_WAVE) + threadIdx.x / KN_WAVE;
    if (q >= nq) {
        return;
    }
    const float tau = gthr[q];
    const int n = n_row[q];
    if (tau == worst_dist<IS_L2>() || n <= 0) {
        return; // (no bound: the query takes the exact kernels on its own)
    }
    // one pass over the query's sample for both forms (qs: half precision -> poor[0]; qs2, when given: integer -> poor[1])
    const float eps = qs[q * 4 + 2] + 64.0f * PF_U * fabsf(tau);
    const float lim = IS_L2 ? tau + 2.0f * eps : tau - 2.0f * eps;
    const float eps2 = qs2 != nullptr ? qs2[q * 4 + 2] + 64.0f * PF_U * fabsf(tau) : 0.f;
    const float lim2 = IS_L2 ? tau + 2.0f * eps2 : tau - 2.0f * eps2;
    float cnt = 0.f, cnt2 = 0.f, rows = 0.f;
    for (int i = lane; i < n; i += KN_WAVE) {
        const float v = dump[q * stride + i];
        cnt += (IS_L2 ? (v > tau && v <= lim) : (v < tau && v >= lim)) ? 1.f : 0.f;
        cnt2 += (IS_L2 ? (v > tau && v <= lim2) : (v < tau && v >= lim2)) ? 1.f : 0.f;
    }
    for (int sl = lane; sl < nprobe; sl += KN_WAVE) {
        const int64_t key = keys[q * nprobe + sl];
        rows += (key >= 0 && key < nlist) ? (float)list_len[key] : 0.f;
    }
#pragma unroll
    for (int dlt = KN_WAVE / 2; dlt > 0; dlt >>= 1) {
        cnt += __shfl_xor(cnt, dlt, KN_WAVE);
        cnt2 += __shfl_xor(cnt2, dlt, KN_WAVE);
        rows += __shfl_xor(rows, dlt, KN_WAVE);
    }
    if (lane == 0) {
        if ((float)k + cnt / (float)n * rows > 0.5f * (float)cap) {
            atomicAdd(poor, 1);
        }
        if (qs2 != nullptr && (float)k + cnt2 / (float)n * rows > 0.5f * (float)cap) {
            atomicAdd(poor + 1, 1);
        }
    }
}

// poor[0]: queries the half-precision form (records qs) would overflow; poor[1]: the same for the records qs2 (integer
// form; nullptr: not evaluated, poor[1] = 0)
hipError_t launch_pqf_predict(const float* dump, int64_t stride, const int32_t* n_row, const float* gthr, const float* qs,
                              const float* qs2, const int64_t* keys, int nprobe, int64_t nlist, const int64_t* list_len,
                              int64_t nq, int cap, int k, bool is_l2, int32_t* poor, hipStream_t s) {
    hipError_t e = hipMemsetAsync(poor, 0, 2 * sizeof(int32_t), s);
    if (e != hipSuccess || nq <= 0) {
        return e;
    }
    const unsigned grid = (unsigned)((nq + 3) / 4);
    if (is_l2) {
        hipLaunchKernelGGL(pqf_predict_kernel<true>, dim3(grid), dim3(256), 0, s, dump, stride, n_row, gthr, qs, qs2, keys,
                           nprobe, nlist, list_len, nq, cap, k, poor);
    } else {
        hipLaunchKernelGGL(pqf_predict_kernel<false>, dim3(grid), dim3(256), 0, s, dump, stride, n_row, gthr, qs, qs2, keys,
                           nprobe, nlist, list_len, nq, cap, k, poor);
    }
    return hipGetLastError();
}

// =====================================================================================================================
// The INTEGER form: int8 tables, 16 queries per LUT entry, v_mfma_i32_16x16x64_i8 (round 3).
//
// The LDS gather (one ds_read_b128 per lane and step, 4 cycles per wave-instruction) and the matrix pipe (16 cycles per
// instruction and SIMD) bind the half-precision form at the same 128 lookups/clk/CU.  Both move 16 bytes per lane and
// step whatever those bytes mean: with ONE byte per query an entry holds 16 queries, the instruction K = 64 bytes, and
// every step delivers twice the lookups -- a unit is (list, <= 16 queries), half as many units per batch.
//
//   Qi_q[m][c] = rint((Qf_q[m][c] - mu_q[m]) / s_q) in [-127, 127],  mu_q[m] = midrange of Qf_q[m][.],
//                s_q = max_m (range of Qf_q[m][.]) / 254            (one step per query: the sums share one accumulator)
//   approx(q, v) = dis0 + psum[v] + sum_m mu_q[m] + s_q * sum_m Qi_q[m][code_m(v)]          (int32 accumulation: exact)
//   |approx - exact| <= eps = 16.4 s_q + 128 * 2^-24 * (max_v sum_m |term2| + A_q) + 64 * 2^-24 * (|dis0| + |tau| + |mu|)
//   (32 entries half a step each, + 2 % for the fp32 roundings of the quantisation itself)
// That is 8 .. 20 x looser than the half-precision form (and 2 x tighter than round 2's integer-scaled half table), so
// this form is chosen per batch by the selectivity guard: the sample pass always runs in half precision, and its dump
// predicts the candidate count under either eps (knhip_api.hip): integer form when that is small, else the half
// form, else the exact kernel.
//
// Lane L = (kb = L >> 4, n = L & 15) fetches for vector n of a group of 16, eight steps per group; at step s it handles
// m = 16 (kb >> 1) + ((pi(n) + 8 (kb & 1) + s) & 15), pi(n) = n ^ 4 for n < 8, n otherwise: the four lanes of a vector
// cover its 32 sub-quantizers once, and the 16 lanes of an LDS service group -- in the documented grouping
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} as well as in contiguous sixteenths -- sit on 16 different bank quads.
// The selector A[i][kb][e] = [e == i] sums the four k blocks: D[i][n] = query i's 32 entries of vector n after 8 steps.
constexpr int PI_Q = 16;
typedef int pf_i4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int pi_lane_m(int L, int s) {
    const int n = L & 15, kb = L >> 4;
    const int pn = n < 8 ? (n ^ 4) : n;
    return 16 * (kb >> 1) + ((pn + 8 * (kb & 1) + s) & 15);
}

// uint4 out[blk][lane]: 8 steps = one group of 16 vectors per block; block offsets shared with the half-precision stream
__global__ void pq_stream16i_kernel(const uint8_t* __restrict__ codes, const int64_t* __restrict__ list_row_off,
                                    const int64_t* __restrict__ list_len, const int64_t* __restrict__ list_sblk_off,
                                    int64_t nlist, uint4* __restrict__ out) {
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
        const int64_t v = blk * 16 + (L & 15);
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const int m = pi_lane_m(L, s);
            uint32_t code = 0;
            if (v < len) {
                code = codes[(row_off + v) * PF_M + m];
            }
            const uint32_t val = (code << 8) | ((uint32_t)m << 3);
            w[s >> 1] |= val << (16 * (s & 1));
        }
        out[(list_sblk_off[l] + blk) * 64 + L] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

hipError_t launch_pq_stream16i(const uint8_t* codes, const int64_t* list_row_off, const int64_t* list_len,
                               const int64_t* list_sblk_off, int64_t nlist, uint4* out, hipStream_t s) {
    if (nlist <= 0) {
        return hipSuccess;
    }
    const unsigned gy = (unsigned)std::min<int64_t>(nlist, 65535);
    const unsigned gz = (unsigned)((nlist + gy - 1) / gy);
    hipLaunchKernelGGL(pq_stream16i_kernel, dim3(8, gy, gz), dim3(256), 0, s, codes, list_row_off, list_len, list_sblk_off,
                       nlist, out);
    return hipGetLastError();
}

// Steps on a lattice.  s_q = w_q * s0 with an integer weight 1 <= w_q <= 127 and ONE power of two s0 = 2^E0 per batch
// (from the largest table range of the batch: pqi_query_range_kernel).  The weight goes into the selector operand
// (A[i][kb][e] = w_i [e == i], negated for inner product), so the matrix instruction delivers Y = w_q * sum_m Qi in units
// of s0 for every query of a unit, and the filter's fast path is integer: with the accumulator started at -T_q,
//   pass  <=>  min_i (Y_i - T_i) <= ceil(-psum[v] / s0)                                (pqi_kernel, finish()).
// Rounding a step up to the lattice costs <= 1 / w_q of eps (w_q >= 64 for the widest query of the batch).
// qis[q] = {s_q, sum_m mu, eps_base, A}; qis[nq] = {bits of the batch's largest range (atomic max), s0, 1 / s0, -}.
constexpr float PI_INT_LIM = 536870912.0f; // 2^29: |psum| / s0 stays below it (else: no bound, the exact kernels)

// the power of two s0 with 127 s0 >= (largest range of the batch) / 254
__device__ __forceinline__ float pqi_base_step(float rmax) {
    if (!(rmax > 0.f) || !(rmax < INFINITY)) {
        return 1.0f;
    }
    int e;
    (void)frexpf(rmax / (254.0f * 127.0f), &e); // value = f 2^e, f in [0.5, 1)
    e = e < -100 ? -100 : e;
    return ldexpf(1.0f, e);
}

// pass 1, one workgroup per query, thread = centroid index c: the per-m midranges mu[q][m], qis[q] = {R (largest range),
// sum_m mu, -, A}, and the batch's largest finite range -> gl[0] (bits of a non-negative float: integer order = float
// order)
template <bool IS_L2>
__global__ __launch_bounds__(PF_KSUB) void pqi_query_stats_kernel(const float* __restrict__ queries,
                                                                  const float4* __restrict__ cb_m, int d,
                                                                  float* __restrict__ qmu, float* __restrict__ qis,
                                                                  uint32_t* __restrict__ gl) {
    __shared__ float sq[PF_M * PF_DSUB];
    __shared__ float smax[PF_M][PF_KSUB / KN_WAVE];
    __shared__ float smin[PF_M][PF_KSUB / KN_WAVE];
    __shared__ float s_mu[PF_M], s_a[PF_M], s_r[PF_M];
    const int64_t q = blockIdx.x;
    const int c = threadIdx.x;
    const int wave = c / KN_WAVE;
    if (c < PF_M * PF_DSUB) {
        sq[c] = queries[q * d + c];
    }
    __syncthreads();
    float hi[PF_M], lo[PF_M];
    pq_table_values<IS_L2>(sq, cb_m, c, hi);
#pragma unroll
    for (int m = 0; m < PF_M; m++) {
        const bool fin = fabsf(hi[m]) < INFINITY; // (false for NaN too: "no bound", the query takes the exact kernels)
        lo[m] = fin ? hi[m] : -INFINITY;
        hi[m] = fin ? hi[m] : INFINITY;
    }
    const float rh = pf_reduce32(hi, [](float x, float y) { return fmaxf(x, y); });
    const float rl = pf_reduce32(lo, [](float x, float y) { return fminf(x, y); });
    if ((lane_id() & 1) == 0) {
        smax[(lane_id() >> 1) & 31][wave] = rh;
        smin[(lane_id() >> 1) & 31][wave] = rl;
    }
    __syncthreads();
    if (c < PF_M) {
        float h = smax[c][0], l = smin[c][0];
        for (int w = 1; w < PF_KSUB / KN_WAVE; w++) {
            h = fmaxf(h, smax[c][w]);
            l = fminf(l, smin[c][w]);
        }
        const float mu = 0.5f * h + 0.5f * l;
        s_mu[c] = mu;
        s_a[c] = fmaxf(fabsf(h), fabsf(l));
        s_r[c] = h - l;
        qmu[q * PF_M + c] = mu;
    }
    __syncthreads();
    if (c == 0) {
        float A = 0.f, R = 0.f, musum = 0.f;
        for (int m = 0; m < PF_M; m++) { // (in this order: the sums are part of the bound's definition)
            musum += s_mu[m];
            A += s_a[m];
            R = fmaxf(R, s_r[m]);
        }
        qis[q * 4 + 0] = R;
        qis[q * 4 + 1] = musum;
        qis[q * 4 + 2] = INFINITY;
        qis[q * 4 + 3] = A;
        if (R > 0.f && R < INFINITY) {
            atomicMax(gl, __float_as_uint(R));
        }
    }
}

// pass 2: step on the batch's lattice, eps, the quantised table.
// Table layout (bytes): qi[q][c >> 2][m & 15][c & 3][m >> 4]: the 8-byte piece thread t = (c >> 2) * 16 + (m & 15) of the
// filter kernel reads holds this query's entries of its 8 (c, m) cells.
template <bool IS_L2>
__global__ __launch_bounds__(PF_KSUB) void pqi_query_table_kernel(const float* __restrict__ queries,
                                                                  const float4* __restrict__ cb_m, int d, float pabs_max,
                                                                  int64_t nq, const float* __restrict__ qmu,
                                                                  uint32_t* __restrict__ qi, float* __restrict__ qis) {
    __shared__ float sq[PF_M * PF_DSUB];
    __shared__ float smu[PF_M];
    __shared__ float s_inv;
    const int64_t q = blockIdx.x;
    const int c = threadIdx.x;
    if (c < PF_M * PF_DSUB) {
        sq[c] = queries[q * d + c];
    }
    if (c < PF_M) {
        smu[c] = qmu[q * PF_M + c];
    }
    if (c == 0) {
        const float s0 = pqi_base_step(__uint_as_float(reinterpret_cast<const uint32_t*>(qis)[nq * 4]));
        const float inv0 = 1.0f / s0; // (a power of two: exact)
        if (q == 0) {
            qis[nq * 4 + 1] = s0;
            qis[nq * 4 + 2] = inv0;
        }
        const float R = qis[q * 4 + 0], musum = qis[q * 4 + 1], A = qis[q * 4 + 3];
        float step = s0, eps = INFINITY;
        if (A < INFINITY && R < INFINITY && pabs_max * inv0 < PI_INT_LIM) {
            // the smallest lattice step >= R / 254 (w = 1 for a constant table; R <= the batch's largest range => w <= 127)
            const float w = fminf(fmaxf(ceilf((R / 254.0f) * inv0), 1.0f), 127.0f);
            step = w * s0;
            eps = 16.4f * step + 128.0f * PF_U * (pabs_max + A) + 64.0f * PF_U * fabsf(musum);
        }
        s_inv = 1.0f / step;
        qis[q * 4 + 0] = step;
        qis[q * 4 + 2] = eps;
    }
    __syncthreads();
    float v[PF_M];
    pq_table_values<IS_L2>(sq, cb_m, c, v);
    const float inv = s_inv;
    uint32_t* out = qi + q * (PF_KSUB * PF_M / 4);
    // word (c >> 2, l16, w): bytes = cells (c & 3 = 2 w, h = 0), (2 w, 1), (2 w + 1, 0), (2 w + 1, 1): thread c owns byte
    // pairs; the four threads of a c >> 2 group combine through LDS-free shuffles: c & 3 = 0..3 are neighbouring lanes
#pragma unroll
    for (int l16 = 0; l16 < 16; l16++) {
        float x0 = rintf((v[l16] - smu[l16]) * inv), x1 = rintf((v[l16 + 16] - smu[l16 + 16]) * inv);
        x0 = fminf(fmaxf(x0, -127.0f), 127.0f);
        x1 = fminf(fmaxf(x1, -127.0f), 127.0f);
        x0 = (x0 == x0) ? x0 : 0.f;
        x1 = (x1 == x1) ? x1 : 0.f;
        const uint32_t pair = ((uint32_t)(int)x0 & 0xffu) | (((uint32_t)(int)x1 & 0xffu) << 8); // (h = 0, h = 1) of cell c
        const uint32_t other = (uint32_t)__shfl_xor((int)pair, 1, KN_WAVE);                      // the cell c ^ 1
        if ((c & 1) == 0) {
            out[((c >> 2) * 16 + l16) * 2 + ((c >> 1) & 1)] = pair | (other << 16);
        }
    }
}

// cb_m: the codebook in FAISS order [m][256][4]; qis: nq * 4 + 4 floats (the batch record behind the per-query
// records); qmu: nq * 32 floats (the midranges between the two passes).  stats_done: pass 1 was part of the sample pass
// (pq_sample_kernel) -- only the tables are written.
hipError_t launch_pqi_query_table(const float* queries, const float4* cb_m, int d, int64_t nq, bool is_l2, float pabs_max,
                                  void* qi, float* qis, float* qmu, bool stats_done, hipStream_t s) {
    if (nq <= 0) {
        return hipSuccess;
    }
    uint32_t* gl = reinterpret_cast<uint32_t*>(qis + nq * 4);
    if (!stats_done) {
        hipError_t e = hipMemsetAsync(qis + nq * 4, 0, 4 * sizeof(float), s);
        if (e != hipSuccess) {
            return e;
        }
        if (is_l2) {
            hipLaunchKernelGGL(pqi_query_stats_kernel<true>, dim3((unsigned)nq), dim3(PF_KSUB), 0, s, queries, cb_m, d,
                               qmu, qis, gl);
        } else {
            hipLaunchKernelGGL(pqi_query_stats_kernel<false>, dim3((unsigned)nq), dim3(PF_KSUB), 0, s, queries, cb_m, d,
                               qmu, qis, gl);
        }
    }
    if (is_l2) {
        hipLaunchKernelGGL(pqi_query_table_kernel<true>, dim3((unsigned)nq), dim3(PF_KSUB), 0, s, queries, cb_m, d,
                           pabs_max, nq, qmu, static_cast<uint32_t*>(qi), qis);
    } else {
        hipLaunchKernelGGL(pqi_query_table_kernel<false>, dim3((unsigned)nq), dim3(PF_KSUB), 0, s, queries, cb_m, d,
                           pabs_max, nq, qmu, static_cast<uint32_t*>(qi), qis);
    }
    return hipGetLastError();
}

// ---- the sample pass, direct ---------------------------------------------------------------------------------------
// tau_q comes from the closest list(s) of each query alone: ~10^4 (query, list) pairs with one query each.  As units of
// the filter kernel every one of them paid a 128 KB LUT for a few thousand rows (0.43 ms per 10^4 queries, plus a work
// table, units and a sample plan built just for it: ~1 ms per batch in the round-3 profile).  Here one workgroup per
// query builds the query's fp32 table once in LDS (32 KB) and walks the sampled rows with plain fp32 lookups, the plan
// (probes in coarse order until `smin` rows, at most PF_SAMPLE) evaluated on the way.  The same pass over the table
// yields the statistics both table forms need, so it also writes qs[q] = {sc, 1 / sc, eps_base, A} (half form) and, when
// asked, qis[q] = {R, sum mu, -, A}, qmu and the batch's largest range (integer form, pass 1: pqi_query_stats_kernel).
// dump[q][col] = fp32 ADC distance made pessimistic by its own rounding bound (no table quantisation: tighter than the
// half-precision sample it replaces); filtered rows dump the neutral value.
// Round 5: PS_THREADS = 512.  The walk is bound by the latency of its table lookups (round-4 PMC pass: LDS array 58 %
// busy, vector issue 47 %, 60 % of the wave cycles waiting), and the 32 KB table caps the CU at four workgroups whatever
// their size: eight waves per workgroup double the waves that cover each other's lookups.  The table is built by the
// first 256 threads (thread = code).  The selection of tau_q no longer bisects over all 8192 keys (as many vector
// instructions as the walk itself): the ksel-th smallest of the PS_THREADS per-thread minima bounds the ksel-th
// smallest key, the few keys under the bound are compacted into LDS and ONE wave bisects over them with ballots -- no
// barriers, no atomics in the steps; same value bit for bit.  (ksel > PS_THREADS, or more than PS_CAND keys under the
// bound -- masses of equal values --: the block-wide interval cuts of round 4.)
constexpr int PS_THREADS = 512;
constexpr int PS_WAVES = PS_THREADS / KN_WAVE;
constexpr int PS_CAND = 1024;
constexpr int PS_ROWS = 2;
// the r-th smallest (r >= 1) of the keys a wave holds in v[] (lane-strided; padding = 0xffffffff), known to lie in [l, h]:
// plain bisection, the counts are ballots (scalar unit), no barrier.  (mid < h <= 0xffffffff: padding is never counted.)
template <int NV>
__device__ __forceinline__ uint32_t ps_wave_select(const uint32_t (&v)[NV], int r, uint32_t l, uint32_t h) {
    while (l < h) {
        const uint32_t mid = l + ((h - l) >> 1);
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < NV; u++) {
            cnt += __popcll(__ballot(v[u] <= mid));
        }
        if (cnt >= r) {
            h = mid;
        } else {
            l = mid + 1;
        }
    }
    return l;
}
template <bool IS_L2>
__global__ __launch_bounds__(PS_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void pq_sample_kernel(MScanArgs a, const int64_t* __restrict__ keys,
                                                            const float4* __restrict__ cb_m, int64_t nlist, int smin,
                                                            int scap, float pabs_max, int32_t* __restrict__ n_row,
                                                            float* __restrict__ qs, float* __restrict__ qis,
                                                            float* __restrict__ qmu, uint32_t* __restrict__ gl,
                                                            float* __restrict__ gthr_out, uint2* __restrict__ gmeta_out,
                                                            int ksel) {
    __shared__ float lut[PF_M * PF_KSUB]; // [m][c]
    __shared__ float sq[PF_M * PF_DSUB];
    __shared__ float s_mu[PF_M], s_a[PF_M], s_r[PF_M];
    __shared__ float s_A;
    const int64_t q = blockIdx.x;
    const int c = threadIdx.x;
    const int wave = c / KN_WAVE;
    if (c < PF_M * PF_DSUB) {
        sq[c] = a.queries[q * a.d + c];
    }
    __syncthreads();
    {
        // the table: thread = (half of the sub-quantizers, code); the operations of pq_table_values
        const int cc = c & (PF_KSUB - 1), m0 = (c / PF_KSUB) * (PF_M / 2);
#pragma unroll
        for (int j = 0; j < PF_M / 2; j++) {
            const int m = m0 + j;
            const float4 y = cb_m[m * PF_KSUB + cc];
            const float4 x = *reinterpret_cast<const float4*>(sq + m * PF_DSUB);
            float t = ip_step(0.f, x.x, y.x);
            t = ip_step(t, x.y, y.y);
            t = ip_step(t, x.z, y.z);
            t = ip_step(t, x.w, y.w);
            lut[m * PF_KSUB + cc] = IS_L2 ? fmul_x(-2.0f, t) : t;
        }
    }
    __syncthreads();
    {
        // per-m extrema: wave w takes m = 4 w .. 4 w + 3, four entries per lane (max / min do not depend on the order)
#pragma unroll
        for (int j = 0; j < PF_M / PS_WAVES; j++) {
            const int m = wave * (PF_M / PS_WAVES) + j;
            float h = -INFINITY, l = INFINITY;
#pragma unroll
            for (int e = 0; e < PF_KSUB / KN_WAVE; e++) {
                const float v = lut[m * PF_KSUB + lane_id() + e * KN_WAVE];
                const bool fin = fabsf(v) < INFINITY; // (false for NaN too: "no bound", the query takes the exact kernels)
                h = fmaxf(h, fin ? v : INFINITY);
                l = fminf(l, fin ? v : -INFINITY);
            }
#pragma unroll
            for (int dlt = KN_WAVE / 2; dlt > 0; dlt >>= 1) {
                h = fmaxf(h, __shfl_xor(h, dlt, KN_WAVE));
                l = fminf(l, __shfl_xor(l, dlt, KN_WAVE));
            }
            if (lane_id() == 0) {
                const float mu = 0.5f * h + 0.5f * l;
                s_mu[m] = mu;
                s_a[m] = fmaxf(fabsf(h), fabsf(l));
                s_r[m] = h - l;
                if (qmu != nullptr) {
                    qmu[q * PF_M + m] = mu;
                }
            }
        }
    }
    __syncthreads();
    if (c == 0) {
        float A = 0.f, R = 0.f, musum = 0.f, gmax = 0.f;
#pragma unroll 4
        for (int m = 0; m < PF_M; m++) { // (in this order: the sums are part of the bounds' definitions)
            musum += s_mu[m];
            A += s_a[m];
            R = fmaxf(R, s_r[m]);
            gmax = fmaxf(gmax, s_a[m]);
        }
        s_A = A;
        // half form: the record pqf_query_table_kernel writes
        float sc = 1.0f, eps = INFINITY;
        if (A < INFINITY) {
            if (gmax > 0.f) {
                int e = 0;
                (void)frexpf(gmax, &e);
                int p2 = 15 - e;
                p2 = p2 > 126 ? 126 : (p2 < -126 ? -126 : p2);
                sc = ldexpf(1.0f, p2);
            }
            const float isc = 1.0f / sc;
            eps = PF_UH * A * 1.001f + 9.5367431640625e-7f * isc + 128.0f * PF_U * (pabs_max + A);
        }
        qs[q * 4 + 0] = sc;
        qs[q * 4 + 1] = 1.0f / sc;
        qs[q * 4 + 2] = eps;
        qs[q * 4 + 3] = A;
        if (qis != nullptr) { // integer form, pass 1
            qis[q * 4 + 0] = R;
            qis[q * 4 + 1] = musum;
            qis[q * 4 + 2] = INFINITY;
            qis[q * 4 + 3] = A;
            if (R > 0.f && R < INFINITY) {
                atomicMax(gl, __float_as_uint(R));
            }
        }
    }
    __syncthreads();
    const float A = s_A;
    // ---- the sampled rows ------------------------------------------------------------------------------------------
    int cum = 0;
    for (int slot = 0; slot < a.nslot && cum < smin && cum < scap; slot++) {
        const int64_t key = keys[q * a.nslot + slot];
        const int64_t len = (key >= 0 && key < nlist) ? a.list_len[key] : 0;
        if (len <= 0) {
            continue;
        }
        const int off = cum;
        const int rows = (int)min(len, (int64_t)(scap - cum));
        cum += rows;
        const float dis0 = a.coarse_dis[q * a.nslot + slot];
        const int64_t row_off = a.list_row_off[key];
        const float* ps = a.pq_psum + a.pq_sblk_off_r[key] * 16;
        const float slack0 = 128.0f * PF_U * (pabs_max + A) + 64.0f * PF_U * fabsf(dis0);
        // PS_ROWS rows per thread and round: their loads are in flight together (64 registers per thread: eight waves per SIMD)
        for (int pos0 = c; pos0 < rows; pos0 += PS_ROWS * PS_THREADS) {
            uint4 c0[PS_ROWS], c1[PS_ROWS];
            float psv[PS_ROWS];
#pragma unroll
            for (int u = 0; u < PS_ROWS; u++) {
                const int pos = min(pos0 + u * PS_THREADS, rows - 1); // (clamped: a valid row, its result is dropped)
                const uint4* cp = reinterpret_cast<const uint4*>(a.pq_codes + (row_off + pos) * PF_M);
                c0[u] = cp[0];
                c1[u] = cp[1];
                psv[u] = IS_L2 ? ps[pos] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < PS_ROWS; u++) {
                const int pos = pos0 + u * PS_THREADS;
                const uint32_t w[8] = {c0[u].x, c0[u].y, c0[u].z, c0[u].w, c1[u].x, c1[u].y, c1[u].z, c1[u].w};
                float acc = 0.f;
#pragma unroll
                for (int m = 0; m < PF_M; m++) {
                    acc += lut[m * PF_KSUB + ((w[m >> 2] >> (8 * (m & 3))) & 0xffu)];
                }
                float val = IS_L2 ? (dis0 + psv[u]) + acc : dis0 + acc;
                const float slack = slack0 + 64.0f * PF_U * fabsf(val);
                val = IS_L2 ? val + slack : val - slack;
                if (!(slack < INFINITY) || val != val) {
                    val = worst_dist<IS_L2>(); // (no bound from this row)
                }
                if (pos < rows) {
                    if (a.bitset != nullptr && bitset_filtered(a.bitset, a.bitset_nbits, a.ids[row_off + pos])) {
                        val = worst_dist<IS_L2>();
                    }
                    a.dump[q * a.dump_stride + off + pos] = val;
                }
            }
        }
    }
    if (c == 0) {
        n_row[q] = cum;
    }
    if (gthr_out == nullptr) {
        return;
    }
    // ---- tau_q = the ksel-th best dump value, and the candidate histogram's range [best, tau] (what a radix select of the
    // dump + ms_tau_kernel produced in round 3: 0.3 ms per batch for a value this workgroup has at hand).  The values are
    // read back into registers (this workgroup wrote them: L1 / L2 hits) as order-preserving integer keys; the ksel-th
    // smallest key comes from a bisection with one counter and one barrier per step.
    __syncthreads();
    constexpr int PER = PF_SAMPLE / PS_THREADS;
    uint32_t key[PER];
    uint32_t mn = 0xffffffffu, mx = 0u;
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const int i = c + u * PS_THREADS;
        key[u] = i < cum ? dist_key<IS_L2>(a.dump[q * a.dump_stride + i]) : 0xffffffffu;
        mn = min(mn, key[u]);
        mx = i < cum ? max(mx, key[u]) : mx;
    }
    constexpr int NSTEP = 18; // 2 bits of the interval per step (16 cover the key space; two spare for the rounding of the cuts)
    __shared__ int s_step[NSTEP][3];
    __shared__ uint32_t s_mn[PS_WAVES], s_mx[PS_WAVES];
    __shared__ uint32_t s_tmin[PS_THREADS];
    __shared__ uint32_t s_cand[PS_CAND];
    __shared__ uint32_t s_bound, s_kth;
    __shared__ int s_ncand;
    s_tmin[c] = mn; // (this thread's minimum, before the reduction below)
#pragma unroll
    for (int dlt = KN_WAVE / 2; dlt > 0; dlt >>= 1) {
        mn = min(mn, (uint32_t)__shfl_xor((int)mn, dlt, KN_WAVE));
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, dlt, KN_WAVE));
    }
    if (lane_id() == 0) {
        s_mn[wave] = mn;
        s_mx[wave] = mx;
    }
    if (c < NSTEP * 3) {
        (&s_step[0][0])[c] = 0;
    }
    if (c == 0) {
        s_ncand = 0;
    }
    __syncthreads();
    uint32_t lo = s_mn[0], hi = s_mx[0];
    for (int w = 1; w < PS_WAVES; w++) {
        lo = min(lo, s_mn[w]);
        hi = max(hi, s_mx[w]);
    }
    const uint32_t best = lo;
    bool fast = ksel <= PS_THREADS && cum >= ksel && lo < hi; // (uniform)
    if (fast) {
        // 1. the bound: ksel threads hold a key <= the ksel-th smallest thread minimum, so the ksel-th smallest key is
        //    <= it (cum >= ksel keys are spread over min(cum, PS_THREADS) >= ksel threads)
        if (wave == 0) {
            uint32_t v[PS_THREADS / KN_WAVE];
#pragma unroll
            for (int u = 0; u < PS_THREADS / KN_WAVE; u++) {
                v[u] = s_tmin[lane_id() + u * KN_WAVE];
            }
            const uint32_t b = ps_wave_select(v, ksel, lo, hi);
            if (lane_id() == 0) {
                s_bound = b;
            }
        }
        __syncthreads();
        // 2. the keys under the bound (a few more than ksel)
        const uint32_t bound = s_bound;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            if (key[u] <= bound) {
                const int at = atomicAdd(&s_ncand, 1);
                if (at < PS_CAND) {
                    s_cand[at] = key[u];
                }
            }
        }
        __syncthreads();
        const int nc = s_ncand;
        if (nc <= PS_CAND) {
            // 3. one wave: the ksel-th smallest of them
            if (wave == 0) {
                uint32_t v[PS_CAND / KN_WAVE];
#pragma unroll
                for (int u = 0; u < PS_CAND / KN_WAVE; u++) {
                    const int i = lane_id() + u * KN_WAVE;
                    v[u] = i < nc ? s_cand[i] : 0xffffffffu;
                }
                const uint32_t kk = ps_wave_select(v, ksel, lo, bound);
                if (lane_id() == 0) {
                    s_kth = kk;
                }
            }
            __syncthreads();
            lo = s_kth;
            hi = lo;
        } else {
            hi = bound; // (the ksel-th smallest is <= the bound: the block-wide cuts start from there)
            fast = false;
        }
    }
    // the ksel-th smallest key lies in [lo, hi] (when cum >= ksel): the interval is cut in four per step -- three counters,
    // one barrier -- until it is a point; the number of steps depends on the spread of the sample only (distances of one
    // sample share their exponent: ~12 steps, against 32 for a bisection of the whole key space)
    for (int it = 0; it < NSTEP && lo < hi; it++) { // (lo, hi are the same in every thread: uniform trip count)
        const uint32_t span = hi - lo;
        const uint32_t q1 = lo + (span >> 2), q2 = lo + (span >> 1), q3 = lo + (span >> 2) + (span >> 1); // lo <= q1 <= q2 <= q3 < hi
        int c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            c1 += key[u] <= q1 ? 1 : 0;
            c2 += key[u] <= q2 ? 1 : 0;
            c3 += key[u] <= q3 ? 1 : 0;
        }
#pragma unroll
        for (int dlt = KN_WAVE / 2; dlt > 0; dlt >>= 1) {
            c1 += __shfl_xor(c1, dlt, KN_WAVE);
            c2 += __shfl_xor(c2, dlt, KN_WAVE);
            c3 += __shfl_xor(c3, dlt, KN_WAVE);
        }
        if (lane_id() == 0) {
            if (c1) atomicAdd(&s_step[it][0], c1);
            if (c2) atomicAdd(&s_step[it][1], c2);
            if (c3) atomicAdd(&s_step[it][2], c3);
        }
        __syncthreads();
        const int t1 = s_step[it][0], t2 = s_step[it][1], t3 = s_step[it][2];
        if (t1 >= ksel) {
            hi = q1;
        } else if (t2 >= ksel) {
            lo = q1 + 1;
            hi = q2;
        } else if (t3 >= ksel) {
            lo = q2 + 1;
            hi = q3;
        } else {
            lo = q3 + 1;
        }
    }
    if (c == 0) {
        // fewer than ksel sampled rows (padding keys counted): no bound, exactly as a selection of ksel values would say
        const float kth = cum >= ksel ? dist_key_inv<IS_L2>(lo) : worst_dist<IS_L2>();
        gthr_out[q] = kth;
        if (gmeta_out != nullptr) {
            uint32_t glo = 0, shift = KN_HIST_OFF;
            if (kth != worst_dist<IS_L2>() && kth == kth) { // (ms_tau_kernel's rule)
                glo = best;
                const uint32_t range = dist_key<IS_L2>(kth) - glo;
                shift = 0;
                while ((range >> shift) >= (uint32_t)(KN_HIST_BINS - 1)) {
                    shift++;
                }
            }
            gmeta_out[q] = make_uint2(glo, shift);
        }
    }
}

// qis / qmu null: the integer form is off (no pass-1 record).  With qis: its batch record qis[nq * 4 ..) is reset here.
// gthr_out (+ gmeta_out) non-null: the kernel also selects tau_q = the ksel-th best sampled value itself.
hipError_t launch_pq_sample(const MScanArgs& a, const int64_t* keys, const float4* cb_m, int64_t nlist, int smin,
                            int scap, float pabs_max, bool is_l2, int32_t* n_row, float* qs, float* qis, float* qmu,
                            hipStream_t s, float* gthr_out, uint2* gmeta_out, int ksel) {
    if (a.nq <= 0) {
        return hipSuccess;
    }
    if (a.dump == nullptr || scap < 1 || scap > PF_SAMPLE || a.dump_stride < scap) {
        return hipErrorInvalidValue;
    }
    uint32_t* gl = nullptr;
    if (qis != nullptr) {
        hipError_t e = hipMemsetAsync(qis + a.nq * 4, 0, 4 * sizeof(float), s);
        if (e != hipSuccess) {
            return e;
        }
        gl = reinterpret_cast<uint32_t*>(qis + a.nq * 4);
    }
    if (is_l2) {
        hipLaunchKernelGGL(pq_sample_kernel<true>, dim3((unsigned)a.nq), dim3(PS_THREADS), 0, s, a, keys, cb_m, nlist, smin,
                           scap, pabs_max, n_row, qs, qis, qmu, gl, gthr_out, gmeta_out, ksel);
    } else {
        hipLaunchKernelGGL(pq_sample_kernel<false>, dim3((unsigned)a.nq), dim3(PS_THREADS), 0, s, a, keys, cb_m, nlist, smin,
                           scap, pabs_max, n_row, qs, qis, qmu, gl, gthr_out, gmeta_out, ksel);
    }
    return hipGetLastError();
}

__global__ void pqi_prepare_kernel(MScanArgs a, int64_t nrec) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < 8 * 16) {
        a.pq_ctr[r] = 0;
    }
    if (r >= nrec || r >= *a.nunits_dev) {
        return;
    }
    P16Rec rec{};
    const KnItem it = a.units[r];
    const int npair = it.npair < PI_Q ? it.npair : PI_Q;
    rec.list = it.list;
    rec.npair = npair;
    rec.len = a.list_len[it.list];
    rec.sblk0 = a.pq_sblk_off_r[it.list];
    rec.row_off = a.list_row_off[it.list];
    for (int j = 0; j < PI_Q; j++) {
        const KnPair p = a.pairs[it.pair0 + (j < npair ? j : npair - 1)];
        rec.q[j] = p.q;
        rec.slot[j] = p.slot;
        rec.dis0[j] = a.coarse_dis[(int64_t)p.q * a.nslot + p.slot];
    }
    a.pq_recs16[r] = rec;
}

// Integer form: a hit inside the scan loop only parks the lane's raw accumulators (32-byte record, one LDS atomic per
// wave); the fp32 re-test, the pessimistic distance and the global side happen here, spread over all 16 waves, at the
// START of the next unit -- behind that unit's table loads, whose latency covers the atomics' round trips.
constexpr int PI_STAGE_CAP = PF_STAGE_CAP / 2; // records of 32 bytes: {x[4], psum, position, first query of the lane}
constexpr int PI_PC_OFF = 576;                 // control block: per-pair constants [parity][16][4], then {query, slot}
constexpr int PI_PC_STRIDE = 384;              // 256 bytes of constants + 128 bytes of (query, slot) per parity
static_assert(PI_PC_OFF + 2 * PI_PC_STRIDE <= PF_CTL_BYTES, "control block layout");

template <bool IS_L2>
__device__ __forceinline__ void pqi_emit_raw(const MScanArgs& a, const float* pcp, const int* pqsp, const int (&x)[4],
                                             float ps, int64_t pos, int qb, float s0, float inv0, int64_t row_off) {
    const int negp = IS_L2 ? (int)ceilf(-ps * inv0) : 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (x[r] <= negp) {
            // the fp32 test of the half-precision form, bit for bit: s_q S = s0 Y (both products are exact)
            const int j = qb + r;
            const int T = __float_as_int(pcp[j * 4 + 3]);
            const int y = IS_L2 ? x[r] + T : -(x[r] + T);
            const float f = (float)y;
            const float v = __fmaf_rn(f, s0, IS_L2 ? ps : 0.f);
            const float th = pcp[j * 4];
            if (IS_L2 ? (v <= th) : (v >= th)) {
                const float pcs = pcp[j * 4 + 2];
                const float pess = IS_L2 ? __fmaf_rn(f, s0, pcs + ps) : __fmaf_rn(f, s0, pcs);
                ms_emit<IS_L2>(a, pqsp[j * 2], pqsp[j * 2 + 1], row_off, pos, pess);
            }
        }
    }
}

// One parked record per thread (PI_STAGE_CAP < PF_THREADS), in two halves: pqi_flush_issue re-tests the record and issues
// the returning atomics on the queries' candidate counters (+ the loads of their histogram origins); pqi_flush_complete
// stores the candidates once those are back.  In between the caller does everything else a unit's start needs, so the
// round trips overlap.  Together they are ms_emit (ms_common.h) of every hit of the record.
struct PiFlush { // the first hit of the thread's record (further hits of a record are rare: written out at once)
    int has;
    int32_t q, slot, n;
    float pess;
    uint2 mt;
    uint32_t pos;
};

template <bool IS_L2>
__device__ __forceinline__ void pqi_flush_issue(const MScanArgs& a, unsigned char* smem, int pp, int64_t row_off, float s0,
                                                float inv0, int wave, int lane, PiFlush& fl) {
    static_assert(PI_STAGE_CAP <= PF_THREADS, "one parked record per thread");
    fl.has = 0;
    const int* cnt = reinterpret_cast<const int*>(smem + PF_LUT_BYTES + PF_CTL_BYTES + PF_STAGE_CAP * 16);
    const int n = min(cnt[pp], PI_STAGE_CAP);
    const int i = wave + PF_WAVES * lane; // (record i -> wave i % 16: every wave takes a share)
    if (i >= n) {
        return;
    }
    const float* pcp = reinterpret_cast<const float*>(smem + PF_LUT_BYTES + PI_PC_OFF + pp * PI_PC_STRIDE);
    const int* pqsp = reinterpret_cast<const int*>(pcp + 64);
    const uint4 r0 = *reinterpret_cast<const uint4*>(smem + PF_LUT_BYTES + PF_CTL_BYTES + i * 32);
    const uint4 r1 = *reinterpret_cast<const uint4*>(smem + PF_LUT_BYTES + PF_CTL_BYTES + i * 32 + 16);
    const int x[4] = {(int)r0.x, (int)r0.y, (int)r0.z, (int)r0.w};
    const float ps = __uint_as_float(r1.x);
    const int qb = (int)r1.z;
    fl.pos = r1.y;
    const int negp = IS_L2 ? (int)ceilf(-ps * inv0) : 0;
    if (a.bitset != nullptr && bitset_filtered(a.bitset, a.bitset_nbits, a.ids[row_off + (int64_t)fl.pos])) {
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (x[r] <= negp) {
            const int j = qb + r;
            const int T = __float_as_int(pcp[j * 4 + 3]);
            const int y = IS_L2 ? x[r] + T : -(x[r] + T);
            const float f = (float)y;
            const float v = __fmaf_rn(f, s0, IS_L2 ? ps : 0.f);
            const float th = pcp[j * 4];
            if (IS_L2 ? (v <= th) : (v >= th)) {
                const float pcs = pcp[j * 4 + 2];
                const float pess = IS_L2 ? __fmaf_rn(f, s0, pcs + ps) : __fmaf_rn(f, s0, pcs);
                const int32_t q = pqsp[j * 2], slot = pqsp[j * 2 + 1];
                if (!fl.has) {
                    fl.has = 1;
                    fl.q = q;
                    fl.slot = slot;
                    fl.pess = pess;
                    fl.n = atomicAdd(a.cand_cnt + q, 1);
                    fl.mt = a.ghist != nullptr ? a.gmeta[q] : make_uint2(0u, KN_HIST_OFF);
                } else {
                    MScanArgs b = a;
                    b.bitset = nullptr; // (tested above)
                    ms_emit<IS_L2>(b, q, slot, row_off, (int64_t)fl.pos, pess);
                }
            }
        }
    }
}

template <bool IS_L2>
__device__ __forceinline__ void pqi_flush_complete(const MScanArgs& a, const PiFlush& fl) {
    if (!fl.has) {
        return;
    }
    const int32_t q = fl.q;
    if (fl.n < a.cap) {
        a.cand[(int64_t)q * a.cap + fl.n] = ((int64_t)fl.slot << 32) | (int64_t)fl.pos;
        if (a.cand_pess != nullptr) {
            a.cand_pess[(int64_t)q * a.cap + fl.n] = fl.pess;
        }
    } else {
        a.overflow[q] = 1;
        a.overflow[a.nq] = 1;
    }
    if (fl.mt.y != KN_HIST_OFF) {
        atomicAdd(a.ghist + (int64_t)q * KN_HIST_BINS + hist_bin(dist_key<IS_L2>(fl.pess), fl.mt.x, fl.mt.y), 1u);
    }
}

// The filter kernel of the integer form.  Same persistent structure as pqf_kernel (two-deep mailbox); the unit's table
// pieces are fetched at its start (L2 hits) so that the scan keeps eight gathers per wave in flight; no sample mode
// (the sample pass stays in half precision).
template <bool IS_L2>
__global__ __launch_bounds__(PF_THREADS) void pqi_kernel(MScanArgs a) {
#ifdef KNHIP_PHASE_TIMERS
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
#endif
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int REC_WORDS = (int)(sizeof(P16Rec) / 4);
    static_assert(REC_WORDS == 64, "mailbox slots of 64 words, one lane per word");
    static_assert(offsetof(P16Rec, npair) == 4 && offsetof(P16Rec, q) == 8 && offsetof(P16Rec, slot) == 72 &&
                  offsetof(P16Rec, dis0) == 136 && offsetof(P16Rec, len) == 200 && offsetof(P16Rec, sblk0) == 208 &&
                  offsetof(P16Rec, row_off) == 216, "P16Rec layout");
    // behind the LUT: ints [0], [1] unit index of mailbox slot 0 / 1, [8 + 64 s ..) the record of slot s;
    // from byte PI_PC_OFF, per unit parity: per-pair constants [16][4] = {t, step, pess const, bits of the integer
    // threshold T}, then [16][2] = {query, slot} (the previous unit's are read by its flush while this unit's are written)
    int* ctl = reinterpret_cast<int*>(smem + PF_LUT_BYTES);
    static_assert(8 * 4 + 2 * REC_WORDS * 4 <= PI_PC_OFF, "control block layout");
    const float s0 = pf_sgpr_f(a.pq_qis[(int64_t)a.nq * 4 + 1]); // the batch's base step (a power of two) and 1 / s0
    const float inv0 = pf_sgpr_f(a.pq_qis[(int64_t)a.nq * 4 + 2]);
    const int lane = lane_id();
    const int wave = pf_sgpr((int)(threadIdx.x / KN_WAVE));
    if ((uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)smem) != 0u) {
        __builtin_trap(); // the 16-bit tokens assume the LUT at LDS offset 0
    }
    const int nunits = (int)*a.nunits_dev;
    const int per = (nunits + 7) / 8;
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int xcd = (int)(xcc & 7u);
    int fetch_t = 0; // thread 0: counters [xcd, xcd + fetch_t) are known to be exhausted
    auto fetch_slow = [&]() -> int {
        while (fetch_t < 8) {
            const int x = (xcd + fetch_t) & 7;
            const int base = x * per;
            const int cnt = min(per, nunits - base);
            if (cnt > 0) {
                const int i = atomicAdd(a.pq_ctr + x * 16, 1);
                if (i < cnt) {
                    return base + i;
                }
            }
            fetch_t++;
        }
        return -1;
    };
    if (wave == 0) {
        int u0 = -1, u1 = -1;
        if (lane == 0) {
            u0 = fetch_slow();
            u1 = u0 >= 0 ? fetch_slow() : -1;
        }
        u0 = __builtin_amdgcn_readlane(u0, 0);
        u1 = __builtin_amdgcn_readlane(u1, 0);
        if (u0 >= 0) {
            ctl[8 + lane] = (int)reinterpret_cast<const uint32_t*>(a.pq_recs16 + u0)[lane];
        }
        if (u1 >= 0) {
            ctl[8 + REC_WORDS + lane] = (int)reinterpret_cast<const uint32_t*>(a.pq_recs16 + u1)[lane];
        }
        if (lane == 0) {
            ctl[0] = u0;
            ctl[1] = u1;
            int* scnt = reinterpret_cast<int*>(smem + PF_LUT_BYTES + PF_CTL_BYTES + PF_STAGE_CAP * 16);
            scnt[0] = 0;
            scnt[1] = 0;
        }
    }
    __syncthreads();
    int cur = pf_sgpr(ctl[0]);
    int par = 0;
    const uint32_t one = 1u;
    const uint2* qi2 = reinterpret_cast<const uint2*>(a.pq_qi);
    // this thread's 8-byte piece of the 16 query tables of the unit in mailbox slot `slot`
    auto load_tables = [&](int slot, uint2 (&t)[PI_Q], int lane_i) {
        const uint32_t rq = (uint32_t)ctl[8 + slot * REC_WORDS + lane_i];
#pragma unroll
        for (int j = 0; j < PI_Q; j++) {
            // (a scalar base per table + one 32-bit lane offset: saddr loads, no 64-bit address arithmetic per table)
            const int64_t q = (int64_t)(int32_t)__builtin_amdgcn_readlane((int)rq, 2 + j);
            const unsigned char* tq = reinterpret_cast<const unsigned char*>(qi2) + q * (PF_KSUB * PF_M);
            t[j] = *reinterpret_cast<const uint2*>(tq + (size_t)((uint32_t)(wave * KN_WAVE + lane_i) * 8u));
        }
    };

    int64_t prev_row_off = -1; // row offset of the unit whose parked hits are still to be written out

    while (cur >= 0) {
        PF_T(5);
        int lane_i = lane;
        asm volatile("" : "+v"(lane_i)); // nothing lane-derived is hoisted out of the unit loop
        int f_x = 0, f_i = 0;
        if (wave == 0 && lane_i == 0 && fetch_t < 8) {
            f_x = (xcd + fetch_t) & 7;
            f_i = atomicAdd(a.pq_ctr + f_x * 16, 1);
        }
        const uint32_t rw = (uint32_t)ctl[8 + par * REC_WORDS + lane_i];
        const int nxt_unit = pf_sgpr(ctl[par ^ 1]);
        auto rl = [&](int i) { return (uint32_t)__builtin_amdgcn_readlane((int)rw, i); };
        const int npair = (int)rl(1);
        const int64_t len = (int64_t)(((uint64_t)rl(51) << 32) | rl(50));
        const int64_t sblk0 = (int64_t)(((uint64_t)rl(53) << 32) | rl(52));
        const int64_t row_off = (int64_t)(((uint64_t)rl(55) << 32) | rl(54));
        // (the registers hold four value buffers in the scan, not the next unit's tables: round 3 experiments)
        uint2 tp[PI_Q];
        load_tables(par, tp, lane_i);
        float* pc = reinterpret_cast<float*>(smem + PF_LUT_BYTES + PI_PC_OFF + par * PI_PC_STRIDE);
        int* pqs = reinterpret_cast<int*>(pc + 64);
        // this wave's groups of 16 vectors (one code block each)
        const int ngroups = (int)((len + 15) / 16);
        const int gbase = ngroups / PF_WAVES, grem = ngroups % PF_WAVES;
        const int G0 = wave * gbase + min(wave, grem);
        const int G1 = G0 + gbase + (wave < grem ? 1 : 0);
        const int nwin = G1 - G0;
        const uint4* cbase = a.pq_codes_i + (sblk0 + (int64_t)G0) * 64 + lane_i;
        auto load_blk = [&](int b) { return cbase[(int64_t)b * 64]; }; // past-the-end blocks exist (slack)
        const float* psb = a.pq_psum + sblk0 * 16 + (int64_t)G0 * 16 + (lane_i & 15);
        uint4 U0 = make_uint4(0, 0, 0, 0), U1 = U0, U2 = U0, U3 = U0;
        if (nwin > 0) {
            U0 = load_blk(0);
            U1 = load_blk(1);
            U2 = load_blk(2);
            U3 = load_blk(3);
        }
        // ---- everything this unit's start waits for is issued first: table pieces and code blocks (above), the pair's
        // constants, the atomics of the previous unit's parked hits -- one round trip instead of five in a row
        // wave j prepares pair j (the lane index of a readlane may be a scalar register)
        const int32_t pq_q = (int32_t)rl(2 + wave), pq_slot = (int32_t)rl(18 + wave);
        const float pq_dis0 = __uint_as_float(rl(34 + wave));
        int32_t pq_qv = pq_q; // (in a vector register: the loads below stay vector loads whose results nobody reads early)
        asm volatile("" : "+v"(pq_qv));
        const float4 s4 = *reinterpret_cast<const float4*>(a.pq_qis + (int64_t)pq_qv * 4);
        float tau_g = a.gthr[pq_qv];
        MsHist hst;
        ms_hist_load(a, pq_qv, hst);
        PiFlush fl;
        fl.has = 0;
        if (prev_row_off >= 0) {
            pqi_flush_issue<IS_L2>(a, smem, par ^ 1, prev_row_off, s0, inv0, wave, lane_i, fl);
        }
        asm volatile("" : "+v"(tau_g)::"memory"); // (the loaded constants are looked at below this line)
        // ---- per-pair constants -------------------------------------------------------------------------------------
        {
            const int32_t q = pq_q, slot = pq_slot;
            const float dis0 = pq_dis0;
            float t = IS_L2 ? -INFINITY : INFINITY, pcst = 0.f; // pass-nothing defaults
            if (wave < npair) {
                const float tau = tighter<IS_L2>(tau_g, ms_hist_eval<IS_L2>(hst, a.k));
                const float eps = s4.z + 64.0f * PF_U * (fabsf(dis0) + fabsf(tau) + fabsf(s4.y));
                if (tau == worst_dist<IS_L2>() || !(eps < INFINITY)) {
                    if (lane_i == 0) {
                        a.overflow[q] = 1;
                        a.overflow[a.nq] = 1;
                    }
                } else {
                    t = IS_L2 ? ((tau + eps) - dis0) - s4.y : ((tau - eps) - dis0) - s4.y;
                    pcst = IS_L2 ? (dis0 + s4.y) + eps : (dis0 + s4.y) - eps;
                }
            }
            // integer threshold in units of s0: whatever passes `ps + s Y <= t` (L2; IP: `s Y >= t`) in fp32 also
            // satisfies Y + floor(ps / s0) <= T (IP: -Y <= T): 2^-22 |t| covers the one rounding of the fp32 form
            int T;
            {
                const float tt = IS_L2 ? t : -t;
                float x = floorf((tt + 2.384185791015625e-7f * fabsf(tt)) * inv0) + 2.0f;
                x = fminf(fmaxf(x, -2.0f * PI_INT_LIM), 2.0f * PI_INT_LIM); // (NaN -> the lower clamp: nothing passes)
                T = (int)x;
            }
            if (lane_i == 0) {
                *reinterpret_cast<float4*>(pc + wave * 4) = make_float4(t, s4.x, pcst, __int_as_float(T));
                pqs[wave * 2] = q;
                pqs[wave * 2 + 1] = slot;
            }
        }
        // ---- LUT[c][m][16 queries]: 16 x 8 bytes transposed in registers, 8 conflict-free 16-byte stores ------------
        {
            const int t = wave * KN_WAVE + lane_i;
            const int c4 = t >> 4, l16 = t & 15;
            uint4* lut = reinterpret_cast<uint4*>(smem);
#pragma unroll
            for (int w = 0; w < 2; w++) { // word w of a piece: cells (cc = 2 w, h = 0), (2 w, 1), (2 w + 1, 0), (2 w + 1, 1)
                uint32_t r[PI_Q];
#pragma unroll
                for (int j = 0; j < PI_Q; j++) {
                    r[j] = w ? tp[j].y : tp[j].x;
                }
                // level 1: queries (2 i, 2 i + 1) -> bytes {b0 q, b0 q', b1 q, b1 q'} and {b2 q, b2 q', b3 q, b3 q'}
                uint32_t lo[8], hi[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    lo[i] = __builtin_amdgcn_perm(r[2 * i + 1], r[2 * i], 0x05010400u);
                    hi[i] = __builtin_amdgcn_perm(r[2 * i + 1], r[2 * i], 0x07030602u);
                }
#pragma unroll
                for (int e = 0; e < 4; e++) { // byte e of the word = cell (cc = 2 w + (e >> 1), h = e & 1)
                    const uint32_t sel = (e & 1) ? 0x07060302u : 0x05040100u;
                    const uint32_t* src = (e & 2) ? hi : lo;
                    uint4 o;
                    o.x = __builtin_amdgcn_perm(src[1], src[0], sel);
                    o.y = __builtin_amdgcn_perm(src[3], src[2], sel);
                    o.z = __builtin_amdgcn_perm(src[5], src[4], sel);
                    o.w = __builtin_amdgcn_perm(src[7], src[6], sel);
                    const int cc = 2 * w + (e >> 1), h = e & 1;
                    lut[(c4 * 4 + cc) * PF_M + l16 + 16 * h] = o;
                }
            }
        }
        pqi_flush_complete<IS_L2>(a, fl);
        PF_T(0);
        __syncthreads();
        PF_T(1);
        if (threadIdx.x == 0) { // the NEXT unit's staging counter (its last user's flush ended before this barrier)
            reinterpret_cast<int*>(smem + PF_LUT_BYTES + PF_CTL_BYTES + PF_STAGE_CAP * 16)[par ^ 1] = 0;
        }
        int nxt = -1;
        uint32_t rw_next = 0;
        if (wave == 0) {
            if (lane_i == 0) {
                if (fetch_t < 8 && nxt_unit >= 0) {
                    const int base = f_x * per;
                    const int cnt = min(per, nunits - base);
                    if (f_i < cnt) {
                        nxt = base + f_i;
                    } else {
                        fetch_t++;
                        nxt = fetch_slow();
                    }
                }
            }
            nxt = __builtin_amdgcn_readlane(nxt, 0);
            if (nxt >= 0) {
                rw_next = reinterpret_cast<const uint32_t*>(a.pq_recs16 + nxt)[lane_i];
            }
        }
        // this lane's 4 accumulator rows are the queries 4 (lane >> 4) + r of vector lane & 15
        const int qb = 4 * (lane_i >> 4);
        pf_i4 negT; // every group's accumulator starts at -T of the lane's four queries
#pragma unroll
        for (int r = 0; r < 4; r++) {
            negT[r] = -__float_as_int(pc[(qb + r) * 4 + 3]);
        }
        // selector: row i = lane & 15 takes byte i of every k block, times the query's step weight (negated for IP:
        // the accumulator then holds -Y and "small is good" holds for both metrics)
        pf_i4 sel;
        {
            const int i = lane_i & 15;
            const int wq = (int)rintf(pc[i * 4 + 1] * inv0); // s_q / s0: an integer 1 .. 127 by construction
            const uint32_t wb = (uint32_t)(IS_L2 ? wq : -wq) & 0xffu;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                sel[w] = (i >> 2) == w ? (int)(wb << (8 * (i & 3))) : 0;
            }
        }
        typedef __attribute__((address_space(3))) const pf_i4 lds_i4;
        auto lut_read = [&](uint32_t addr) -> pf_i4 { return *reinterpret_cast<lds_i4*>(addr); };
        auto issue2 = [&](uint32_t w0, pf_i4 (&v)[2]) {
            v[0] = lut_read(pf_addr_lo(w0, one));
            v[1] = lut_read(pf_addr_hi(w0, one));
        };
        PF_T(2);
        if (nwin > 0) {
            pf_i4 B0[2], B1[2], B2[2], B3[2];
            uint4 W0 = U0;
            issue2(W0.x, B0); // (in this order: the loop's first unit waits for the oldest two reads only)
            __builtin_amdgcn_sched_barrier(0);
            issue2(W0.y, B1);
            __builtin_amdgcn_sched_barrier(0);
            issue2(W0.z, B2);
            __builtin_amdgcn_sched_barrier(0);
            issue2(W0.w, B3);
            __builtin_amdgcn_sched_barrier(0);
            pf_i4 e0 = negT, o0 = negT; // even / odd windows: one accumulator chain each, started at -T
            const int vec = lane_i & 15;
            float ps_cur[4] = {0.f, 0.f, 0.f, 0.f}, ps_prev = 0.f;
            if (IS_L2) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    ps_cur[u] = psb[u * 16];
                }
            }

#define PI_UNIT(ACC, CIN, BUF, WORD)                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    ACC = __builtin_amdgcn_mfma_i32_16x16x64_i8(sel, BUF[0], CIN, 0, 0, 0);                  \
    ACC = __builtin_amdgcn_mfma_i32_16x16x64_i8(sel, BUF[1], ACC, 0, 0, 0);                  \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    issue2(WORD, BUF);

            auto finish = [&](const pf_i4& x, int G, float ps) {
#ifdef KNHIP_EXPERIMENT_NOFINISH
                asm volatile("" ::"v"(x), "v"(ps));
                return;
#endif
                // fast path, integer: x[r] = (+-)Y_r - T_r in units of s0; something passes iff the smallest of the lane's
                // four is <= ceil(-ps / s0) (a superset of the fp32 test below: see the per-pair constants).  Whether the
                // row exists at all is only looked at when something passed.
                const int negp = IS_L2 ? (int)ceilf(-ps * inv0) : 0;
                const int mn = min(min(x[0], x[1]), min(x[2], x[3]));
                if (__builtin_expect(__ballot(mn <= negp) != 0ull, 0)) { // (out of line: the hot path falls through)
#ifdef KNHIP_EXPERIMENT_NOSLOW
                    asm volatile("s_nop 0");
                    return;
#endif
                    const int64_t pos = (int64_t)G * 16 + vec;
                    if (G < G1 && mn <= negp && pos < len) {
                        int* cnt = reinterpret_cast<int*>(smem + PF_LUT_BYTES + PF_CTL_BYTES + PF_STAGE_CAP * 16);
                        const int n = atomicAdd(cnt + par, 1); // (one LDS atomic per wave: the compiler aggregates)
                        if (n < PI_STAGE_CAP) {
                            unsigned char* rec = smem + PF_LUT_BYTES + PF_CTL_BYTES + n * 32;
                            *reinterpret_cast<uint4*>(rec) = make_uint4((uint32_t)x[0], (uint32_t)x[1], (uint32_t)x[2],
                                                                        (uint32_t)x[3]);
                            *reinterpret_cast<uint4*>(rec + 16) = make_uint4(__float_as_uint(ps), (uint32_t)pos,
                                                                             (uint32_t)qb, 0u);
                        } else { // (staging full: straight to the candidate list)
                            const int xs[4] = {x[0], x[1], x[2], x[3]};
                            pqi_emit_raw<IS_L2>(a, pc, pqs, xs, ps, pos, qb, s0, inv0, row_off);
                        }
                    }
                }
            };

            const int niter = (nwin + 3) >> 2;
            for (int i = 0; i < niter; i++) {
                pf_setprio((i + (wave >> 2)) & 3);
                const float ps0 = ps_cur[0], ps1 = ps_cur[1], ps2 = ps_cur[2], ps3 = ps_cur[3];
                if (IS_L2) {
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        ps_cur[u] = psb[(int64_t)(4 * i + 4 + u) * 16]; // (the slack groups behind a list exist)
                    }
                }
                // four value buffers: a read is consumed four units (eight steps) after it was issued
                PI_UNIT(e0, negT, B0, U1.x)
                PI_UNIT(e0, e0, B1, U1.y)
                if (i > 0) {
                    finish(o0, G0 + 4 * i - 1, ps_prev);
                }
                PI_UNIT(e0, e0, B2, U1.z)
                PI_UNIT(e0, e0, B3, U1.w)
                W0 = load_blk(4 * i + 4);
                PI_UNIT(o0, negT, B0, U2.x)
                PI_UNIT(o0, o0, B1, U2.y)
                finish(e0, G0 + 4 * i, ps0);
                PI_UNIT(o0, o0, B2, U2.z)
                PI_UNIT(o0, o0, B3, U2.w)
                U1 = load_blk(4 * i + 5);
                PI_UNIT(e0, negT, B0, U3.x)
                PI_UNIT(e0, e0, B1, U3.y)
                finish(o0, G0 + 4 * i + 1, ps1);
                PI_UNIT(e0, e0, B2, U3.z)
                PI_UNIT(e0, e0, B3, U3.w)
                U2 = load_blk(4 * i + 6);
                PI_UNIT(o0, negT, B0, W0.x)
                PI_UNIT(o0, o0, B1, W0.y)
                finish(e0, G0 + 4 * i + 2, ps2);
                PI_UNIT(o0, o0, B2, W0.z)
                PI_UNIT(o0, o0, B3, W0.w)
                U3 = load_blk(4 * i + 7);
                __builtin_amdgcn_sched_barrier(0);
                ps_prev = ps3;
            }
            finish(o0, G0 + 4 * niter - 1, ps_prev);
#undef PI_UNIT
            __builtin_amdgcn_s_setprio(0);
        }
        PF_T(3);
        PF_COUNT(6, nwin);
        if (wave == 0) { // park the unit after the next
            ctl[8 + par * REC_WORDS + lane_i] = (int)rw_next;
            if (lane_i == 0) {
                ctl[par] = nxt;
            }
        }
        __syncthreads();
        PF_T(4);
        PF_COUNT(7, 1);
        prev_row_off = row_off; // (its parked hits go out at the start of the next unit, or below)
        cur = nxt_unit;
        par ^= 1;
    }
    if (prev_row_off >= 0) { // the last unit's parked hits
        PiFlush fl;
        pqi_flush_issue<IS_L2>(a, smem, par ^ 1, prev_row_off, s0, inv0, wave, lane, fl);
        pqi_flush_complete<IS_L2>(a, fl);
    }
#ifdef KNHIP_PHASE_TIMERS
    if (lane == 0) {
        for (int i = 0; i < 8; i++) {
            atomicAdd(&g_pf_prof[wave * 8 + i], tacc[i]);
        }
    }
#endif
}


hipError_t launch_pqi(const MScanArgs& a, bool is_l2, int64_t units_bound, hipStream_t s) {
    if (units_bound <= 0) {
        return hipSuccess;
    }
    auto kern = is_l2 ? pqi_kernel<true> : pqi_kernel<false>;
    const int nthreads = PF_THREADS;
    const size_t sm = pqf_smem();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)sm);
    if (e != hipSuccess) {
        return e;
    }
    hipLaunchKernelGGL(pqi_prepare_kernel, dim3((unsigned)((std::max<int64_t>(units_bound, 128) + 255) / 256)), dim3(256),
                       0, s, a, units_bound);
    int dev = 0, ncu = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (ncu <= 0) {
        ncu = 256;
    }
    const int64_t wgs = std::max<int64_t>(1, std::min<int64_t>(ncu, units_bound));
#ifdef KNHIP_PHASE_TIMERS
    static unsigned long long zero[128] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pf_prof), zero, sizeof(zero));
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(nthreads), sm, s, a);
#ifdef KNHIP_PHASE_TIMERS
    (void)hipStreamSynchronize(s);
    unsigned long long h[128];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pf_prof), sizeof(h));
    fprintf(stderr, "[pqf timers] int8 filter ticks per unit: wave | lut wait1 setup windows wait2 top | windows/unit\n");
    for (int w = 0; w < nthreads / KN_WAVE; w++) {
        const unsigned long long* r = h + w * 8;
        const double n = r[7] ? (double)r[7] : 1.0;
        fprintf(stderr, "[pqf timers] %2d | %6.0f %6.0f %6.0f %6.0f %6.0f %5.0f | %.2f   (units %llu)\n", w, r[0] / n, r[1] / n,
                r[2] / n, r[3] / n, r[4] / n, r[5] / n, r[6] / n, r[7]);
    }
#endif
    return hipGetLastError();
}

size_t pqf_smem() {
    return (size_t)PF_LUT_BYTES + PF_CTL_BYTES + PF_STAGE_BYTES;
}
static_assert(PF_LUT_BYTES + PF_CTL_BYTES + PF_STAGE_BYTES <= 160 * 1024, "LDS of one workgroup");

bool pqf_supports(int M, int d) {
    return M == PF_M && d == PF_M * PF_DSUB;
}

// units_bound: upper bound of *a.nunits_dev (sizes pq_recs)
hipError_t launch_pqf(const MScanArgs& a, bool is_l2, int64_t units_bound, hipStream_t s) {
    if (units_bound <= 0) {
        return hipSuccess;
    }
    const bool dump = a.dump != nullptr;
    if (dump && a.dump_stride != PF_SAMPLE) {
        return hipErrorInvalidValue; // (the sample pass writes at most PF_SAMPLE columns per query)
    }
    auto kern = is_l2 ? (dump ? pqf_kernel<true, true> : pqf_kernel<true, false>)
                      : (dump ? pqf_kernel<false, true> : pqf_kernel<false, false>);
    const size_t sm = pqf_smem();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)sm);
    if (e != hipSuccess) {
        return e;
    }
    hipLaunchKernelGGL(pqf_prepare_kernel, dim3((unsigned)((std::max<int64_t>(units_bound, 128) + 255) / 256)), dim3(256),
                       0, s, a, units_bound);
    int dev = 0, ncu = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (ncu <= 0) {
        ncu = 256;
    }
    const int64_t wgs = std::max<int64_t>(1, std::min<int64_t>(ncu, units_bound));
#ifdef KNHIP_PHASE_TIMERS
    static unsigned long long zero[128] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pf_prof), zero, sizeof(zero));
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(PF_THREADS), sm, s, a);
#ifdef KNHIP_PHASE_TIMERS
    (void)hipStreamSynchronize(s);
    unsigned long long h[128];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pf_prof), sizeof(h));
    fprintf(stderr, "[pqf timers] %s ticks per unit: wave | lut wait1 setup windows wait2 top | windows/unit\n",
            dump ? "sample" : "filter");
    for (int w = 0; w < 16; w++) {
        const unsigned long long* r = h + w * 8;
        const double n = r[7] ? (double)r[7] : 1.0;
        fprintf(stderr, "[pqf timers] %2d | %6.0f %6.0f %6.0f %6.0f %6.0f %5.0f | %.2f   (units %llu)\n", w, r[0] / n, r[1] / n,
                r[2] / n, r[3] / n, r[4] / n, r[5] / n, r[6] / n, r[7]);
    }
#endif
    return hipGetLastError();
}

} // namespace knhip
