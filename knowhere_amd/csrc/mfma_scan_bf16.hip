// knowhere_amd/csrc/mfma_scan_bf16.hip -- the IVF-Flat filter pass of mfma_scan.hip on the bf16 matrix pipe (gfx950, round 5).
//
// mscan_flat_kernel (mfma_scan.hip) multiplies fp32 rows by fp32 queries on v_mfma_f32_32x32x2_f32: 64 flop per cycle and
// SIMD, the kernel sits at 0.61 of that peak and cannot go further.  But the pass is a PREFILTER (every survivor is
// recomputed in the reference's arithmetic by mscan_finish_kernel): it needs a bounded error, not fp32 products.  Here both
// operands are split into two bf16 terms, x = hi + lo + r (bf16 carries 8 significant bits and the conversion rounds to
// nearest: |x - hi| <= 2^-8 |x|, |r| <= 2^-16 |x|), and the product is taken as hi hi + hi lo + lo hi on
// v_mfma_f32_32x32x16_bf16 -- bf16 products are exact in fp32, the accumulation is fp32 -- three instructions of a pipe
// sixteen times as fast.  Dropped: lo lo + r_q x + q r_x, so
//     |dot - exact| <= (3 * 2^-16 + 3 d 2^-24) ||q|| ||x||,
// and the caller's eps_scale carries 2^-14 on top of the fp32 kernel's term (L2: the distance has twice the dot's error and
// 2 ab <= a^2 + b^2; tests/test_coarse_bf16_bound.py replays the arithmetic).  Everything around the product is the fp32
// kernel's: units of (list, <= QT queries), rows read straight from the interleaved fp32 blocks (a half-wave = 32 rows x
// two 16-byte chunks = 8 dimensions per lane: exactly one lane's share of the A operand), the accumulator started at
// -||x||^2 / 2 by one fp32 matrix instruction, one compare per (row, query), candidates through ms_emit.  What changes with
// the faster pipe is the balance: the rows now arrive too slowly for 64 queries per unit (25 GB per launch at C2), so a
// unit takes up to 128 queries (four tiles of 32; tiles beyond the unit's pairs are skipped, so a short unit costs what
// its pairs cost) and a list is streamed half as often.  The rows are converted in registers (two v_cvt_pk_bf16_f32, two
// shifts, two subtractions per pair of values -- issued beside the matrix instructions of the previous step); the queries
// are split once per unit into LDS.
//
// Reference semantics replaced: as mscan_flat_kernel (IVFFlatScanner::scan_codes, thirdparty/faiss/faiss/cppcontrib/
// knowhere/IndexIVFFlat.cpp:193-236) -- only WHICH rows reach the exact finish, never a returned value.
#include "common.h"
#include "kernels.h"
#include "ms_common.h"

namespace knhip {

typedef float mb_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 mb_bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 mb_bf4 __attribute__((ext_vector_type(4)));

constexpr int MB_WAVES = 4;
constexpr int MB_THREADS = MB_WAVES * KN_WAVE;
constexpr int MB_HITS = 448; // parked hits per unit (7 KB of LDS; more are appended on the spot)

// bytes of one query's split row in LDS: per step of 16 dimensions 16 hi + 16 lo bf16, + 16 bytes so that the rows of
// consecutive queries start an odd number of 16-byte quads apart (conflict-free ds_read_b128 across a query tile)
__host__ __device__ inline int mb_pitch(int nstep) {
    return nstep * 64 + 16;
}

__device__ __forceinline__ void mb_split8(const float4& f0, const float4& f1, mb_bf8& hi, mb_bf8& lo) {
    const float v[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const __bf16 h = (__bf16)v[e];
        hi[e] = h;
        lo[e] = (__bf16)(v[e] - (float)h); // (inf / NaN rows give NaN: never >= a threshold, never a candidate)
    }
}

template <bool IS_L2, int NQT>
__device__ __forceinline__ void mscan_flatb_unit(const MScanArgs a, const int64_t u, unsigned char* smem) {
    constexpr int QT = 32 * NQT;
    const int lane = lane_id();
    const int wave = threadIdx.x / KN_WAVE;
    const KnItem it = a.units[u];
    const int npair = it.npair;
    const int ntq = (npair + 31) >> 5; // query tiles in use (uniform)
    const int64_t list = it.list;
    const int64_t len = a.list_len[list];
    const int64_t blk0 = a.list_blk_off[list];
    const int64_t row_off = a.list_row_off[list];
    const int nchunk = a.nchunk;
    const int nstep = a.nstep; // steps of 16 dims (4 chunks)
    const int pitch = mb_pitch(nstep);

    unsigned char* sQ = smem;                                        // [QT][pitch]
    float* sT = reinterpret_cast<float*>(smem + (size_t)QT * pitch); // [QT] accumulator threshold
    float* sC = sT + QT;                                             // [QT] pessimistic distance = c - 2 acc (L2) / acc - c (IP)
    int32_t* sPq = reinterpret_cast<int32_t*>(sC + QT);              // [QT] query of the pair (-1: none)
    int32_t* sPs = sPq + QT;                                         // [QT] slot of the pair
    uint4* sHit = reinterpret_cast<uint4*>(sPs + QT);                  // [MB_HITS] parked hits {pair, position, value bits, -}
    int32_t* sNhit = reinterpret_cast<int32_t*>(sHit + MB_HITS);
    if (threadIdx.x == 0) {
        *sNhit = 0;
    }
    // thread per pair (32 ntq <= 128 of them): every pair's record, norm, tau and histogram row are requested together --
    // with a wave per pair, as the fp32 kernel has it, the 32 pairs a wave walks cost a memory round trip each, one
    // after the other: longer than the unit's scan on this pipe
    for (int j = threadIdx.x; j < 32 * ntq; j += MB_THREADS) {
        float t = INFINITY, c = 0.f;
        int32_t q = -1, slot = 0;
        if (j < npair) {
            const KnPair p = a.pairs[it.pair0 + j];
            q = p.q;
            slot = p.slot;
            const float qn = a.qnorm[q];
            const float eps = a.eps_scale * (IS_L2 ? (qn + a.xnorm_max) : sqrtf(qn * a.xnorm_max)) + 1e-30f;
            c = IS_L2 ? qn + eps : eps;
            float tau = a.gthr[q];
            tau = tighter<IS_L2>(tau, ms_hist_bound_lane<IS_L2>(a, q, a.k));
            if (tau == worst_dist<IS_L2>()) {
                // no bound (fewer than k unfiltered rows in the sample): every row would pass -> exact fallback
                a.overflow[q] = 1;
                a.overflow[a.nq] = 1;
            } else {
                t = IS_L2 ? (qn - tau - eps) * 0.5f : tau - eps;
            }
        }
        sT[j] = t;
        sC[j] = c;
        sPq[j] = q;
        sPs[j] = slot;
    }
    __syncthreads();
    // the queries, split: thread = (pair, chunk of 4 dims); dims past d and pairs past npair are zero
    const bool vec4 = (a.d & 3) == 0 && (reinterpret_cast<uintptr_t>(a.queries) & 15) == 0; // (one load per chunk)
#pragma unroll 4
    for (int t = threadIdx.x; t < 32 * ntq * nstep * 4; t += MB_THREADS) {
        const int j = t / (nstep * 4), c = t % (nstep * 4);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (j < npair) {
            const float* src = a.queries + (int64_t)sPq[j] * a.d + c * 4;
            if (vec4) {
                if (c * 4 < a.d) {
                    const float4 f = *reinterpret_cast<const float4*>(src);
                    v[0] = f.x;
                    v[1] = f.y;
                    v[2] = f.z;
                    v[3] = f.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (c * 4 + e < a.d) {
                        v[e] = src[e];
                    }
                }
            }
        }
        mb_bf4 h, l;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const __bf16 hh = (__bf16)v[e];
            h[e] = hh;
            l[e] = (__bf16)(v[e] - (float)hh);
        }
        unsigned char* dst = sQ + (size_t)j * pitch + (c >> 2) * 64 + (c & 3) * 8;
        *reinterpret_cast<mb_bf4*>(dst) = h;
        *reinterpret_cast<mb_bf4*>(dst + 32) = l;
    }
    __syncthreads();

    const int hi = lane >> 5, lr = lane & 31;
    const int64_t nblk = (len + 63) >> 6;
    if (nblk <= 0) {
        return;
    }
    const float4* rows = reinterpret_cast<const float4*>(a.rows) + blk0 * (int64_t)nchunk * 64;
    // A operand of one step (16 dims): lane (row lr of tile t, half hi) takes chunks 4 s + 2 hi and + 1 = its 8 dims.
    // Branch-free as in mscan_flat_kernel: a prefetch past this wave's last block re-reads that block (never used); a chunk
    // past the last one re-reads the last chunk, whose query operand is zero in LDS.
    auto load_step = [&](int64_t b, int s, float4 (&A)[2][2]) {
        const int64_t bb = min(b, nblk - 1);
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int c = min(4 * s + 2 * hi + e, nchunk - 1);
            const float4* p = rows + (bb * nchunk + c) * 64 + lr;
            A[0][e] = p[0];
            A[1][e] = p[32];
        }
    };
    float4 A[2][2][2]; // two statically rotating row buffers
    int64_t lb = wave; // load cursor
    int ls = 0;
    auto issue = [&](float4 (&dst)[2][2]) {
        load_step(lb, ls, dst);
        if (++ls == nstep) {
            ls = 0;
            lb += MB_WAVES;
        }
    };
    mb_f32x16 acc[2][NQT];
    auto init_acc = [&](int64_t b) {
        // L2: acc = -||x||^2 / 2 for the tile's rows (one fp32 matrix instruction: k = 0 carries the norm, k = 1 nothing)
        mb_f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            z[r] = 0.f;
        }
        mb_f32x16 i0 = z, i1 = z;
        if (IS_L2) {
            const int64_t bb = min(b, nblk - 1);
            float xn0 = a.xnorm[(blk0 + bb) * 64 + lr];
            float xn1 = a.xnorm[(blk0 + bb) * 64 + 32 + lr];
            xn0 = hi == 0 ? xn0 : 0.f;
            xn1 = hi == 0 ? xn1 : 0.f;
            const float mh = hi == 0 ? -0.5f : 0.f;
            i0 = __builtin_amdgcn_mfma_f32_32x32x2f32(xn0, mh, z, 0, 0, 0);
            i1 = __builtin_amdgcn_mfma_f32_32x32x2f32(xn1, mh, z, 0, 0, 0);
        }
#pragma unroll
        for (int qt = 0; qt < NQT; qt++) {
            acc[0][qt] = i0;
            acc[1][qt] = i1;
        }
    };
    auto compute = [&](const float4 (&Ac)[2][2], int s) {
        mb_bf8 ah[2], al[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            mb_split8(Ac[t][0], Ac[t][1], ah[t], al[t]);
        }
        const unsigned char* bq = sQ + (size_t)lr * pitch + s * 64 + hi * 16;
#pragma unroll
        for (int qt = 0; qt < NQT; qt++) {
            if (qt < ntq) { // (uniform)
                const mb_bf8 bh = *reinterpret_cast<const mb_bf8*>(bq + (size_t)qt * 32 * pitch);
                const mb_bf8 bl = *reinterpret_cast<const mb_bf8*>(bq + (size_t)qt * 32 * pitch + 32);
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    acc[t][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bh, acc[t][qt], 0, 0, 0);
                    acc[t][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bl, acc[t][qt], 0, 0, 0);
                    acc[t][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t], bh, acc[t][qt], 0, 0, 0);
                }
            }
        }
    };
    auto epilogue = [&](int64_t b) {
        // one compare per (row, query); the slow path only where something passes
#pragma unroll
        for (int qt = 0; qt < NQT; qt++) {
            if (qt < ntq) {
                const float thr = sT[qt * 32 + lr];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    float m = acc[t][qt][0];
#pragma unroll
                    for (int r = 1; r < 16; r++) {
                        m = fmaxf(m, acc[t][qt][r]);
                    }
                    if (__ballot(m >= thr) != 0ull) {
                        // (rare: a few dozen rows per query and batch.  One append site per tile: the hits of a lane as
                        // a bit mask, their values picked out of the accumulator by a select chain)
                        uint32_t hits = 0;
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            hits |= acc[t][qt][r] >= thr ? (1u << r) : 0u;
                        }
                        if (hits != 0u) {
                            const int32_t q = sPq[qt * 32 + lr], slot = sPs[qt * 32 + lr];
                            const float c = sC[qt * 32 + lr];
                            while (hits != 0u) {
                                const int r = __ffs((int)hits) - 1;
                                hits &= hits - 1u;
                                float v = acc[t][qt][0];
#pragma unroll
                                for (int r2 = 1; r2 < 16; r2++) {
                                    v = r == r2 ? acc[t][qt][r2] : v;
                                }
                                const int64_t pos = b * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                                if (pos < len) {
                                    // parked: the append's global atomics (a round trip each, with the wave waiting) run at
                                    // the unit's end, all in flight together
                                    const int at = atomicAdd(sNhit, 1);
                                    if (at < MB_HITS) {
                                        sHit[at] = make_uint4((uint32_t)(qt * 32 + lr), (uint32_t)pos, __float_as_uint(v), 0u);
                                    } else {
                                        ms_emit<IS_L2>(a, q, slot, row_off, pos, IS_L2 ? c - 2.0f * v : v - c);
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
    };
    // (loads one step ahead.  Two steps ahead -- three buffers, 240 registers -- was measured and changes nothing: 2.845
    // against 2.815 ms at C2; the bytes in flight are not what limits the kernel)
    issue(A[0]);
    int64_t b = wave; // compute cursor
    int s = 0;
    const int64_t nbw = nblk > wave ? (nblk - wave + MB_WAVES - 1) / MB_WAVES : 0;
    const int64_t G = nbw * nstep;
    init_acc(b);
    for (int64_t g = 0; g < G; g += 2) {
#pragma unroll
        for (int u2 = 0; u2 < 2; u2++) {
            issue(A[u2 ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
            compute(A[u2], s);
            if (++s == nstep) {
                if (b < nblk) {
                    epilogue(b);
                }
                s = 0;
                b += MB_WAVES;
                init_acc(b);
            }
        }
    }
    __syncthreads();
    const int nhit = min(*sNhit, MB_HITS);
    for (int i = threadIdx.x; i < nhit; i += MB_THREADS) {
        const uint4 h = sHit[i];
        const float v = __uint_as_float(h.z), c = sC[h.x];
        ms_emit<IS_L2>(a, sPq[h.x], sPs[h.x], row_off, (int64_t)h.y, IS_L2 ? c - 2.0f * v : v - c);
    }
}

// One unit per workgroup in XCD-aware order; LOOP: a fixed grid walks a unit table whose size only the device knows (the
// retry round of overflowed queries).  Every exit inside a unit is workgroup-uniform.
template <bool IS_L2, int NQT, bool LOOP>
__global__ __launch_bounds__(MB_THREADS, 2) void mscan_flatb_kernel(MScanArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int64_t nunits = *a.nunits_dev;
    if (LOOP) {
        for (int64_t u = blockIdx.x; u < nunits; u += gridDim.x) {
            mscan_flatb_unit<IS_L2, NQT>(a, u, smem);
            __syncthreads();
        }
    } else {
        if ((int64_t)blockIdx.x >= ((nunits + 7) / 8) * 8) {
            return;
        }
        const int64_t u = xcd_item(blockIdx.x, nunits);
        if (u >= nunits) {
            return;
        }
        mscan_flatb_unit<IS_L2, NQT>(a, u, smem);
    }
}

// queries per unit: 128 while two workgroups' split queries fit a CU's LDS side by side with room to spare, else 64;
// 0 = the shape is not served (the fp32 kernel's own limit, d <= 608)
int mscan_flat_bf16_qt(int nstep) {
    auto bytes = [&](int qt) { return (size_t)qt * mb_pitch(nstep) + (size_t)qt * 16 + (size_t)MB_HITS * 16 + 16; };
    if (bytes(128) <= 80 * 1024) {
        return 128;
    }
    return bytes(64) <= 160 * 1024 - 512 ? 64 : 0;
}

size_t mscan_flat_bf16_smem(int nstep) {
    const int qt = mscan_flat_bf16_qt(nstep);
    return (size_t)qt * mb_pitch(nstep) + (size_t)qt * 16 + (size_t)MB_HITS * 16 + 16;
}

template <int NQT>
static hipError_t launch_flatb(const MScanArgs& a, bool is_l2, int64_t units_bound, size_t sm, hipStream_t s) {
    auto kern = is_l2 ? mscan_flatb_kernel<true, NQT, false> : mscan_flatb_kernel<false, NQT, false>;
    if (a.unit_loop) { // the retry round's one-query units
        kern = is_l2 ? mscan_flatb_kernel<true, NQT, true> : mscan_flatb_kernel<false, NQT, true>;
    }
    if (sm > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) {
            return e;
        }
    }
    const int64_t grid = a.unit_loop ? std::min<int64_t>(units_bound, 2048) : ((units_bound + 7) / 8) * 8;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(MB_THREADS), sm, s, a);
    return hipGetLastError();
}

// filter pass only (a.dump == nullptr); the units must have been cut for mscan_flat_bf16_qt(a.nstep) queries
hipError_t launch_mscan_flat_bf16(const MScanArgs& a, bool is_l2, int64_t units_bound, hipStream_t s) {
    if (units_bound <= 0) {
        return hipSuccess;
    }
    const int qt = mscan_flat_bf16_qt(a.nstep);
    if (a.dump != nullptr || qt == 0) {
        return hipErrorInvalidValue;
    }
    const size_t sm = mscan_flat_bf16_smem(a.nstep);
    return qt == 128 ? launch_flatb<4>(a, is_l2, units_bound, sm, s) : launch_flatb<2>(a, is_l2, units_bound, sm, s);
}

} // namespace knhip
