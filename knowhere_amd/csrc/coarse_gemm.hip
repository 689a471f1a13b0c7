// knowhere_amd/csrc/coarse_gemm.hip -- coarse quantizer as an fp32 MFMA GEMM (gfx950).
//
// The query x centroid distance matrix is a true dense contraction (SURVEY.md 8a row a3:
// 2*nq*nlist*d flops, e.g. 4.2e10 per 10k-query batch at nlist=16384, 1.0e12 at nlist=65536,
// d=768), so it runs on the matrix cores: v_mfma_f32_32x32x2_f32, exact-f32 products with an
// f32 k-ordered accumulate, 157 TF peak (no xf32/TF32 exists on CDNA4).
//
// The reference computes the same distances with the DIRECT form at n=1
// (thirdparty/faiss/faiss/utils/distances.cpp:326-362: fvec_L2sqr per centroid) and feeds
// coarse_dis into the final IVF-PQ / IVF-SQ8 distance (IVFPQ_QueryTables.cpp:138), so the
// expanded-form GEMM  ||q||^2 + ||c||^2 - 2 q.c  (what the reference's BLAS path uses for big
// batches, distances.cpp:425-512) cannot be the final word if results are to stay bit-equal.
// It is used as a PREFILTER:
//   1. coarse_gemm        approx[q][c]                       (MFMA, this file)
//   2. row_select         the nprobe+margin best approx per query          (topk.hip)
//   3. coarse_rerank      exact reference-order distances of those candidates, canonical sort,
//                         top-nprobe, and a CERTIFICATE: every unselected centroid has
//                         approx >= T (the worst selected approx), hence exact >= T - eps;
//                         if T - eps is still worse than the exact nprobe-th distance the
//                         selection provably equals the exact one.  eps bounds |approx - exact|
//                         from the standard fp32 dot-product error (gamma_d = d * 2^-24).
//   4. queries whose certificate fails are recomputed with the exact all-pairs kernel
//      (flat_full + row_select restricted to flagged rows): rare, and exactness never depends
//      on the margin being "big enough".
#include "common.h"
#include "kernels.h"

namespace knhip {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- squared norms (any summation order will do: they only feed the prefilter) -----------------
__global__ __launch_bounds__(256) void row_norms_kernel(const float* __restrict__ x, int64_t n, int d,
                                                        float* __restrict__ out) {
    const int lane = lane_id();
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x / KN_WAVE) + threadIdx.x / KN_WAVE;
    if (row >= n) {
        return;
    }
    float acc = 0.f;
    for (int i = lane; i < d; i += KN_WAVE) {
        const float v = x[row * d + i];
        acc += v * v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        acc += __shfl_xor(acc, off, KN_WAVE);
    }
    if (lane == 0) {
        out[row] = acc;
    }
}

// ---- 128 x 128 output tile per 256-thread workgroup; wave w owns the 64 x 64 quadrant ----------
// (w >> 1, w & 1) as 2 x 2 MFMA tiles of 32 x 32.  K is consumed in slabs of 32 staged in LDS
// k-major (sA[k][row]) so the A/B operand fetch -- lane l needs A[row0 + (l & 31)][k + (l >> 5)] --
// is a conflict-free ds_read_b32.
// LD = 129 (= 1 mod 32): the staging stores of a 32-lane group -- 8 k-quads x 4 rows, element e of each -- hit the
// banks 4 * quad + row + e: conflict-free ds_write_b32 (LD = 132 made them 4-way conflicts: half of the kernel's LDS
// cycles in the round-2 PMC pass); the operand reads walk 32 consecutive rows of one k and stay conflict-free.
constexpr int CG_BM = 128, CG_BN = 128, CG_BK = 32, CG_LD = 129;

template <bool IS_L2>
__global__ __launch_bounds__(256) void coarse_gemm_kernel(const float* __restrict__ Q,
                                                          const float* __restrict__ qn,
                                                          const float* __restrict__ Cm,
                                                          const float* __restrict__ cn, int64_t nq,
                                                          int64_t nlist, int d, int64_t tiles_n,
                                                          int64_t ntiles, float* __restrict__ out) {
    __shared__ float sA[CG_BK * CG_LD];
    __shared__ float sB[CG_BK * CG_LD];
    // XCD-aware tile order: consecutive tile ids share the query panel; give each XCD a run
    // (grid is rounded up to a multiple of 8; ids beyond the last tile exit)
    const int64_t per = (ntiles + 7) / 8;
    const int64_t tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (tile >= ntiles) {
        return;
    }
    const int64_t tm = tile / tiles_n, tn = tile % tiles_n;
    const int64_t q0 = tm * CG_BM, c0 = tn * CG_BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                acc[i][j][r] = 0.f;
            }
        }
    }
    const int lrow = tid >> 3; // 0..31 (+32 per pass)
    const int lkq = tid & 7;   // which float4 of the 32-wide k slab
    const bool vec_ok = (d & 3) == 0;
    // one k slab of both operands in registers: the loads of slab s + 1 are in flight while slab s is multiplied
    float4 va[4], vb[4];
    auto fetch = [&](int k0) {
        const int kk = k0 + lkq * 4;
#pragma unroll
        for (int pass = 0; pass < 4; pass++) {
            const int row = lrow + pass * 32;
            va[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
            vb[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q0 + row < nq) {
                const float* p = Q + (q0 + row) * d + kk;
                if (kk + 3 < d && vec_ok) {
                    va[pass] = *reinterpret_cast<const float4*>(p);
                } else {
                    if (kk + 0 < d) va[pass].x = p[0];
                    if (kk + 1 < d) va[pass].y = p[1];
                    if (kk + 2 < d) va[pass].z = p[2];
                    if (kk + 3 < d) va[pass].w = p[3];
                }
            }
            if (c0 + row < nlist) {
                const float* p = Cm + (c0 + row) * d + kk;
                if (kk + 3 < d && vec_ok) {
                    vb[pass] = *reinterpret_cast<const float4*>(p);
                } else {
                    if (kk + 0 < d) vb[pass].x = p[0];
                    if (kk + 1 < d) vb[pass].y = p[1];
                    if (kk + 2 < d) vb[pass].z = p[2];
                    if (kk + 3 < d) vb[pass].w = p[3];
                }
            }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < d; k0 += CG_BK) {
#pragma unroll
        for (int pass = 0; pass < 4; pass++) {
            const int row = lrow + pass * 32;
            sA[(lkq * 4 + 0) * CG_LD + row] = va[pass].x;
            sA[(lkq * 4 + 1) * CG_LD + row] = va[pass].y;
            sA[(lkq * 4 + 2) * CG_LD + row] = va[pass].z;
            sA[(lkq * 4 + 3) * CG_LD + row] = va[pass].w;
            sB[(lkq * 4 + 0) * CG_LD + row] = vb[pass].x;
            sB[(lkq * 4 + 1) * CG_LD + row] = vb[pass].y;
            sB[(lkq * 4 + 2) * CG_LD + row] = vb[pass].z;
            sB[(lkq * 4 + 3) * CG_LD + row] = vb[pass].w;
        }
        __syncthreads();
        if (k0 + CG_BK < d) {
            fetch(k0 + CG_BK);
        }
#pragma unroll
        for (int k = 0; k < CG_BK; k += 2) {
            const int kr = k + (lane >> 5);
            const float a0 = sA[kr * CG_LD + wm + (lane & 31)];
            const float a1 = sA[kr * CG_LD + wm + 32 + (lane & 31)];
            const float b0 = sB[kr * CG_LD + wn + (lane & 31)];
            const float b1 = sB[kr * CG_LD + wn + 32 + (lane & 31)];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    // epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int64_t col = c0 + wn + j * 32 + (lane & 31);
            const float cnv = (IS_L2 && col < nlist) ? cn[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int64_t row = q0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < nq && col < nlist) {
                    float v = acc[i][j][r];
                    if (IS_L2) {
                        v = qn[row] + cnv - 2.0f * v;
                    }
                    out[row * nlist + col] = v;
                }
            }
        }
    }
}

// ---- exact re-rank of the candidates + certificate ----------------------------------------------
// one 256-thread workgroup per query; ncand <= 4096
template <bool IS_L2>
__device__ __forceinline__ uint32_t cr_key(float f) {
    uint32_t b = __float_as_uint(f);
    uint32_t asc = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return IS_L2 ? asc : ~asc;
}
template <bool IS_L2>
__device__ __forceinline__ float cr_unkey(uint32_t key) {
    uint32_t asc = IS_L2 ? key : ~key;
    uint32_t b = (asc & 0x80000000u) ? (asc & 0x7fffffffu) : ~asc;
    return __uint_as_float(b);
}

template <bool IS_L2>
__global__ __launch_bounds__(256) void coarse_rerank_kernel(
        const float* __restrict__ queries, const float* __restrict__ centroids, int d, int64_t nlist,
        int ncand, int kp, const int64_t* __restrict__ cand_keys, const float* __restrict__ cand_approx,
        int nprobe, const float* __restrict__ qnorm, float cnorm_max, int64_t* __restrict__ out_keys,
        float* __restrict__ out_d, int32_t* __restrict__ fail_flags, unsigned long long* __restrict__ nfail) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(smem); // [kp]
    float* sq = reinterpret_cast<float*>(smem + (size_t)kp * 8);           // [d]
    const int tid = threadIdx.x;
    const int64_t q = blockIdx.x;
    for (int i = tid; i < d; i += 256) {
        sq[i] = queries[q * d + i];
    }
    for (int i = tid; i < kp; i += 256) {
        cand[i] = ~0ull;
    }
    __syncthreads();
    for (int c = tid; c < ncand; c += 256) {
        const int64_t key = cand_keys[q * ncand + c];
        if (key < 0) {
            continue;
        }
        const float* y = centroids + key * d;
        float acc = 0.f;
        for (int i = 0; i < d; i++) {
            acc = IS_L2 ? l2_step(acc, sq[i], y[i]) : ip_step(acc, sq[i], y[i]);
        }
        const uint32_t tie = IS_L2 ? (uint32_t)key : ~(uint32_t)key;
        cand[c] = ((unsigned long long)cr_key<IS_L2>(acc) << 32) | tie;
    }
    __syncthreads();
    for (int size = 2; size <= kp; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < kp / 2; t += 256) {
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
    for (int e = tid; e < nprobe; e += 256) {
        const unsigned long long c = cand[e];
        if (c == ~0ull) {
            out_keys[q * nprobe + e] = -1;
            out_d[q * nprobe + e] = worst_dist<IS_L2>();
        } else {
            const uint32_t tie = (uint32_t)c;
            out_keys[q * nprobe + e] = IS_L2 ? (int64_t)tie : (int64_t)(~tie);
            out_d[q * nprobe + e] = cr_unkey<IS_L2>((uint32_t)(c >> 32));
        }
    }
    if (tid == 0) {
        int fail = 0;
        if ((int64_t)ncand < nlist) {
            // T = worst selected approx (row_select output is sorted best-first)
            const float T = cand_approx[q * ncand + ncand - 1];
            const unsigned long long c = cand[nprobe - 1];
            const float en = cr_unkey<IS_L2>((uint32_t)(c >> 32));
            // |approx - exact| <= eps, gamma_d = d * 2^-24 with a 8x safety factor
            const float scale = IS_L2 ? (qnorm[q] + cnorm_max) : sqrtf(qnorm[q] * cnorm_max);
            const float eps = 8.0f * (float)d * 5.9604645e-8f * scale + 1e-30f;
            if (c == ~0ull) {
                fail = 1;
            } else if (IS_L2) {
                fail = !(T - eps > en);
            } else {
                fail = !(T + eps < en);
            }
        }
        fail_flags[q] = fail;
        if (fail) {
            fail_flags[gridDim.x] = 1; // (summary entry [nq]: the fallback kernel returns at once while it is 0)
            if (nfail != nullptr) {
                atomicAdd(nfail, 1ull);
            }
        }
    }
}

hipError_t launch_row_norms(const float* x, int64_t n, int d, float* out, hipStream_t s) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(row_norms_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, x, n, d, out);
    return hipGetLastError();
}

hipError_t launch_coarse_gemm(const float* q, const float* qnorm, const float* c, const float* cnorm,
                              int64_t nq, int64_t nlist, int d, bool is_l2, float* out, hipStream_t s) {
    if (nq <= 0 || nlist <= 0) {
        return hipSuccess;
    }
    const int64_t tm = (nq + CG_BM - 1) / CG_BM, tn = (nlist + CG_BN - 1) / CG_BN;
    const int64_t ntiles = tm * tn;
    const unsigned grid = (unsigned)(((ntiles + 7) / 8) * 8);
    if (is_l2) {
        hipLaunchKernelGGL((coarse_gemm_kernel<true>), dim3(grid), dim3(256), 0, s, q, qnorm, c, cnorm, nq,
                           nlist, d, tn, ntiles, out);
    } else {
        hipLaunchKernelGGL((coarse_gemm_kernel<false>), dim3(grid), dim3(256), 0, s, q, qnorm, c, cnorm, nq,
                           nlist, d, tn, ntiles, out);
    }
    return hipGetLastError();
}

hipError_t launch_coarse_rerank(const float* queries, const float* centroids, int d, int64_t nq,
                                int64_t nlist, int ncand, const int64_t* cand_keys,
                                const float* cand_approx, int nprobe, bool is_l2, const float* qnorm,
                                float cnorm_max, int64_t* out_keys, float* out_d, int32_t* fail_flags,
                                unsigned long long* nfail, hipStream_t s) {
    if (nq <= 0) {
        return hipSuccess;
    }
    int kp = 2;
    while (kp < ncand) {
        kp <<= 1;
    }
    const size_t sm = (size_t)kp * 8 + (size_t)d * 4;
    auto kern = is_l2 ? coarse_rerank_kernel<true> : coarse_rerank_kernel<false>;
    if (sm > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) {
            return e;
        }
    }
    hipError_t e0 = hipMemsetAsync(fail_flags + nq, 0, sizeof(int32_t), s); // fail_flags: [nq + 1], the last = any flag
    if (e0 != hipSuccess) {
        return e0;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nq), dim3(256), sm, s, queries, centroids, d, nlist, ncand, kp,
                       cand_keys, cand_approx, nprobe, qnorm, cnorm_max, out_keys, out_d, fail_flags, nfail);
    return hipGetLastError();
}

} // namespace knhip
