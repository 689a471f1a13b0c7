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

// ---- round 5: the same prefilter on the bf16 matrix pipe, selection fused, no distance matrix ------------------------
// The certificate + exact re-rank make the GEMM a prefilter: it needs a bounded error, not fp32 products.  Every operand is
// split into two bf16 terms (bf16 carries 8 significant bits, v_cvt_pk_bf16_f32 rounds to nearest: |x - hi| <= 2^-8 |x|,
// x = hi + lo + r with |r| <= 2^-16 |x|) and the product taken as hi hi + hi lo + lo hi on v_mfma_f32_32x32x16_bf16 (the
// bf16 products are exact in fp32, fp32 accumulation): three instructions of a pipe sixteen times as fast as the fp32
// one.  What is dropped is lo lo + r_q c + q r_c: |dot - exact| <= (3 * 2^-16 + 3 d 2^-24) ||q|| ||c||; the certificate's
// eps takes 2^-14 (||q||^2 + max ||c||^2) (L2: twice the dot's error, 2 ab <= a^2 + b^2) / 2^-14 ||q|| max ||c|| (IP).  And the nq x nlist matrix (655 MB at C3, written
// once and read twice by the select) is never written: the GEMM runs TWICE with different epilogues --
//   pass 1: the minimum over every group of 32 centroids (one MFMA tile's rows: in the C layout a lane = a query, its 16
//           registers + the partner lane's = the group) -> gmin[group][query]; the ncand-th smallest group minimum B_q
//           bounds the row's ncand-th value (ncand groups hold a value <= B_q);
//   pass 2: every centroid with approx <= B_q is appended to the query's candidate list (~1.2 ncand of them).
// Everything unselected has approx > B_q: B_q is the certificate's T.  M = centroids, N = queries (A = centroids).
// order-preserving float <-> uint32 keys ("better" = smaller key): used by the bound selection above the re-rank kernel's own
template <bool IS_L2>
__device__ __forceinline__ uint32_t cr_key_fwd(float f) {
    const uint32_t b = __float_as_uint(f);
    const uint32_t asc = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return IS_L2 ? asc : ~asc;
}
template <bool IS_L2>
__device__ __forceinline__ float cr_unkey_fwd(uint32_t key) {
    const uint32_t asc = IS_L2 ? key : ~key;
    const uint32_t b = (asc & 0x80000000u) ? (asc & 0x7fffffffu) : ~asc;
    return __uint_as_float(b);
}

typedef __bf16 cg_bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 cg_bf4 __attribute__((ext_vector_type(4)));
constexpr int CB_BM = 256, CB_BN = 128, CB_BK = 32; // rows (centroids) x queries per workgroup tile, dimensions per slab
constexpr int CB_WI = 4;                            // a wave holds CB_WI x 2 MFMA tiles: rows (wave >> 1) * 128 .. + 128, queries
                                                    // (wave & 1) * 64 .. + 64.  Round 6: 2 x 2 tiles per wave read 8 operand
                                                    // fragments per 12 matrix instructions -- with the staging stores the LDS
                                                    // was as busy as the matrix pipe (both at half their rate); 4 x 2 tiles read
                                                    // 12 per 24
constexpr int CB_ROW = 144; // bytes per staged row: 32 hi + 32 lo bf16 + 16 of padding (16 rows -> 16 different bank quads)

// fp32 rows [n][d] -> the split operand rows the GEMM stages without touching them: per row and k slab of 32 dimensions
// 32 hi then 32 lo bf16 (128 bytes; dimensions past d are zero).  The centroids' copy is built once with the index
// (8 MB at C3, 201 MB at C5), the queries' once per batch.
__global__ void cb_split_rows_kernel(const float* __restrict__ x, int64_t n, int d, int nslab, unsigned char* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // one thread per (row, slab, group of 4 dimensions)
    const int64_t total = n * nslab * 8;
    if (t >= total) {
        return;
    }
    const int g4 = (int)(t & 7);
    const int64_t rs = t >> 3;
    const int slab = (int)(rs % nslab);
    const int64_t row = rs / nslab;
    const int k = slab * CB_BK + g4 * 4;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        v[e] = k + e < d ? x[row * d + k + e] : 0.f;
    }
    cg_bf4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const __bf16 h = (__bf16)v[e];
        hi[e] = h;
        lo[e] = (__bf16)(v[e] - (float)h); // (inf / NaN inputs give NaN here: nothing is selected, the query falls back)
    }
    unsigned char* o = out + (row * nslab + slab) * 128 + g4 * 8;
    *reinterpret_cast<cg_bf4*>(o) = hi;
    *reinterpret_cast<cg_bf4*>(o + 64) = lo;
}

// MODE 1: gmin[query][group] group minima (maxima for IP).  MODE 2: candidates appended where approx <= bound.
template <bool IS_L2, int MODE>
__global__ __launch_bounds__(256, 2) void coarse_bf16_kernel(const unsigned char* __restrict__ Qs, const float* __restrict__ qn,
                                                          const unsigned char* __restrict__ Cs, const float* __restrict__ cn,
                                                          int64_t nq, int64_t nlist, int nslab, int64_t tiles_q,
                                                          int64_t ntiles, float* __restrict__ gmin, int G,
                                                          const float* __restrict__ bound, int32_t* __restrict__ cand_cnt,
                                                          int64_t* __restrict__ cand, int cap, int g16) {
    __align__(16) __shared__ unsigned char sA[CB_BM * CB_ROW];
    __align__(16) __shared__ unsigned char sB[CB_BN * CB_ROW];
    __shared__ float s_cn[CB_BM];
    __shared__ float s_gm[CB_BM / 16][CB_BN]; // group minima of the tile: 8 groups of 32 centroids, or 16 of 16 (g16)
    // XCD-aware tile order: consecutive tile ids share the centroid panel
    const int64_t per = (ntiles + 7) / 8;
    const int64_t tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (tile >= ntiles) {
        return;
    }
    const int64_t tc = tile / tiles_q, tq = tile % tiles_q;
    const int64_t c0 = tc * CB_BM, q0 = tq * CB_BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * (32 * CB_WI), wn = (wave & 1) * 64;
    static_assert(CB_BM == 256 && 2 * 32 * CB_WI == CB_BM, "one s_cn entry per thread, two wave rows");
    if (IS_L2) {
        s_cn[tid] = c0 + tid < nlist ? cn[c0 + tid] : 0.f;
    }
    f32x16 acc[CB_WI][2];
#pragma unroll
    for (int i = 0; i < CB_WI; i++) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                acc[i][j][r] = 0.f;
            }
        }
    }
    // staging: 2048 + 1024 pieces of 16 bytes per slab, eight + four per thread; rows past the end are zero
    constexpr int NA = CB_BM * 8 / 256, NB = CB_BN * 8 / 256;
    uint4 va[NA], vb[NB];
    auto fetch = [&](int slab) {
#pragma unroll
        for (int j = 0; j < NA; j++) {
            const int p = tid + 256 * j, row = p >> 3, piece = p & 7;
            va[j] = make_uint4(0u, 0u, 0u, 0u);
            if (c0 + row < nlist) {
                va[j] = *reinterpret_cast<const uint4*>(Cs + ((c0 + row) * nslab + slab) * 128 + piece * 16);
            }
        }
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const int p = tid + 256 * j, row = p >> 3, piece = p & 7;
            vb[j] = make_uint4(0u, 0u, 0u, 0u);
            if (q0 + row < nq) {
                vb[j] = *reinterpret_cast<const uint4*>(Qs + ((q0 + row) * nslab + slab) * 128 + piece * 16);
            }
        }
    };
    fetch(0);
    for (int slab = 0; slab < nslab; slab++) {
#pragma unroll
        for (int j = 0; j < NA; j++) {
            const int p = tid + 256 * j, row = p >> 3, piece = p & 7;
            *reinterpret_cast<uint4*>(sA + row * CB_ROW + piece * 16) = va[j];
        }
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const int p = tid + 256 * j, row = p >> 3, piece = p & 7;
            *reinterpret_cast<uint4*>(sB + row * CB_ROW + piece * 16) = vb[j];
        }
        __syncthreads();
        if (slab + 1 < nslab) {
            fetch(slab + 1);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ks++) { // two k steps of 16: lane (row = lane & 31, k half = lane >> 5) holds 8 consecutive k
            const int ko = (ks * 16 + (lane >> 5) * 8) * 2;
            cg_bf8 ah[CB_WI], al[CB_WI], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < CB_WI; i++) {
                const unsigned char* pa = sA + (wm + i * 32 + (lane & 31)) * CB_ROW + ko;
                ah[i] = *reinterpret_cast<const cg_bf8*>(pa);
                al[i] = *reinterpret_cast<const cg_bf8*>(pa + 64);
            }
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const unsigned char* pb = sB + (wn + i * 32 + (lane & 31)) * CB_ROW + ko;
                bh[i] = *reinterpret_cast<const cg_bf8*>(pb);
                bl[i] = *reinterpret_cast<const cg_bf8*>(pb + 64);
            }
#pragma unroll
            for (int i = 0; i < CB_WI; i++) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // epilogue: C/D layout col (query) = lane & 31, row (centroid) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int64_t query = q0 + wn + j * 32 + (lane & 31);
        const bool qok = query < nq;
        const float qnv = (IS_L2 && qok) ? qn[query] : 0.f;
        float bnd = 0.f;
        if (MODE == 2) {
            bnd = qok ? bound[query] : (IS_L2 ? -INFINITY : INFINITY);
        }
        uint32_t hit[CB_WI] = {}; // MODE 2: which of the lane's CB_WI x 16 values pass (bit r of block i)
#pragma unroll
        for (int i = 0; i < CB_WI; i++) {
            float best = IS_L2 ? INFINITY : -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rowl = wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int64_t cen = c0 + rowl;
                float v = acc[i][j][r];
                if (IS_L2) {
                    v = qnv + s_cn[rowl] - 2.0f * v;
                }
                if (cen >= nlist) {
                    v = IS_L2 ? INFINITY : -INFINITY;
                }
                if (MODE == 1) {
                    best = IS_L2 ? fminf(best, v) : fmaxf(best, v);
                } else if (qok && (IS_L2 ? v <= bnd : v >= bnd)) {
                    hit[i] |= 1u << r;
                }
            }
            if (MODE == 1) {
                if (g16) {
                    // groups of 16: the 16 rows this lane holds of the 32-row block (any fixed partition of the centroids
                    // serves the bound) -- twice the groups, for nlist where 32-row groups are fewer than 2 ncand
                    s_gm[2 * ((wm >> 5) + i) + (lane >> 5)][wn + j * 32 + (lane & 31)] = best;
                } else {
                    const float other = __shfl_xor(best, 32, KN_WAVE);
                    best = IS_L2 ? fminf(best, other) : fmaxf(best, other);
                    if (lane < 32) {
                        s_gm[(wm >> 5) + i][wn + j * 32 + lane] = best;
                    }
                }
            }
        }
        if (MODE == 2) {
            // ONE returning atomic per lane for all its hits of this query block (a returning atomic per hit made the wave
            // wait a memory round trip ~40 times per tile: 465 us of the stage at C3), then the ids go to their slots
            int nh = 0;
#pragma unroll
            for (int i = 0; i < CB_WI; i++) {
                nh += __popc(hit[i]);
            }
            if (nh > 0) {
                int n = atomicAdd(cand_cnt + query, nh);
#pragma unroll
                for (int i = 0; i < CB_WI; i++) {
                    uint32_t m = hit[i];
                    while (m != 0u) {
                        const int r = __ffs((int)m) - 1;
                        m &= m - 1u;
                        if (n < cap) {
                            cand[query * cap + n] = c0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        }
                        n++;
                    }
                }
            }
        }
    }
    if (MODE == 1) {
        // the tile's 128 queries x 8 (16) groups: 16-byte stores per query (gmin [nq][G]: a row per query for the bound kernel)
        __syncthreads();
        if (tid < CB_BN && q0 + tid < nq) {
            const int ng = g16 ? CB_BM / 16 : CB_BM / 32;
            float* o = gmin + (q0 + tid) * G + c0 / (g16 ? 16 : 32);
            for (int g = 0; g < ng; g += 4) {
                *reinterpret_cast<float4*>(o + g) = make_float4(s_gm[g][tid], s_gm[g + 1][tid], s_gm[g + 2][tid], s_gm[g + 3][tid]);
            }
        }
    }
}

// B_q = the ncand-th best of the query's G group minima (gmin [nq][G]): one wave per query, the row in registers,
// bisection over the order-preserving integer keys with ballot counts.  G <= 4096.
template <bool IS_L2>
__global__ __launch_bounds__(256) void coarse_bound_kernel(const float* __restrict__ gmin, int G, int64_t nq, int ncand,
                                                           float* __restrict__ bound) {
    constexpr int RMAX = 4096 / KN_WAVE;
    const int lane = lane_id();
    const int64_t q = (int64_t)blockIdx.x * (blockDim.x / KN_WAVE) + threadIdx.x / KN_WAVE;
    if (q >= nq) {
        return;
    }
    uint32_t key[RMAX];
    const int nreg = (G + KN_WAVE - 1) / KN_WAVE;
#pragma unroll
    for (int r = 0; r < RMAX; r++) {
        const int g = r * KN_WAVE + lane;
        key[r] = (r < nreg && g < G) ? cr_key_fwd<IS_L2>(gmin[q * G + g]) : 0xffffffffu;
    }
    // smallest key x with count(key <= x) >= ncand; keys ascend with "better"
    uint32_t lo = 0u, hi = 0xffffffffu;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        int cnt = 0;
#pragma unroll
        for (int r = 0; r < RMAX; r++) {
            if (r < nreg) {
                cnt += __popcll(__ballot(key[r] <= mid));
            }
        }
        if (cnt >= ncand) {
            hi = mid;
        } else {
            lo = mid + 1;
        }
    }
    if (lane == 0) {
        bound[q] = cr_unkey_fwd<IS_L2>(lo);
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
        float* __restrict__ out_d, int32_t* __restrict__ fail_flags, unsigned long long* __restrict__ nfail,
        const int32_t* __restrict__ cand_cnt, const float* __restrict__ bound, float eps_rel, int need_kth) {
    // cand_cnt / bound non-null (the bf16 prefilter): the row holds cand_cnt[q] unordered candidates (capacity ncand: more
    // is an overflow -> exact fallback) = EVERY centroid with approx <= bound[q], so bound[q] is the certificate's T
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
    const int nc = cand_cnt != nullptr ? min(cand_cnt[q], ncand) : ncand;
    for (int c = tid; c < nc; c += 256) {
        const int64_t key = cand_keys[q * ncand + c];
        if (key < 0) {
            continue;
        }
        const float* y = centroids + key * d;
        float acc = 0.f;
        int i = 0;
        if ((d & 3) == 0 && (reinterpret_cast<uintptr_t>(centroids) & 15) == 0) {
            // 16-byte loads, eight in flight per lane (every lane reads its own row: 4-byte loads were one cache-line request
            // per element -- 2.6 ms of the stage at C5); the accumulation order stays i = 0, 1, 2, ... (reference order)
            const float4* y4 = reinterpret_cast<const float4*>(y);
            const float4* q4 = reinterpret_cast<const float4*>(sq);
            const int n4 = d >> 2;
            int j = 0;
            for (; j + 8 <= n4; j += 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    v[u] = y4[j + u];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const float4 x = q4[j + u];
                    acc = IS_L2 ? l2_step(acc, x.x, v[u].x) : ip_step(acc, x.x, v[u].x);
                    acc = IS_L2 ? l2_step(acc, x.y, v[u].y) : ip_step(acc, x.y, v[u].y);
                    acc = IS_L2 ? l2_step(acc, x.z, v[u].z) : ip_step(acc, x.z, v[u].z);
                    acc = IS_L2 ? l2_step(acc, x.w, v[u].w) : ip_step(acc, x.w, v[u].w);
                }
            }
            i = j * 4;
        }
        for (; i < d; i++) {
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
            // T = worst selected approx (row_select output is sorted best-first) / the bound everything selected is under
            const float T = bound != nullptr ? bound[q] : cand_approx[q * ncand + ncand - 1];
            const unsigned long long c = cand[nprobe - 1];
            const float en = cr_unkey<IS_L2>((uint32_t)(c >> 32));
            // |approx - exact| <= eps = eps_rel * magnitude (fp32 GEMM: gamma_d = d * 2^-24 with a 8x safety factor; the
            // bf16 split adds 2^-14 >= 3 * 2^-16, see coarse_bf16_kernel)
            const float scale = IS_L2 ? (qnorm[q] + cnorm_max) : sqrtf(qnorm[q] * cnorm_max);
            const float eps = eps_rel * scale + 1e-30f;
            if (!need_kth) {
                // the bound came from OUTSIDE (a k-th best found elsewhere, widened by eps: coarse_ext_bound_kernel): every row
                // that can matter is among the candidates unless they overflowed; fewer than nprobe of them is no failure
                fail = (cand_cnt != nullptr && cand_cnt[q] > ncand) || !(eps < INFINITY) || !(T == T);
            } else if (c == ~0ull || (cand_cnt != nullptr && cand_cnt[q] > ncand) || !(eps < INFINITY)) {
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
                                unsigned long long* nfail, hipStream_t s, const int32_t* cand_cnt, const float* bound,
                                bool need_kth) {
    if (nq <= 0) {
        return hipSuccess;
    }
    const float eps_rel = 8.0f * (float)d * 5.9604645e-8f + (bound != nullptr ? 6.103515625e-5f : 0.f);
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
                       cand_keys, cand_approx, nprobe, qnorm, cnorm_max, out_keys, out_d, fail_flags, nfail, cand_cnt, bound,
                       eps_rel, need_kth ? 1 : 0);
    return hipGetLastError();
}

// the bf16 prefilter: group minima -> bound per query -> candidates under the bound (see coarse_bf16_kernel)
// groups of 32 centroids where there are at least 2 ncand of them (enough for a tight bound), else groups of 16 (round 5:
// nlist 4096 with nprobe 64 -- C2 -- has 128 groups of 32 for 96 candidates); 0: the shape is not served
static int coarse_bf16_group_rows(int64_t nlist, int ncand) {
    if (nlist < 2048) {
        return 0;
    }
    const int64_t g32 = (nlist + 31) / 32, g16 = (nlist + 15) / 16;
    if (g32 >= 2 * (int64_t)ncand && g32 <= 4096) {
        return 32;
    }
    return (g16 >= 2 * (int64_t)ncand && g16 <= 4096) ? 16 : 0;
}

bool coarse_bf16_supports(int64_t nlist, int ncand) {
    return coarse_bf16_group_rows(nlist, ncand) != 0;
}

int coarse_bf16_slabs(int d) {
    return (d + CB_BK - 1) / CB_BK;
}

// split operand rows of n fp32 rows (coarse_bf16_slabs(d) * 128 bytes per row)
hipError_t launch_coarse_bf16_split(const float* x, int64_t n, int d, void* out, hipStream_t s) {
    if (n <= 0) {
        return hipSuccess;
    }
    const int nslab = coarse_bf16_slabs(d);
    const int64_t total = n * nslab * 8;
    hipLaunchKernelGGL(cb_split_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, n, d, nslab,
                       static_cast<unsigned char*>(out));
    return hipGetLastError();
}

hipError_t launch_coarse_bf16(const void* q_split, const float* qnorm, const void* c_split, const float* cnorm, int64_t nq,
                              int64_t nlist, int d, bool is_l2, int ncand, int cap, float* gmin, float* bound,
                              int32_t* cand_cnt, int64_t* cand, hipStream_t s, bool bound_given) {
    // bound_given: `bound` already holds every query's selection bound (coarse_ext_bound): no group-minima pass, no bound
    // kernel -- ONE pass over the rows
    if (nq <= 0 || nlist <= 0) {
        return hipSuccess;
    }
    const int64_t tc = (nlist + CB_BM - 1) / CB_BM, tq = (nq + CB_BN - 1) / CB_BN;
    const int64_t ntiles = tc * tq;
    const unsigned grid = (unsigned)(((ntiles + 7) / 8) * 8);
    const int grows = coarse_bf16_group_rows(nlist, ncand);
    if (grows == 0) {
        return hipErrorInvalidValue;
    }
    const int g16 = grows == 16 ? 1 : 0;
    const int G = (int)(tc * (CB_BM / grows)); // (groups of the padded tiles: the padding's minima are the neutral value)
    const int nslab = coarse_bf16_slabs(d);
    const unsigned char* Qs = static_cast<const unsigned char*>(q_split);
    const unsigned char* Cs = static_cast<const unsigned char*>(c_split);
    hipError_t e = hipMemsetAsync(cand_cnt, 0, (size_t)nq * sizeof(int32_t), s);
    if (e != hipSuccess) {
        return e;
    }
    if (is_l2) {
        if (!bound_given) {
            hipLaunchKernelGGL((coarse_bf16_kernel<true, 1>), dim3(grid), dim3(256), 0, s, Qs, qnorm, Cs, cnorm, nq, nlist, nslab, tq,
                               ntiles, gmin, G, nullptr, nullptr, nullptr, 0, g16);
            hipLaunchKernelGGL((coarse_bound_kernel<true>), dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, gmin, G, nq, ncand, bound);
        }
        hipLaunchKernelGGL((coarse_bf16_kernel<true, 2>), dim3(grid), dim3(256), 0, s, Qs, qnorm, Cs, cnorm, nq, nlist, nslab, tq,
                           ntiles, nullptr, G, bound, cand_cnt, cand, cap, g16);
    } else {
        if (!bound_given) {
            hipLaunchKernelGGL((coarse_bf16_kernel<false, 1>), dim3(grid), dim3(256), 0, s, Qs, qnorm, Cs, cnorm, nq, nlist, nslab, tq,
                               ntiles, gmin, G, nullptr, nullptr, nullptr, 0, g16);
            hipLaunchKernelGGL((coarse_bound_kernel<false>), dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, gmin, G, nq, ncand, bound);
        }
        hipLaunchKernelGGL((coarse_bf16_kernel<false, 2>), dim3(grid), dim3(256), 0, s, Qs, qnorm, Cs, cnorm, nq, nlist, nslab, tq,
                           ntiles, nullptr, G, bound, cand_cnt, cand, cap, g16);
    }
    return hipGetLastError();
}

// ---- a selection bound from OUTSIDE: the best k-th distance found so far (BRUTE_FORCE: over the chunks already searched) ----
// kth [nq]: the running k-th best EXACT distance per query (worst value: none yet).  chunk_d [nq][k] (may be null): a chunk's
// exact result, best first -- its k-th entry, where filled, updates kth.  bound_out[q] = kth widened by the prefilter's eps
// (the same eps_rel * magnitude the certificate uses, times 1.001): every row whose exact distance is at least as good as
// kth has approx within the bound.
template <bool IS_L2>
__global__ void coarse_ext_bound_kernel(float* __restrict__ kth, const float* __restrict__ chunk_d, int k, int64_t nq,
                                        const float* __restrict__ qnorm, float cnorm_max, float eps_rel,
                                        float* __restrict__ bound_out) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) {
        return;
    }
    float b = kth[q];
    if (chunk_d != nullptr) {
        const float v = chunk_d[q * k + k - 1];
        if (v == v && fabsf(v) < FLT_MAX) {
            b = IS_L2 ? fminf(b, v) : fmaxf(b, v);
        }
        kth[q] = b;
    }
    const float scale = IS_L2 ? (qnorm[q] + cnorm_max) : sqrtf(qnorm[q] * cnorm_max);
    const float eps = 1.001f * (eps_rel * scale + 1e-30f);
    bound_out[q] = IS_L2 ? b + eps : b - eps; // (no k-th yet: +-FLT_MAX stays out of reach of every approx -> everything passes
                                              // -> the candidate lists overflow -> the exact fallback: never taken, chunk 0
                                              // runs the two-pass form)
}

hipError_t launch_coarse_ext_bound(float* kth, const float* chunk_d, int k, int64_t nq, const float* qnorm, float cnorm_max, int d,
                                   bool is_l2, float* bound_out, hipStream_t s) {
    if (nq <= 0) {
        return hipSuccess;
    }
    const float eps_rel = 8.0f * (float)d * 5.9604645e-8f + 6.103515625e-5f; // (launch_coarse_rerank's, bf16 form)
    if (is_l2) {
        hipLaunchKernelGGL(coarse_ext_bound_kernel<true>, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, kth, chunk_d, k, nq,
                           qnorm, cnorm_max, eps_rel, bound_out);
    } else {
        hipLaunchKernelGGL(coarse_ext_bound_kernel<false>, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, kth, chunk_d, k, nq,
                           qnorm, cnorm_max, eps_rel, bound_out);
    }
    return hipGetLastError();
}

// upper bound of the group count (the gmin scratch: [nq][groups]; groups of 16 at most)
int64_t coarse_bf16_groups(int64_t nlist) {
    return ((nlist + CB_BM - 1) / CB_BM) * (CB_BM / 16);
}

} // namespace knhip
