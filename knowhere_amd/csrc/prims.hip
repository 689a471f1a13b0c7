// knowhere_amd/csrc/prims.hip -- HIP equivalents of the src/simd distance primitives.
//
// The reference selects these through a function-pointer table at run time
// (src/simd/hook.h:33-139, hook.cc:163-382); the scalar definitions in
// src/simd/distances_ref.cc are the known-answer its tests compare every SIMD level against
// (tests/ut/test_simd.cc:259-568).  The kernels below reproduce the scalar definitions
// operation for operation (sequential over the dimension, one rounding per operation), so
// they are bit-equal to *_ref:
//   fvec_L2sqr_ny / fvec_inner_products_ny   distances_ref.cc:67-81   (one x against ny rows)
//   fvec_norm_L2sqr (rows)                   faiss float accumulator form
//                                            (thirdparty/faiss/faiss/utils/simd_impl/distances_autovec-inl.h:28-39)
//   fvec_madd                                thirdparty/faiss/faiss/utils/distances_simd.cpp:33-43
//   int8_vec_L2sqr / int8_vec_inner_product  distances_ref.cc:386-404 (int32 accumulate, cast)
//
// Row-major y is the ABI's layout (as in the reference); each wave stages a 64-row x 64-column
// tile through LDS with coalesced global loads, then lane r walks row r sequentially (row pitch
// 65 words: conflict-free column access).
#include "common.h"
#include "kernels.h"

namespace knhip {

constexpr int PR_WAVES = 2;

template <int MODE> // 0: L2(x,y)  1: IP(x,y)  2: norm(y)
__global__ __launch_bounds__(PR_WAVES * KN_WAVE) void fvec_rows_kernel(float* __restrict__ out,
                                                                      const float* __restrict__ x,
                                                                      const float* __restrict__ y,
                                                                      int64_t d, int64_t ny) {
    __shared__ float tile[PR_WAVES][64 * 65];
    __shared__ float sx[PR_WAVES][64];
    const int lane = lane_id();
    const int wave = threadIdx.x / KN_WAVE;
    const int64_t row0 = ((int64_t)blockIdx.x * PR_WAVES + wave) * 64;
    float* t = tile[wave];
    float acc = 0.f;
    for (int64_t c0 = 0; c0 < d; c0 += 64) {
        const int64_t c = c0 + lane;
        if (MODE != 2) {
            sx[wave][lane] = (c < d) ? x[c] : 0.f;
        }
        for (int r = 0; r < 64; r++) {
            const int64_t row = row0 + r;
            t[r * 65 + lane] = (row < ny && c < d) ? y[row * d + c] : 0.f;
        }
        __syncthreads(); // tile + x chunk visible
        const int cn = (int)min((int64_t)64, d - c0);
        for (int i = 0; i < cn; i++) {
            const float yv = t[lane * 65 + i];
            if (MODE == 0) {
                acc = l2_step(acc, sx[wave][i], yv);
            } else if (MODE == 1) {
                acc = ip_step(acc, sx[wave][i], yv);
            } else {
                acc = ip_step(acc, yv, yv);
            }
        }
        __syncthreads();
    }
    if (row0 + lane < ny) {
        out[row0 + lane] = acc;
    }
}

template <bool IS_L2>
__global__ __launch_bounds__(256) void int8_rows_kernel(float* __restrict__ out,
                                                        const int8_t* __restrict__ x,
                                                        const int8_t* __restrict__ y, int64_t d,
                                                        int64_t ny) {
    // integer arithmetic is associative: any order gives the reference's int32 result
    const int lane = lane_id();
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x / KN_WAVE) + threadIdx.x / KN_WAVE;
    if (row >= ny) {
        return;
    }
    int32_t acc = 0;
    for (int64_t i = lane; i < d; i += KN_WAVE) {
        const int32_t a = (int32_t)x[i], b = (int32_t)y[row * d + i];
        if (IS_L2) {
            const int32_t t = a - b;
            acc += t * t;
        } else {
            acc += a * b;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        acc += __shfl_xor(acc, off, KN_WAVE);
    }
    if (lane == 0) {
        out[row] = (float)acc;
    }
}

__global__ void fvec_madd_kernel(int64_t n, const float* __restrict__ a, float bf,
                                 const float* __restrict__ b, float* __restrict__ c) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        c[i] = fadd_x(a[i], fmul_x(bf, b[i]));
    }
}

hipError_t launch_fvec_ny(float* out, const float* x, const float* y, int64_t d, int64_t ny,
                          bool is_l2, hipStream_t s) {
    if (ny <= 0) {
        return hipSuccess;
    }
    const unsigned grid = (unsigned)((ny + PR_WAVES * 64 - 1) / (PR_WAVES * 64));
    if (is_l2) {
        hipLaunchKernelGGL((fvec_rows_kernel<0>), dim3(grid), dim3(PR_WAVES * KN_WAVE), 0, s, out, x, y,
                           d, ny);
    } else {
        hipLaunchKernelGGL((fvec_rows_kernel<1>), dim3(grid), dim3(PR_WAVES * KN_WAVE), 0, s, out, x, y,
                           d, ny);
    }
    return hipGetLastError();
}

hipError_t launch_fvec_norms(float* out, const float* x, int64_t d, int64_t n, hipStream_t s) {
    if (n <= 0) {
        return hipSuccess;
    }
    const unsigned grid = (unsigned)((n + PR_WAVES * 64 - 1) / (PR_WAVES * 64));
    hipLaunchKernelGGL((fvec_rows_kernel<2>), dim3(grid), dim3(PR_WAVES * KN_WAVE), 0, s, out, nullptr,
                       x, d, n);
    return hipGetLastError();
}

hipError_t launch_fvec_madd(int64_t n, const float* a, float bf, const float* b, float* c,
                            hipStream_t s) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(fvec_madd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, a, bf,
                       b, c);
    return hipGetLastError();
}

hipError_t launch_int8_ny(float* out, const int8_t* x, const int8_t* y, int64_t d, int64_t ny,
                          bool is_l2, hipStream_t s) {
    if (ny <= 0) {
        return hipSuccess;
    }
    const unsigned grid = (unsigned)((ny + 3) / 4);
    if (is_l2) {
        hipLaunchKernelGGL((int8_rows_kernel<true>), dim3(grid), dim3(256), 0, s, out, x, y, d, ny);
    } else {
        hipLaunchKernelGGL((int8_rows_kernel<false>), dim3(grid), dim3(256), 0, s, out, x, y, d, ny);
    }
    return hipGetLastError();
}

} // namespace knhip
