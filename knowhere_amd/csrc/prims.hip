// knowhere_amd/csrc/prims.hip -- HIP equivalents of the src/simd distance primitives (the whole hook table).
//
// The reference selects these through a function-pointer table at run time (src/simd/hook.h:33-123,
// hook.cc:163-382); the scalar definitions in src/simd/distances_ref.cc are the known answer its tests compare
// every SIMD level against (tests/ut/test_simd.cc:259-568).  The kernels below reproduce the scalar definitions
// operation for operation (sequential over the dimension, one rounding per operation), so they are bit-equal
// to *_ref:
//   fvec_L2sqr / inner_product / L1 / Linf (+ _ny)   distances_ref.cc:21-55, 67-81  (one x against ny rows)
//   fvec_norm_L2sqr                                  faiss float accumulator form
//                                                    (thirdparty/faiss/faiss/utils/simd_impl/distances_autovec-inl.h:28-39,
//                                                    what the IVF-PQ tables use) and the _ref form (double accumulator,
//                                                    distances_ref.cc:57-64)
//   fvec_L2sqr_ny_transposed                         distances_ref.cc:84-101  (y column-major, expanded form)
//   fvec_L2sqr_ny_nearest / _nearest_y_transposed    distances_ref.cc:106-145 (first strict minimum below +inf)
//   fvec_madd / fvec_madd_and_argmin                 distances_ref.cc:147-168
//   fvec_{inner_product,L2sqr}_batch_4               distances_ref.cc:170-210 (four independent sequential sums)
//   ivec_*, int8_vec_*                               distances_ref.cc:217-233, 386-456 (int32 accumulate, cast)
//   fp16_vec_* / bf16_vec_*                          distances_ref.cc:236-384 (convert to float, float arithmetic)
//
// Row-major y is the ABI's layout (as in the reference); each wave stages a 64-row x 64-column tile through LDS with
// coalesced global loads (converted to float on the way in: fp16 / bf16 / int8 -> float is exact), then lane r walks
// row r sequentially (row pitch 65 words: conflict-free column access).  Every kernel is a streaming read of y:
// algorithmic bytes = ny * d * sizeof(T), HBM-bound.
#include "common.h"
#include "kernels.h"

#include <hip/hip_fp16.h>

namespace knhip {

constexpr int PR_WAVES = 4;

struct bf16_bits {
    uint16_t v;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__half v) { return __half2float(v); }
__device__ __forceinline__ float to_f32(bf16_bits v) { return __uint_as_float((uint32_t)v.v << 16); }
__device__ __forceinline__ float to_f32(int8_t v) { return (float)v; }

// LDS word of one element: the float value (fp32 / fp16 / bf16 -> float is exact) or, for int8, the int32 value
template <typename T>
__device__ __forceinline__ uint32_t to_word(T v) {
    if (sizeof(T) == 1) {
        return (uint32_t)(int32_t)to_f32(v);
    }
    return __float_as_uint(to_f32(v));
}

// four consecutive elements with one load (the caller guarantees alignment and bounds)
__device__ __forceinline__ void load4(const float* p, uint32_t w[4]) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    w[0] = __float_as_uint(v.x);
    w[1] = __float_as_uint(v.y);
    w[2] = __float_as_uint(v.z);
    w[3] = __float_as_uint(v.w);
}
__device__ __forceinline__ void load4(const __half* p, uint32_t w[4]) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    const __half2 a = *reinterpret_cast<const __half2*>(&v.x), b = *reinterpret_cast<const __half2*>(&v.y);
    w[0] = __float_as_uint(__low2float(a));
    w[1] = __float_as_uint(__high2float(a));
    w[2] = __float_as_uint(__low2float(b));
    w[3] = __float_as_uint(__high2float(b));
}
__device__ __forceinline__ void load4(const bf16_bits* p, uint32_t w[4]) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    w[0] = v.x << 16;
    w[1] = v.x & 0xffff0000u;
    w[2] = v.y << 16;
    w[3] = v.y & 0xffff0000u;
}
__device__ __forceinline__ void load4(const int8_t* p, uint32_t w[4]) {
    const uint32_t v = *reinterpret_cast<const uint32_t*>(p);
#pragma unroll
    for (int e = 0; e < 4; e++) {
        w[e] = (uint32_t)(int32_t)(int8_t)(v >> (8 * e));
    }
}

enum { PR_L2 = 0, PR_IP = 1, PR_NORM = 2, PR_L1 = 3, PR_LINF = 4, PR_NORM_REF = 5 };

// One wave = one tile of 64 rows x 64 columns at a time: 16 independent 4-element loads per lane (1 KB per wave
// instruction, 16 KB in flight per wave) transposed into LDS at row pitch 65 words (bank = row + col: the 4-element
// stores of a wave instruction and the later per-row walks are both conflict-free), then lane r walks row r in the
// reference's order.  A wave touches only its own tile and LDS operations of a wave execute in order: no barrier.
// VEC = 0 is the element-wise fallback for d % 4 != 0 or unaligned bases.
template <typename T, int OP, typename OutT, bool VEC>
__global__ __launch_bounds__(PR_WAVES* KN_WAVE) void rows_kernel(OutT* __restrict__ out, const T* __restrict__ x,
                                                                 const T* __restrict__ y, int64_t d, int64_t ny) {
    __shared__ uint32_t tile[PR_WAVES][64 * 65];
    __shared__ uint32_t sx[PR_WAVES][64];
    constexpr bool UNARY = OP == PR_NORM || OP == PR_NORM_REF;
    constexpr bool INT = sizeof(T) == 1;
    const int lane = lane_id();
    const int wave = threadIdx.x / KN_WAVE;
    const int64_t row0 = ((int64_t)blockIdx.x * PR_WAVES + wave) * 64;
    if (row0 >= ny) {
        return;
    }
    uint32_t* t = tile[wave];
    float acc = 0.f;
    double dacc = 0.0;
    int32_t iacc = 0;
    for (int64_t c0 = 0; c0 < d; c0 += 64) {
        const int64_t c = c0 + lane;
        if (!UNARY) {
            sx[wave][lane] = (c < d) ? to_word(x[c]) : 0u;
        }
        if (VEC) {
            uint32_t w[16][4];
#pragma unroll
            for (int it = 0; it < 16; it++) {
                const int e = it * 64 + lane;
                const int64_t row = row0 + (e >> 4);
                const int64_t col = c0 + 4 * (e & 15);
                if (row < ny && col < d) {
                    load4(y + row * d + col, w[it]);
                } else {
                    w[it][0] = w[it][1] = w[it][2] = w[it][3] = 0u;
                }
            }
#pragma unroll
            for (int it = 0; it < 16; it++) {
                const int e = it * 64 + lane;
                uint32_t* o = t + (e >> 4) * 65 + 4 * (e & 15);
                o[0] = w[it][0];
                o[1] = w[it][1];
                o[2] = w[it][2];
                o[3] = w[it][3];
            }
        } else {
            for (int r = 0; r < 64; r++) {
                const int64_t row = row0 + r;
                t[r * 65 + lane] = (row < ny && c < d) ? to_word(y[row * d + c]) : 0u;
            }
        }
        const int cn = (int)min((int64_t)64, d - c0);
        for (int i = 0; i < cn; i++) {
            const uint32_t yw = t[lane * 65 + i];
            const uint32_t xw = UNARY ? 0u : sx[wave][i];
            if (INT) {
                const int32_t a = (int32_t)xw, b = (int32_t)yw;
                if (OP == PR_L2) {
                    iacc += (a - b) * (a - b);
                } else if (OP == PR_IP) {
                    iacc += a * b;
                } else {
                    iacc += b * b;
                }
                continue;
            }
            const float yv = __uint_as_float(yw), xv = __uint_as_float(xw);
            if (OP == PR_L2) {
                acc = l2_step(acc, xv, yv);
            } else if (OP == PR_IP) {
                acc = ip_step(acc, xv, yv);
            } else if (OP == PR_NORM) {
                acc = ip_step(acc, yv, yv);
            } else if (OP == PR_L1) {
                acc = fadd_x(acc, fabsf(fsub_x(xv, yv)));
            } else if (OP == PR_LINF) {
                acc = fmaxf(acc, fabsf(fsub_x(xv, yv)));
            } else { // float product summed in a double, rounded once at the end
                dacc = dacc + (double)fmul_x(yv, yv);
            }
        }
    }
    if (row0 + lane < ny) {
        out[row0 + lane] = INT ? (OutT)iacc : (OutT)(OP == PR_NORM_REF ? (float)dacc : acc);
    }
}

// ---- argmin plumbing: one 64-bit cell, key = (order-preserving float bits << 32) | index ------------------------
__device__ __forceinline__ unsigned long long argmin_key(float v, int64_t i) {
    v = v + 0.0f; // -0 -> +0: the reference's `<` does not tell them apart, the first index wins
    uint32_t b = __float_as_uint(v);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((unsigned long long)b << 32) | (unsigned long long)(uint32_t)i;
}

__device__ __forceinline__ void argmin_publish(unsigned long long key, unsigned long long* cell) {
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(key, off, KN_WAVE);
        key = o < key ? o : key;
    }
    if (lane_id() == 0 && key != ~0ull) {
        atomicMin(cell, key);
    }
}

// entries >= limit (and NaN) never win: `dis[i] < min_dis` starting from limit
__global__ __launch_bounds__(256) void argmin_kernel(const float* __restrict__ v, int64_t n, float limit,
                                                     unsigned long long* __restrict__ cell) {
    unsigned long long key = ~0ull;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = v[i];
        if (x < limit) {
            const unsigned long long k2 = argmin_key(x, i);
            key = k2 < key ? k2 : key;
        }
    }
    argmin_publish(key, cell);
}

__global__ void argmin_decode_kernel(unsigned long long* cell, int64_t none_value) {
    const unsigned long long key = *cell;
    *reinterpret_cast<int64_t*>(cell) = key == ~0ull ? none_value : (int64_t)(key & 0xffffffffull);
}

__global__ __launch_bounds__(256) void fvec_madd_kernel(int64_t n, const float* __restrict__ a, float bf,
                                                        const float* __restrict__ b, float* __restrict__ c,
                                                        unsigned long long* __restrict__ cell) {
    unsigned long long key = ~0ull;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = fadd_x(a[i], fmul_x(bf, b[i]));
        c[i] = v;
        if (cell != nullptr && v < 1e20f) {
            const unsigned long long k2 = argmin_key(v, i);
            key = k2 < key ? k2 : key;
        }
    }
    if (cell != nullptr) {
        argmin_publish(key, cell);
    }
}

// y is [d][d_offset]: vector i in column i (lanes read consecutive columns: coalesced); x is read uniformly
__global__ __launch_bounds__(256) void l2_transposed_kernel(float* __restrict__ dis, const float* __restrict__ x,
                                                            const float* __restrict__ y,
                                                            const float* __restrict__ y_sqlen, int64_t d,
                                                            int64_t d_offset, int64_t ny) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ny) {
        return;
    }
    float x_sqlen = 0.f, dp = 0.f;
    for (int64_t j = 0; j < d; j++) {
        const float xv = x[j];
        x_sqlen = ip_step(x_sqlen, xv, xv);
        dp = ip_step(dp, xv, y[i + j * d_offset]);
    }
    dis[i] = fsub_x(fadd_x(x_sqlen, y_sqlen[i]), fmul_x(2.f, dp));
}

// four rows sharing x: thread r walks row r (a leaf call of the reference's scanners; not a throughput kernel)
template <typename T, bool IS_L2>
__global__ void batch4_kernel(const T* __restrict__ x, const T* y0, const T* y1, const T* y2, const T* y3, int64_t d,
                              float* __restrict__ out) {
    const int r = threadIdx.x;
    if (r >= 4) {
        return;
    }
    const T* y = r == 0 ? y0 : (r == 1 ? y1 : (r == 2 ? y2 : y3));
    if (sizeof(T) == 1) { // int8: int32 accumulate, cast at the end
        int32_t acc = 0;
        for (int64_t i = 0; i < d; i++) {
            const int32_t a = (int32_t)to_f32(x[i]), b = (int32_t)to_f32(y[i]);
            acc += IS_L2 ? (a - b) * (a - b) : a * b;
        }
        out[r] = (float)acc;
        return;
    }
    float acc = 0.f;
    for (int64_t i = 0; i < d; i++) {
        acc = IS_L2 ? l2_step(acc, to_f32(x[i]), to_f32(y[i])) : ip_step(acc, to_f32(x[i]), to_f32(y[i]));
    }
    out[r] = acc;
}

// ---- launchers ----------------------------------------------------------------------------------------------------
template <typename T, int OP, typename OutT>
static void launch_rows_op(OutT* out, const T* x, const T* y, int64_t d, int64_t ny, hipStream_t s) {
    const dim3 grid((unsigned)((ny + PR_WAVES * 64 - 1) / (PR_WAVES * 64))), block(PR_WAVES * KN_WAVE);
    const size_t align = 4 * sizeof(T);
    const bool vec = (d % 4) == 0 && (reinterpret_cast<uintptr_t>(y) % align) == 0;
    if (vec) {
        hipLaunchKernelGGL((rows_kernel<T, OP, OutT, true>), grid, block, 0, s, out, x, y, d, ny);
    } else {
        hipLaunchKernelGGL((rows_kernel<T, OP, OutT, false>), grid, block, 0, s, out, x, y, d, ny);
    }
}

template <typename T>
static hipError_t launch_rows_t(int op, float* out, const T* x, const T* y, int64_t d, int64_t ny, hipStream_t s) {
    if (ny <= 0) {
        return hipSuccess;
    }
    switch (op) {
        case PR_L2: launch_rows_op<T, PR_L2, float>(out, x, y, d, ny, s); break;
        case PR_IP: launch_rows_op<T, PR_IP, float>(out, x, y, d, ny, s); break;
        case PR_NORM: launch_rows_op<T, PR_NORM, float>(out, x, y, d, ny, s); break;
        case PR_L1: launch_rows_op<T, PR_L1, float>(out, x, y, d, ny, s); break;
        case PR_LINF: launch_rows_op<T, PR_LINF, float>(out, x, y, d, ny, s); break;
        case PR_NORM_REF: launch_rows_op<T, PR_NORM_REF, float>(out, x, y, d, ny, s); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_fvec_rows(int op, float* out, const float* x, const float* y, int64_t d, int64_t ny, hipStream_t s) {
    return launch_rows_t<float>(op, out, x, y, d, ny, s);
}

hipError_t launch_fvec_ny(float* out, const float* x, const float* y, int64_t d, int64_t ny, bool is_l2,
                          hipStream_t s) {
    return launch_rows_t<float>(is_l2 ? PR_L2 : PR_IP, out, x, y, d, ny, s);
}

hipError_t launch_fvec_norms(float* out, const float* x, int64_t d, int64_t n, hipStream_t s) {
    return launch_rows_t<float>(PR_NORM, out, nullptr, x, d, n, s);
}

// dtype 0 fp16, 1 bf16, 2 int8; op 0 L2sqr, 1 inner product, 2 norm_L2sqr (the _ref forms: double accumulator for the
// 16-bit norms, int32 for int8)
hipError_t launch_typed_rows(int dtype, int op, float* out, const void* x, const void* y, int64_t d, int64_t ny,
                             hipStream_t s) {
    if (ny <= 0) {
        return hipSuccess;
    }
    if (op < 0 || op > 2) {
        return hipErrorInvalidValue;
    }
    if (dtype == 0) {
        return launch_rows_t<__half>(op == 2 ? PR_NORM_REF : op, out, static_cast<const __half*>(x),
                                     static_cast<const __half*>(y), d, ny, s);
    }
    if (dtype == 1) {
        return launch_rows_t<bf16_bits>(op == 2 ? PR_NORM_REF : op, out, static_cast<const bf16_bits*>(x),
                                        static_cast<const bf16_bits*>(y), d, ny, s);
    }
    if (dtype != 2) {
        return hipErrorInvalidValue;
    }
    return launch_rows_t<int8_t>(op, out, static_cast<const int8_t*>(x), static_cast<const int8_t*>(y), d, ny, s);
}

hipError_t launch_int8_ny(float* out, const int8_t* x, const int8_t* y, int64_t d, int64_t ny, bool is_l2,
                          hipStream_t s) {
    return launch_typed_rows(2, is_l2 ? 0 : 1, out, x, y, d, ny, s);
}

hipError_t launch_ivec_ny(int32_t* out, const int8_t* x, const int8_t* y, int64_t d, int64_t ny, bool is_l2,
                          hipStream_t s) {
    if (ny <= 0) {
        return hipSuccess;
    }
    if (is_l2) {
        launch_rows_op<int8_t, PR_L2, int32_t>(out, x, y, d, ny, s);
    } else {
        launch_rows_op<int8_t, PR_IP, int32_t>(out, x, y, d, ny, s);
    }
    return hipGetLastError();
}

static unsigned stream_grid(int64_t n) {
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 256 * 8));
}

// d_idx receives the first index whose value is minimal and below `limit`, else none_value
hipError_t launch_argmin(const float* v, int64_t n, float limit, int64_t none_value, int64_t* d_idx, hipStream_t s) {
    hipError_t e = hipMemsetAsync(d_idx, 0xff, sizeof(int64_t), s);
    if (e != hipSuccess) {
        return e;
    }
    auto* cell = reinterpret_cast<unsigned long long*>(d_idx);
    if (n > 0) {
        hipLaunchKernelGGL(argmin_kernel, dim3(stream_grid(n)), dim3(256), 0, s, v, n, limit, cell);
    }
    hipLaunchKernelGGL(argmin_decode_kernel, dim3(1), dim3(1), 0, s, cell, none_value);
    return hipGetLastError();
}

hipError_t launch_fvec_madd(int64_t n, const float* a, float bf, const float* b, float* c, hipStream_t s) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(fvec_madd_kernel, dim3(stream_grid(n)), dim3(256), 0, s, n, a, bf, b, c,
                       static_cast<unsigned long long*>(nullptr));
    return hipGetLastError();
}

hipError_t launch_fvec_madd_and_argmin(int64_t n, const float* a, float bf, const float* b, float* c, int64_t* d_imin,
                                       hipStream_t s) {
    hipError_t e = hipMemsetAsync(d_imin, 0xff, sizeof(int64_t), s);
    if (e != hipSuccess) {
        return e;
    }
    auto* cell = reinterpret_cast<unsigned long long*>(d_imin);
    if (n > 0) {
        hipLaunchKernelGGL(fvec_madd_kernel, dim3(stream_grid(n)), dim3(256), 0, s, n, a, bf, b, c, cell);
    }
    hipLaunchKernelGGL(argmin_decode_kernel, dim3(1), dim3(1), 0, s, cell, (int64_t)-1);
    return hipGetLastError();
}

hipError_t launch_l2_transposed(float* dis, const float* x, const float* y, const float* y_sqlen, int64_t d,
                                int64_t d_offset, int64_t ny, hipStream_t s) {
    if (ny <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(l2_transposed_kernel, dim3((unsigned)((ny + 255) / 256)), dim3(256), 0, s, dis, x, y, y_sqlen, d,
                       d_offset, ny);
    return hipGetLastError();
}

// dtype -1 fp32, 0 fp16, 1 bf16, 2 int8
hipError_t launch_batch4(int dtype, bool is_l2, const void* x, const void* y0, const void* y1, const void* y2,
                         const void* y3, int64_t d, float* out, hipStream_t s) {
#define B4(T_)                                                                                                      \
    do {                                                                                                            \
        if (is_l2) {                                                                                                \
            hipLaunchKernelGGL((batch4_kernel<T_, true>), dim3(1), dim3(64), 0, s, static_cast<const T_*>(x),       \
                               static_cast<const T_*>(y0), static_cast<const T_*>(y1), static_cast<const T_*>(y2),  \
                               static_cast<const T_*>(y3), d, out);                                                 \
        } else {                                                                                                    \
            hipLaunchKernelGGL((batch4_kernel<T_, false>), dim3(1), dim3(64), 0, s, static_cast<const T_*>(x),      \
                               static_cast<const T_*>(y0), static_cast<const T_*>(y1), static_cast<const T_*>(y2),  \
                               static_cast<const T_*>(y3), d, out);                                                 \
        }                                                                                                           \
    } while (0)
    switch (dtype) {
        case -1: B4(float); break;
        case 0: B4(__half); break;
        case 1: B4(bf16_bits); break;
        case 2: B4(int8_t); break;
        default: return hipErrorInvalidValue;
    }
#undef B4
    return hipGetLastError();
}

} // namespace knhip
