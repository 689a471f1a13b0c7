// tests/hipemu/hip/hip_runtime.h -- a HOST stand-in for <hip/hip_runtime.h>, test infrastructure only.
//
// Purpose: run the *source* of selected kernels (knowhere_amd/csrc/pq_filter.hip, the finish kernel of mfma_scan.hip)
// on the CPU when no GPU is at hand, to check their index arithmetic, work protocol and epilogues against the numpy
// model of tests/test_pqf_model.py.  It is NOT a product path and nothing under knowhere_amd/ includes it: the test
// compiles a patched copy of the kernel files against this directory (tests/hipemu/emu_build.py).
//
// Execution model: a workgroup = blockDim.x OS threads running the kernel body; workgroups run one after another
// (function-local `__shared__` arrays are plain statics, valid for one workgroup at a time).
//   __syncthreads()                      -> barrier over the workgroup's live threads (a thread that returns drops out)
//   __shfl / __shfl_up / __shfl_xor / __ballot / readlane / readfirstlane
//                                        -> exchange through a per-wave buffer + barrier over the wave's 64 threads.
//     They are only valid where all 64 lanes of the wave take part (wave-uniform control flow) -- which is how the
//     kernels under test use them.
//   atomics                              -> GCC/Clang __atomic builtins on host memory
// What it cannot model: the hardware's memory model and LDS hazards, EXEC-masked cross-lane operations, instruction
// scheduling.  Hardware-specific lines (LDS offset addressing, inline ISA) are rewritten by emu_build.py.
#pragma once
#include <atomic>
#include <barrier>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

// ---- qualifiers ---------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define __align__(n) alignas(n)
#define HIP_SYMBOL(x) x
#define __HIP_MEMORY_SCOPE_AGENT 0

// ---- vector types ---------------------------------------------------------------------------------------------------
struct alignas(8) uint2 { uint32_t x, y; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct alignas(8) int2 { int32_t x, y; };
struct alignas(8) float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
inline int2 make_int2(int32_t x, int32_t y) { return int2{x, y}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ---- runtime API subset -----------------------------------------------------------------------------------------------
typedef int hipError_t;
typedef void* hipStream_t;
constexpr hipError_t hipSuccess = 0;
constexpr hipError_t hipErrorInvalidValue = 1;
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
namespace hipemu { extern int g_ncu; }
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = hipemu::g_ncu; return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }  // (launches are synchronous here)
// memory / streams / events of the orchestration (knhip_api.hip): "device" memory is host memory, everything is synchronous
constexpr hipError_t hipErrorOutOfMemory = 2;
constexpr hipError_t hipErrorNotReady = 600;
constexpr unsigned hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000;
typedef void* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
constexpr unsigned hipStreamNonBlocking = 1;
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "emulated error"; }
inline hipError_t hipMalloc(void** p, size_t n) {
    // (slack behind every allocation: kernels prefetch past the end by design; filled with a pattern, not zero)
    *p = std::aligned_alloc(256, ((n + 255) / 256) * 256 + 4096);
    if (*p == nullptr) {
        return hipErrorOutOfMemory;
    }
    std::memset(*p, 0xcd, ((n + 255) / 256) * 256 + 4096);
    return hipSuccess;
}
template <class T>
inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), n); }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* p, int v, size_t n) { std::memset(p, v, n); return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)64 << 30; *t = (size_t)64 << 30; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
constexpr unsigned hipEventDisableTiming = 2;
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; } // (launches run in order)
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; } // (everything enqueued has already run)
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = std::malloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
[[noreturn]] inline void hipemu_unreachable() { // stands in for inline ISA (kernels that hold it are not run here)
    std::fprintf(stderr, "hipemu: reached inline ISA that the emulation does not model\n");
    std::abort();
}

// ---- execution model ------------------------------------------------------------------------------------------------
namespace hipemu {

struct Wave {
    std::barrier<> bar;
    uint64_t x[2][64];
    alignas(16) unsigned char wide[2][64][32]; // 32-byte payloads (matrix-core operands)
    explicit Wave(int n) : bar(n) {}
};
struct Group {
    std::barrier<> bar;
    std::vector<std::unique_ptr<Wave>> waves;
    unsigned char* smem = nullptr;
    explicit Group(int n) : bar(n) {}
};
struct Ctx {
    dim3 tid, bid, bdim, gdim;
    Group* g = nullptr;
    Wave* w = nullptr;
    int lane = 0;
    unsigned op = 0;
};
extern thread_local Ctx tl;

inline uint64_t xchg(uint64_t v, int src) { // every lane of the wave calls this; returns lane `src`'s value
    Wave& w = *tl.w;
    const unsigned k = tl.op++ & 1u;
    w.x[k][tl.lane] = v;
    w.bar.arrive_and_wait();
    return w.x[k][src & 63];
}
inline unsigned long long ballot(bool p) {
    Wave& w = *tl.w;
    const unsigned k = tl.op++ & 1u;
    w.x[k][tl.lane] = p ? 1u : 0u;
    w.bar.arrive_and_wait();
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++) {
        m |= (unsigned long long)(w.x[k][i] & 1u) << i;
    }
    return m;
}
// every lane publishes n <= 32 bytes; returns the wave's 64 payloads (valid until this lane's next-but-one exchange)
inline const unsigned char (*xchg_wide(const void* p, size_t n))[32] {
    Wave& w = *tl.w;
    const unsigned k = tl.op++ & 1u;
    std::memcpy(w.wide[k][tl.lane], p, n);
    w.bar.arrive_and_wait();
    return w.wide[k];
}
template <class T>
inline uint64_t to_bits(T v) {
    uint64_t b = 0;
    static_assert(sizeof(T) <= 8, "shuffle payload");
    std::memcpy(&b, &v, sizeof(T));
    return b;
}
template <class T>
inline T from_bits(uint64_t b) {
    T v;
    std::memcpy(&v, &b, sizeof(T));
    return v;
}

// run one kernel launch: workgroups in sequence, threads of a workgroup concurrently
void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body);

} // namespace hipemu

#define threadIdx (hipemu::tl.tid)
#define blockIdx (hipemu::tl.bid)
#define blockDim (hipemu::tl.bdim)
#define gridDim (hipemu::tl.gdim)
#define hipLaunchKernelGGL(kern, grid, block, smem, stream, ...) \
    hipemu::launch((grid), (block), (size_t)(smem), [&]() { (kern)(__VA_ARGS__); })

inline void __syncthreads() { hipemu::tl.g->bar.arrive_and_wait(); }
inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }
inline void __threadfence_system() { std::atomic_thread_fence(std::memory_order_seq_cst); }

// ---- cross-lane ---------------------------------------------------------------------------------------------------
template <class T>
inline T __shfl(T v, int src, int = 64) { return hipemu::from_bits<T>(hipemu::xchg(hipemu::to_bits(v), src)); }
template <class T>
inline T __shfl_up(T v, unsigned d, int = 64) {
    const int l = hipemu::tl.lane;
    const T o = hipemu::from_bits<T>(hipemu::xchg(hipemu::to_bits(v), l >= (int)d ? l - (int)d : l));
    return l >= (int)d ? o : v;
}
template <class T>
inline T __shfl_down(T v, unsigned d, int = 64) {
    const int l = hipemu::tl.lane;
    const T o = hipemu::from_bits<T>(hipemu::xchg(hipemu::to_bits(v), l + (int)d < 64 ? l + (int)d : l));
    return l + (int)d < 64 ? o : v;
}
template <class T>
inline T __shfl_xor(T v, int m, int = 64) { return hipemu::from_bits<T>(hipemu::xchg(hipemu::to_bits(v), hipemu::tl.lane ^ m)); }
inline unsigned long long __ballot(bool p) { return hipemu::ballot(p); }
#define __builtin_amdgcn_readlane(v, l) ((int)hipemu::xchg((uint64_t)(uint32_t)(v), (l)))
#define __builtin_amdgcn_readfirstlane(v) ((int)hipemu::xchg((uint64_t)(uint32_t)(v), 0))
// DPP: only wave_shr:1 (control 0x138, no bound control) is used by the kernels: lane i takes lane i - 1's value,
// lane 0 keeps `old`
inline int hipemu_update_dpp(int old, int src, int ctrl, int, int, bool) {
    if (ctrl != 0x138) {
        std::abort();
    }
    const int l = hipemu::tl.lane;
    const int v = (int)hipemu::xchg((uint64_t)(uint32_t)src, l > 0 ? l - 1 : 0);
    return l > 0 ? v : old;
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) hipemu_update_dpp((old), (src), (ctrl), (rm), (bm), (bc))
inline uint32_t hipemu_perm(uint32_t s0, uint32_t s1, uint32_t sel) { // v_perm_b32: bytes 0..3 = S1, 4..7 = S0
    const uint64_t src = ((uint64_t)s0 << 32) | s1;
    uint32_t out = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t c = (sel >> (8 * i)) & 0xffu;
        uint32_t b;
        if (c <= 7) {
            b = (uint32_t)(src >> (8 * c)) & 0xffu;
        } else if (c == 0x0c) {
            b = 0;
        } else {
            std::abort(); // (selector forms the kernels do not use)
        }
        out |= b << (8 * i);
    }
    return out;
}
#define __builtin_amdgcn_perm(a, b, sel) hipemu_perm((a), (b), (sel))
// matrix cores, 32x32 output tile of a wave: lane l supplies A[i = l % 32][k-slab l / 32] and B[k-slab l / 32][j = l % 32]
// and holds, in element r of its 16 accumulators, D[i = (r & 3) + 8 (r >> 2) + 4 (l / 32)][j = l % 32].
// v_mfma_f32_32x32x2_f32: one fp32 per lane and operand (K = 2).  v_mfma_f32_32x32x16_f16: 8 halves per lane and operand
// (K = 16, slab = 8 consecutive k).  Products are added to the fp32 accumulator one by one here; the hardware rounds
// differently inside -- the kernels use these results only as a prefilter, never as a returned value.
template <class C>
inline C hipemu_mfma_32x32x2_f32(float a, float b, C c, int, int, int) {
    const float ab[2] = {a, b};
    const unsigned char (*all)[32] = hipemu::xchg_wide(ab, sizeof(ab));
    const int l = hipemu::tl.lane, j = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; r++) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int kk = 0; kk < 2; kk++) {
            float av, bv;
            std::memcpy(&av, all[i + 32 * kk], 4);
            std::memcpy(&bv, all[j + 32 * kk] + 4, 4);
            acc += av * bv;
        }
        c[r] = acc;
    }
    return c;
}
template <class H8, class C>
inline C hipemu_mfma_32x32x16_f16(H8 a, H8 b, C c, int, int, int) {
    static_assert(sizeof(H8) == 16, "8 halves per lane");
    unsigned char ab[32];
    std::memcpy(ab, &a, 16);
    std::memcpy(ab + 16, &b, 16);
    const unsigned char (*all)[32] = hipemu::xchg_wide(ab, 32);
    const int l = hipemu::tl.lane, j = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; r++) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int kk = 0; kk < 2; kk++) {
            _Float16 av[8], bv[8];
            std::memcpy(av, all[i + 32 * kk], 16);
            std::memcpy(bv, all[j + 32 * kk] + 16, 16);
            for (int e = 0; e < 8; e++) {
                acc += (float)av[e] * (float)bv[e];
            }
        }
        c[r] = acc;
    }
    return c;
}
// v_mfma_f32_32x32x16_bf16: the same lane maps with 8 bf16 per lane and operand
template <class B8, class C>
inline C hipemu_mfma_32x32x16_bf16(B8 a, B8 b, C c, int, int, int) {
    static_assert(sizeof(B8) == 16, "8 bf16 per lane");
    unsigned char ab[32];
    std::memcpy(ab, &a, 16);
    std::memcpy(ab + 16, &b, 16);
    const unsigned char (*all)[32] = hipemu::xchg_wide(ab, 32);
    const int l = hipemu::tl.lane, j = l & 31, hi = l >> 5;
    auto widen = [](uint16_t h) {
        const uint32_t u = (uint32_t)h << 16;
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    };
    for (int r = 0; r < 16; r++) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int kk = 0; kk < 2; kk++) {
            uint16_t av[8], bv[8];
            std::memcpy(av, all[i + 32 * kk], 16);
            std::memcpy(bv, all[j + 32 * kk] + 16, 16);
            for (int e = 0; e < 8; e++) {
                acc += widen(av[e]) * widen(bv[e]);
            }
        }
        c[r] = acc;
    }
    return c;
}
// v_mfma_f32_16x16x32_f16: A lane l holds row i = l & 15, k block l >> 4 (8 halves); B lane l column n = l & 15, k
// block l >> 4; D lane l, register r = D[4 (l >> 4) + r][l & 15].  Element e of k block kb of A meets element e of k
// block kb of B -- which k index that is does not matter to a sum over k.
template <class H8, class C>
inline C hipemu_mfma_16x16x32_f16(H8 a, H8 b, C c, int, int, int) {
    static_assert(sizeof(H8) == 16, "8 halves per lane");
    unsigned char ab[32];
    std::memcpy(ab, &a, 16);
    std::memcpy(ab + 16, &b, 16);
    const unsigned char (*all)[32] = hipemu::xchg_wide(ab, 32);
    const int l = hipemu::tl.lane, n = l & 15;
    for (int r = 0; r < 4; r++) {
        const int i = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int kb = 0; kb < 4; kb++) {
            _Float16 av[8], bv[8];
            std::memcpy(av, all[i + 16 * kb], 16);
            std::memcpy(bv, all[n + 16 * kb] + 16, 16);
            for (int e = 0; e < 8; e++) {
                acc += (float)av[e] * (float)bv[e];
            }
        }
        c[r] = acc;
    }
    return c;
}
// v_mfma_i32_16x16x64_i8: 16 signed bytes per lane and operand, int32 accumulator; same lane maps
template <class I4>
inline I4 hipemu_mfma_i32_16x16x64_i8(I4 a, I4 b, I4 c, int, int, int) {
    static_assert(sizeof(I4) == 16, "16 bytes per lane");
    unsigned char ab[32];
    std::memcpy(ab, &a, 16);
    std::memcpy(ab + 16, &b, 16);
    const unsigned char (*all)[32] = hipemu::xchg_wide(ab, 32);
    const int l = hipemu::tl.lane, n = l & 15;
    for (int r = 0; r < 4; r++) {
        const int i = 4 * (l >> 4) + r;
        int acc = c[r];
        for (int kb = 0; kb < 4; kb++) {
            const signed char* av = reinterpret_cast<const signed char*>(all[i + 16 * kb]);
            const signed char* bv = reinterpret_cast<const signed char*>(all[n + 16 * kb] + 16);
            for (int e = 0; e < 16; e++) {
                acc += (int)av[e] * (int)bv[e];
            }
        }
        c[r] = acc;
    }
    return c;
}
#define __builtin_amdgcn_mfma_i32_16x16x64_i8(...) hipemu_mfma_i32_16x16x64_i8(__VA_ARGS__)
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(...) hipemu_mfma_16x16x32_f16(__VA_ARGS__)
#define __builtin_amdgcn_mfma_f32_32x32x2f32(...) hipemu_mfma_32x32x2_f32(__VA_ARGS__)
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(...) hipemu_mfma_32x32x16_f16(__VA_ARGS__)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(...) hipemu_mfma_32x32x16_bf16(__VA_ARGS__)
// buffer descriptors (pq_decode.hip): base + byte count, loads past the end return zeros like the hardware's bounds check
struct hipemu_rsrc {
    const unsigned char* base;
    uint32_t bytes;
};
#define __amdgpu_buffer_rsrc_t hipemu_rsrc
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, num, flags) \
    hipemu_rsrc{reinterpret_cast<const unsigned char*>(p), (uint32_t)(num)}
template <class V>
inline V hipemu_buffer_load(hipemu_rsrc r, uint32_t off) {
    V v{};
    if ((uint64_t)off + sizeof(V) <= r.bytes) {
        std::memcpy(&v, r.base + off, sizeof(V));
    }
    return v;
}
typedef uint32_t hipemu_u4 __attribute__((ext_vector_type(4)));
#define __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, aux) hipemu_buffer_load<hipemu_u4>((r), (uint32_t)((voff) + (soff)))
#define __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, aux) hipemu_buffer_load<uint32_t>((r), (uint32_t)((voff) + (soff)))
#define __builtin_amdgcn_s_setprio(p) ((void)0)
// a wave runs in lockstep on the hardware: LDS traffic between its lanes needs only the compiler's attention there.  Here the
// lanes are separate threads of execution: the wave barrier is a real rendezvous (an exchange every lane takes part in)
#define __builtin_amdgcn_wave_barrier() ((void)hipemu::xchg(0ull, 0))
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_sched_barrier(m) ((void)0)
#define __builtin_amdgcn_s_memtime() 0ull
#define __builtin_amdgcn_s_sleep(n) ((void)0)

// ---- scalar helpers ---------------------------------------------------------------------------------------------------
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fdiv_rn(float a, float b) { return a / b; }
inline float __fmaf_rn(float a, float b, float c) { return std::fmaf(a, b, c); }
inline uint32_t __float_as_uint(float f) { return hipemu::from_bits<uint32_t>(hipemu::to_bits(f)); }
inline int __float_as_int(float f) { return (int)__float_as_uint(f); }
inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline float __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned int v) { return __builtin_popcount(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
template <class T>
inline T min(T a, T b) { return a < b ? a : b; }
template <class T>
inline T max(T a, T b) { return a > b ? a : b; }
inline int64_t min(int64_t a, int b) { return a < b ? a : (int64_t)b; }
inline int64_t min(int a, int64_t b) { return a < b ? (int64_t)a : b; }
inline int64_t max(int64_t a, int b) { return a > b ? a : (int64_t)b; }

// ---- atomics -----------------------------------------------------------------------------------------------------------
template <class T>
inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned* p, int v) { return __atomic_fetch_add(p, (unsigned)v, __ATOMIC_RELAXED); }
inline unsigned long long atomicAdd(unsigned long long* p, int v) { return __atomic_fetch_add(p, (unsigned long long)v, __ATOMIC_RELAXED); }
template <class T>
inline T atomicMax(T* p, T v) {
    T o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (o < v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
    return o;
}
template <class T>
inline T atomicMin(T* p, T v) {
    T o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (o > v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
    return o;
}
template <class T>
inline T hipemu_aload(const T* p) {
    T v;
    __atomic_load(const_cast<T*>(p), &v, __ATOMIC_RELAXED);
    return v;
}
#define __hip_atomic_load(p, order, scope) hipemu_aload(p)
#define __hip_atomic_fetch_min(p, v, order, scope) atomicMin((p), (v))
#define __hip_atomic_fetch_max(p, v, order, scope) atomicMax((p), (v))
