// tools/ubench/mfma_stream.hip -- how fast does ONE wave per SIMD issue v_mfma_f32_32x32x16_f16 back to back, with the
// accumulators in VGPRs (-mllvm -amdgpu-mfma-vgpr-form) or in AGPRs (default)?  Built twice by the run line below.
//   hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-mfma-vgpr-form] mfma_stream.hip -o mfma_stream_{v,a}
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC, int FILL>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* ticks, int iters) {
    extern __shared__ unsigned char smem[];
    h8 a, b[NACC];
    for (int e = 0; e < 8; e++) {
        a[e] = (_Float16)(threadIdx.x * 0.001f + e);
        for (int j = 0; j < NACC; j++) b[j][e] = (_Float16)(threadIdx.x * 0.002f + e + j);
    }
    f16v acc[NACC];
    for (int j = 0; j < NACC; j++)
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    float f = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int s = 0; s < 8; s++) {
#pragma unroll
            for (int j = 0; j < NACC; j++) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[j], acc[j], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < FILL; u++) {
                    f = f * 1.0001f + 0.5f; // independent VALU filler
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = f;
    for (int j = 0; j < NACC; j++)
        for (int r = 0; r < 16; r++) sum += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int NACC, int FILL>
void run(const char* name) {
    float* out;
    unsigned long long* ticks;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&ticks, 256 * 8);
    const int iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<NACC, FILL>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    k<NACC, FILL><<<256, 256, 100 * 1024>>>(out, ticks, iters);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256; i++) s += (double)h[i];
    printf("%s NACC=%d FILL=%d: %.1f ticks per MFMA\n", name, NACC, FILL, s / 256 / iters / (8.0 * NACC));
    hipFree(out);
    hipFree(ticks);
}

// the same stream for several milliseconds, timed from outside: the rate a whole chip of matrix pipes SUSTAINS (ticks of
// s_memtime follow the shader clock; wall time shows what that clock was)
template <int NACC>
void run_long(const char* name, int iters) {
    float* out;
    unsigned long long* ticks;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&ticks, 256 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<NACC, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<NACC, 0><<<256, 256, 100 * 1024>>>(out, ticks, 200); // warm
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        k<NACC, 0><<<256, 256, 100 * 1024>>>(out, ticks, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[256];
        hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0;
        for (int i = 0; i < 256; i++) s += (double)h[i];
        const double n = (double)iters * 8.0 * NACC; // matrix instructions per wave
        printf("%s long NACC=%d: %.2f ms, %.1f ticks per MFMA, %.2f ns per MFMA = %.2f GHz at 32 cycles each, %.0f TFLOP/s on 1024 pipes\n",
               name, NACC, ms, s / 256 / n, ms * 1e6 / n, 32.0 / (ms * 1e6 / n), 1024.0 * n * 32768.0 / (ms * 1e-3) / 1e12);
    }
    hipFree(out);
    hipFree(ticks);
}

int main(int argc, char** argv) {
    const char* name = argc > 1 ? argv[1] : "?";
    if (argc > 2) {
        run_long<3>(name, 3000);   // ~1 ms
        run_long<3>(name, 15000);  // ~5 ms
        run_long<3>(name, 60000);  // ~20 ms
        return 0;
    }
    run<1, 0>(name);
    run<2, 0>(name);
    run<3, 0>(name);
    run<4, 0>(name);
    run<3, 2>(name);
    run<3, 4>(name);
    run<3, 6>(name);
    run<4, 4>(name);
    return 0;
}
