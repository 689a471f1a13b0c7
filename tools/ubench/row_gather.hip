// tools/ubench/row_gather.hip -- how should a wave gather 64 random 512-byte rows out of a 51 GB array?
//
// refine.hip re-scores k_base = 100 candidates per query against raw fp32 rows: 10^6 random 512-byte rows per 10k-query batch
// out of 51.2 GB (C3).  Its loads are lane = row: every wave-level 16-byte load touches 64 different rows (64 different
// pages), 32 such loads per row set.  Measured 0.45 ms = 1.1 TB/s; PMC says 79 % of the wave cycles wait on memory
// instructions and DESIGN suspected address translation.  This measures the alternatives on the same access stream:
//   mode 0  lane = row, 16-byte loads, 8 in flight per lane (refine.hip today)
//   mode 1  half-wave = row: one wave-level load covers two whole rows (2 pages per instruction instead of 64), 16 in flight
//   mode 2  as 1, the row ids sorted per wave (no effect expected: rows stay random across waves)
// Every variant sums what it reads into a per-row checksum so that nothing is optimised away; the checksums must agree.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/row_gather tools/ubench/row_gather.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                    \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

constexpr int D4 = 32; // float4 pieces per row (d = 128)

__global__ void fill_kernel(float4* x, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = (float)((i * 2654435761ull) & 1023) * 0.001f;
        x[i] = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
    }
}

// mode 0: lane = row
__global__ __launch_bounds__(256) void gather_lane_kernel(const float4* __restrict__ base, const int64_t* __restrict__ ids,
                                                          int64_t n, float* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) {
        return;
    }
    const float4* y = base + ids[r] * D4;
    float acc = 0.f;
    for (int j = 0; j < D4; j += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            v[u] = y[j + u];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            acc += v[u].x + v[u].y + v[u].z + v[u].w;
        }
    }
    out[r] = acc;
}

// mode 1 / 2: half-wave = row; a wave takes 64 rows, instruction j covers rows 2 j and 2 j + 1
__global__ __launch_bounds__(256) void gather_coop_kernel(const float4* __restrict__ base, const int64_t* __restrict__ ids,
                                                          int64_t n, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t r0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~63ll;
    if (r0 >= n) {
        return;
    }
    const int64_t myid = r0 + lane < n ? ids[r0 + lane] : ids[r0];
    const int half = lane >> 5, p = lane & 31;
    float sum[2] = {0.f, 0.f}; // (this lane's share of rows 2 j + half, j even / odd kept apart only to shorten the chain)
    float part[32];
#pragma unroll
    for (int j0 = 0; j0 < 32; j0 += 16) {
        float4 v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int64_t id = __shfl(myid, 2 * (j0 + u) + half, 64);
            v[u] = base[id * D4 + p];
        }
#pragma unroll
        for (int u = 0; u < 16; u++) {
            part[j0 + u] = v[u].x + v[u].y + v[u].z + v[u].w;
        }
    }
    // row 2 j + half: the sum over the 32 lanes of its half-wave (order differs from mode 0: compared with a tolerance)
#pragma unroll
    for (int j = 0; j < 32; j++) {
        float s = part[j];
#pragma unroll
        for (int dlt = 16; dlt > 0; dlt >>= 1) {
            s += __shfl_xor(s, dlt, 64);
        }
        if (p == 0 && r0 + 2 * j + half < n) {
            out[r0 + 2 * j + half] = s;
        }
    }
    (void)sum;
}

int main(int argc, char** argv) {
    const int64_t nrows = argc > 1 ? atoll(argv[1]) : 100000000ll; // 51.2 GB
    const int64_t n = 1000000;                                       // rows gathered per launch
    float4* base = nullptr;
    CK(hipMalloc(&base, (size_t)nrows * D4 * sizeof(float4)));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, base, nrows * D4);
    CK(hipDeviceSynchronize());
    std::mt19937_64 rng(7);
    std::vector<int64_t> ids((size_t)n), sorted;
    for (auto& v : ids) {
        v = (int64_t)(rng() % (uint64_t)nrows);
    }
    sorted = ids;
    for (int64_t w = 0; w + 64 <= n; w += 64) {
        std::sort(sorted.begin() + w, sorted.begin() + w + 64);
    }
    int64_t *d_ids = nullptr, *d_sorted = nullptr;
    float *o0 = nullptr, *o1 = nullptr;
    CK(hipMalloc(&d_ids, n * 8));
    CK(hipMalloc(&d_sorted, n * 8));
    CK(hipMalloc(&o0, n * 4));
    CK(hipMalloc(&o1, n * 4));
    CK(hipMemcpy(d_ids, ids.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_sorted, sorted.data(), n * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((n + 255) / 256);
    for (int mode = 0; mode < 3; mode++) {
        float best = 1e9f;
        for (int it = 0; it < 6; it++) {
            CK(hipEventRecord(e0, 0));
            if (mode == 0) {
                hipLaunchKernelGGL(gather_lane_kernel, dim3(grid), dim3(256), 0, 0, base, d_ids, n, o0);
            } else {
                hipLaunchKernelGGL(gather_coop_kernel, dim3(grid), dim3(256), 0, 0, base, mode == 1 ? d_ids : d_sorted, n, o1);
            }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (it > 0) {
                best = std::min(best, ms);
            }
        }
        printf("mode %d: %.3f ms for %lld rows of 512 B out of %.1f GB = %.2f TB/s\n", mode, best, (long long)n,
               (double)nrows * 512 / 1e9, (double)n * 512 / best / 1e9);
        if (mode == 1) {
            std::vector<float> a((size_t)n), b((size_t)n);
            CK(hipMemcpy(a.data(), o0, n * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), o1, n * 4, hipMemcpyDeviceToHost));
            int64_t bad = 0;
            for (int64_t i = 0; i < n; i++) {
                bad += std::abs(a[(size_t)i] - b[(size_t)i]) > 1e-3f * std::abs(a[(size_t)i]) + 1e-3f;
            }
            printf("checksums differing between mode 0 and mode 1: %lld\n", (long long)bad);
        }
    }
    return 0;
}
