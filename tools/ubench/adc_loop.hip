// tools/ubench/adc_loop.hip -- issue-rate microbenchmark of the pq_scan_q4 window loop in isolation:
// one 16-wave workgroup per CU, 128 KB LUT in LDS, random tokens held in registers, no global traffic.
// Modes knock out one ingredient at a time to show which unit bounds the loop.
//   0 full (SDWA + ds_read_b128 + split/plain accumulates)      1 no EXEC flips (every step plain)
//   2 no LDS reads (values = registers)                          3 no SDWA (addresses precomputed)
//   4 only ds_read_b128 (no accumulate)                          5 only accumulates (plain + split), no LDS/SDWA
//   6 f16 table, 8 queries per ds_read_b128, 4 v_pk_add_f16 per lookup, no EXEC flips: the loop an approximate
//     (prefilter) ADC pass could run -- lanes rotated through m instead of delayed, half the LDS bytes per query
//   7 the same with v_pk_add_f16 only (no LDS, no SDWA): its VALU floor
//   8 f16 table, the additions on the matrix cores: every ds_read_b128 result (8 halves per lane) is the B operand of one
//     v_mfma_f32_16x16x32_f16 whose A operand is a constant selector (A[i][k] = [i == k mod 8]): fp32 accumulation of
//     512 lookups per instruction, no VALU on the data path (two accumulators alternate)
//   9 the same with a single accumulator (dependent MFMAs back to back)
//  10 mode 8 without the LDS reads (MFMA + SDWA only)
//  11 mode 8 with two value buffers (4 reads in flight instead of 8): what pq_filter.hip's registers allow
//  12 int8 table, 16 queries per entry: one v_mfma_i32_16x16x64_i8 per ds_read_b128 (1024 lookups per instruction)
//  13 int8 table, one v_smfmac_i32_16x16x128_i8 (2:4-sparse selector) per TWO ds_read_b128 (2048 lookups per instruction)
//  14 int8 table, EIGHT queries per entry (8 bytes, 64 KB LUT): two ds_read_b64 (plain and / shift addresses: the 16-bit
//     token IS the byte address) feed one v_mfma_i32_16x16x64_i8 -- the loop of two 8-wave workgroups per CU, each with
//     its own 64 KB LUT (one can rebuild its LUT while the other scans); run as 512 blocks x 512 threads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../knowhere_amd/csrc/kernels.h"

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
using knhip::pq_stream_mask;

#define SPLIT(MK, X, Y) "s_mov_b32 exec_lo, " MK "\n s_mov_b32 exec_hi, " MK "\n v_pk_add_f32 %0, %0, " X "\n v_pk_add_f32 %1, %1, " Y "\n s_not_b64 exec, exec\n v_pk_add_f32 %2, %2, " X "\n v_pk_add_f32 %3, %3, " Y "\n"
#define PLAIN(X, Y) "v_pk_add_f32 %0, %0, " X "\n v_pk_add_f32 %1, %1, " Y "\n"
#define VALS(v) "v"(__builtin_shufflevector(v[0], v[0], 0, 1)), "v"(__builtin_shufflevector(v[0], v[0], 2, 3)), "v"(__builtin_shufflevector(v[1], v[1], 0, 1)), "v"(__builtin_shufflevector(v[1], v[1], 2, 3))

template <int U, bool FLIP>
__device__ __forceinline__ void accum2(f2& n01, f2& n23, f2& o01, f2& o23, const f4 (&v)[2]) {
    if constexpr (FLIP && U < 7) {
        asm volatile(SPLIT("%8", "%4", "%5") SPLIT("%9", "%6", "%7") "s_mov_b64 exec, -1\n"
                     : "+v"(n01), "+v"(n23), "+v"(o01), "+v"(o23) : VALS(v), "i"(pq_stream_mask(2 * U)), "i"(pq_stream_mask(2 * U + 1)));
    } else if constexpr (FLIP && U == 7) {
        asm volatile(SPLIT("%8", "%4", "%5") "s_mov_b64 exec, -1\n" PLAIN("%6", "%7")
                     : "+v"(n01), "+v"(n23), "+v"(o01), "+v"(o23) : VALS(v), "i"(pq_stream_mask(14)));
    } else {
        asm volatile(PLAIN("%4", "%5") PLAIN("%6", "%7") : "+v"(n01), "+v"(n23), "+v"(o01), "+v"(o23) : VALS(v));
    }
}

// modes 6 / 7: the 16 bytes of a lookup are 8 half-precision table values (8 queries); four packed accumulators
__device__ __forceinline__ void accum2_h(uint32_t& h0, uint32_t& h1, uint32_t& h2, uint32_t& h3, const f4 (&v)[2]) {
    asm volatile("v_pk_add_f16 %0, %0, %4\n v_pk_add_f16 %1, %1, %5\n v_pk_add_f16 %2, %2, %6\n v_pk_add_f16 %3, %3, %7\n"
                 "v_pk_add_f16 %0, %0, %8\n v_pk_add_f16 %1, %1, %9\n v_pk_add_f16 %2, %2, %10\n v_pk_add_f16 %3, %3, %11\n"
                 : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3)
                 : "v"(v[0].x), "v"(v[0].y), "v"(v[0].z), "v"(v[0].w), "v"(v[1].x), "v"(v[1].y), "v"(v[1].z), "v"(v[1].w));
}

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void accum2_mfma(f4& a0, f4& a1, const h8 sel, const f4 (&v)[2], bool one_acc) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(sel, __builtin_bit_cast(h8, v[0]), a0, 0, 0, 0);
    if (one_acc) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(sel, __builtin_bit_cast(h8, v[1]), a0, 0, 0, 0);
    } else {
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(sel, __builtin_bit_cast(h8, v[1]), a1, 0, 0, 0);
    }
}

typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void accum2_i8(i4v& a0, i4v& a1, const i4v sel, const f4 (&v)[2]) {
    a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(sel, __builtin_bit_cast(i4v, v[0]), a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(sel, __builtin_bit_cast(i4v, v[1]), a1, 0, 0, 0);
}
__device__ __forceinline__ void accum2_smfmac(i4v& a0, const i4v sel, int idx, const f4 (&v)[2]) {
    const i4v lo = __builtin_bit_cast(i4v, v[0]), hi = __builtin_bit_cast(i4v, v[1]);
    const i8v b = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    a0 = __builtin_amdgcn_smfmac_i32_16x16x128_i8(sel, b, a0, idx, 0, 0);
}

__device__ __forceinline__ uint32_t addr_lo(uint32_t w, uint32_t one) {
    uint32_t a;
    asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(a) : "v"(w), "s"(one));
    return a;
}
__device__ __forceinline__ uint32_t addr_hi(uint32_t w, uint32_t one) {
    uint32_t a;
    asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(a) : "v"(w), "s"(one));
    return a;
}

// mode 14: 8 waves, 64 KB LUT of 8-byte entries; per token word two ds_read_b64 and one MFMA
typedef int i2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512) void k14(float* out, const uint32_t* tok, int nwin, unsigned long long* cyc) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* lut = reinterpret_cast<float*>(smem);
    for (int i = threadIdx.x; i < 16384; i += 512) lut[i] = (float)(i & 1023) * 0.001f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t T[16];
    for (int i = 0; i < 16; i++) T[i] = tok[((blockIdx.x & 255) * 1024 + threadIdx.x) * 16 + i];
    typedef __attribute__((address_space(3))) const i2v lds_i2;
    i4v im0 = {0, 0, 0, 0}, im1 = {0, 0, 0, 0}, isel = {lane & 1, (lane >> 1) & 1, 0, 1};
    i4v B0, B1, B2, B3, B4, B5, B6, B7;
    auto issue = [&](uint32_t w, i4v& v) {
        const i2v lo = *reinterpret_cast<lds_i2*>(w & 0xffffu);
        const i2v hi = *reinterpret_cast<lds_i2*>(w >> 16);
        v = i4v{lo[0], lo[1], hi[0], hi[1]};
    };
    issue(T[0], B0); issue(T[1], B1); issue(T[2], B2); issue(T[3], B3);
    issue(T[4], B4); issue(T[5], B5); issue(T[6], B6); issue(T[7], B7);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define U14(ACC, BUF, WORD)                                                          \
    __builtin_amdgcn_sched_barrier(0);                                               \
    ACC = __builtin_amdgcn_mfma_i32_16x16x64_i8(isel, BUF, ACC, 0, 0, 0);            \
    __builtin_amdgcn_sched_barrier(0);                                               \
    issue(WORD, BUF);
    for (int w = 0; w < nwin; w++) {
        U14(im0, B0, T[8]) U14(im1, B1, T[9]) U14(im0, B2, T[10]) U14(im1, B3, T[11])
        U14(im0, B4, T[12]) U14(im1, B5, T[13]) U14(im0, B6, T[14]) U14(im1, B7, T[15])
        U14(im0, B0, T[0]) U14(im1, B1, T[1]) U14(im0, B2, T[2]) U14(im1, B3, T[3])
        U14(im0, B4, T[4]) U14(im1, B5, T[5]) U14(im0, B6, T[6]) U14(im1, B7, T[7])
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(blockIdx.x & 255) * 1024 + threadIdx.x] = (float)(im0[0] + im0[1] + im0[2] + im0[3] + im1[0] + im1[1] + im1[2] + im1[3] +
                                                          B0[0] + B1[0] + B2[0] + B3[0] + B4[0] + B5[0] + B6[0] + B7[0]);
    if (lane == 0) atomicAdd(cyc + (threadIdx.x >> 6), t1 - t0);
}

void run14(const uint32_t* dtok, float* out, unsigned long long* dcyc, int blocks) {
    const int nwin = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k14), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k14, dim3(blocks), dim3(512), 65536, 0, out, dtok, 10, dcyc);
    hipDeviceSynchronize();
    hipMemset(dcyc, 0, 16 * 8);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k14, dim3(blocks), dim3(512), 65536, 0, out, dtok, nwin, dcyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16];
    hipMemcpy(h, dcyc, sizeof(h), hipMemcpyDeviceToHost);
    // per workgroup: 8 waves x nwin windows x 16 MFMAs x 1024 lookups
    const double lookups = (double)blocks * 8 * nwin * 16 * 1024;
    printf("int8 x 8 queries, 2 x ds_read_b64 per MFMA, %d blocks of 8 waves  %8.3f ms  %5.1f lookups/ns/CU  ticks/window: w0 %.0f w7 %.0f\n",
           blocks, ms, lookups / 256 / (ms * 1e6), (double)h[0] / blocks / nwin, (double)h[7] / blocks / nwin);
}

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, const uint32_t* tok, int nwin, unsigned long long* cyc) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* lut = reinterpret_cast<float*>(smem);
    for (int i = threadIdx.x; i < 32768; i += 1024) lut[i] = (float)(i & 1023) * 0.001f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // 16 token words per lane: code random, m = (step - phase) mod 32 as in the real stream
    uint32_t T[16];
    for (int i = 0; i < 16; i++) T[i] = tok[(blockIdx.x * 1024 + threadIdx.x) * 16 + i];
    const uint32_t one = 1;
    typedef __attribute__((address_space(3))) const f4 lds_f4;
    f2 n01 = {0, 0}, n23 = {0, 0}, o01 = {0, 0}, o23 = {0, 0};
    f4 B0[2], B1[2], B2[2], B3[2];
    uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0;
    f4 m0 = {0, 0, 0, 0}, m1 = {0, 0, 0, 0};
    i4v im0 = {0, 0, 0, 0}, im1 = {0, 0, 0, 0}, isel = {lane & 1, (lane >> 1) & 1, 0, 1};
    const int iidx = 0x44444444 ^ (lane & 3);
    h8 sel;
    for (int e = 0; e < 8; e++) sel[e] = (_Float16)(((lane & 15) == e) ? 1.0f : 0.0f);
    auto rd = [&](uint32_t a) -> f4 {
        if (MODE == 2 || MODE == 5 || MODE == 7 || MODE == 10) return f4{__uint_as_float(a), 1.f, 2.f, 3.f};
        return *reinterpret_cast<lds_f4*>(a);
    };
    auto issue2 = [&](uint32_t w, f4 (&v)[2]) {
        if (MODE == 3 || MODE == 5 || MODE == 7) {
            v[0] = rd(w & 0x1fff0u);
            v[1] = rd((w >> 15) & 0x1fff0u);
        } else {
            v[0] = rd(addr_lo(w, one));
            v[1] = rd(addr_hi(w, one));
        }
    };
    if (MODE == 3 || MODE == 5) {
        // addresses precomputed once: (mode 3 keeps the ds_reads, drops the per-step VALU address op)
    }
    issue2(T[0], B0); issue2(T[1], B1); issue2(T[2], B2); issue2(T[3], B3);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define UNIT(U, BUF, WORD)                                                   \
    __builtin_amdgcn_sched_barrier(0);                                       \
    if (MODE == 12) accum2_i8(im0, im1, isel, BUF);                          \
    else if (MODE == 13) accum2_smfmac(im0, isel, iidx, BUF);                \
    else if (MODE >= 8) accum2_mfma(m0, m1, sel, BUF, MODE == 9);            \
    else if (MODE == 6 || MODE == 7) accum2_h(h0, h1, h2, h3, BUF);          \
    else if (MODE != 4) accum2<U, MODE != 1>(n01, n23, o01, o23, BUF);       \
    else asm volatile("" :: "v"(BUF[0]), "v"(BUF[1]));                       \
    __builtin_amdgcn_sched_barrier(0);                                       \
    issue2(WORD, BUF);
    if (MODE == 11) {
        for (int w = 0; w < nwin; w++) {
            UNIT(0, B0, T[2]) UNIT(1, B1, T[3]) UNIT(2, B0, T[4]) UNIT(3, B1, T[5])
            UNIT(4, B0, T[6]) UNIT(5, B1, T[7]) UNIT(6, B0, T[8]) UNIT(7, B1, T[9])
            UNIT(8, B0, T[10]) UNIT(9, B1, T[11]) UNIT(10, B0, T[12]) UNIT(11, B1, T[13])
            UNIT(12, B0, T[14]) UNIT(13, B1, T[15]) UNIT(14, B0, T[0]) UNIT(15, B1, T[1])
            __builtin_amdgcn_sched_barrier(0);
        }
    } else
    for (int w = 0; w < nwin; w++) {
        UNIT(0, B0, T[4]) UNIT(1, B1, T[5]) UNIT(2, B2, T[6]) UNIT(3, B3, T[7])
        UNIT(4, B0, T[8]) UNIT(5, B1, T[9]) UNIT(6, B2, T[10]) UNIT(7, B3, T[11])
        UNIT(8, B0, T[12]) UNIT(9, B1, T[13]) UNIT(10, B2, T[14]) UNIT(11, B3, T[15])
        UNIT(12, B0, T[0]) UNIT(13, B1, T[1]) UNIT(14, B2, T[2]) UNIT(15, B3, T[3])
        __builtin_amdgcn_sched_barrier(0);
        o01 = n01; o23 = n23;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 1024 + threadIdx.x] = n01.x + n01.y + n23.x + n23.y + o01.x + o23.y + B0[0].x + B1[0].x + B2[0].x + B3[0].x +
                                           __uint_as_float(h0 ^ h1 ^ h2 ^ h3) + m0.x + m0.y + m0.z + m0.w + m1.x + m1.y + m1.z + m1.w +
                                           (float)(im0[0] + im0[1] + im0[2] + im0[3] + im1[0] + im1[1] + im1[2] + im1[3]);
    if (lane == 0) atomicAdd(cyc + (threadIdx.x >> 6), t1 - t0);
}

template <int MODE>
void run(const char* name, const uint32_t* dtok, float* out, unsigned long long* dcyc) {
    const int nwin = 2000, blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipMemset(dcyc, 0, 16 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 131072, 0, out, dtok, 10, dcyc);
    hipDeviceSynchronize();
    hipMemset(dcyc, 0, 16 * 8);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 131072, 0, out, dtok, nwin, dcyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16];
    hipMemcpy(h, dcyc, sizeof(h), hipMemcpyDeviceToHost);
    // per CU: 16 waves x nwin windows; lookups per window per wave = 64 lanes x 32 steps x 4 queries (8 in the f16 modes)
    const double lookups = 256.0 * 16 * nwin * 64 * 32 * ((MODE >= 12) ? 16 : (MODE >= 6) ? 8 : 4);
    printf("%-34s %8.3f ms  %6.1f ns per window-round (16 waves)  %5.1f lookups/ns/CU (LDS peak 64/clk)  ticks/window: w0 %.0f w15 %.0f\n", name, ms,
           ms * 1e6 / nwin, lookups / 256 / (ms * 1e6), (double)h[0] / blocks / nwin, (double)h[15] / blocks / nwin);
}

int main() {
    const size_t n = (size_t)256 * 1024 * 16;
    uint32_t* htok = (uint32_t*)malloc(n * 4);
    srand(1);
    for (size_t t = 0; t < (size_t)256 * 1024; t++) {
        const int lane = t & 63;
        const int ph = knhip::pq_stream_phase(lane);
        for (int i = 0; i < 16; i++) {
            uint32_t w = 0;
            for (int h = 0; h < 2; h++) {
                const int step = 2 * i + h;
                const uint32_t m = (uint32_t)((step - ph) & 31), code = rand() & 255;
                w |= ((code << 8) | (m << 3)) << (16 * h);
            }
            htok[t * 16 + i] = w;
        }
    }
    uint32_t* dtok; float* out; unsigned long long* dcyc;
    hipMalloc(&dtok, n * 4); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&dcyc, 16 * 8);
    hipMemcpy(dtok, htok, n * 4, hipMemcpyHostToDevice);
    run<0>("full", dtok, out, dcyc);
    run<1>("no exec flips", dtok, out, dcyc);
    run<2>("no LDS reads", dtok, out, dcyc);
    run<3>("no SDWA (and+shift addresses)", dtok, out, dcyc);
    run<4>("only SDWA + ds_read_b128", dtok, out, dcyc);
    run<5>("only accumulates", dtok, out, dcyc);
    run<6>("f16 table, 8 queries, no flips", dtok, out, dcyc);
    run<7>("f16: only v_pk_add_f16", dtok, out, dcyc);
    run<8>("f16 table, MFMA 16x16x32 adds", dtok, out, dcyc);
    run<9>("f16 table, MFMA adds, one acc", dtok, out, dcyc);
    run<10>("MFMA adds + SDWA, no LDS", dtok, out, dcyc);
    run<11>("MFMA adds, 4 reads in flight", dtok, out, dcyc);
    run<12>("int8 table, MFMA i8 16x16x64", dtok, out, dcyc);
    run<13>("int8 table, SMFMAC i8 16x16x128", dtok, out, dcyc);
    // mode 14 tokens: lane (kb = L >> 4, n = L & 15) walks m = 16 (kb & 1) + ((n + step) & 15): the 32 lanes of either half
    // of the wave touch 32 different sub-quantizers = 32 different bank pairs of the 8-byte entries at every step
    for (size_t t = 0; t < (size_t)256 * 1024; t++) {
        const int L = t & 63;
        for (int i = 0; i < 16; i++) {
            uint32_t w = 0;
            for (int h = 0; h < 2; h++) {
                const int step = 2 * i + h;
                const uint32_t m = (uint32_t)(16 * ((L >> 4) & 1) + ((L + step) & 15)), code = rand() & 255;
                w |= ((code << 8) | (m << 3)) << (16 * h);
            }
            htok[t * 16 + i] = w;
        }
    }
    hipMemcpy(dtok, htok, n * 4, hipMemcpyHostToDevice);
    run14(dtok, out, dcyc, 512);  // two workgroups per CU
    run14(dtok, out, dcyc, 256);  // one workgroup per CU (half the waves): what one scans at while the other builds its LUT
    return 0;
}
