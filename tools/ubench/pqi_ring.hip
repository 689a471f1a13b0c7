// tools/ubench/pqi_ring.hip -- round 5: can the per-unit LUT phase of pqi_kernel (pq_filter.hip) leave the critical path?
// Three questions, each answered by one kernel (no product code involved; random tokens, synthetic tables):
//
//   fill   : how long does it take a 16-wave workgroup to bring a ready-made 128 KB LUT image from global memory into LDS
//            offsets [0, 128 K) with LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPR round trip)?
//            Sources: the workgroup's own slot every time (L2-hot) or a slot that rotates over 256 slots = 32 MB (MALL /
//            HBM).  The image is verified word for word (M0 above 64 KiB is not documented in the guides on file).
//   scan   : the int8 x 16-query gather + v_mfma_i32_16x16x64_i8 loop (adc_loop.hip mode 12) with 16, 12 and 8 scanning
//            waves per CU: what does the loop lose when some of the CU's waves do something else?
//   build  : 12 scanning waves + 4 BUILDER waves in one workgroup: the builders assemble the NEXT unit's LUT image in
//            global memory (16 x 8-byte table pieces per thread-piece from L2 / MALL, 64 v_perm_b32, eight 16-byte stores)
//            while the others scan.  Reports the scan rate beside idle / busy builders and the cycles one image takes.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o pqi_ring pqi_ring.hip ; run on the GPU box (tools/gpu_ubench_ring.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

typedef int i4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

constexpr int LUT_BYTES = 131072;

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

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

// ---- fill ----------------------------------------------------------------------------------------------------------------
// img: 256 slots x 128 KB.  One fill = 128 wave-instructions of 1 KiB (8 per wave).  ROT: slot = (block + it * 37) & 255.
template <bool ROT>
__global__ __launch_bounds__(1024) void kfill(const uint4* __restrict__ img, int iters, unsigned long long* cyc,
                                              uint32_t* bad) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    unsigned long long acc = 0;
    int slot = blockIdx.x & 255;
    for (int it = 0; it < iters; it++) {
        slot = ROT ? ((blockIdx.x + it * 37) & 255) : (blockIdx.x & 255);
        const uint4* src = img + (size_t)slot * (LUT_BYTES / 16);
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int piece = r * 16 + wave; // 1 KiB pieces
            __builtin_amdgcn_global_load_lds((glb_void*)(src + piece * 64 + lane), (lds_void*)(smem + piece * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += __builtin_amdgcn_s_memtime() - t0;
    }
    // verify the last image word for word
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(img + (size_t)slot * (LUT_BYTES / 16));
    uint32_t nbad = 0;
    for (int i = threadIdx.x; i < LUT_BYTES / 4; i += 1024) {
        nbad += reinterpret_cast<const uint32_t*>(smem)[i] != s32[i];
    }
    if (nbad) {
        atomicAdd(bad, nbad);
    }
    if (threadIdx.x == 0) {
        atomicAdd(cyc, acc);
    }
}

// ---- scan (+ builders) -------------------------------------------------------------------------------------------------------
// NSCAN scanning waves (first), NBUILD builder waves (last).  BMODE 0: builders absent (NBUILD = 0) or parked in s_sleep,
// 1: builders assemble images back to back, 2: one image per `nwin_per_image` windows of wave 0 (the real cadence).
template <int NSCAN, int NBUILD, int BMODE>
__global__ __launch_bounds__((NSCAN + NBUILD) * 64) void kscan(float* out, const uint32_t* tok, int nwin,
                                                               const uint2* __restrict__ tables, int ntables,
                                                               uint4* __restrict__ scratch, unsigned long long* cyc,
                                                               unsigned long long* bcyc, unsigned int* bcount) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NT = (NSCAN + NBUILD) * 64;
    uint32_t* lut = reinterpret_cast<uint32_t*>(smem);
    for (int i = threadIdx.x; i < LUT_BYTES / 4; i += NT) {
        lut[i] = (uint32_t)(i * 2654435761u);
    }
    volatile int* stop = reinterpret_cast<volatile int*>(smem + LUT_BYTES);
    if (threadIdx.x == 0) {
        *stop = 0;
        stop[1] = 0; // scanners still running
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wave < NSCAN) {
        uint32_t T[16];
        for (int i = 0; i < 16; i++) {
            T[i] = tok[((blockIdx.x & 255) * 1024 + threadIdx.x) * 16 + i];
        }
        const uint32_t one = 1;
        typedef __attribute__((address_space(3))) const i4v lds_i4;
        i4v im0 = {0, 0, 0, 0}, im1 = {0, 0, 0, 0}, isel = {lane & 1, (lane >> 1) & 1, 0, 1};
        i4v B0[2], B1[2], B2[2], B3[2];
        auto issue2 = [&](uint32_t w, i4v(&v)[2]) {
            v[0] = *reinterpret_cast<lds_i4*>(addr_lo(w, one));
            v[1] = *reinterpret_cast<lds_i4*>(addr_hi(w, one));
        };
        issue2(T[0], B0);
        issue2(T[1], B1);
        issue2(T[2], B2);
        issue2(T[3], B3);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define UNIT(BUF, WORD)                                                          \
    __builtin_amdgcn_sched_barrier(0);                                           \
    im0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(isel, BUF[0], im0, 0, 0, 0);     \
    im1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(isel, BUF[1], im1, 0, 0, 0);     \
    __builtin_amdgcn_sched_barrier(0);                                           \
    issue2(WORD, BUF);
        for (int w = 0; w < nwin; w++) {
            UNIT(B0, T[4]) UNIT(B1, T[5]) UNIT(B2, T[6]) UNIT(B3, T[7])
            UNIT(B0, T[8]) UNIT(B1, T[9]) UNIT(B2, T[10]) UNIT(B3, T[11])
            UNIT(B0, T[12]) UNIT(B1, T[13]) UNIT(B2, T[14]) UNIT(B3, T[15])
            UNIT(B0, T[0]) UNIT(B1, T[1]) UNIT(B2, T[2]) UNIT(B3, T[3])
            __builtin_amdgcn_sched_barrier(0);
        }
#undef UNIT
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        out[(blockIdx.x & 255) * 1024 + threadIdx.x] = (float)(im0[0] + im0[1] + im0[2] + im0[3] + im1[0] + im1[1] + im1[2] +
                                                              im1[3] + B0[0][0] + B1[0][0] + B2[0][0] + B3[0][0]);
        if (lane == 0) {
            atomicAdd(cyc + wave, t1 - t0);
            atomicAdd(const_cast<int*>(stop) + 1, 1);
        }
    } else if (NBUILD > 0) {
        if (BMODE == 0) {
            while (stop[1] < NSCAN) {
                __builtin_amdgcn_s_sleep(32);
            }
        } else {
            // one image = 1024 thread-pieces; builder wave b takes 1024 / NBUILD of them in rounds of 64 lanes
            const int b = wave - NSCAN;
            uint4* dst = scratch + (size_t)blockIdx.x * (LUT_BYTES / 16);
            unsigned int nimg = 0;
            unsigned long long tb = 0;
            uint32_t rng = blockIdx.x * 977u + 13u;
            while (stop[1] < NSCAN) {
                const unsigned long long t0 = __builtin_amdgcn_s_memtime();
                // the unit's 16 queries (wave-uniform pseudo-random picks out of ntables)
                int q[16];
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    rng = rng * 1664525u + 1013904223u;
                    q[j] = __builtin_amdgcn_readfirstlane((int)((rng >> 8) % (uint32_t)ntables));
                }
                for (int r = 0; r < 16 / NBUILD; r++) {
                    const int t = b * (1024 / NBUILD) + r * 64 + lane; // thread-piece: c4 = t >> 4, l16 = t & 15
                    uint2 tp[16];
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        tp[j] = tables[(size_t)q[j] * 1024 + t];
                    }
                    const int c4 = t >> 4, l16 = t & 15;
#pragma unroll
                    for (int w = 0; w < 2; w++) {
                        uint32_t rr[16];
#pragma unroll
                        for (int j = 0; j < 16; j++) {
                            rr[j] = w ? tp[j].y : tp[j].x;
                        }
                        uint32_t lo[8], hi[8];
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            lo[i] = __builtin_amdgcn_perm(rr[2 * i + 1], rr[2 * i], 0x05010400u);
                            hi[i] = __builtin_amdgcn_perm(rr[2 * i + 1], rr[2 * i], 0x07030602u);
                        }
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const uint32_t sel = (e & 1) ? 0x07060302u : 0x05040100u;
                            const uint32_t* s = (e & 2) ? hi : lo;
                            uint4 o;
                            o.x = __builtin_amdgcn_perm(s[1], s[0], sel);
                            o.y = __builtin_amdgcn_perm(s[3], s[2], sel);
                            o.z = __builtin_amdgcn_perm(s[5], s[4], sel);
                            o.w = __builtin_amdgcn_perm(s[7], s[6], sel);
                            const int cc = 2 * w + (e >> 1), h = e & 1;
                            dst[(c4 * 4 + cc) * 32 + l16 + 16 * h] = o;
                        }
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                tb += __builtin_amdgcn_s_memtime() - t0;
                nimg++;
                if (BMODE == 2) { // the real cadence: one image per unit of ~30 k cycles
                    const unsigned long long tw = __builtin_amdgcn_s_memtime();
                    while (stop[1] < NSCAN && __builtin_amdgcn_s_memtime() - tw < 25000ull) {
                        __builtin_amdgcn_s_sleep(16);
                    }
                }
            }
            if (lane == 0 && b == 0) {
                atomicAdd(bcyc, tb);
                atomicAdd(bcount, nimg);
            }
        }
    }
}

template <int NSCAN, int NBUILD, int BMODE>
void run_scan(const char* name, const uint32_t* dtok, float* out, const uint2* tables, int ntables, uint4* scratch,
              unsigned long long* dcyc) {
    const int nwin = 2000, blocks = 256;
    constexpr int NT = (NSCAN + NBUILD) * 64;
    const size_t sm = LUT_BYTES + 64;
    auto kern = kscan<NSCAN, NBUILD, BMODE>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), sm, 0, out, dtok, 10, tables, ntables, scratch, dcyc, dcyc + 20, (unsigned int*)(dcyc + 21));
    CK(hipDeviceSynchronize());
    CK(hipMemset(dcyc, 0, 32 * 8));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), sm, 0, out, dtok, nwin, tables, ntables, scratch, dcyc, dcyc + 20, (unsigned int*)(dcyc + 21));
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[32];
    CK(hipMemcpy(h, dcyc, sizeof(h), hipMemcpyDeviceToHost));
    const double lookups = 256.0 * NSCAN * nwin * 64 * 32 * 16;
    const unsigned int nimg = (unsigned int)(h[21] & 0xffffffffu);
    printf("%-58s %8.3f ms  %6.1f lookups/ns/CU  ticks/window w0 %.0f", name, ms, lookups / 256 / (ms * 1e6),
           (double)h[0] / blocks / nwin);
    if (NBUILD > 0 && BMODE > 0) {
        printf("  | images built %u (%.2f per block), %.0f s_memtime ticks per image", nimg,
               (double)nimg / blocks, nimg ? (double)h[20] / nimg : 0.0);
    }
    printf("\n");
}

// v_cvt_flr_i32_f32 (one instruction) against (int)floorf(x): pqi8_kernel's group test relies on it
__global__ void kflr(const float* x, int* o, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        int r;
        asm("v_cvt_flr_i32_f32_e32 %0, %1" : "=v"(r) : "v"(x[i]));
        o[i] = r;
    }
}

static void check_flr() {
    std::vector<float> h;
    for (int i = -4100; i <= 4100; i++) {
        for (float f : {0.0f, 0.25f, 0.5f, 0.75f, 0.99999994f}) {
            h.push_back((float)i + f);
            h.push_back(((float)i + f) * 65536.0f);
        }
    }
    for (float f : {536870912.0f, -536870912.0f, 536870880.0f, -536870880.0f, 1e-30f, -1e-30f, -0.0f}) {
        h.push_back(f);
    }
    srand(7);
    for (int i = 0; i < 200000; i++) {
        h.push_back(((float)rand() / RAND_MAX - 0.5f) * 1.0e9f);
        h.push_back(((float)rand() / RAND_MAX - 0.5f) * 3.0e3f);
    }
    const int n = (int)h.size();
    float* dx;
    int* dout;
    CK(hipMalloc(&dx, n * 4));
    CK(hipMalloc(&dout, n * 4));
    CK(hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kflr, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dout, n);
    std::vector<int> o(n);
    CK(hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < n; i++) {
        const int want = (int)floorf(h[i]);
        if (o[i] != want) {
            if (bad < 5) {
                printf("  v_cvt_flr_i32_f32(%.9g) = %d, floorf -> %d\n", h[i], o[i], want);
            }
            bad++;
        }
    }
    printf("v_cvt_flr_i32_f32 == (int)floorf on %d values (|x| < 2^29.9): %d mismatches\n", n, bad);
}

int main() {
    check_flr();
    // tokens as adc_loop.hip: code random, m from the stream's phase (any conflict-free pattern does for a rate test)
    const size_t n = (size_t)256 * 1024 * 16;
    std::vector<uint32_t> htok(n);
    srand(1);
    for (size_t t = 0; t < (size_t)256 * 1024; t++) {
        const int L = t & 63;
        const int nn = L & 15, kb = L >> 4;
        const int pn = nn < 8 ? (nn ^ 4) : nn;
        for (int i = 0; i < 16; i++) {
            uint32_t w = 0;
            for (int h = 0; h < 2; h++) {
                const int s = 2 * i + h;
                const uint32_t m = (uint32_t)(16 * (kb >> 1) + ((pn + 8 * (kb & 1) + s) & 15)), code = rand() & 255;
                w |= ((code << 8) | (m << 3)) << (16 * h);
            }
            htok[t * 16 + i] = w;
        }
    }
    uint32_t* dtok;
    float* out;
    unsigned long long* dcyc;
    CK(hipMalloc(&dtok, n * 4));
    CK(hipMalloc(&out, 256 * 1024 * 4));
    CK(hipMalloc(&dcyc, 32 * 8));
    CK(hipMemcpy(dtok, htok.data(), n * 4, hipMemcpyHostToDevice));

    // ---- fill
    const size_t img_bytes = (size_t)256 * LUT_BYTES;
    std::vector<uint32_t> himg(img_bytes / 4);
    for (size_t i = 0; i < himg.size(); i++) {
        himg[i] = (uint32_t)(i * 2246822519u + 374761393u);
    }
    uint4* dimg;
    uint32_t* dbad;
    CK(hipMalloc(&dimg, img_bytes));
    CK(hipMalloc(&dbad, 4));
    CK(hipMemcpy(dimg, himg.data(), img_bytes, hipMemcpyHostToDevice));
    for (int rot = 0; rot < 2; rot++) {
        const int iters = 200;
        auto kern = rot ? kfill<true> : kfill<false>;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LUT_BYTES));
        CK(hipMemset(dcyc, 0, 32 * 8));
        CK(hipMemset(dbad, 0, 4));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), LUT_BYTES, 0, dimg, iters, dcyc, dbad);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h;
        uint32_t bad;
        CK(hipMemcpy(&h, dcyc, 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost));
        printf("fill 128 KB by LDS-DMA, %-34s %8.3f ms for %d fills x 256 CUs: %.2f us per fill, %.0f memtime ticks per fill, "
               "%.2f TB/s chip-wide, mismatching words %u\n",
               rot ? "source rotates over 32 MB (MALL / HBM)" : "same slot every time (L2-hot)", ms, iters, ms * 1e3 / iters,
               (double)h / 256 / iters, 256.0 * LUT_BYTES * iters / (ms * 1e-3) / 1e12, bad);
    }

    // ---- scan / build
    const int ntables = 10000; // 10^4 queries x 8 KB = 80 MB of int8 tables, as at C3
    uint2* dtab;
    uint4* dscr;
    CK(hipMalloc(&dtab, (size_t)ntables * 8192));
    CK(hipMemset(dtab, 0x5a, (size_t)ntables * 8192));
    CK(hipMalloc(&dscr, img_bytes));
    run_scan<16, 0, 0>("scan: 16 waves", dtok, out, dtab, ntables, dscr, dcyc);
    run_scan<12, 0, 0>("scan: 12 waves", dtok, out, dtab, ntables, dscr, dcyc);
    run_scan<8, 0, 0>("scan:  8 waves", dtok, out, dtab, ntables, dscr, dcyc);
    run_scan<12, 4, 0>("scan: 12 waves + 4 parked (s_sleep) waves", dtok, out, dtab, ntables, dscr, dcyc);
    run_scan<12, 4, 1>("scan: 12 waves + 4 builders back to back", dtok, out, dtab, ntables, dscr, dcyc);
    run_scan<12, 4, 2>("scan: 12 waves + 4 builders, one image per ~25 k ticks", dtok, out, dtab, ntables, dscr, dcyc);
    run_scan<14, 2, 1>("scan: 14 waves + 2 builders back to back", dtok, out, dtab, ntables, dscr, dcyc);
    return 0;
}
