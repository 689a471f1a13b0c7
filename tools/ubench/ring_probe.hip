// tools/ubench/ring_probe.hip -- does the hand-issued ring of pq_decode.hip deliver what it loads?  One wave walks `ntile` tiles
// of a buffer whose dword i holds i; every taken slot is written out and compared on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t pd_u4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t pd_rsrc;
#define KN_WAVE 64
typedef int pd_i4 __attribute__((ext_vector_type(4)));
#if defined(__HIP_DEVICE_COMPILE__)
#define PD_RING_CLOBBER                                                                                                      \
    "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250",  \
            "a251", "a252", "a253", "a254", "a255", "memory"
struct PdRing {}; // (nothing: the slots are the named registers)
// slot I <- 16 code bytes (+ the start value).  s_nop 4: the scalar offsets are computed right in front of this statement, and
// a vector memory instruction that reads an SGPR needs five wait states behind the scalar instruction that wrote it -- the
// compiler pads that for its own instructions, not for inline ISA (without the padding the loads took the PREVIOUS tile's
// offset now and then: tiles scanned twice, tiles missed)
template <int I, bool WITH_P>
__device__ __forceinline__ void pd_ring_load(PdRing&, pd_i4 rc, int voff_c, int soff_c, pd_i4 rp, int voff_p, int soff_p) {
    static_assert(I >= 0 && I < 4, "four slots");
#define PD_LD(WREG, PREG)                                                                                                    \
    if (WITH_P) {                                                                                                            \
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 " WREG ", %0, %1, %2 offen\n\tbuffer_load_dword " PREG ", %3, %4, %5 offen" \
                     :                                                                                                       \
                     : "v"(voff_c), "s"(rc), "s"(soff_c), "v"(voff_p), "s"(rp), "s"(soff_p)                                  \
                     : PD_RING_CLOBBER);                                                                                     \
    } else {                                                                                                                 \
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 " WREG ", %0, %1, %2 offen" : : "v"(voff_c), "s"(rc), "s"(soff_c) : PD_RING_CLOBBER); \
    }
    if (I == 0) {
        PD_LD("a[236:239]", "a252")
    } else if (I == 1) {
        PD_LD("a[240:243]", "a253")
    } else if (I == 2) {
        PD_LD("a[244:247]", "a254")
    } else {
        PD_LD("a[248:251]", "a255")
    }
#undef PD_LD
}
// wait until at most N loads are in flight, then slot I -> (w, p)
template <int I, int N, bool WITH_P>
__device__ __forceinline__ void pd_ring_take(PdRing&, pd_u4& w, float& p) {
    uint32_t w0, w1, w2, w3;
    float pp = 0.f;
#define PD_TK(W0, W1, W2, W3, PREG)                                                                                          \
    if (WITH_P) {                                                                                                            \
        asm volatile("s_waitcnt vmcnt(%5)\n\tv_accvgpr_read_b32 %0, " W0 "\n\tv_accvgpr_read_b32 %1, " W1                   \
                     "\n\tv_accvgpr_read_b32 %2, " W2 "\n\tv_accvgpr_read_b32 %3, " W3 "\n\tv_accvgpr_read_b32 %4, " PREG   \
                     : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(pp)                                                      \
                     : "n"(N)                                                                                                \
                     : PD_RING_CLOBBER);                                                                                     \
    } else {                                                                                                                 \
        asm volatile("s_waitcnt vmcnt(%4)\n\tv_accvgpr_read_b32 %0, " W0 "\n\tv_accvgpr_read_b32 %1, " W1                   \
                     "\n\tv_accvgpr_read_b32 %2, " W2 "\n\tv_accvgpr_read_b32 %3, " W3                                      \
                     : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3)                                                                \
                     : "n"(N)                                                                                                \
                     : PD_RING_CLOBBER);                                                                                     \
    }
    if (I == 0) {
        PD_TK("a236", "a237", "a238", "a239", "a252")
    } else if (I == 1) {
        PD_TK("a240", "a241", "a242", "a243", "a253")
    } else if (I == 2) {
        PD_TK("a244", "a245", "a246", "a247", "a254")
    } else {
        PD_TK("a248", "a249", "a250", "a251", "a255")
    }
#undef PD_TK
    w[0] = w0;
    w[1] = w1;
    w[2] = w2;
    w[3] = w3;
    p = pp;
}
// nothing of the ring is in flight any more (its registers are the compiler's again)
__device__ __forceinline__ void pd_ring_drain(PdRing&) { asm volatile("s_waitcnt vmcnt(0)" : : : PD_RING_CLOBBER); }
#else
struct PdRing {
    pd_u4 w[4];
    float p[4];
};
template <int I, bool WITH_P>
__device__ __forceinline__ void pd_ring_load(PdRing& r, pd_i4 rc, int voff_c, int soff_c, pd_i4 rp, int voff_p, int soff_p) {
    const pd_rsrc dc = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<void*>(((uint64_t)(uint32_t)rc[1] << 32) | (uint32_t)rc[0]), 0, rc[2], rc[3]);
    r.w[I] = __builtin_amdgcn_raw_buffer_load_b128(dc, voff_c, soff_c, 0);
    r.p[I] = 0.f;
    if (WITH_P) {
        const pd_rsrc dp = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<void*>(((uint64_t)(uint32_t)rp[1] << 32) | (uint32_t)rp[0]), 0, rp[2], rp[3]);
        r.p[I] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dp, voff_p, soff_p, 0));
    }
}
template <int I, int N, bool WITH_P>
__device__ __forceinline__ void pd_ring_take(PdRing& r, pd_u4& w, float& p) {
    w = r.w[I];
    p = r.p[I];
}
__device__ __forceinline__ void pd_ring_drain(PdRing&) {}
#endif


__global__ __launch_bounds__(256, 1) void probe(const uint32_t* codes, const float* ps, int ntile, uint32_t* out_w, float* out_p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64));
    const int lr = lane & 31, hi = lane >> 5;
    auto make_rsrc = [](const void* p, int bytes) -> pd_i4 {
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        pd_i4 r;
        r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)v);
        r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) & 0xffff;
        r[2] = __builtin_amdgcn_readfirstlane(bytes);
        r[3] = 0x00020000;
        return r;
    };
    const pd_i4 rc = make_rsrc(codes, ntile * 1024);
    const pd_i4 rp = make_rsrc(ps, ntile * 128);
    const int voff_c = lr * 32 + hi * 16, voff_p = lr * 4;
    const int t_end = ntile - 1;
    PdRing ring;
    pd_u4 w0;
    float p0;
    pd_ring_load<0, true>(ring, rc, voff_c, min(wave, t_end) * 1024, rp, voff_p, min(wave, t_end) * 128);
    pd_ring_take<0, 0, true>(ring, w0, p0);
    pd_ring_load<0, true>(ring, rc, voff_c, min(wave + 4, t_end) * 1024, rp, voff_p, min(wave + 4, t_end) * 128);
    pd_ring_load<1, true>(ring, rc, voff_c, min(wave + 8, t_end) * 1024, rp, voff_p, min(wave + 8, t_end) * 128);
    pd_ring_load<2, true>(ring, rc, voff_c, min(wave + 12, t_end) * 1024, rp, voff_p, min(wave + 12, t_end) * 128);
    pd_ring_load<3, true>(ring, rc, voff_c, min(wave + 16, t_end) * 1024, rp, voff_p, min(wave + 16, t_end) * 128);
    auto put = [&](int t, pd_u4 w, float p) {
        for (int e = 0; e < 4; e++) out_w[((size_t)t * 64 + lane) * 4 + e] = w[e];
        out_p[(size_t)t * 64 + lane] = p;
    };
    put(wave, w0, p0);
    int t = wave;
    for (;;) {
        bool done = false;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (!done) {
                pd_u4 wn;
                float pn;
                if (i == 0) pd_ring_take<0, 6, true>(ring, wn, pn);
                else if (i == 1) pd_ring_take<1, 6, true>(ring, wn, pn);
                else if (i == 2) pd_ring_take<2, 6, true>(ring, wn, pn);
                else pd_ring_take<3, 6, true>(ring, wn, pn);
                if (t + 4 < ntile) put(t + 4, wn, pn);
                const int tn = min(t + 20, t_end);
                if (i == 0) pd_ring_load<0, true>(ring, rc, voff_c, tn * 1024, rp, voff_p, tn * 128);
                else if (i == 1) pd_ring_load<1, true>(ring, rc, voff_c, tn * 1024, rp, voff_p, tn * 128);
                else if (i == 2) pd_ring_load<2, true>(ring, rc, voff_c, tn * 1024, rp, voff_p, tn * 128);
                else pd_ring_load<3, true>(ring, rc, voff_c, tn * 1024, rp, voff_p, tn * 128);
                t += 4;
                done = t >= ntile;
            }
        }
        if (done) break;
    }
    pd_ring_drain(ring);
}

int main() {
    const int ntile = 40;
    std::vector<uint32_t> hc((size_t)ntile * 256);
    std::vector<float> hp((size_t)ntile * 32);
    for (size_t i = 0; i < hc.size(); i++) hc[i] = (uint32_t)i;
    for (size_t i = 0; i < hp.size(); i++) hp[i] = (float)i;
    uint32_t *dc, *dw;
    float *dp, *dpo;
    hipMalloc(&dc, hc.size() * 4);
    hipMalloc(&dp, hp.size() * 4);
    hipMalloc(&dw, (size_t)ntile * 64 * 16);
    hipMalloc(&dpo, (size_t)ntile * 64 * 4);
    hipMemcpy(dc, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dp, hp.data(), hp.size() * 4, hipMemcpyHostToDevice);
    hipMemset(dw, 0xff, (size_t)ntile * 64 * 16);
    probe<<<1, 256>>>(dc, dp, ntile, dw, dpo);
    hipDeviceSynchronize();
    std::vector<uint32_t> ow((size_t)ntile * 256);
    std::vector<float> op((size_t)ntile * 64);
    hipMemcpy(ow.data(), dw, ow.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(op.data(), dpo, op.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < ntile; t++)
        for (int l = 0; l < 64; l++) {
            const int lr = l & 31, hi = l >> 5;
            for (int e = 0; e < 4; e++) {
                const uint32_t want = (uint32_t)(t * 256 + lr * 8 + hi * 4 + e), got = ow[((size_t)t * 64 + l) * 4 + e];
                if (want != got && bad++ < 10) printf("tile %d lane %d word %d: got %u want %u\n", t, l, e, got, want);
            }
            const float wantp = (float)(t * 32 + lr), gotp = op[(size_t)t * 64 + l];
            if (wantp != gotp && bad++ < 10) printf("tile %d lane %d start: got %g want %g\n", t, l, gotp, wantp);
        }
    printf("ring_probe: %d mismatches\n", bad);
    return bad != 0;
}
