// tools/ubench/valu_rates.hip -- issue-rate microbenchmark for the instruction mix of the ADC scan.
// Each kernel runs N iterations of 16 independent copies of one instruction per wave; 4 waves/SIMD.
// Reports cycles per wave-instruction per SIMD (wall clock x measured clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X

template <int MODE>
__global__ __launch_bounds__(256, 4) void k(float* out, int iters, float s) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = s, b1 = s + 1;
    unsigned u0 = threadIdx.x, u1 = u0 * 3;
    unsigned long long acc = 0;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {  // v_add_f32
            asm volatile(REP16("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                               "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
        } else if (MODE == 1) {  // v_pk_add_f32 (4 pairs)
            asm volatile(REP16("v_pk_add_f32 v[10:11], v[10:11], v[18:19]\n v_pk_add_f32 v[12:13], v[12:13], v[18:19]\n"
                               "v_pk_add_f32 v[14:15], v[14:15], v[18:19]\n v_pk_add_f32 v[16:17], v[16:17], v[18:19]\n"
                               "v_pk_add_f32 v[10:11], v[10:11], v[18:19]\n v_pk_add_f32 v[12:13], v[12:13], v[18:19]\n"
                               "v_pk_add_f32 v[14:15], v[14:15], v[18:19]\n v_pk_add_f32 v[16:17], v[16:17], v[18:19]\n")
                         ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19");
        } else if (MODE == 2) {  // v_fmac_f32_dpp wave_shr:1 (8 independent chains)
            asm volatile(REP16("s_nop 1\n v_fmac_f32_dpp %0, %1, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_fmac_f32_dpp %2, %3, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_fmac_f32_dpp %4, %5, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_fmac_f32_dpp %6, %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "s_nop 1\n v_fmac_f32_dpp %1, %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_fmac_f32_dpp %3, %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_fmac_f32_dpp %5, %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_fmac_f32_dpp %7, %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
        } else if (MODE == 3) {  // v_perm_b32
            asm volatile(REP16("v_perm_b32 %0, %0, %2, %3\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %0, %0, %2, %3\n v_perm_b32 %1, %1, %2, %3\n"
                               "v_perm_b32 %0, %0, %2, %3\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %0, %0, %2, %3\n v_perm_b32 %1, %1, %2, %3\n")
                         : "+v"(u0), "+v"(u1) : "v"(u0), "s"(0x0c0c0500u));
        } else if (MODE == 4) {  // v_cmp_le_f32_e64 -> sgpr pair
            unsigned long long m;
            asm volatile(REP16("v_cmp_le_f32_e64 %0, %1, %2\n v_cmp_le_f32_e64 %0, %2, %1\n v_cmp_le_f32_e64 %0, %1, %2\n v_cmp_le_f32_e64 %0, %2, %1\n"
                               "v_cmp_le_f32_e64 %0, %1, %2\n v_cmp_le_f32_e64 %0, %2, %1\n v_cmp_le_f32_e64 %0, %1, %2\n v_cmp_le_f32_e64 %0, %2, %1\n")
                         : "=s"(m) : "v"(a0), "v"(b0));
            acc += m;
        } else if (MODE == 5) {  // v_add_f32 with an exec flip before each (s_mov_b64 exec)
            asm volatile(REP16("s_mov_b64 exec, %9\n v_add_f32 %0, %0, %8\n s_mov_b64 exec, %10\n v_add_f32 %1, %1, %8\n"
                               "s_mov_b64 exec, %9\n v_add_f32 %2, %2, %8\n s_mov_b64 exec, %10\n v_add_f32 %3, %3, %8\n"
                               "s_mov_b64 exec, %9\n v_add_f32 %4, %4, %8\n s_mov_b64 exec, %10\n v_add_f32 %5, %5, %8\n"
                               "s_mov_b64 exec, %9\n v_add_f32 %6, %6, %8\n s_mov_b64 exec, %10\n v_add_f32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(b0), "s"(0x00ff00ff00ff00ffull), "s"(0xff00ff00ff00ff00ull));
            asm volatile("s_mov_b64 exec, -1");
        } else if (MODE == 6) {  // v_min3_f32
            asm volatile(REP16("v_min3_f32 %0, %0, %8, %1\n v_min3_f32 %1, %1, %8, %2\n v_min3_f32 %2, %2, %8, %3\n v_min3_f32 %3, %3, %8, %4\n"
                               "v_min3_f32 %4, %4, %8, %5\n v_min3_f32 %5, %5, %8, %6\n v_min3_f32 %6, %6, %8, %7\n v_min3_f32 %7, %7, %8, %0\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
        } else if (MODE == 7) {  // v_pk_add_f32 with exec flips
            asm volatile(REP16("s_mov_b64 exec, %0\n v_pk_add_f32 v[10:11], v[10:11], v[18:19]\n s_mov_b64 exec, %1\n v_pk_add_f32 v[12:13], v[12:13], v[18:19]\n"
                               "s_mov_b64 exec, %0\n v_pk_add_f32 v[14:15], v[14:15], v[18:19]\n s_mov_b64 exec, %1\n v_pk_add_f32 v[16:17], v[16:17], v[18:19]\n"
                               "s_mov_b64 exec, %0\n v_pk_add_f32 v[10:11], v[10:11], v[18:19]\n s_mov_b64 exec, %1\n v_pk_add_f32 v[12:13], v[12:13], v[18:19]\n"
                               "s_mov_b64 exec, %0\n v_pk_add_f32 v[14:15], v[14:15], v[18:19]\n s_mov_b64 exec, %1\n v_pk_add_f32 v[16:17], v[16:17], v[18:19]\n")
                         :: "s"(0x00ff00ff00ff00ffull), "s"(0xff00ff00ff00ff00ull)
                         : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19");
            asm volatile("s_mov_b64 exec, -1");
        } else if (MODE == 9) {  // v_lshlrev_b32_sdwa WORD_1 (token -> address in pq_scan_q4)
            asm volatile(REP16("v_lshlrev_b32_sdwa %0, %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                               "v_lshlrev_b32_sdwa %1, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                               "v_lshlrev_b32_sdwa %0, %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                               "v_lshlrev_b32_sdwa %1, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                               "v_lshlrev_b32_sdwa %0, %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                               "v_lshlrev_b32_sdwa %1, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                               "v_lshlrev_b32_sdwa %0, %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                               "v_lshlrev_b32_sdwa %1, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n")
                         : "+v"(u0), "+v"(u1) : "s"(1u));
        } else if (MODE == 15) {  // SDWA shift with an inline-constant shift amount
            asm volatile(REP16("v_lshlrev_b32_sdwa %0, 1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                               "v_lshlrev_b32_sdwa %1, 1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                               "v_lshlrev_b32_sdwa %0, 1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                               "v_lshlrev_b32_sdwa %1, 1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                               "v_lshlrev_b32_sdwa %0, 1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                               "v_lshlrev_b32_sdwa %1, 1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                               "v_lshlrev_b32_sdwa %0, 1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                               "v_lshlrev_b32_sdwa %1, 1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n")
                         : "+v"(u0), "+v"(u1));
        } else if (MODE == 16) {  // SDWA shift with a VGPR shift amount
            unsigned one = 1u;
            asm volatile("" : "+v"(one));
            asm volatile(REP16("v_lshlrev_b32_sdwa %0, %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                               "v_lshlrev_b32_sdwa %1, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                               "v_lshlrev_b32_sdwa %0, %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                               "v_lshlrev_b32_sdwa %1, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                               "v_lshlrev_b32_sdwa %0, %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                               "v_lshlrev_b32_sdwa %1, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                               "v_lshlrev_b32_sdwa %0, %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                               "v_lshlrev_b32_sdwa %1, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n")
                         : "+v"(u0), "+v"(u1) : "v"(one));
        } else if (MODE == 17) {  // v_and_b32 with a VGPR mask
            unsigned msk = 0x1fff0u;
            asm volatile("" : "+v"(msk));
            asm volatile(REP16("v_and_b32 %0, %2, %0\n v_and_b32 %1, %2, %1\n v_and_b32 %0, %2, %0\n v_and_b32 %1, %2, %1\n"
                               "v_and_b32 %0, %2, %0\n v_and_b32 %1, %2, %1\n v_and_b32 %0, %2, %0\n v_and_b32 %1, %2, %1\n")
                         : "+v"(u0), "+v"(u1) : "v"(msk));
        } else if (MODE == 18) {  // v_cmp_le_f32 e32 -> vcc
            asm volatile(REP16("v_cmp_le_f32 vcc, %0, %1\n v_cmp_le_f32 vcc, %1, %0\n v_cmp_le_f32 vcc, %0, %1\n v_cmp_le_f32 vcc, %1, %0\n"
                               "v_cmp_le_f32 vcc, %0, %1\n v_cmp_le_f32 vcc, %1, %0\n v_cmp_le_f32 vcc, %0, %1\n v_cmp_le_f32 vcc, %1, %0\n")
                         :: "v"(a0), "v"(b0) : "vcc");
        } else if (MODE == 10) {  // v_and_b32 with an SGPR mask (VOP2)
            asm volatile(REP16("v_and_b32 %0, %2, %0\n v_and_b32 %1, %2, %1\n v_and_b32 %0, %2, %0\n v_and_b32 %1, %2, %1\n"
                               "v_and_b32 %0, %2, %0\n v_and_b32 %1, %2, %1\n v_and_b32 %0, %2, %0\n v_and_b32 %1, %2, %1\n")
                         : "+v"(u0), "+v"(u1) : "s"(0x1fff0u));
        } else if (MODE == 11) {  // v_lshrrev_b32 by an inline constant (VOP2)
            asm volatile(REP16("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n"
                               "v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n")
                         : "+v"(u0), "+v"(u1));
        } else if (MODE == 12) {  // v_bfe_u32 (VOP3)
            asm volatile(REP16("v_bfe_u32 %0, %0, 1, 17\n v_bfe_u32 %1, %1, 1, 17\n v_bfe_u32 %0, %0, 1, 17\n v_bfe_u32 %1, %1, 1, 17\n"
                               "v_bfe_u32 %0, %0, 1, 17\n v_bfe_u32 %1, %1, 1, 17\n v_bfe_u32 %0, %0, 1, 17\n v_bfe_u32 %1, %1, 1, 17\n")
                         : "+v"(u0), "+v"(u1));
        } else if (MODE == 13) {  // v_and_b32 with a 32-bit literal (VOP2 + literal dword)
            asm volatile(REP16("v_and_b32 %0, 0x1fff0, %0\n v_and_b32 %1, 0x1fff0, %1\n v_and_b32 %0, 0x1fff0, %0\n v_and_b32 %1, 0x1fff0, %1\n"
                               "v_and_b32 %0, 0x1fff0, %0\n v_and_b32 %1, 0x1fff0, %1\n v_and_b32 %0, 0x1fff0, %0\n v_and_b32 %1, 0x1fff0, %1\n")
                         : "+v"(u0), "+v"(u1));
        } else if (MODE == 14) {  // v_mov_b32
            asm volatile(REP16("v_mov_b32 %0, %1\n v_mov_b32 %1, %0\n v_mov_b32 %0, %1\n v_mov_b32 %1, %0\n"
                               "v_mov_b32 %0, %1\n v_mov_b32 %1, %0\n v_mov_b32 %0, %1\n v_mov_b32 %1, %0\n")
                         : "+v"(u0), "+v"(u1));
        } else if (MODE == 8) {  // v_fmac_f32 plain (no dpp)
            asm volatile(REP16("v_fmac_f32 %0, %1, %8\n v_fmac_f32 %2, %3, %8\n v_fmac_f32 %4, %5, %8\n v_fmac_f32 %6, %7, %8\n"
                               "v_fmac_f32 %1, %0, %8\n v_fmac_f32 %3, %2, %8\n v_fmac_f32 %5, %4, %8\n v_fmac_f32 %7, %6, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b1 + (float)u0 + (float)u1 + (float)acc;
}

template <int MODE>
double run(const char* name, int per_iter_insts) {
    const int blocks = 256 * 4, iters = 2000;  // 4 blocks of 256 threads per CU = 4 waves / SIMD
    float* out;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves x iters x per_iter_insts wave-instructions
    double insts_per_simd = 4.0 * iters * per_iter_insts;
    double ns_per_inst = ms * 1e6 / insts_per_simd;
    printf("%-28s %8.3f ms  %6.3f ns per wave-instruction per SIMD  (= %5.2f cycles @2.4GHz, %5.2f @2.0GHz)\n", name, ms,
           ns_per_inst, ns_per_inst * 2.4, ns_per_inst * 2.0);
    hipFree(out);
    return ns_per_inst;
}

// does ds_read_b128 ignore the low 4 address bits?
__global__ void k_mis(float* out) {
    __shared__ __align__(16) float lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (float)i;
    __syncthreads();
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const f4 lds_f4;
    const unsigned base = (unsigned)(size_t)((__attribute__((address_space(3))) float*)lds);
    const unsigned mis[4] = {0, 1, 2, 8};
    for (int t = 0; t < 4; t++) {
        f4 v;
        const unsigned addr = base + 64 + mis[t];
        asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
        if (threadIdx.x == 0) { out[t * 4 + 0] = v.x; out[t * 4 + 1] = v.y; out[t * 4 + 2] = v.z; out[t * 4 + 3] = v.w; }
    }
}
void misaligned_b128() {
    float* out; hipMalloc(&out, 64);
    hipLaunchKernelGGL(k_mis, dim3(1), dim3(64), 0, 0, out);
    float h[16]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    const int mis[4] = {0, 1, 2, 8};
    for (int t = 0; t < 4; t++) printf("ds_read_b128 at aligned+%d -> %g %g %g %g (aligned data: 16 17 18 19)\n", mis[t], h[t*4], h[t*4+1], h[t*4+2], h[t*4+3]);
}

int main() {
    run<0>("v_add_f32", 128);
    run<8>("v_fmac_f32", 128);
    run<1>("v_pk_add_f32", 128);
    run<2>("v_fmac_f32_dpp wave_shr (+nop)", 128);
    run<3>("v_perm_b32", 128);
    run<4>("v_cmp_le_f32_e64 ->sgpr", 128);
    run<6>("v_min3_f32", 128);
    run<5>("v_add_f32 + exec flip", 128);
    run<7>("v_pk_add_f32 + exec flip", 128);
    run<9>("v_lshlrev_b32_sdwa", 128);
    run<15>("v_lshlrev_b32_sdwa (inline const)", 128);
    run<16>("v_lshlrev_b32_sdwa (vgpr amount)", 128);
    run<10>("v_and_b32 (sgpr mask)", 128);
    run<17>("v_and_b32 (vgpr mask)", 128);
    run<18>("v_cmp_le_f32 e32 -> vcc", 128);
    run<13>("v_and_b32 (literal)", 128);
    run<11>("v_lshrrev_b32", 128);
    run<12>("v_bfe_u32", 128);
    run<14>("v_mov_b32", 128);
    misaligned_b128();
    return 0;
}
