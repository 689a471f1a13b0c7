// tools/probe/smfmac_probe.hip -- operand layout of v_smfmac_i32_16x16x128_i8 on gfx950, found by experiment:
// one-hot compressed A element (lane la, byte p) x one-hot B element (lane lb, byte e) under a uniform index pattern;
// every (la, p, lb, e) whose product reaches the accumulator is printed with the accumulator cell it lands in.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i8v __attribute__((ext_vector_type(8)));

struct Hit { int la, p, lb, e, lane, r, val; };

__global__ void probe(int idxword, Hit* hits, int* nhits, int cap, int la_lo, int la_hi) {
    const int lane = threadIdx.x;
    for (int la = la_lo; la < la_hi; la++) {
        for (int p = 0; p < 16; p++) {
            i4 a = {0, 0, 0, 0};
            if (lane == la) a[p >> 2] = 1 << (8 * (p & 3));
            for (int lb = 0; lb < 64; lb += 16) {
                for (int e = 0; e < 32; e++) {
                    i8v b = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (lane == lb) b[e >> 2] = 1 << (8 * (e & 3));
                    i4 c = {0, 0, 0, 0};
                    c = __builtin_amdgcn_smfmac_i32_16x16x128_i8(a, b, c, idxword, 0, 0);
                    for (int r = 0; r < 4; r++) {
                        if (c[r] != 0) {
                            const int n = atomicAdd(nhits, 1);
                            if (n < cap) hits[n] = Hit{la, p, lb, e, lane, r, c[r]};
                        }
                    }
                }
            }
        }
    }
}

int main(int argc, char** argv) {
    const int cap = 1 << 20;
    Hit* d; int* dn;
    hipMalloc(&d, cap * sizeof(Hit)); hipMalloc(&dn, 4);
    // index patterns: 4 bits per group of 4 k's = {idx0, idx1}; uniform over the 8 groups of a lane
    const int pats[4][2] = {{0, 1}, {2, 3}, {0, 3}, {1, 2}};
    for (int pi = 0; pi < 4; pi++) {
        int w = 0;
        for (int g = 0; g < 8; g++) w |= (pats[pi][0] | (pats[pi][1] << 2)) << (4 * g);
        hipMemset(dn, 0, 4);
        // only A lanes 0, 1, 16, 17, 33 (rows 0, 1; k blocks 0, 1, 2): enough to see the structure
        const int las[5] = {0, 1, 16, 17, 33};
        for (int t = 0; t < 5; t++) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, w, d, dn, cap, las[t], las[t] + 1);
        }
        hipDeviceSynchronize();
        int n; hipMemcpy(&n, dn, 4, hipMemcpyDeviceToHost);
        std::vector<Hit> h(n < cap ? n : cap);
        hipMemcpy(h.data(), d, h.size() * sizeof(Hit), hipMemcpyDeviceToHost);
        printf("pattern idx0=%d idx1=%d word=%08x hits=%d\n", pats[pi][0], pats[pi][1], (unsigned)w, n);
        for (size_t i = 0; i < h.size() && i < 400; i++) {
            printf("  A(lane %2d byte %2d) x B(lane %2d byte %2d) -> D(lane %2d reg %d) = %d\n", h[i].la, h[i].p, h[i].lb, h[i].e,
                   h[i].lane, h[i].r, h[i].val);
        }
    }
    return 0;
}
