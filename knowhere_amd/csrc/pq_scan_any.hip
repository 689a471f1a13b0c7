// knowhere_amd/csrc/pq_scan_any.hip -- exact IVF-PQ ADC scan for ANY number of 8-bit sub-quantizers.
//
// The reference's IVF_PQ accepts every m that divides dim (src/index/ivf/ivf_config.h:118, faiss::ProductQuantizer);
// the fast kernels of this backend (pq_filter.hip, pq_scan_q4.hip, pq_scan_v2.hip, pq_scan.hip) are built for
// m in {8, 16, 32, 64}.  This kernel serves every other width (m = 1 .. 128: 4, 12, 24, 48, 96, ...) with the same
// results as the reference:
//   * tables   precompute_list_tables_L2 / _IP and the residual tables (IVFPQ_QueryTables.cpp:110-230): the same
//              arithmetic as pq_scan.hip's LUT build and range.hip::pq_adc_dump_kernel
//   * distance PQCodeDistanceScalar::distance_single_code (pq_code_distance-inl.h:69-90): the m table values summed from 0
//              in m order, then dis0 + sum (IVFPQScanner_impl.h:109-181)
//   * top-k    per wave a canonical (distance, id) top-k; the four waves of a workgroup write four sorted partial lists
//              (partial slot = 4 * probe rank + wave), merged by topk.hip::merge_partials -- boundary ties are then
//              resolved like for every other kernel (knhip_api.hip::search_batch_ties)
// One workgroup per (query, probed list): the (query, list) table [m][256] fp32 in LDS (m KB; 160 KB LDS holds m = 128),
// one thread per stored vector reading its m code bytes from the list-sorted AoS codes.  Not a tuned kernel: LDS
// gathers with random bank conflicts, a table build per (query, list); it is the completeness path, the headline
// shapes never reach it.
#include "common.h"
#include "kernels.h"

namespace knhip {

constexpr int PA_THREADS = 256;
constexpr int PA_WAVES = PA_THREADS / KN_WAVE;

int pq_scan_any_supports(int M, int d) {
    // LDS: the table (M KB here; 256 * dsub floats in the encoder, build.hip::launch_pq_encode)
    return M >= 1 && M <= 128 && d % M == 0 && (size_t)256 * (d / M) * sizeof(float) <= 144 * 1024;
}

template <bool IS_L2, int R>
__global__ __launch_bounds__(PA_THREADS) void pq_scan_any_kernel(PqAnyArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* lut = reinterpret_cast<float*>(smem); // [M][256]
    const int lane = lane_id();
    const int wave = threadIdx.x / KN_WAVE;
    const int64_t q = blockIdx.x / a.nprobe;
    const int slot = (int)(blockIdx.x % a.nprobe);
    float* pd = a.partial_d + ((q * a.nprobe + slot) * PA_WAVES + wave) * (int64_t)a.k;
    int64_t* pi = a.partial_i + ((q * a.nprobe + slot) * PA_WAVES + wave) * (int64_t)a.k;
    WaveTopK<IS_L2, R> top;
    top.init(a.k);
    const int64_t list = a.keys[q * a.nprobe + slot];
    const int64_t len = (list >= 0 && list < a.nlist) ? a.list_len[list] : 0;
    if (len <= 0) { // (uniform over the workgroup) an empty partial list
        top.store(pd, pi);
        return;
    }
    const int M = a.M, dsub = a.d / a.M;
    for (int e = threadIdx.x; e < M * 256; e += PA_THREADS) {
        const int m = e >> 8, c = e & 255;
        float t;
        if (a.lut_mode == PQ_LUT_RESIDUAL) { // ||(q - c_list)_m - cb[m][c]||^2
            const float* y = a.cb + ((int64_t)m * 256 + c) * dsub;
            const float* x = a.queries + q * a.d + m * dsub;
            const float* cl = a.centroids + list * a.d + m * dsub;
            t = 0.f;
            for (int i = 0; i < dsub; i++) {
                t = l2_step(t, fsub_x(x[i], cl[i]), y[i]);
            }
        } else {
            t = a.t2t[(q * 256 + c) * M + m]; // <q_m, cb[m][c]>
            if (a.lut_mode == PQ_LUT_PRECOMP) {
                t = fadd_x(a.precomp_t[(list * 256 + c) * M + m], fmul_x(-2.0f, t));
            }
        }
        lut[e] = t;
    }
    __syncthreads();
    const float dis0 = a.lut_mode == PQ_LUT_RESIDUAL ? 0.f : a.coarse_dis[q * a.nprobe + slot];
    const int64_t row_off = a.list_row_off[list];
    float kd = worst_dist<IS_L2>();
    int64_t ki = -1;
    const bool words = (M & 3) == 0; // (rows of M bytes start 4-byte aligned)
    for (int64_t p0 = (int64_t)wave * KN_WAVE; p0 < len; p0 += PA_THREADS) {
        const int64_t pos = p0 + lane;
        bool ok = pos < len;
        float dis = 0.f;
        int64_t id = -1;
        if (ok) {
            id = a.ids[row_off + pos];
            ok = !bitset_filtered(a.bitset, a.bitset_nbits, id);
        }
        if (ok) {
            const uint8_t* code = a.codes + (row_off + pos) * M;
            float acc = 0.f;
            if (words) {
                const uint32_t* cw = reinterpret_cast<const uint32_t*>(code);
                for (int m = 0; m < M; m += 4) {
                    const uint32_t w = cw[m >> 2];
                    acc = fadd_x(acc, lut[(m + 0) * 256 + (w & 0xffu)]);
                    acc = fadd_x(acc, lut[(m + 1) * 256 + ((w >> 8) & 0xffu)]);
                    acc = fadd_x(acc, lut[(m + 2) * 256 + ((w >> 16) & 0xffu)]);
                    acc = fadd_x(acc, lut[(m + 3) * 256 + (w >> 24)]);
                }
            } else {
                for (int m = 0; m < M; m++) {
                    acc = fadd_x(acc, lut[m * 256 + code[m]]);
                }
            }
            dis = fadd_x(dis0, acc);
        }
        unsigned long long mk = __ballot(ok && top.admits(dis, id, kd, ki));
        while (mk) {
            const int l = __ffsll((long long)mk) - 1;
            mk &= mk - 1;
            const float cd = readlane_f(dis, l);
            const int64_t ci = readlane_i64(id, l);
            if (top.admits(cd, ci, kd, ki)) {
                top.insert(cd, ci);
                kd = top.kth_dist();
                ki = top.kth_idx();
            }
        }
    }
    top.store(pd, pi);
}

hipError_t launch_pq_scan_any(const PqAnyArgs& a, int64_t nq, bool is_l2, hipStream_t s) {
    if (nq <= 0 || a.nprobe <= 0) {
        return hipSuccess;
    }
    if (!pq_scan_any_supports(a.M, a.d) || a.k <= 0 || a.k > KN_MAX_K) {
        return hipErrorInvalidValue;
    }
    const size_t sm = (size_t)a.M * 256 * sizeof(float);
    const unsigned grid = (unsigned)(nq * a.nprobe);
    KN_DISPATCH_R(a.k, {
        auto kern = is_l2 ? pq_scan_any_kernel<true, R_> : pq_scan_any_kernel<false, R_>;
        if (sm > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)sm);
            if (e != hipSuccess) {
                return e;
            }
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(PA_THREADS), sm, s, a);
    });
    return hipGetLastError();
}

} // namespace knhip
