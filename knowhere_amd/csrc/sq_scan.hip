// knowhere_amd/csrc/sq_scan.hip -- IVF-SQ8 list scan for gfx950 (exact reference arithmetic).
//
// Replaces, on the device:
//   * BaselineIVFSQScannerIP / BaselineIVFSQScannerL2::scan_codes
//       (reference thirdparty/faiss/faiss/cppcontrib/knowhere/IndexScalarQuantizer.cpp:196-400)
//   * DCTemplate<Quantizer, Similarity>::compute_distance
//       (thirdparty/faiss/faiss/impl/scalar_quantizer/distance_computers.h:37-45)
//   * Codec8bit::decode_component  (.../scalar_quantizer/codecs.h:37-41):  xi = (code + 0.5f) / 255.0f
//   * QuantizerTemplate<NON_UNIFORM>::reconstruct_component (.../quantizers.h:139-145):
//       x = vmin[i] + xi * vdiff[i]
//   * SimilarityL2 / SimilarityIP::add_component (.../similarities.h:46-49, 80-82)
// by_residual (IndexScalarQuantizer.h:47):
//   IP : dis = coarse_dis + sum_i q_i * x_i
//   L2 : dis = sum_i ((q - c_list)_i - x_i)^2
// Every operation is rounded once, in the reference's order (sequential over i).
//
// HBM layout: rows in blocks of 64, block = uint4 blk[nchunk16][64] with blk[c][r] = code bytes
// 16c..16c+15 of row 64b+r, so a lane reads 16 dims of its own row per 16-byte coalesced load.
// Work item = (list, up to QG queries that probe it): the codes are decoded once for the QG
// queries.  Algorithmic bytes per item = rows * d * npair.
#include "common.h"
#include "kernels.h"

namespace knhip {

constexpr int SQ_WAVES = 4;
constexpr int SQ_THREADS = SQ_WAVES * KN_WAVE;
// queries per work item: the per-query top-k state is R = ceil(k / 64) (distance, position) registers per lane,
// so the group shrinks as k grows (same schedule as the flat scan; refine asks for k x refine_k candidates)
int sq_scan_qg(int k) {
    return k <= 128 ? 8 : (k <= 256 ? 4 : (k <= 512 ? 2 : 1));
}

// one work item (`item_or_block` is the item when a.item_loop is set, else the block index that xcd_item maps to an
// item).  Every exit is workgroup-uniform.
template <bool IS_L2, int QG, int R>
__device__ __forceinline__ void sq_scan_item(const SqScanArgs& a, const int64_t item_or_block, unsigned char* smem,
                                             float* tab) {
    const int lane = lane_id();
    const int wave = threadIdx.x / KN_WAVE;
    const int dpad = a.nchunk16 * 16;

    const int64_t nitems = *a.nitems_dev;
    int64_t item;
    if (a.item_loop) {
        item = item_or_block;
    } else {
        if (item_or_block >= ((nitems + 7) / 8) * 8) {
            return;
        }
        item = xcd_item(item_or_block, nitems);
    }
    if (item >= nitems) {
        return;
    }
    const KnItem it = a.items[item];
    const int npair = it.npair;
    const int64_t list = it.list;
    const int64_t blk0 = a.list_blk_off[list];
    const int64_t len = a.list_len[list];
    const int64_t row_off = a.list_row_off[list];
    int32_t q_of[QG], slot_of[QG];
    float accu0[QG];
#pragma unroll
    for (int j = 0; j < QG; j++) {
        const KnPair p = a.pairs[it.pair0 + min(j, npair - 1)];
        q_of[j] = p.q;
        slot_of[j] = p.slot;
        accu0[j] = IS_L2 ? 0.f : a.coarse_dis[(int64_t)p.q * a.nslot + p.slot];
    }

    // LDS: y[QG][dpad] (query, or query residual for L2), vmin[dpad], vdiff[dpad]
    float* sy = reinterpret_cast<float*>(smem);
    float* svmin = sy + QG * dpad;
    float* svdiff = svmin + dpad;
    if (threadIdx.x < 256) {
        tab[threadIdx.x] = __fdiv_rn((float)threadIdx.x + 0.5f, 255.0f);
    }
    for (int t = threadIdx.x; t < QG * dpad; t += SQ_THREADS) {
        const int j = t / dpad, i = t % dpad;
        float v = 0.f;
        if (i < a.d) {
            v = a.queries[(int64_t)q_of[j] * a.d + i];
            if (IS_L2) {
                v = fsub_x(v, a.centroids[list * a.d + i]); // compute_residual: x - centroid
            }
        }
        sy[t] = v;
    }
    for (int i = threadIdx.x; i < dpad; i += SQ_THREADS) {
        // padded dims decode to 0 and meet y = 0: they add exactly +0 to the sums
        svmin[i] = (i < a.d) ? a.trained[i] : 0.f;
        svdiff[i] = (i < a.d) ? a.trained[a.d + i] : 0.f;
    }
    __syncthreads();

    WaveTopK<IS_L2, R> top[QG];
    float kd[QG], gt[QG];
    int64_t ki[QG];
#pragma unroll
    for (int j = 0; j < QG; j++) {
        top[j].init(a.k);
        kd[j] = worst_dist<IS_L2>();
        ki[j] = -1;
        gt[j] = worst_dist<IS_L2>();
    }

    const int64_t nblk = (len + 63) / 64;
    for (int64_t b = wave; b < nblk; b += SQ_WAVES) {
        const int64_t row = b * 64 + lane;
#pragma unroll
        for (int j = 0; j < QG; j++) {
            gt[j] = gthr_load<IS_L2>(a.gthr + q_of[j]);
        }
        bool valid = row < len;
        if (a.bitset != nullptr && valid) {
            valid = !bitset_filtered(a.bitset, a.bitset_nbits, a.ids[row_off + row]);
        }
        const uint4* p = a.rows + (blk0 + b) * (int64_t)a.nchunk16 * 64 + lane;
        float acc[QG];
#pragma unroll
        for (int j = 0; j < QG; j++) {
            acc[j] = 0.f;
        }
        constexpr int UF = QG <= 2 ? 4 : 1; // (few-query items: keep several code loads in flight)
#pragma unroll UF
        for (int c = 0; c < a.nchunk16; c++) {
            const uint4 w = p[(int64_t)c * 64];
            const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const uint32_t code = (ww[e >> 2] >> (8 * (e & 3))) & 0xffu;
                const int i = c * 16 + e;
                const float xi = tab[code];
                const float x = fadd_x(svmin[i], fmul_x(xi, svdiff[i]));
#pragma unroll
                for (int j = 0; j < QG; j++) {
                    const float y = sy[j * dpad + i];
                    if (IS_L2) {
                        acc[j] = l2_step(acc[j], y, x); // tmp = y - x; accu += tmp * tmp
                    } else {
                        acc[j] = ip_step(acc[j], y, x); // accu += y * x
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < QG; j++) {
            if (j < npair && a.dump != nullptr) {
                // range search: every distance of the list, filtered rows as the sentinel
                if (row < len) {
                    const float dis = IS_L2 ? acc[j] : fadd_x(accu0[j], acc[j]);
                    a.dump[(int64_t)q_of[j] * a.dump_stride + row_off + row] = valid ? dis : worst_dist<IS_L2>();
                }
            } else if (j < npair) {
                const float dis = IS_L2 ? acc[j] : fadd_x(accu0[j], acc[j]);
                const bool pass = valid && within_gthr<IS_L2>(dis, gt[j]) &&
                                  top[j].admits(dis, row, kd[j], ki[j]);
                unsigned long long m = __ballot(pass);
                bool tightened = false;
                while (m) {
                    const int l = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const float cd = readlane_f(dis, l);
                    const int64_t ci = b * 64 + l;
                    if (top[j].admits(cd, ci, kd[j], ki[j])) {
                        top[j].insert(cd, ci);
                        kd[j] = top[j].kth_dist();
                        ki[j] = top[j].kth_idx();
                        tightened = true;
                    }
                }
                if (tightened && ki[j] >= 0 && lane == 0) {
                    gthr_publish<IS_L2>(a.gthr + q_of[j], kd[j]);
                }
            }
        }
    }

    if (a.dump != nullptr) {
        return;
    }
    // ---- merge waves (same scheme as flat_scan) ----
    __syncthreads();
    const int k = a.k;
    const int qr = max(1, min(QG, (int)(48 * 1024 / (SQ_WAVES * k * 12))));
    float* md = reinterpret_cast<float*>(smem);
    int64_t* mi = reinterpret_cast<int64_t*>(smem + (((size_t)qr * SQ_WAVES * k * 4 + 7) & ~(size_t)7));
    for (int j0 = 0; j0 < QG; j0 += qr) {
#pragma unroll
        for (int j = 0; j < QG; j++) {
            if (j >= j0 && j < j0 + qr) {
                top[j].store(md + ((j - j0) * SQ_WAVES + wave) * k, mi + ((j - j0) * SQ_WAVES + wave) * k);
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < QG; j++) {
            if (j >= j0 && j < j0 + qr && j < npair && (j % SQ_WAVES) == wave) {
                for (int w = 1; w < SQ_WAVES; w++) {
                    const int ow = (wave + w) % SQ_WAVES;
                    const float* od = md + ((j - j0) * SQ_WAVES + ow) * k;
                    const int64_t* oi = mi + ((j - j0) * SQ_WAVES + ow) * k;
                    for (int e = 0; e < k; e++) {
                        const float cd = od[e];
                        const int64_t ci = oi[e];
                        if (ci < 0 || !top[j].admits(cd, ci, kd[j], ki[j])) {
                            break;
                        }
                        top[j].insert(cd, ci);
                        kd[j] = top[j].kth_dist();
                        ki[j] = top[j].kth_idx();
                    }
                }
                // the merged list bounds the query's final k-th with the whole list behind it: far
                // tighter than any single wave's slice (matters most for large k)
                if (ki[j] >= 0 && lane == 0) {
                    gthr_publish<IS_L2>(a.gthr + q_of[j], kd[j]);
                }
                float* pd = a.partial_d + ((int64_t)q_of[j] * a.nslot + slot_of[j]) * k;
                int64_t* pi = a.partial_i + ((int64_t)q_of[j] * a.nslot + slot_of[j]) * k;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int e = r * KN_WAVE + lane;
                    if (e < k) {
                        const int64_t pos = top[j].i[r];
                        pd[e] = top[j].d[r];
                        pi[e] = pos >= 0 ? a.ids[row_off + pos] : -1;
                    }
                }
            }
        }
        __syncthreads();
    }
}

template <bool IS_L2, int QG, int R>
__global__ __launch_bounds__(SQ_THREADS) void sq_scan_kernel(SqScanArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ float tab[256]; // (c + 0.5f) / 255.0f, correctly rounded
    if (a.item_loop) {
        // a fixed grid walks an item table whose size only the device knows (mfma_scan.hip fallback)
        const int64_t nitems = *a.nitems_dev;
        for (int64_t item = blockIdx.x; item < nitems; item += gridDim.x) {
            sq_scan_item<IS_L2, QG, R>(a, item, smem, tab);
            __syncthreads();
        }
        return;
    }
    sq_scan_item<IS_L2, QG, R>(a, blockIdx.x, smem, tab);
}

// AoS codes [len][d] (sorted by list) -> interleaved blocks
__global__ void sq_interleave_kernel(const uint8_t* __restrict__ codes,
                                     const int64_t* __restrict__ list_row_off,
                                     const int64_t* __restrict__ list_len,
                                     const int64_t* __restrict__ list_blk_off, int64_t nlist, int d,
                                     int nchunk16, uint4* __restrict__ out) {
    const int64_t l = blockIdx.y + (int64_t)blockIdx.z * gridDim.y;
    if (l >= nlist) {
        return;
    }
    const int64_t len = list_len[l];
    const int64_t nblk = (len + 63) / 64;
    const int64_t row_off = list_row_off[l];
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nblk * nchunk16 * 64;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / ((int64_t)nchunk16 * 64);
        const int rem = (int)(t % ((int64_t)nchunk16 * 64));
        const int c = rem / 64, r = rem % 64;
        const int64_t row = b * 64 + r;
        uint32_t w[4] = {0, 0, 0, 0};
        if (row < len) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int i = c * 16 + e;
                if (i < d) {
                    w[e >> 2] |= (uint32_t)codes[(row_off + row) * d + i] << (8 * (e & 3));
                }
            }
        }
        out[(list_blk_off[l] + b) * (int64_t)nchunk16 * 64 + rem] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

hipError_t launch_sq_scan(const SqScanArgs& a, bool is_l2, int64_t grid, hipStream_t s, int qg_override) {
    if (grid <= 0) {
        return hipSuccess;
    }
    if (!a.item_loop) {
        grid = (grid + 7) / 8 * 8; // (xcd_item spreads the items over the blocks [0, round_up(nitems, 8)))
    }
    const int dpad = a.nchunk16 * 16;
    const int k = a.k;
    if (k > KN_MAX_K) {
        return hipErrorInvalidValue;
    }
#define SQ_LAUNCH(L2_, QG_, R_)                                                                    \
    do {                                                                                           \
        const size_t ybytes = (size_t)(QG_ + 2) * dpad * 4;                                        \
        const int qr = std::max(1, std::min(QG_, (int)(48 * 1024 / (SQ_WAVES * k * 12))));         \
        const size_t mbytes =                                                                      \
            (((size_t)qr * SQ_WAVES * k * 4 + 7) & ~(size_t)7) + (size_t)qr * SQ_WAVES * k * 8;    \
        const size_t sm = std::max(ybytes, mbytes);                                                \
        auto kern = sq_scan_kernel<L2_, QG_, R_>;                                                  \
        if (sm > 48 * 1024) {                                                                      \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); \
            if (e != hipSuccess) return e;                                                         \
        }                                                                                          \
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(SQ_THREADS), sm, s, a);                \
    } while (0)
#define SQ_BY_K(L2_)                                                                               \
    do {                                                                                           \
        if (qg_override == 1) { /* one query per item: the compact table of the MFMA prefilter's fallback */ \
            if (k <= 64) SQ_LAUNCH(L2_, 1, 1);                                                     \
            else if (k <= 128) SQ_LAUNCH(L2_, 1, 2);                                               \
            else if (k <= 256) SQ_LAUNCH(L2_, 1, 4);                                               \
            else if (k <= 512) SQ_LAUNCH(L2_, 1, 8);                                               \
            else SQ_LAUNCH(L2_, 1, 16);                                                            \
        } else                                                                                     \
        if (k <= 64) SQ_LAUNCH(L2_, 8, 1);                                                         \
        else if (k <= 128) SQ_LAUNCH(L2_, 8, 2);                                                   \
        else if (k <= 256) SQ_LAUNCH(L2_, 4, 4);                                                   \
        else if (k <= 512) SQ_LAUNCH(L2_, 2, 8);                                                   \
        else SQ_LAUNCH(L2_, 1, 16);                                                                \
    } while (0)
    if (is_l2) {
        SQ_BY_K(true);
    } else {
        SQ_BY_K(false);
    }
#undef SQ_BY_K
#undef SQ_LAUNCH
    return hipGetLastError();
}

hipError_t launch_sq_interleave(const uint8_t* codes, const int64_t* list_row_off,
                                const int64_t* list_len, const int64_t* list_blk_off, int64_t nlist,
                                int d, uint4* out, hipStream_t s) {
    if (nlist <= 0) {
        return hipSuccess;
    }
    const int nchunk16 = (d + 15) / 16;
    const unsigned gy = (unsigned)std::min<int64_t>(nlist, 32768);
    const unsigned gz = (unsigned)((nlist + gy - 1) / gy);
    hipLaunchKernelGGL(sq_interleave_kernel, dim3(8, gy, gz), dim3(256), 0, s, codes, list_row_off,
                       list_len, list_blk_off, nlist, d, nchunk16, out);
    return hipGetLastError();
}

} // namespace knhip
