// knowhere_amd/csrc/flat_scan.hip -- exact fp32 row scan for gfx950.
//
// Replaces, on the device:
//   * IVFFlatScanner::scan_codes          (reference thirdparty/faiss/faiss/cppcontrib/knowhere/
//                                          IndexIVFFlat.cpp:193-236; baseline IndexIVFFlat.cpp)
//   * exhaustive_L2sqr_seq / _inner_product_seq  (thirdparty/faiss/faiss/utils/distances.cpp:283-362)
//     used by FLAT / BruteForce and by the coarse quantizer
//   * fvec_L2sqr / fvec_inner_product     (src/simd/distances_ref.cc:21-37) -- same operation
//     order, one rounding per operation, so distances are bit-equal to the scalar reference.
//
// HBM layout ("row-interleaved"): rows are stored in blocks of 64; block b holds
//   float4 blk[nchunk][64]   with blk[c][r] = dims 4c..4c+3 of row 64b+r   (nchunk = ceil(d/4))
// so lane r of a wave reads its own row with perfectly coalesced 16-byte loads (1 KiB per
// wave instruction) and keeps a private, sequentially accumulated distance per query.
// A work item is (one list or base chunk) x (up to QG queries): the rows are read ONCE for the
// QG queries (queries that probe the same list are grouped by the host, see worktable.hip), so
// HBM/L2 traffic is 1/QG of the algorithmic bytes and the kernel is VALU-bound by design:
// 3 VALU ops per (row, query, dim) for L2, 2 for IP, nothing fused (exact mode).
//
// Roofline: algorithmic bytes per item = rows * d * 4 * npair (SURVEY.md 8d counts every
// (query, probe) pair's list bytes, no credit for reuse).
#include "common.h"
#include "kernels.h"

namespace knhip {

constexpr int FS_WAVES = 4;
constexpr int FS_THREADS = FS_WAVES * KN_WAVE;


// one work item (TABLE: `item_or_block` is the item when a.item_loop is set, else the block index that xcd_item
// maps to an item; DENSE: the block index).  Every exit is workgroup-uniform.
template <bool IS_L2, int QG, int R, bool DENSE>
__device__ __forceinline__ void flat_scan_item(const FlatScanArgs& a, const int64_t item_or_block, unsigned char* smem) {
    const int lane = lane_id();
    const int wave = threadIdx.x / KN_WAVE;
    const int dpad = a.nchunk * 4;

    // ---- decode the work item ----------------------------------------------------------
    int64_t item;
    int npair;
    int64_t blk0, len, row_off = 0;
    int32_t q_of[QG];
    int32_t slot_of[QG];
    int64_t row_base = 0; // DENSE: first row of the chunk
    if (DENSE) {
        item = item_or_block;
        if (item >= a.nitems_dense) {
            return;
        }
        const int64_t chunk = item / a.ngroups;
        const int64_t g = item % a.ngroups;
        row_base = chunk * a.chunk_rows;
        blk0 = row_base / 64;
        len = min(a.chunk_rows, a.nrows - row_base);
        npair = (int)min((int64_t)QG, a.nq - g * QG);
#pragma unroll
        for (int j = 0; j < QG; j++) {
            q_of[j] = (int32_t)min(g * QG + j, a.nq - 1);
            slot_of[j] = (int32_t)chunk;
        }
    } else {
        const int64_t nitems = *a.nitems_dev;
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
        npair = it.npair;
        blk0 = a.list_blk_off[it.list];
        len = a.list_len[it.list];
        row_off = a.list_row_off[it.list];
#pragma unroll
        for (int j = 0; j < QG; j++) {
            const KnPair p = a.pairs[it.pair0 + min(j, npair - 1)];
            q_of[j] = p.q;
            slot_of[j] = p.slot;
        }
    }

    // ---- stage the QG queries in LDS (zero padded to dpad) --------------------------------
    float* sq = reinterpret_cast<float*>(smem); // [QG][dpad]
    for (int t = threadIdx.x; t < QG * dpad; t += FS_THREADS) {
        const int j = t / dpad, i = t % dpad;
        sq[t] = (i < a.d) ? a.queries[(int64_t)q_of[j] * a.d + i] : 0.f;
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
    for (int64_t b = wave; b < nblk; b += FS_WAVES) {
        const int64_t row = b * 64 + lane; // row inside the list / chunk
#pragma unroll
        for (int j = 0; j < QG; j++) {
            gt[j] = gthr_load<IS_L2>(a.gthr + q_of[j]);
        }
        bool valid = row < len;
        int64_t id_for_filter = -1;
        if (a.bitset != nullptr && valid) {
            id_for_filter = DENSE ? (row_base + row + a.id_offset) : a.ids[row_off + row]; // (the bitset lives in the id domain)
            valid = !bitset_filtered(a.bitset, a.bitset_nbits, id_for_filter);
        }
        const float4* p = a.rows + (blk0 + b) * (int64_t)a.nchunk * 64 + lane;
        float acc[QG];
#pragma unroll
        for (int j = 0; j < QG; j++) {
            acc[j] = 0.f;
        }
        // (one- and two-query items keep 8 row loads in flight: with so little arithmetic per load the scan is
        // latency-bound otherwise)
        constexpr int UF = QG <= 2 ? 8 : 2;
#pragma unroll UF
        for (int c = 0; c < a.nchunk; c++) {
            const float4 y = p[(int64_t)c * 64];
#pragma unroll
            for (int j = 0; j < QG; j++) {
                const float4 q = *reinterpret_cast<const float4*>(sq + j * dpad + c * 4);
                if (IS_L2) {
                    acc[j] = l2_step(acc[j], q.x, y.x);
                    acc[j] = l2_step(acc[j], q.y, y.y);
                    acc[j] = l2_step(acc[j], q.z, y.z);
                    acc[j] = l2_step(acc[j], q.w, y.w);
                } else {
                    acc[j] = ip_step(acc[j], q.x, y.x);
                    acc[j] = ip_step(acc[j], q.y, y.y);
                    acc[j] = ip_step(acc[j], q.z, y.z);
                    acc[j] = ip_step(acc[j], q.w, y.w);
                }
            }
        }
        if (!IS_L2 && a.cos_mode != 0) { // COSINE with stored norms (FlatScanArgs)
            const float sc = a.row_scale[(blk0 + b) * 64 + lane];
#pragma unroll
            for (int j = 0; j < QG; j++) {
                acc[j] = cosine_finish(acc[j], sc, a.cos_mode);
            }
        }
        // ---- candidates: ties are ordered by the row position, which is the id order
        //      (lists are stored sorted by id; DENSE ids are row + offset) ----------------
#pragma unroll
        for (int j = 0; j < QG; j++) {
            if (j < npair) {
                bool pass = valid && within_gthr<IS_L2>(acc[j], gt[j]) &&
                            top[j].admits(acc[j], row, kd[j], ki[j]);
                unsigned long long m = __ballot(pass);
                bool tightened = false;
                while (m) {
                    const int l = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const float cd = readlane_f(acc[j], l);
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

    // ---- merge the 4 waves' lists (wave w merges queries j = w, w+4, ...) through LDS ----
    __syncthreads(); // queries in LDS no longer needed
    float* md = reinterpret_cast<float*>(smem);                         // [qr][WAVES][k]
    // keep the int64 array 8-byte aligned
    const int k = a.k;
    const int qr = max(1, min(QG, (int)(48 * 1024 / (FS_WAVES * k * 12))));
    int64_t* mi = reinterpret_cast<int64_t*>(smem + (((size_t)qr * FS_WAVES * k * 4 + 7) & ~(size_t)7));
    for (int j0 = 0; j0 < QG; j0 += qr) {
#pragma unroll
        for (int j = 0; j < QG; j++) {
            if (j >= j0 && j < j0 + qr) {
                top[j].store(md + ((j - j0) * FS_WAVES + wave) * k, mi + ((j - j0) * FS_WAVES + wave) * k);
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < QG; j++) {
            if (j >= j0 && j < j0 + qr && j < npair && (j % FS_WAVES) == wave) {
                // start from this wave's own list, fold in the other three
                for (int w = 1; w < FS_WAVES; w++) {
                    const int ow = (wave + w) % FS_WAVES;
                    const float* od = md + ((j - j0) * FS_WAVES + ow) * k;
                    const int64_t* oi = mi + ((j - j0) * FS_WAVES + ow) * k;
                    for (int e = 0; e < k; e++) {
                        const float cd = od[e];
                        const int64_t ci = oi[e];
                        if (ci < 0 || !top[j].admits(cd, ci, kd[j], ki[j])) {
                            break; // sorted best-first: nothing further can enter
                        }
                        top[j].insert(cd, ci);
                        kd[j] = top[j].kth_dist();
                        ki[j] = top[j].kth_idx();
                    }
                }
                // row position -> id, write the (query, slot) partial
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
                        int64_t pos = top[j].i[r];
                        int64_t id = -1;
                        if (pos >= 0) {
                            id = DENSE ? (row_base + pos + a.id_offset) : a.ids[row_off + pos];
                        }
                        pd[e] = top[j].d[r];
                        pi[e] = id;
                    }
                }
            }
        }
        __syncthreads();
    }
}

template <bool IS_L2, int QG, int R, bool DENSE>
__global__ __launch_bounds__(FS_THREADS) void flat_scan_kernel(FlatScanArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (!DENSE && a.item_loop) {
        // a fixed grid walks an item table whose size only the device knows (mfma_scan.hip fallback)
        const int64_t nitems = *a.nitems_dev;
        for (int64_t item = blockIdx.x; item < nitems; item += gridDim.x) {
            flat_scan_item<IS_L2, QG, R, DENSE>(a, item, smem);
            __syncthreads();
        }
        return;
    }
    flat_scan_item<IS_L2, QG, R, DENSE>(a, blockIdx.x, smem);
}

// ---- all-pairs exact distances (coarse quantizer, exact mode / fallback) ----------------------
// out[q][row] for every query and every row of a DENSE row set.  Same arithmetic as above.
template <bool IS_L2, int QG>
__global__ __launch_bounds__(FS_THREADS) void flat_full_kernel(FlatScanArgs a, float* out,
                                                               const int32_t* q_subset,
                                                               int64_t nq_subset,
                                                               const int32_t* row_flags) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = lane_id();
    const int wave = threadIdx.x / KN_WAVE;
    const int dpad = a.nchunk * 4;
    const int64_t nq = q_subset ? nq_subset : a.nq;
    const int64_t ngroups = (nq + QG - 1) / QG;
    // With row flags (the exact fallback of the MFMA prefilter) the grid is small and walks the items: row_flags[a.nq] is
    // the "any query flagged" summary -- normally 0, and the launch then costs a few hundred workgroups that return at once
    // instead of one per item (0.16 ms of the C3 coarse stage in the round-2 profile).
    if (row_flags != nullptr && row_flags[a.nq] == 0) {
        return;
    }
    const int64_t nitems = ngroups * ((a.nrows + a.chunk_rows - 1) / a.chunk_rows);
    for (int64_t item = blockIdx.x; item < nitems; item += gridDim.x) {
    __syncthreads(); // (the query tile in LDS is rewritten per item)
    const int64_t chunk = item / ngroups;
    const int64_t g = item % ngroups;
    const int64_t row_base = chunk * a.chunk_rows;
    const int64_t len = min(a.chunk_rows, a.nrows - row_base);
    const int npair = (int)min((int64_t)QG, nq - g * QG);
    int32_t q_of[QG];
#pragma unroll
    for (int j = 0; j < QG; j++) {
        const int64_t qi = min(g * QG + j, nq - 1);
        q_of[j] = q_subset ? q_subset[qi] : (int32_t)qi;
    }
    if (row_flags != nullptr) {
        // exact fallback of the MFMA prefilter: only queries whose certificate failed are recomputed
        bool any = false;
#pragma unroll
        for (int j = 0; j < QG; j++) {
            any |= (j < npair) && row_flags[q_of[j]] != 0;
        }
        if (!any) {
            continue;
        }
    }
    float* sq = reinterpret_cast<float*>(smem);
    for (int t = threadIdx.x; t < QG * dpad; t += FS_THREADS) {
        const int j = t / dpad, i = t % dpad;
        sq[t] = (i < a.d) ? a.queries[(int64_t)q_of[j] * a.d + i] : 0.f;
    }
    __syncthreads();
    const int64_t nblk = (len + 63) / 64;
    for (int64_t b = wave; b < nblk; b += FS_WAVES) {
        const int64_t row = b * 64 + lane;
        const float4* p = a.rows + (row_base / 64 + b) * (int64_t)a.nchunk * 64 + lane;
        float acc[QG];
#pragma unroll
        for (int j = 0; j < QG; j++) {
            acc[j] = 0.f;
        }
#pragma unroll 2
        for (int c = 0; c < a.nchunk; c++) {
            const float4 y = p[(int64_t)c * 64];
#pragma unroll
            for (int j = 0; j < QG; j++) {
                const float4 q = *reinterpret_cast<const float4*>(sq + j * dpad + c * 4);
                if (IS_L2) {
                    acc[j] = l2_step(acc[j], q.x, y.x);
                    acc[j] = l2_step(acc[j], q.y, y.y);
                    acc[j] = l2_step(acc[j], q.z, y.z);
                    acc[j] = l2_step(acc[j], q.w, y.w);
                } else {
                    acc[j] = ip_step(acc[j], q.x, y.x);
                    acc[j] = ip_step(acc[j], q.y, y.y);
                    acc[j] = ip_step(acc[j], q.z, y.z);
                    acc[j] = ip_step(acc[j], q.w, y.w);
                }
            }
        }
        if (!IS_L2 && a.cos_mode != 0 && row < len) {
            const float sc = a.row_scale[row_base + row];
#pragma unroll
            for (int j = 0; j < QG; j++) {
                acc[j] = cosine_finish(acc[j], sc, a.cos_mode);
            }
        }
        if (row < len) {
#pragma unroll
            for (int j = 0; j < QG; j++) {
                if (j < npair && (row_flags == nullptr || row_flags[q_of[j]] != 0)) {
                    out[(int64_t)q_of[j] * a.nrows + row_base + row] = acc[j];
                }
            }
        }
    }
    } // items
}

// ---- row-major [n][d] -> interleaved blocks -------------------------------------------------
// dst block index = dst_blk0 + (row / 64); rows beyond n in the last block are zero filled.
__global__ void interleave_rows_kernel(const float* __restrict__ src, int64_t n, int d, int nchunk,
                                       float4* __restrict__ dst, int64_t dst_blk0) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // one float4 slot each
    const int64_t nblk = (n + 63) / 64;
    const int64_t total = nblk * nchunk * 64;
    if (t >= total) {
        return;
    }
    const int64_t b = t / ((int64_t)nchunk * 64);
    const int rem = (int)(t % ((int64_t)nchunk * 64));
    const int c = rem / 64, r = rem % 64;
    const int64_t row = b * 64 + r;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (row < n) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int i = c * 4 + e;
            if (i < d) {
                v[e] = src[row * d + i];
            }
        }
    }
    dst[(dst_blk0 + b) * (int64_t)nchunk * 64 + rem] = make_float4(v[0], v[1], v[2], v[3]);
}

// ---- list-sorted row-major [ntotal][d] -> per-list interleaved blocks --------------------------
__global__ void interleave_lists_kernel(const float* __restrict__ src,
                                        const int64_t* __restrict__ list_row_off,
                                        const int64_t* __restrict__ list_len,
                                        const int64_t* __restrict__ list_blk_off, int64_t nlist, int d,
                                        int nchunk, float4* __restrict__ dst) {
    const int64_t l = blockIdx.y + (int64_t)blockIdx.z * gridDim.y;
    if (l >= nlist) {
        return;
    }
    const int64_t len = list_len[l];
    const int64_t nblk = (len + 63) / 64;
    const int64_t row_off = list_row_off[l];
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nblk * nchunk * 64;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / ((int64_t)nchunk * 64);
        const int rem = (int)(t % ((int64_t)nchunk * 64));
        const int c = rem / 64, r = rem % 64;
        const int64_t row = b * 64 + r;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (row < len) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int i = c * 4 + e;
                if (i < d) {
                    v[e] = src[(row_off + row) * d + i];
                }
            }
        }
        dst[(list_blk_off[l] + b) * (int64_t)nchunk * 64 + rem] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
static size_t flat_scan_smem(int QG, int dpad, int k) {
    const size_t qbytes = (size_t)QG * dpad * 4;
    const int qr = std::max(1, std::min(QG, (int)(48 * 1024 / (FS_WAVES * k * 12))));
    const size_t mbytes = (((size_t)qr * FS_WAVES * k * 4 + 7) & ~(size_t)7) + (size_t)qr * FS_WAVES * k * 8;
    return std::max(qbytes, mbytes);
}

int flat_scan_qg(int k) {
    return k <= 128 ? 8 : (k <= 256 ? 4 : (k <= 512 ? 2 : 1));
}

template <bool IS_L2, bool DENSE>
static hipError_t launch_flat_scan_t(const FlatScanArgs& a, int64_t grid, hipStream_t s, int qg_override) {
    const int dpad = a.nchunk * 4;
    const int k = a.k;
#define FS_LAUNCH(QG_, R_)                                                                         \
    do {                                                                                           \
        const size_t sm = flat_scan_smem(QG_, dpad, k);                                            \
        auto kern = flat_scan_kernel<IS_L2, QG_, R_, DENSE>;                                       \
        if (sm > 48 * 1024) {                                                                      \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); \
            if (e != hipSuccess) return e;                                                         \
        }                                                                                          \
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(FS_THREADS), sm, s, a);                \
    } while (0)
    if (!DENSE && qg_override == 1) { // one query per item: the compact table of the MFMA prefilter's fallback
        if (k <= 64) {
            FS_LAUNCH(1, 1);
        } else if (k <= 128) {
            FS_LAUNCH(1, 2);
        } else if (k <= 256) {
            FS_LAUNCH(1, 4);
        } else if (k <= 512) {
            FS_LAUNCH(1, 8);
        } else {
            FS_LAUNCH(1, 16);
        }
    } else if (k <= 64) {
        FS_LAUNCH(8, 1);
    } else if (k <= 128) {
        FS_LAUNCH(8, 2);
    } else if (k <= 256) {
        FS_LAUNCH(4, 4);
    } else if (k <= 512) {
        FS_LAUNCH(2, 8);
    } else {
        FS_LAUNCH(1, 16);
    }
#undef FS_LAUNCH
    return hipGetLastError();
}

hipError_t launch_flat_scan(const FlatScanArgs& a, bool is_l2, bool dense, int64_t grid,
                            hipStream_t s, int qg_override) {
    if (grid <= 0) {
        return hipSuccess;
    }
    if (!a.item_loop) {
        grid = (grid + 7) / 8 * 8; // (xcd_item spreads the items over the blocks [0, round_up(nitems, 8)))
    }
    if (is_l2) {
        return dense ? launch_flat_scan_t<true, true>(a, grid, s, 0)
                     : launch_flat_scan_t<true, false>(a, grid, s, qg_override);
    }
    return dense ? launch_flat_scan_t<false, true>(a, grid, s, 0)
                 : launch_flat_scan_t<false, false>(a, grid, s, qg_override);
}

hipError_t launch_flat_full(const FlatScanArgs& a, bool is_l2, float* out, const int32_t* q_subset,
                            int64_t nq_subset, const int32_t* row_flags, hipStream_t s) {
    constexpr int QG = 8;
    const int64_t nq = q_subset ? nq_subset : a.nq;
    if (nq <= 0 || a.nrows <= 0) {
        return hipSuccess;
    }
    const int64_t ngroups = (nq + QG - 1) / QG;
    const int64_t nchunks = (a.nrows + a.chunk_rows - 1) / a.chunk_rows;
    const size_t sm = (size_t)QG * a.nchunk * 4 * 4;
    auto kern = is_l2 ? flat_full_kernel<true, QG> : flat_full_kernel<false, QG>;
    if (sm > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) return e;
    }
    // (row_flags: [nq + 1], the last entry = any flag set; the grid then walks the items)
    const int64_t grid = row_flags != nullptr ? std::min<int64_t>(ngroups * nchunks, 512) : ngroups * nchunks;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(FS_THREADS), sm, s, a, out, q_subset, nq_subset, row_flags);
    return hipGetLastError();
}

hipError_t launch_interleave_rows(const float* src, int64_t n, int d, float4* dst, int64_t dst_blk0,
                                  hipStream_t s) {
    const int nchunk = (d + 3) / 4;
    const int64_t total = ((n + 63) / 64) * (int64_t)nchunk * 64;
    if (total == 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(interleave_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       src, n, d, nchunk, dst, dst_blk0);
    return hipGetLastError();
}

hipError_t launch_interleave_lists(const float* src, const int64_t* list_row_off,
                                   const int64_t* list_len, const int64_t* list_blk_off, int64_t nlist,
                                   int d, float4* dst, hipStream_t s) {
    if (nlist <= 0) {
        return hipSuccess;
    }
    const int nchunk = (d + 3) / 4;
    const unsigned gy = (unsigned)std::min<int64_t>(nlist, 32768);
    const unsigned gz = (unsigned)((nlist + gy - 1) / gy);
    hipLaunchKernelGGL(interleave_lists_kernel, dim3(8, gy, gz), dim3(256), 0, s, src, list_row_off,
                       list_len, list_blk_off, nlist, d, nchunk, dst);
    return hipGetLastError();
}

} // namespace knhip
