// knowhere_amd/csrc/worktable.hip -- turn the coarse quantizer's (query, probe-rank) -> list
// assignment into list-major work items.
//
// The reference walks probes query by query (thirdparty/faiss/faiss/IndexIVF.cpp:642-662: for
// each query, for ik in 0..nprobe: scan_one_list(keys[i*nprobe+ik], ...)).  On the GPU the same
// (query, list) scans are regrouped by LIST: all queries of the batch that probe list l become
// ceil(count_l / qg) work items of up to qg queries each, consecutive in the item array, so that
// a list's codes are fetched from HBM once and then served from L2 / shared between the qg
// queries of an item.  The per-(query, probe) results are written back to slot `ik` of the
// query's partial-result row, so the final merge is independent of this regrouping.
//
// Everything stays on the device: no host synchronisation between coarse search and scan.
#include "common.h"
#include "kernels.h"

namespace knhip {

__global__ void wt_zero_kernel(int32_t* list_count, int32_t* list_cursor, int64_t nlist,
                               double* scan_bytes, int64_t* nitems) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nlist) {
        list_count[t] = 0;
        list_cursor[t] = 0;
    }
    if (t == 0) {
        *nitems = 0;
    }
    (void)scan_bytes;
}

// Virtual list id: the rank-0 probe of every query (its closest list) goes to virtual lists
// [0, nlist), all other probes to [nlist, 2*nlist).  Items are emitted in virtual-list order, so
// the scans most likely to contain a query's true neighbours are dispatched first and publish a
// tight per-query threshold (common.h gthr_*) before the bulk of the probes run.  Pure
// scheduling: results do not depend on it.
// rank0_slot = -1: no split, every pair goes to [nlist, 2*nlist) (mfma_scan.hip scans all probes of a list together).
// cls != nullptr: the first class is given per pair instead (cls[t] >= 0: mfma_scan.hip's sample pairs).
__device__ __forceinline__ int64_t wt_vlist(int64_t key, int64_t t, int nprobe, int64_t nlist, int rank0_slot,
                                            const int32_t* cls) {
    const bool first = cls != nullptr ? cls[t] >= 0 : (t % nprobe) == rank0_slot;
    return key + (first ? 0 : nlist);
}

// Pairs whose list is empty on this device (always the case for the lists another rank owns when the index
// is list-sharded) produce no work item at all: their partial slot is just marked empty by wt_scatter_kernel.
__global__ void wt_count_kernel(const int64_t* __restrict__ keys, int64_t npairs, int nprobe,
                                int64_t nlist, const int64_t* __restrict__ list_len, int32_t* list_count,
                                int rank0_slot, const int32_t* __restrict__ cls) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npairs) {
        return;
    }
    const int64_t key = keys[t];
    if (key >= 0 && key < nlist && list_len[key] > 0) {
        atomicAdd(&list_count[wt_vlist(key, t, nprobe, nlist, rank0_slot, cls)], 1);
    }
}

// single workgroup: exclusive scans of pair counts and item counts over the lists.  Round 6: the lists are walked in chunks of
// 4096, four CONSECUTIVE lists per thread (16 bytes of counts per thread: coalesced), a wave scan by shuffles and 16 wave
// totals through LDS per chunk.  (Before: every thread walked its own run of nlist / 1024 lists -- each load instruction
// touched 1024 cache lines, twice per list: 0.31 ms at 2 x 16384 virtual lists, 0.40 ms at 2 x 65536.)
constexpr int WT_SCAN_THREADS = 1024;
constexpr int WT_SCAN_PER = 4;
__global__ __launch_bounds__(WT_SCAN_THREADS) void wt_scan_kernel(
        const int32_t* __restrict__ list_count, const int64_t* __restrict__ list_len, int64_t nlist,
        int64_t nreal, int qg0, int qg1, int64_t code_size, int64_t* list_pair_off, int64_t* list_item_off,
        int64_t* nitems, double* scan_bytes) {
    constexpr int NW = WT_SCAN_THREADS / KN_WAVE;
    __shared__ long long s_wp[NW], s_wi[NW];
    __shared__ double s_wb[NW], s_wb0[NW];
    const int tid = threadIdx.x, lane = tid & (KN_WAVE - 1), wave = tid / KN_WAVE;
    long long carry_p = 0, carry_i = 0; // (the same in every thread)
    double nb = 0.0, nb0 = 0.0;
    for (int64_t c0 = 0; c0 < nlist; c0 += (int64_t)WT_SCAN_THREADS * WT_SCAN_PER) {
        const int64_t l0 = c0 + (int64_t)tid * WT_SCAN_PER;
        long long p[WT_SCAN_PER], it[WT_SCAN_PER], tp = 0, ti = 0;
#pragma unroll
        for (int e = 0; e < WT_SCAN_PER; e++) {
            const int64_t l = l0 + e;
            p[e] = 0;
            it[e] = 0;
            if (l < nlist) {
                const long long c = list_count[l];
                const int qg = l < nreal ? qg0 : qg1;
                p[e] = c;
                it[e] = (c + qg - 1) / qg;
                const double b = (double)c * (double)list_len[l % nreal] * (double)code_size;
                nb += b;
                if (l < nreal) {
                    nb0 += b; // rank-0 probes (virtual lists [0, nreal))
                }
            }
            tp += p[e];
            ti += it[e];
        }
        long long ip = tp, ii = ti; // inclusive scan over the wave's lanes
        for (int d = 1; d < KN_WAVE; d <<= 1) {
            const long long up = __shfl_up(ip, d, KN_WAVE), ui = __shfl_up(ii, d, KN_WAVE);
            if (lane >= d) {
                ip += up;
                ii += ui;
            }
        }
        if (lane == KN_WAVE - 1) {
            s_wp[wave] = ip;
            s_wi[wave] = ii;
        }
        __syncthreads();
        long long wp = 0, wi = 0, cp = 0, ci = 0;
        for (int w = 0; w < NW; w++) {
            const long long a = s_wp[w], b = s_wi[w];
            if (w < wave) {
                wp += a;
                wi += b;
            }
            cp += a;
            ci += b;
        }
        long long pp = carry_p + wp + (ip - tp), qq = carry_i + wi + (ii - ti); // exclusive prefixes of the thread's first list
#pragma unroll
        for (int e = 0; e < WT_SCAN_PER; e++) {
            const int64_t l = l0 + e;
            if (l < nlist) {
                list_pair_off[l] = pp;
                list_item_off[l] = qq;
            }
            pp += p[e];
            qq += it[e];
        }
        carry_p += cp;
        carry_i += ci;
        __syncthreads(); // (the wave totals are rewritten by the next chunk)
    }
    // the byte counters: exact in double whatever the order (integers far below 2^53)
    for (int off = KN_WAVE / 2; off > 0; off >>= 1) {
        nb += __shfl_xor(nb, off, KN_WAVE);
        nb0 += __shfl_xor(nb0, off, KN_WAVE);
    }
    if (lane == 0) {
        s_wb[wave] = nb;
        s_wb0[wave] = nb0;
    }
    __syncthreads();
    if (tid == 0) {
        double t = 0.0, t0 = 0.0;
        for (int w = 0; w < NW; w++) {
            t += s_wb[w];
            t0 += s_wb0[w];
        }
        list_pair_off[nlist] = carry_p;
        list_item_off[nlist] = carry_i;
        *nitems = carry_i;
        scan_bytes[0] += t; // accumulated across query batches; reset by the host
        scan_bytes[1] += t0;
    }
}

__global__ void wt_scatter_kernel(const int64_t* __restrict__ keys, int64_t npairs, int nprobe,
                                  int64_t nlist, const int64_t* __restrict__ list_len,
                                  const int64_t* __restrict__ list_pair_off, int32_t* list_cursor, KnPair* pairs,
                                  int64_t* empty_mark, int k, int rank0_slot, const int32_t* __restrict__ cls) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npairs) {
        return;
    }
    const int64_t key = keys[t];
    if (key < 0 || key >= nlist || list_len[key] == 0) {
        if (empty_mark != nullptr) {
            empty_mark[t * k] = -1; // sentinel-terminated partial list of (query, slot) t: empty
        }
        return;
    }
    const int64_t vl = wt_vlist(key, t, nprobe, nlist, rank0_slot, cls);
    const int64_t pos = list_pair_off[vl] + atomicAdd(&list_cursor[vl], 1);
    KnPair p;
    p.q = (int32_t)(t / nprobe);
    p.slot = (int32_t)(t % nprobe);
    pairs[pos] = p;
}

__global__ void wt_items_kernel(const int32_t* __restrict__ list_count,
                                const int64_t* __restrict__ list_pair_off,
                                const int64_t* __restrict__ list_item_off, int64_t nlist,
                                int64_t nreal, int qg0, int qg1, KnItem* items) {
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nlist) {
        return;
    }
    const int qg = l < nreal ? qg0 : qg1;
    const int64_t c = list_count[l];
    const int64_t p0 = list_pair_off[l];
    int64_t it = list_item_off[l];
    for (int64_t i = 0; i < c; i += qg, it++) {
        KnItem x;
        x.list = (int32_t)(l % nreal);
        x.npair = (int32_t)min((int64_t)qg, c - i);
        x.pair0 = p0 + i;
        items[it] = x;
    }
}

// qg0: queries per item of the rank-0 virtual lists, qg1: of all other probes (the two phases of the IVF-PQ
// search may run different kernels)
hipError_t launch_build_worktable(const int64_t* keys, int64_t nq, int nprobe, int64_t nlist, int qg0, int qg1,
                                  const int64_t* list_len, int64_t code_size, const WorkTable& wt,
                                  hipStream_t s, int rank0_slot, const int32_t* cls) {
    const int64_t npairs = nq * nprobe;
    const int64_t nvl = 2 * nlist; // virtual lists, see wt_vlist
    const unsigned gl = (unsigned)((nvl + 255) / 256);
    const unsigned gp = (unsigned)((npairs + 255) / 256);
    hipLaunchKernelGGL(wt_zero_kernel, dim3(gl), dim3(256), 0, s, wt.list_count, wt.list_cursor, nvl,
                       wt.scan_bytes, wt.nitems);
    if (npairs > 0) {
        hipLaunchKernelGGL(wt_count_kernel, dim3(gp), dim3(256), 0, s, keys, npairs, nprobe, nlist, list_len,
                           wt.list_count, rank0_slot, cls);
    }
    hipLaunchKernelGGL(wt_scan_kernel, dim3(1), dim3(WT_SCAN_THREADS), 0, s, wt.list_count, list_len,
                       nvl, nlist, qg0, qg1, code_size, wt.list_pair_off, wt.list_item_off, wt.nitems,
                       wt.scan_bytes);
    if (npairs > 0) {
        hipLaunchKernelGGL(wt_scatter_kernel, dim3(gp), dim3(256), 0, s, keys, npairs, nprobe, nlist, list_len,
                           wt.list_pair_off, wt.list_cursor, wt.pairs, wt.empty_mark, wt.k, rank0_slot, cls);
    }
    hipLaunchKernelGGL(wt_items_kernel, dim3(gl), dim3(256), 0, s, wt.list_count, wt.list_pair_off,
                       wt.list_item_off, nvl, nlist, qg0, qg1, wt.items);
    return hipGetLastError();
}

// ---- a handful of pairs: one item per pair, no table ------------------------------------------------------------------------
// The grouping above costs five launches, one of them a single workgroup walking 2 nlist virtual lists (0.11 ms at nlist
// 16384): too much for the dump pass of ONE or two queries (the boundary rule's flagged queries, a small RangeSearch).
// thread per (query, slot): the pairs of non-empty lists become items {list, 1 pair}, in atomic order (a dump does not care)
__global__ void wt_direct_items_kernel(const int64_t* __restrict__ keys, int64_t npairs, int nprobe, int64_t nlist,
                                       const int64_t* __restrict__ list_len, KnItem* __restrict__ items,
                                       KnPair* __restrict__ pairs, int64_t* __restrict__ nitems) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npairs) {
        return;
    }
    const int64_t key = keys[t];
    if (key < 0 || key >= nlist || list_len[key] == 0) {
        return;
    }
    const int64_t pos = (int64_t)atomicAdd(reinterpret_cast<unsigned long long*>(nitems), 1ull);
    KnPair p;
    p.q = (int32_t)(t / nprobe);
    p.slot = (int32_t)(t % nprobe);
    pairs[pos] = p;
    KnItem it;
    it.list = (int32_t)key;
    it.npair = 1;
    it.pair0 = pos;
    items[pos] = it;
}

hipError_t launch_direct_items(const int64_t* keys, int64_t nq, int nprobe, int64_t nlist, const int64_t* list_len,
                               const WorkTable& wt, hipStream_t s) {
    hipError_t e = hipMemsetAsync(wt.nitems, 0, sizeof(int64_t), s);
    const int64_t npairs = nq * nprobe;
    if (e != hipSuccess || npairs <= 0) {
        return e;
    }
    hipLaunchKernelGGL(wt_direct_items_kernel, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, s, keys, npairs, nprobe,
                       nlist, list_len, wt.items, wt.pairs, wt.nitems);
    return hipGetLastError();
}

__global__ void fill_f32_kernel(float* p, int64_t n, float v) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        p[t] = v;
    }
}

hipError_t launch_fill_f32(float* p, int64_t n, float v, hipStream_t s) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(fill_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, v);
    return hipGetLastError();
}

} // namespace knhip
