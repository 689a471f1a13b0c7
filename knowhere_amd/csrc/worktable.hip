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

// single workgroup: exclusive scans of pair counts and item counts over the lists
constexpr int WT_SCAN_THREADS = 1024;
__global__ __launch_bounds__(WT_SCAN_THREADS) void wt_scan_kernel(
        const int32_t* __restrict__ list_count, const int64_t* __restrict__ list_len, int64_t nlist,
        int64_t nreal, int qg0, int qg1, int64_t code_size, int64_t* list_pair_off, int64_t* list_item_off,
        int64_t* nitems, double* scan_bytes) {
    __shared__ int64_t s_pairs[WT_SCAN_THREADS];
    __shared__ int64_t s_items[WT_SCAN_THREADS];
    __shared__ double s_bytes[WT_SCAN_THREADS];
    const int tid = threadIdx.x;
    const int64_t per = (nlist + WT_SCAN_THREADS - 1) / WT_SCAN_THREADS;
    const int64_t l0 = (int64_t)tid * per;
    const int64_t l1 = min(l0 + per, nlist);
    __shared__ double s_bytes0[WT_SCAN_THREADS];
    int64_t np = 0, ni = 0;
    double nb = 0.0, nb0 = 0.0;
    for (int64_t l = l0; l < l1; l++) {
        const int64_t c = list_count[l];
        const int qg = l < nreal ? qg0 : qg1;
        np += c;
        ni += (c + qg - 1) / qg;
        const double b = (double)c * (double)list_len[l % nreal] * (double)code_size;
        nb += b;
        if (l < nreal) {
            nb0 += b; // rank-0 probes (virtual lists [0, nreal))
        }
    }
    s_bytes0[tid] = nb0;
    s_pairs[tid] = np;
    s_items[tid] = ni;
    s_bytes[tid] = nb;
    __syncthreads();
    // Hillis-Steele inclusive scan over the 1024 per-thread totals
    for (int off = 1; off < WT_SCAN_THREADS; off <<= 1) {
        int64_t a = 0, b = 0;
        double c = 0.0;
        if (tid >= off) {
            a = s_pairs[tid - off];
            b = s_items[tid - off];
            c = s_bytes[tid - off];
        }
        __syncthreads();
        s_pairs[tid] += a;
        s_items[tid] += b;
        s_bytes[tid] += c;
        __syncthreads();
    }
    int64_t pp = s_pairs[tid] - np; // exclusive prefix
    int64_t ii = s_items[tid] - ni;
    for (int64_t l = l0; l < l1; l++) {
        const int64_t c = list_count[l];
        const int qg = l < nreal ? qg0 : qg1;
        list_pair_off[l] = pp;
        list_item_off[l] = ii;
        pp += c;
        ii += (c + qg - 1) / qg;
    }
    if (tid == WT_SCAN_THREADS - 1) {
        list_pair_off[nlist] = s_pairs[tid];
        list_item_off[nlist] = s_items[tid];
        *nitems = s_items[tid];
        *scan_bytes += s_bytes[tid]; // accumulated across query batches; reset by the host
        double t0 = 0.0;
        for (int i = 0; i < WT_SCAN_THREADS; i++) {
            t0 += s_bytes0[i];
        }
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
