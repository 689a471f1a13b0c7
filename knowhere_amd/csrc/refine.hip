// knowhere_amd/csrc/refine.hip -- exact re-rank of candidate ids against the raw fp32 vectors.
//
// Replaces faiss::IndexRefine::search's second stage (reference
// thirdparty/faiss/faiss/IndexRefine.cpp:104-140; Knowhere wraps IVF_PQ / IVF_SQ8 in it when
// `refine` is set: src/index/ivf/ivf.cc:1073-1103, src/index/refine/refine_utils.cc:99):
//   for each candidate label (in order, stopping at the first -1): dis = metric(q, base[label])
//   then the k best of the k_base re-scored candidates: reorder_2_heaps (utils/Heap.h:657) pushes them IN CANDIDATE ORDER
//   through a heap with strict-improve admission (Heap.cpp addn_with_ids -> heap_replace_top), so among candidates tied
//   at the k-th distance v the ones kept are decided by arrival: a tie is eligible iff it is among the first k
//   candidates with distance <= v (knhip_api.hip, search_batch_ties states the equivalence), and the result is the
//   canonical top-k of {better than v} U {eligible ties}.  Applied here whenever more than k candidates reach v.
// Distances use the reference's sequential fp32 order (fvec_L2sqr / fvec_inner_product), so they
// are bit-equal to the CPU refine.  288 GB of HBM3E holds the raw vectors of a 100M x 128 index
// (51 GB) next to its codes, so refine is a ~0.5 GB random gather per 10k-query batch: noise
// next to the scan, and what lifts PQ32 recall@10 past 0.95.
#include "common.h"
#include "kernels.h"

namespace knhip {

template <bool IS_L2, int R>
__global__ __launch_bounds__(256) void refine_kernel(const float* __restrict__ base, int64_t nbase,
                                                     int64_t id_base, int d,
                                                     const float* __restrict__ queries, int64_t nq,
                                                     const int64_t* __restrict__ cand, int kbase, int k,
                                                     float* __restrict__ out_d,
                                                     int64_t* __restrict__ out_i) {
    extern __shared__ __align__(16) float sq[]; // [4][d] queries, then [4][kbase] the candidates' distances
    const int lane = lane_id();
    const int wave = threadIdx.x / KN_WAVE;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    const bool live = q < nq;
    float* myq = sq + wave * d;
    float* mydis = sq + 4 * d + wave * kbase;
    if (live) {
        for (int i = lane; i < d; i += KN_WAVE) {
            myq[i] = queries[q * d + i];
        }
    }
    __syncthreads();
    if (!live) {
        return;
    }
    WaveTopK<IS_L2, R> top;
    top.init(k);
    float kd = worst_dist<IS_L2>();
    int64_t ki = -1;
    bool ended = false; // a -1 label ends the candidate list (IndexRefine.cpp:119-121)
    bool skipped = false; // a slot whose row lives on another shard: its distance is unknown here
    int nend = kbase;     // candidates before the end of the list
    for (int c0 = 0; c0 < kbase && !ended; c0 += KN_WAVE) {
        const int c = c0 + lane;
        int64_t id = -1;
        if (c < kbase) {
            id = cand[q * kbase + c];
        }
        // -1 ends the row; any other negative id is a skipped slot (sharded refine: not owned here)
        const unsigned long long neg = __ballot(c < kbase && id == -1);
        int nvalid = min(KN_WAVE, kbase - c0);
        if (neg) {
            nvalid = __ffsll((long long)neg) - 1;
            ended = true;
            nend = c0 + nvalid;
        }
        const bool ok = lane < nvalid && id >= 0 && (id - id_base) >= 0 && (id - id_base) < nbase;
        skipped = skipped || __ballot(lane < nvalid && !ok) != 0ull;
        float acc = 0.f;
        if (ok) {
            const float* y = base + (id - id_base) * d;
            int i = 0;
            if ((d & 3) == 0 && (reinterpret_cast<uintptr_t>(base) & 15) == 0) {
                // 16-byte loads, sixteen in flight per lane (every lane reads its own row: scalar loads made each of them a
                // 64-line request); the accumulation order stays i = 0, 1, 2, ... (reference order)
                const float4* y4 = reinterpret_cast<const float4*>(y);
                const float4* q4 = reinterpret_cast<const float4*>(myq);
                const int n4 = d >> 2;
                int j = 0;
                for (; j + 16 <= n4; j += 16) {
                    float4 v[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        v[u] = y4[j + u];
                    }
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        const float4 x = q4[j + u];
                        acc = IS_L2 ? l2_step(acc, x.x, v[u].x) : ip_step(acc, x.x, v[u].x);
                        acc = IS_L2 ? l2_step(acc, x.y, v[u].y) : ip_step(acc, x.y, v[u].y);
                        acc = IS_L2 ? l2_step(acc, x.z, v[u].z) : ip_step(acc, x.z, v[u].z);
                        acc = IS_L2 ? l2_step(acc, x.w, v[u].w) : ip_step(acc, x.w, v[u].w);
                    }
                }
                for (; j < n4; j++) { // (the last n4 mod 16 pieces)
                    const float4 v1 = y4[j];
                    const float4 x = q4[j];
                    acc = IS_L2 ? l2_step(acc, x.x, v1.x) : ip_step(acc, x.x, v1.x);
                    acc = IS_L2 ? l2_step(acc, x.y, v1.y) : ip_step(acc, x.y, v1.y);
                    acc = IS_L2 ? l2_step(acc, x.z, v1.z) : ip_step(acc, x.z, v1.z);
                    acc = IS_L2 ? l2_step(acc, x.w, v1.w) : ip_step(acc, x.w, v1.w);
                }
                i = j * 4;
            }
            for (; i < d; i++) {
                acc = IS_L2 ? l2_step(acc, myq[i], y[i]) : ip_step(acc, myq[i], y[i]);
            }
        }
        if (c < kbase) {
            mydis[c] = acc;
        }
        unsigned long long m = __ballot(ok && top.admits(acc, id, kd, ki));
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const float cd = readlane_f(acc, l);
            const int64_t ci = readlane_i64(id, l);
            if (top.admits(cd, ci, kd, ki)) {
                top.insert(cd, ci);
                kd = top.kth_dist();
                ki = top.kth_idx();
            }
        }
    }
    // ---- the reference's admission at the k-th boundary (see the header): only when a full list leaves tied candidates out
    if (ki >= 0 && !skipped) {
        const float v = kd;
        int nqual = 0;
        for (int c0 = 0; c0 < nend; c0 += KN_WAVE) {
            const int c = c0 + lane;
            const float dis = c < nend ? mydis[c] : 0.f;
            nqual += __popcll(__ballot(c < nend && (IS_L2 ? dis <= v : dis >= v)));
        }
        if (nqual > k) {
            top.init(k);
            kd = worst_dist<IS_L2>();
            ki = -1;
            int seen = 0;
            for (int c0 = 0; c0 < nend; c0 += KN_WAVE) {
                const int c = c0 + lane;
                const float dis = c < nend ? mydis[c] : 0.f;
                const int64_t id = c < nend ? cand[q * kbase + c] : -1;
                const bool qual = c < nend && (IS_L2 ? dis <= v : dis >= v);
                const unsigned long long qm = __ballot(qual);
                const int rank = seen + __popcll(qm & ((1ull << lane) - 1ull)); // arrivals with distance <= v before this one
                seen += __popcll(qm);
                const bool elig = qual && (dis != v || rank < k);
                unsigned long long m = __ballot(elig && top.admits(dis, id, kd, ki));
                while (m) {
                    const int l = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const float cd = readlane_f(dis, l);
                    const int64_t ci = readlane_i64(id, l);
                    if (top.admits(cd, ci, kd, ki)) {
                        top.insert(cd, ci);
                        kd = top.kth_dist();
                        ki = top.kth_idx();
                    }
                }
            }
        }
    }
    top.store(out_d + q * k, out_i + q * k);
}

hipError_t launch_refine(const float* base, int64_t nbase, int64_t id_base, int d, const float* queries,
                         int64_t nq, const int64_t* cand, int kbase, int k, bool is_l2, float* out_d,
                         int64_t* out_i, hipStream_t s) {
    if (nq <= 0) {
        return hipSuccess;
    }
    const unsigned grid = (unsigned)((nq + 3) / 4);
    const size_t sm = ((size_t)4 * d + (size_t)4 * kbase) * sizeof(float);
    KN_DISPATCH_R(k, {
        if (is_l2) {
            hipLaunchKernelGGL((refine_kernel<true, R_>), dim3(grid), dim3(256), sm, s, base, nbase, id_base,
                               d, queries, nq, cand, kbase, k, out_d, out_i);
        } else {
            hipLaunchKernelGGL((refine_kernel<false, R_>), dim3(grid), dim3(256), sm, s, base, nbase, id_base,
                               d, queries, nq, cand, kbase, k, out_d, out_i);
        }
    });
    return hipGetLastError();
}

} // namespace knhip
