// knowhere_amd/csrc/range.hip -- range search epilogue: distance matrix -> (lims, ids, distances).
//
// Reference semantics (IvfIndexNode::RangeSearch, src/index/ivf/ivf.cc:1231-1420 ->
// IndexIVF::range_search_preassigned, thirdparty/faiss/faiss/IndexIVF.cpp:812-990, parallel_mode 0):
// lists are visited in coarse order; a vector is reported when C::cmp(radius, dis) (L2: dis < radius, IP:
// dis > radius) and the id selector admits it; after each list the count of consecutive lists that added
// nothing is bumped or reset, and the loop stops once it reaches max_empty_result_buckets (> 0).  Results
// of one query are emitted list by list in that order, storage order inside a list.
//
// The scan kernels (flat_full / pq_scan_v2 dump mode / pq_adc_dump below) have already written every distance of
// every probed list to dist[q][column]; what is left is data-parallel bookkeeping:
//   range_count  : hits per (query, probe rank)                      one workgroup per pair
//   range_plan   : early-stop cut + running offsets per query       one thread per query (nprobe steps)
//   range_emit   : ordered compaction of the surviving lists         one workgroup per pair
//
// Rank waves (IVF kinds with max_empty_result_buckets > 0): range search probes all nlist lists, but the reference stops
// a query after max_empty consecutive lists without a hit -- usually a few dozen ranks in.  The probes are therefore
// scanned in waves of coarse ranks [r0, r1) (64 ranks first, doubling): range_wave_gather compacts the wave's lists of
// the queries still running, the scan kernels dump only those, range_count counts them, range_wave_state advances each
// query's run of empty lists exactly as the plan kernel will.  The cost follows the early stop instead of nlist.
#include "common.h"
#include "kernels.h"

namespace knhip {

constexpr int RG_THREADS = 256;

template <bool IS_L2>
__device__ __forceinline__ bool range_hit(const RangeArgs& a, int64_t q, int64_t col, int64_t id_pos, float* dis) {
    const float v = a.dist[q * a.ncol + col];
    *dis = v;
    const float rad = a.radius_q ? a.radius_q[q] : a.radius;
    if (a.inclusive) {
        if (!(IS_L2 ? (v <= rad) : (v >= rad))) {
            return false;
        }
    } else if (!(IS_L2 ? (v < rad) : (v > rad))) {
        return false;
    }
    const int64_t id = a.ids ? a.ids[id_pos] : id_pos + a.id_offset;
    return !bitset_filtered(a.bitset, a.bitset_nbits, id);
}

// segment of (q, rank): columns [col0, col0 + len), ids at [idp0, idp0 + len)
__device__ __forceinline__ bool range_segment(const RangeArgs& a, int64_t q, int rank, int64_t* col0, int64_t* idp0,
                                              int64_t* len) {
    const int64_t key = a.order ? a.order[q * a.nprobe + rank] : rank;
    if (key < 0) {
        return false;
    }
    *col0 = a.seg_col[key];
    *idp0 = a.seg_idpos[key];
    *len = a.seg_len[key];
    return *len > 0;
}

// (rank0, nrank): the ranks counted by this launch -- all of them, or one wave
template <bool IS_L2>
__global__ __launch_bounds__(RG_THREADS) void range_count_kernel(RangeArgs a, int32_t* __restrict__ cnt, int rank0,
                                                                 int nrank, const int32_t* __restrict__ qstate) {
    const int64_t q = blockIdx.x / nrank;
    const int rank = rank0 + (int)(blockIdx.x % nrank);
    if (qstate != nullptr && qstate[q * 2 + 1] != 0) {
        return; // (stopped before this wave: its lists were not scanned; cnt stays 0 behind the cut)
    }
    __shared__ int s_tot;
    if (threadIdx.x == 0) {
        s_tot = 0;
    }
    __syncthreads();
    int64_t col0, idp0, len;
    int mine = 0;
    if (range_segment(a, q, rank, &col0, &idp0, &len)) {
        for (int64_t i = threadIdx.x; i < len; i += RG_THREADS) {
            float dis;
            mine += range_hit<IS_L2>(a, q, col0 + i, idp0 + i, &dis) ? 1 : 0;
        }
    }
    // wave reduce, then one LDS atomic per wave
    for (int off = KN_WAVE / 2; off > 0; off >>= 1) {
        mine += __shfl_down(mine, off, KN_WAVE);
    }
    if (lane_id() == 0 && mine) {
        atomicAdd(&s_tot, mine);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        cnt[q * a.nprobe + rank] = s_tot;
    }
}

// keys_w[q][j] = list at rank r0 + j of query q (-1: the query has stopped, or the rank does not exist), cdis_w alike
__global__ void range_wave_gather_kernel(const int64_t* __restrict__ keys, const float* __restrict__ cdis, int64_t nq,
                                         int nprobe, int r0, int W, const int32_t* __restrict__ qstate,
                                         int64_t* __restrict__ keys_w, float* __restrict__ cdis_w) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nq * W) {
        return;
    }
    const int64_t q = t / W;
    const int r = r0 + (int)(t % W);
    const bool live = r < nprobe && qstate[q * 2 + 1] == 0;
    keys_w[t] = live ? keys[q * nprobe + r] : -1;
    cdis_w[t] = live ? cdis[q * nprobe + r] : 0.f;
}

// qstate[q] = {consecutive empty lists so far, stopped}: the recurrence of range_plan_kernel over ranks [r0, r1);
// *alive = queries still running after the wave
__global__ void range_wave_state_kernel(const int32_t* __restrict__ cnt, int64_t nq, int nprobe, int r0, int r1,
                                        int max_empty, int32_t* __restrict__ qstate, int32_t* __restrict__ alive) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) {
        return;
    }
    int ndup = qstate[q * 2];
    bool stopped = qstate[q * 2 + 1] != 0;
    for (int r = r0; r < r1 && !stopped; r++) {
        const int c = cnt[q * nprobe + r];
        ndup = c == 0 ? ndup + 1 : 0;
        stopped = ndup >= max_empty;
    }
    qstate[q * 2] = ndup;
    qstate[q * 2 + 1] = stopped ? 1 : 0;
    if (!stopped) {
        atomicAdd(alive, 1);
    }
}

// ---- IVF-Flat: exact distances of the wave's lists -> dist[q][column] (the arithmetic of flat_full_kernel, one
// workgroup per (query, rank of the wave); column = padded position of the row in the interleaved store) --------------
template <bool IS_L2>
__global__ __launch_bounds__(RG_THREADS) void range_flat_dump_kernel(FlatScanArgs a, const int64_t* __restrict__ keys_w,
                                                                     int W, int64_t nlist,
                                                                     const int64_t* __restrict__ seg_col,
                                                                     const int64_t* __restrict__ seg_len,
                                                                     float* __restrict__ dist, int64_t ncol) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int64_t q = blockIdx.x / W;
    const int64_t key = keys_w[blockIdx.x];
    if (key < 0 || key >= nlist) {
        return;
    }
    const int64_t len = seg_len[key];
    if (len <= 0) {
        return;
    }
    const int64_t col0 = seg_col[key]; // (a multiple of 64: lists start on a block)
    const int dpad = a.nchunk * 4;
    float* sq = reinterpret_cast<float*>(smem);
    for (int i = threadIdx.x; i < dpad; i += RG_THREADS) {
        sq[i] = (i < a.d) ? a.queries[q * a.d + i] : 0.f;
    }
    __syncthreads();
    const int lane = lane_id();
    const int wave = threadIdx.x / KN_WAVE;
    const int64_t nblk = (len + 63) / 64;
    for (int64_t b = wave; b < nblk; b += RG_THREADS / KN_WAVE) {
        const int64_t row = b * 64 + lane;
        const float4* p = a.rows + (col0 / 64 + b) * (int64_t)a.nchunk * 64 + lane;
        float acc = 0.f;
#pragma unroll 4
        for (int c = 0; c < a.nchunk; c++) {
            const float4 y = p[(int64_t)c * 64];
            const float4 x = *reinterpret_cast<const float4*>(sq + c * 4);
            if (IS_L2) {
                acc = l2_step(acc, x.x, y.x);
                acc = l2_step(acc, x.y, y.y);
                acc = l2_step(acc, x.z, y.z);
                acc = l2_step(acc, x.w, y.w);
            } else {
                acc = ip_step(acc, x.x, y.x);
                acc = ip_step(acc, x.y, y.y);
                acc = ip_step(acc, x.z, y.z);
                acc = ip_step(acc, x.w, y.w);
            }
        }
        if (row < len) {
            if (!IS_L2 && a.cos_mode != 0) {
                acc = cosine_finish(acc, a.row_scale[col0 + row], a.cos_mode);
            }
            dist[q * ncol + col0 + row] = acc;
        }
    }
}

// off[q][rank] = running offset of the rank's hits inside the query's result, or -1 behind the early stop
__global__ void range_plan_kernel(const int32_t* __restrict__ cnt, int64_t nq, int nprobe, int max_empty,
                                  int64_t* __restrict__ off, int64_t* __restrict__ total) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) {
        return;
    }
    int64_t run = 0;
    int ndup = 0;
    bool stopped = false;
    for (int r = 0; r < nprobe; r++) {
        if (stopped) {
            off[q * nprobe + r] = -1;
            continue;
        }
        const int c = cnt[q * nprobe + r];
        off[q * nprobe + r] = run;
        run += c;
        if (max_empty > 0) {
            ndup = c == 0 ? ndup + 1 : 0;
            stopped = ndup >= max_empty;
        }
    }
    total[q] = run;
}

// cap > 0: only the first `cap` hits of a query are kept, at out[q * cap + position] (qbase unused): the tie rule needs the
// first k arrivals only
template <bool IS_L2>
__global__ __launch_bounds__(RG_THREADS) void range_emit_kernel(RangeArgs a, const int64_t* __restrict__ off,
                                                                const int64_t* __restrict__ qbase,
                                                                int64_t* __restrict__ out_ids,
                                                                float* __restrict__ out_dis, int64_t cap,
                                                                int64_t* __restrict__ out_key, int64_t key_base) {
    const int64_t q = blockIdx.x / a.nprobe;
    const int rank = (int)(blockIdx.x % a.nprobe);
    const int64_t o = off[blockIdx.x];
    int64_t col0, idp0, len;
    if (o < 0 || (cap > 0 && o >= cap) || !range_segment(a, q, rank, &col0, &idp0, &len)) {
        return;
    }
    __shared__ int s_wave[RG_THREADS / KN_WAVE];
    __shared__ int64_t s_base;
    const int lane = lane_id(), wave = threadIdx.x / KN_WAVE;
    const int64_t qb0 = cap > 0 ? q * cap : qbase[q];
    if (threadIdx.x == 0) {
        s_base = qb0 + o;
    }
    __syncthreads();
    for (int64_t i0 = 0; i0 < len; i0 += RG_THREADS) {
        const int64_t i = i0 + threadIdx.x;
        float dis = 0.f;
        const bool hit = i < len && range_hit<IS_L2>(a, q, col0 + i, idp0 + i, &dis);
        const unsigned long long m = __ballot(hit);
        if (lane == 0) {
            s_wave[wave] = __popcll(m);
        }
        __syncthreads();
        int before = __popcll(m & ((1ull << lane) - 1ull)), tot = 0;
        for (int w = 0; w < RG_THREADS / KN_WAVE; w++) {
            if (w < wave) {
                before += s_wave[w];
            }
            tot += s_wave[w];
        }
        const int64_t base = s_base;
        if (hit && (cap <= 0 || base + before - qb0 < cap)) {
            out_ids[base + before] = a.ids ? a.ids[idp0 + i] : idp0 + i + a.id_offset;
            out_dis[base + before] = dis;
            if (out_key != nullptr) {
                // the arrival's place in the reference's scan order, comparable ACROSS shards: (probe rank, position in the
                // list) for the IVF kinds -- every shard ranks the lists alike --, the row number for brute force
                // (key_base = the shard's first row)
                out_key[base + before] = a.order != nullptr ? (((int64_t)rank << 40) | i) : key_base + idp0 + i;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            s_base = base + tot;
        }
        __syncthreads();
    }
}

// ---- IVF-PQ, any M x 8 bit: every exact ADC distance of every probed list -> dist[q][list_row_off + position] -------
// The fast ADC kernels (stream16 layouts, M = 32) have a dump mode of their own; this plain kernel serves the other
// code widths.  One workgroup per (query, probe): the (query, list) table in LDS, then thread per stored vector, the
// table entries summed from 0 in m order and the coarse term added last -- PQCodeDistanceScalar / scan_list_with_table
// (thirdparty/faiss/faiss/impl/pq_code_distance/pq_code_distance-inl.h:69-90, IVFPQScanner_impl.h:109-181) with the
// tables of IVFPQ_QueryTables.cpp:110-230 (same arithmetic as pq_scan_q4.hip::p4_build_lut).
template <bool IS_L2>
__global__ __launch_bounds__(RG_THREADS) void pq_adc_dump_kernel(PqDumpArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* lut = reinterpret_cast<float*>(smem); // [M][256]
    const int64_t q = blockIdx.x / a.nprobe;
    const int slot = (int)(blockIdx.x % a.nprobe);
    const int64_t list = a.keys[q * a.nprobe + slot];
    if (list < 0 || list >= a.nlist) {
        return;
    }
    const int64_t len = a.list_len[list];
    if (len <= 0) {
        return;
    }
    const int M = a.M, dsub = a.d / a.M;
    for (int e = threadIdx.x; e < M * 256; e += RG_THREADS) {
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
    float* out = a.dist + q * a.ncol + row_off;
    for (int64_t pos = threadIdx.x; pos < len; pos += RG_THREADS) {
        const uint8_t* code = a.codes + (row_off + pos) * M;
        float acc = 0.f;
        for (int m = 0; m < M; m++) {
            acc = fadd_x(acc, lut[m * 256 + code[m]]);
        }
        out[pos] = fadd_x(dis0, acc);
    }
}

hipError_t launch_pq_adc_dump(const PqDumpArgs& a, int64_t nq, bool is_l2, hipStream_t s) {
    if (nq <= 0 || a.nprobe <= 0) {
        return hipSuccess;
    }
    if (a.M <= 0 || a.M > 128 || a.d % a.M != 0) {
        return hipErrorInvalidValue;
    }
    const size_t sm = (size_t)a.M * 256 * sizeof(float);
    auto kern = is_l2 ? pq_adc_dump_kernel<true> : pq_adc_dump_kernel<false>;
    if (sm > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)sm);
        if (e != hipSuccess) {
            return e;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(nq * a.nprobe)), dim3(RG_THREADS), sm, s, a);
    return hipGetLastError();
}

hipError_t launch_range_count(const RangeArgs& a, int64_t nq, bool is_l2, int32_t* cnt, hipStream_t s, int rank0,
                              int nrank, const int32_t* qstate) {
    if (nrank < 0) {
        rank0 = 0;
        nrank = a.nprobe;
    }
    nrank = std::min(nrank, a.nprobe - rank0);
    if (nq <= 0 || nrank <= 0) {
        return hipSuccess;
    }
    const unsigned grid = (unsigned)(nq * nrank);
    if (is_l2) {
        hipLaunchKernelGGL((range_count_kernel<true>), dim3(grid), dim3(RG_THREADS), 0, s, a, cnt, rank0, nrank, qstate);
    } else {
        hipLaunchKernelGGL((range_count_kernel<false>), dim3(grid), dim3(RG_THREADS), 0, s, a, cnt, rank0, nrank, qstate);
    }
    return hipGetLastError();
}

hipError_t launch_range_wave_gather(const int64_t* keys, const float* cdis, int64_t nq, int nprobe, int r0, int W,
                                    const int32_t* qstate, int64_t* keys_w, float* cdis_w, hipStream_t s) {
    if (nq <= 0 || W <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(range_wave_gather_kernel, dim3((unsigned)((nq * W + 255) / 256)), dim3(256), 0, s, keys, cdis, nq,
                       nprobe, r0, W, qstate, keys_w, cdis_w);
    return hipGetLastError();
}

hipError_t launch_range_wave_state(const int32_t* cnt, int64_t nq, int nprobe, int r0, int r1, int max_empty,
                                   int32_t* qstate, int32_t* alive, hipStream_t s) {
    if (nq <= 0) {
        return hipSuccess;
    }
    hipError_t e = hipMemsetAsync(alive, 0, sizeof(int32_t), s);
    if (e != hipSuccess) {
        return e;
    }
    hipLaunchKernelGGL(range_wave_state_kernel, dim3((unsigned)((nq + 63) / 64)), dim3(64), 0, s, cnt, nq, nprobe, r0,
                       std::min(r1, nprobe), max_empty, qstate, alive);
    return hipGetLastError();
}

hipError_t launch_range_flat_dump(const FlatScanArgs& a, const int64_t* keys_w, int64_t nq, int W, int64_t nlist,
                                  const int64_t* seg_col, const int64_t* seg_len, float* dist, int64_t ncol, bool is_l2,
                                  hipStream_t s) {
    if (nq <= 0 || W <= 0) {
        return hipSuccess;
    }
    const size_t sm = (size_t)a.nchunk * 4 * sizeof(float);
    auto kern = is_l2 ? range_flat_dump_kernel<true> : range_flat_dump_kernel<false>;
    if (sm > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) {
            return e;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(nq * W)), dim3(RG_THREADS), sm, s, a, keys_w, W, nlist, seg_col, seg_len,
                       dist, ncol);
    return hipGetLastError();
}

hipError_t launch_range_plan(const int32_t* cnt, int64_t nq, int nprobe, int max_empty, int64_t* off, int64_t* total,
                             hipStream_t s) {
    if (nq <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(range_plan_kernel, dim3((unsigned)((nq + 63) / 64)), dim3(64), 0, s, cnt, nq, nprobe, max_empty,
                       off, total);
    return hipGetLastError();
}

// ---- k-th-boundary ties: the search ran with kk = k + 1 results per query; the (k + 1)-th tells whether the k-th distance
// is shared by a candidate that did not make the canonical top-k -- only then can the reference's first-come admission
// (ResultHandler.h:258-279, Heap.h:113-151) differ from the canonical answer, and only those queries are resolved
__global__ void tie_detect_kernel(const float* __restrict__ d, const int64_t* __restrict__ i, int64_t nq, int k,
                                  float* __restrict__ out_d, int64_t* __restrict__ out_i, int32_t* __restrict__ flagged,
                                  int32_t* __restrict__ nflag) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int kk = k + 1;
    if (t < nq * k) {
        const int64_t q = t / k;
        const int j = (int)(t % k);
        out_d[t] = d[q * kk + j];
        out_i[t] = i[q * kk + j];
    }
    if (t < nq) {
        const bool tie = i[t * kk + k] >= 0 && i[t * kk + k - 1] >= 0 &&
                         __float_as_uint(d[t * kk + k]) == __float_as_uint(d[t * kk + k - 1]);
        if (tie) {
            flagged[atomicAdd(nflag, 1)] = (int32_t)t;
        }
    }
}

hipError_t launch_tie_detect(const float* d, const int64_t* i, int64_t nq, int k, float* out_d, int64_t* out_i,
                             int32_t* flagged, int32_t* nflag, hipStream_t s) {
    if (nq <= 0) {
        return hipSuccess;
    }
    const int64_t n = nq * (int64_t)k;
    hipLaunchKernelGGL(tie_detect_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d, i, nq, k, out_d, out_i,
                       flagged, nflag);
    return hipGetLastError();
}

// flagged[] comes out of tie_detect in atomic order; the shards of a group must walk the same list in the same order:
// ascending (the entries are distinct query numbers: an entry's place = how many are smaller)
__global__ void tie_sort_flags_kernel(const int32_t* __restrict__ in, int n, int32_t* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) {
        return;
    }
    const int32_t v = in[t];
    int r = 0;
    for (int o = 0; o < n; o++) {
        r += in[o] < v ? 1 : 0;
    }
    out[r] = v;
}

hipError_t launch_tie_sort_flags(const int32_t* in, int n, int32_t* out, hipStream_t s) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(tie_sort_flags_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, n, out);
    return hipGetLastError();
}

// sharded refine: every candidate's distance from the one shard that holds its row (parts [nsh][n]; REFINE_NOT_HERE = the
// all-ones pattern elsewhere, kept when no shard holds it)
__global__ void refine_combine_kernel(const uint32_t* __restrict__ parts, int nsh, int64_t n, uint32_t* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) {
        return;
    }
    uint32_t v = 0xffffffffu;
    for (int sh = 0; sh < nsh; sh++) {
        const uint32_t x = parts[(int64_t)sh * n + t];
        v = x != 0xffffffffu ? x : v;
    }
    out[t] = v;
}

hipError_t launch_refine_combine(const float* parts, int nsh, int64_t n, float* out, hipStream_t s) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(refine_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const uint32_t*>(parts), nsh, n, reinterpret_cast<uint32_t*>(out));
    return hipGetLastError();
}

// ---- k-th-boundary ties, resolved on the device (knhip_api.hip, search_batch_ties) -----------------------------------------
// gather the flagged queries (rows of the batch's queries, coarse keys and coarse distances) into dense arrays
__global__ void tie_gather_kernel(const int32_t* __restrict__ flagged, int nflag, const float* __restrict__ q, int d,
                                  const int64_t* __restrict__ keys, const float* __restrict__ cdis, int nprobe,
                                  float* __restrict__ q_out, int64_t* __restrict__ keys_out, float* __restrict__ cdis_out,
                                  const float* __restrict__ can_d, int k, float* __restrict__ rad_out) {
    const int f = blockIdx.x;
    if (f >= nflag) {
        return;
    }
    const int64_t src = flagged[f];
    if (threadIdx.x == 0) {
        rad_out[f] = can_d[src * (k + 1) + k - 1]; // v: the query's k-th distance = its (inclusive) radius
    }
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
        q_out[(int64_t)f * d + i] = q[src * d + i];
    }
    if (keys != nullptr) {
        for (int i = threadIdx.x; i < nprobe; i += blockDim.x) {
            keys_out[(int64_t)f * nprobe + i] = keys[src * nprobe + i];
            cdis_out[(int64_t)f * nprobe + i] = cdis[src * nprobe + i];
        }
    }
}

// One workgroup per flagged query.  Shard s (nsh = 1: the index itself) reports arr_*[s][f][0 .. min(k, arr_n[s][f])) = ITS
// first arrivals with distance <= v (>= v for IP) in the reference's scan order (range_count / range_plan / range_emit over
// the dump of the probed lists it holds, capped at k), v = the canonical k-th distance, each with its place in the GLOBAL
// scan order (arr_key: (probe rank, position) -- every shard ranks the lists alike -- or the row number for brute force).
// The first k arrivals overall are among these (a shard's (k + 1)-th arrival has k earlier ones in its own shard), so:
//   eligible tie = an arrival with distance v that fewer than k arrivals precede (by key, over all shards);
//   result = canonical top-k of {canonical entries better than v} U {eligible ties}
// -- the closed form of the reference's heap (tests/test_tie_rule.py) --, written over the query's output row.  With one
// shard the arrivals are in order already (arr_key may be null).  The reference's own sharding contract is ids-equal
// (tests/ut/test_bruteforce.cc:128-181): this is what makes a list-sharded search return the single index's answer.
template <bool IS_L2>
__global__ __launch_bounds__(RG_THREADS) void tie_resolve_kernel(const int32_t* __restrict__ flagged, int nflag, int nsh,
                                                                 const float* __restrict__ can_d,
                                                                 const int64_t* __restrict__ can_i, int k,
                                                                 const float* __restrict__ arr_d,
                                                                 const int64_t* __restrict__ arr_i,
                                                                 const int64_t* __restrict__ arr_key,
                                                                 const int64_t* __restrict__ arr_n, int64_t arr_n_stride,
                                                                 float* __restrict__ out_d, int64_t* __restrict__ out_i,
                                                                 int32_t* __restrict__ anomalies) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* pool_d = reinterpret_cast<float*>(smem);                         // [2 k]
    int64_t* pool_i = reinterpret_cast<int64_t*>(smem + (size_t)2 * k * 4); // [2 k] (2 k * 4 is a multiple of 8)
    __shared__ int s_n, s_valid;
    const int64_t f = blockIdx.x; // row of the arrivals
    const int64_t q = flagged[f]; // row of the batch
    const int kk = k + 1;
    const int tid = threadIdx.x;
    const float v = can_d[q * kk + k - 1];
    if (tid == 0) {
        s_n = 0;
        s_valid = 0;
    }
    __syncthreads();
    // the canonical entries better than v ...
    for (int e = tid; e < k; e += RG_THREADS) {
        const float de = can_d[q * kk + e];
        const int64_t ie = can_i[q * kk + e];
        if (ie >= 0) {
            atomicAdd(&s_valid, 1);
            if (de != v) {
                const int p = atomicAdd(&s_n, 1);
                pool_d[p] = de;
                pool_i[p] = ie;
            }
        }
    }
    // ... and the ties among the first k arrivals (flattened index a = s * k + e over the shards' lists)
    const int na = nsh * k;
    for (int a = tid; a < na; a += RG_THREADS) {
        const int sh = a / k, e = a - sh * k;
        const int64_t cnt = min((int64_t)k, arr_n[(int64_t)sh * arr_n_stride + f]);
        if (e >= cnt) {
            continue;
        }
        const int64_t at = ((int64_t)sh * nflag + f) * k + e;
        if (arr_d[at] != v) {
            continue;
        }
        int before = e; // arrivals of this shard in front of it
        if (nsh > 1) {
            const int64_t key = arr_key[at];
            for (int o = 0; o < nsh && before < k; o++) {
                if (o == sh) {
                    continue;
                }
                const int64_t ocnt = min((int64_t)k, arr_n[(int64_t)o * arr_n_stride + f]);
                const int64_t* okey = arr_key + ((int64_t)o * nflag + f) * k;
                // (a shard's arrivals are in key order: count by bisection)
                int lo = 0, hi = (int)ocnt;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (okey[mid] < key) {
                        lo = mid + 1;
                    } else {
                        hi = mid;
                    }
                }
                before += lo;
            }
        }
        if (before < k) {
            const int p = atomicAdd(&s_n, 1);
            pool_d[p] = v;
            pool_i[p] = arr_i[at];
        }
    }
    __syncthreads();
    const int n = s_n; // (>= k whenever the query was flagged: k entries at or below v exist and arrive)
    if (n < s_valid) {
        // Fewer entries than the canonical row holds: the dump pass did not reproduce a tied distance bit for bit (it runs
        // through other kernels than the search) or the bitset / k + 1 interplay left fewer than k arrivals.  The row keeps
        // its canonical copy (written by tie_detect) -- a partial overwrite would return duplicate or misordered ids.
        if (tid == 0 && anomalies != nullptr) {
            atomicAdd(anomalies, 1);
        }
        return;
    }
    for (int e = tid; e < n; e += RG_THREADS) {
        const float de = pool_d[e];
        const int64_t ie = pool_i[e];
        int rank = 0;
        for (int o = 0; o < n; o++) {
            rank += better<IS_L2>(pool_d[o], pool_i[o], de, ie) ? 1 : 0;
        }
        if (rank < k) {
            out_d[q * k + rank] = de;
            out_i[q * k + rank] = ie;
        }
    }
}

hipError_t launch_tie_gather(const int32_t* flagged, int nflag, const float* q, int d, const int64_t* keys, const float* cdis,
                             int nprobe, float* q_out, int64_t* keys_out, float* cdis_out, const float* can_d, int k,
                             float* rad_out, hipStream_t s) {
    if (nflag <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(tie_gather_kernel, dim3((unsigned)nflag), dim3(128), 0, s, flagged, nflag, q, d, keys, cdis, nprobe,
                       q_out, keys_out, cdis_out, can_d, k, rad_out);
    return hipGetLastError();
}

hipError_t launch_tie_resolve(const int32_t* flagged, int nflag, int nsh, const float* can_d, const int64_t* can_i, int k,
                              bool is_l2, const float* arr_d, const int64_t* arr_i, const int64_t* arr_key,
                              const int64_t* arr_n, int64_t arr_n_stride, float* out_d, int64_t* out_i, int32_t* anomalies,
                              hipStream_t s) {
    if (nflag <= 0) {
        return hipSuccess;
    }
    if (nsh <= 0 || (nsh > 1 && arr_key == nullptr)) {
        return hipErrorInvalidValue;
    }
    const size_t sm = (size_t)2 * k * 12;
    auto kern = is_l2 ? tie_resolve_kernel<true> : tie_resolve_kernel<false>;
    if (sm > 48 * 1024) {
        return hipErrorInvalidValue; // (k <= 1023: 24 KB)
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nflag), dim3(RG_THREADS), sm, s, flagged, nflag, nsh, can_d, can_i, k, arr_d, arr_i,
                       arr_key, arr_n, arr_n_stride, out_d, out_i, anomalies);
    return hipGetLastError();
}

hipError_t launch_range_emit(const RangeArgs& a, int64_t nq, bool is_l2, const int64_t* off, const int64_t* qbase,
                             int64_t* out_ids, float* out_dis, hipStream_t s, int64_t cap, int64_t* out_key,
                             int64_t key_base) {
    if (nq <= 0 || a.nprobe <= 0) {
        return hipSuccess;
    }
    const unsigned grid = (unsigned)(nq * a.nprobe);
    if (is_l2) {
        hipLaunchKernelGGL((range_emit_kernel<true>), dim3(grid), dim3(RG_THREADS), 0, s, a, off, qbase, out_ids,
                           out_dis, cap, out_key, key_base);
    } else {
        hipLaunchKernelGGL((range_emit_kernel<false>), dim3(grid), dim3(RG_THREADS), 0, s, a, off, qbase, out_ids,
                           out_dis, cap, out_key, key_base);
    }
    return hipGetLastError();
}

} // namespace knhip
