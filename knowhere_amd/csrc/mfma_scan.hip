// knowhere_amd/csrc/mfma_scan.hip -- IVF-Flat / IVF-SQ8 list scan as an MFMA PREFILTER + exact finish (gfx950).
//
// The exact row scans (flat_scan.hip, sq_scan.hip) reproduce the reference's scalar arithmetic for EVERY
// (row, query, dim) triple: 2-3 separately rounded VALU operations each, so they are VALU-bound by construction
// (BASELINE config C2: 36 ms per 10k-query batch at 0.19 of the fp32 vector peak; C5: 3.2 s).  But a (rows of one
// list) x (queries that probe it) x d block is a dense contraction, and only the k best rows of a query need the
// reference's exact arithmetic.  So, exactly as the coarse quantizer does (coarse_gemm.hip):
//
//   1. sample         the rows of a query's first probes, in coarse order until max(1024, 8 k) rows are covered (one list
//                     when the closest list is long enough, several when it is short or empty), at most MS_SAMPLE in
//                     all, go through the same MFMA kernel in DUMP mode: it writes a pessimistic distance (approx
//                     widened by the error bound eps, filtered rows as the neutral value) per (query, row);
//                     row_select picks the k-th best per query = tau_q.  k unfiltered rows are provably at least that
//                     good, so tau_q bounds the query's final k-th distance.
//   2. mscan_*_kernel every (query, list) pair: approximate distances on the matrix cores
//                     (v_mfma_f32_32x32x2_f32 for fp32 rows; v_mfma_f32_32x32x16_f16 for SQ8 codes), one
//                     compare per (row, query) against tau_q widened by eps; rows that pass are appended to the
//                     query's candidate list and counted in its 64-bin histogram, which units starting later read
//                     to tighten tau_q.  Every row whose EXACT distance is <= tau_q passes (|approx - exact| <= eps),
//                     in particular every row of the final top-k.
//   3. mscan_finish   per query: exact reference-order distances of its candidates (the same l2_step / ip_step /
//                     SQ8 decode sequence as the exact kernels), canonical sort, top-k.  Bit-equal to the exact
//                     scan: the candidates are a superset of the true top-k and their distances are the reference's.
//   4. overflow       a query whose candidate list overflows (no bound: fewer than k unfiltered rows in the sample;
//                     or a very loose one) is flagged; its (query, list) pairs are compacted into one-query work
//                     items and scanned by the exact kernels (flat_scan / sq_scan), then merged as usual.  Exactness
//                     never depends on the capacity being "big enough"; with no flagged query these kernels return
//                     at once.
//
// Reference semantics replaced: IVFFlatScanner::scan_codes (thirdparty/faiss/faiss/cppcontrib/knowhere/
// IndexIVFFlat.cpp:193-236), BaselineIVFSQScannerIP/L2::scan_codes (.../IndexScalarQuantizer.cpp:196-400), the
// per-query heap (impl/ResultHandler.h:258-279).
//
// Layout notes.  fp32 rows: the interleaved blocks of flat_scan.hip (float4 blk[chunk][64 rows]).  A wave takes a
// block of 64 rows = two 32-row MFMA tiles; for the K-slab of 8 dims made of chunks (2s, 2s+1) lane l loads
// chunk 2s + (l >> 5) of row (l & 31): each half-wave reads 512 contiguous bytes.  The four components of that
// float4 feed four v_mfma_f32_32x32x2_f32 (k = 0 <-> lanes 0-31, k = 1 <-> lanes 32-63; any assignment of dims to
// k is fine as long as the query operand uses the same one).  Queries (B operand, lane l = query l & 31) sit in LDS
// as [query][d] rows padded to an odd number of 16-byte quads, so the ds_read_b128 of a slab is conflict-free.
// D layout: lane l holds query l & 31 and the 16 rows (r & 3) + 8 (r >> 2) + 4 (l >> 5): one threshold per lane.
// L2: the accumulator starts at -||x||^2 / 2 (one extra MFMA with A = ||x||^2, B = -0.5), so that
//   ||q||^2 + ||x||^2 - 2 q.x <= tau + eps   <=>   acc >= (||q||^2 - tau - eps) / 2  =: t_q,
// and IP is acc >= tau - eps: the epilogue is one compare per element in both metrics.
#include "common.h"
#include "kernels.h"
#include "ms_common.h"

namespace knhip {

typedef float ms_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 ms_f16x8 __attribute__((ext_vector_type(8)));

constexpr int MS_WAVES = 4;
constexpr int MS_THREADS = MS_WAVES * KN_WAVE;
constexpr int MS_NQT = 2;          // query tiles of 32 per unit (fp32 rows)
constexpr int MS_QT = 32 * MS_NQT; // queries per unit
constexpr int MS_SAMPLE = 8192;    // rows that feed tau_q, at most (a multiple of 64, <= row_select's limit)

// ---- ||x||^2 per stored row position (padded block layout), and the maximum ----------------------------------
__global__ void ms_block_norms_kernel(const float4* __restrict__ rows, int64_t total_blk, int nchunk,
                                      float* __restrict__ out, float* __restrict__ out_max) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    if (t < total_blk * 64) {
        const int64_t b = t >> 6;
        const int r = (int)(t & 63);
        const float4* p = rows + b * (int64_t)nchunk * 64 + r;
        for (int c = 0; c < nchunk; c++) {
            const float4 v = p[(int64_t)c * 64];
            acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        out[t] = acc;
    }
    // norms are >= 0: their bit patterns order like ints
    float m = acc;
    for (int off = 32; off > 0; off >>= 1) {
        m = fmaxf(m, __shfl_xor(m, off, KN_WAVE));
    }
    if (lane_id() == 0 && m > 0.f) {
        atomicMax(reinterpret_cast<int*>(out_max), __float_as_int(m));
    }
}

hipError_t launch_ms_block_norms(const float4* rows, int64_t total_blk, int nchunk, float* out, float* out_max,
                                 hipStream_t s) {
    hipError_t e = hipMemsetAsync(out_max, 0, sizeof(float), s);
    if (e != hipSuccess || total_blk <= 0) {
        return e;
    }
    hipLaunchKernelGGL(ms_block_norms_kernel, dim3((unsigned)((total_blk * 64 + 255) / 256)), dim3(256), 0, s, rows,
                       total_blk, nchunk, out, out_max);
    return hipGetLastError();
}

// ---- units: (bulk virtual list, group of up to qt of its pairs) -----------------------------------------------
// (`*_v` arrays are the work table's, offset to the virtual-list range the units are made from: [0, nlist) = the
// rank-0 probes, [nlist, 2 nlist) = the others)
// single workgroup: exclusive scan of ceil(count / qt) over the lists -- chunks of 4096 lists, four consecutive lists per
// thread (coalesced), wave scans by shuffles (worktable.hip::wt_scan_kernel: the per-thread runs of the first form cost a
// cache line per thread and load)
constexpr int MS_SCAN_THREADS = 1024;
constexpr int MS_SCAN_PER = 4;
__global__ __launch_bounds__(MS_SCAN_THREADS) void ms_unit_scan_kernel(const int32_t* __restrict__ list_count_v,
                                                                       int64_t nlist, int qt,
                                                                       int64_t* __restrict__ unit_off,
                                                                       int64_t* __restrict__ nunits) {
    constexpr int NW = MS_SCAN_THREADS / KN_WAVE;
    __shared__ long long s_w[NW];
    const int tid = threadIdx.x, lane = tid & (KN_WAVE - 1), wave = tid / KN_WAVE;
    long long carry = 0; // (the same in every thread)
    for (int64_t c0 = 0; c0 < nlist; c0 += (int64_t)MS_SCAN_THREADS * MS_SCAN_PER) {
        const int64_t l0 = c0 + (int64_t)tid * MS_SCAN_PER;
        long long n[MS_SCAN_PER], tn = 0;
#pragma unroll
        for (int e = 0; e < MS_SCAN_PER; e++) {
            const int64_t l = l0 + e;
            n[e] = l < nlist ? (list_count_v[l] + qt - 1) / qt : 0;
            tn += n[e];
        }
        long long in = tn; // inclusive scan over the wave's lanes
        for (int d = 1; d < KN_WAVE; d <<= 1) {
            const long long up = __shfl_up(in, d, KN_WAVE);
            if (lane >= d) {
                in += up;
            }
        }
        if (lane == KN_WAVE - 1) {
            s_w[wave] = in;
        }
        __syncthreads();
        long long before = 0, total = 0;
        for (int w = 0; w < NW; w++) {
            const long long a = s_w[w];
            before += w < wave ? a : 0;
            total += a;
        }
        long long u = carry + before + (in - tn);
#pragma unroll
        for (int e = 0; e < MS_SCAN_PER; e++) {
            const int64_t l = l0 + e;
            if (l < nlist) {
                unit_off[l] = u;
            }
            u += n[e];
        }
        carry += total;
        __syncthreads(); // (the wave totals are rewritten by the next chunk)
    }
    if (tid == 0) {
        unit_off[nlist] = carry;
        *nunits = carry;
    }
}

__global__ void ms_units_kernel(const int32_t* __restrict__ list_count_v, const int64_t* __restrict__ list_pair_off_v,
                                const int64_t* __restrict__ unit_off, int64_t nlist, int qt, KnItem* __restrict__ units,
                                const int64_t* __restrict__ list_len, int64_t code_size, double* unit_bytes) {
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nlist) {
        return;
    }
    const int64_t c = list_count_v[l];
    if (c > 0 && unit_bytes != nullptr) {
        // bytes the prefilter streams: every unit reads its list once for up to qt queries
        atomicAdd(unit_bytes, (double)((c + qt - 1) / qt) * (double)list_len[l] * (double)code_size);
    }
    const int64_t p0 = list_pair_off_v[l];
    int64_t u = unit_off[l];
    for (int64_t i = 0; i < c; i += qt, u++) {
        KnItem x;
        x.list = (int32_t)l;
        x.npair = (int32_t)min((int64_t)qt, c - i);
        x.pair0 = p0 + i;
        units[u] = x;
    }
}

hipError_t launch_ms_units(const int32_t* list_count_v, const int64_t* list_pair_off_v, int64_t nlist, int qt,
                           int64_t* unit_off, int64_t* nunits, KnItem* units, const int64_t* list_len,
                           int64_t code_size, double* unit_bytes, hipStream_t s) {
    hipLaunchKernelGGL(ms_unit_scan_kernel, dim3(1), dim3(MS_SCAN_THREADS), 0, s, list_count_v, nlist, qt, unit_off,
                       nunits);
    hipLaunchKernelGGL(ms_units_kernel, dim3((unsigned)((nlist + 255) / 256)), dim3(256), 0, s, list_count_v,
                       list_pair_off_v, unit_off, nlist, qt, units, list_len, code_size, unit_bytes);
    return hipGetLastError();
}

// ---- fp32 rows --------------------------------------------------------------------------------------------------
// DUMP = false: filter mode (candidates).  DUMP = true: the sample pass over the closest lists, every (query, row)
// of the first min(len, MS_SAMPLE) rows gets its pessimistic distance written to dump[q * dump_stride + row].
// Per-pair constant in LDS (sT): filter: the accumulator threshold t; dump: c with value = c - 2 acc (L2: c = ||q||^2
// + eps) or value = acc - c (IP: c = eps).
// NQT = query tiles of 32 per unit (2 in filter mode; 1 in the sample pass, whose units hold few queries).
template <bool IS_L2, bool DUMP, int NQT>
__device__ __forceinline__ void mscan_flat_unit(const MScanArgs a, const int64_t u, unsigned char* smem) {
    constexpr int QT = 32 * NQT;
    const int lane = lane_id();
    const int wave = threadIdx.x / KN_WAVE;
    const KnItem it = a.units[u];
    const int npair = it.npair;
    const int64_t list = it.list;
    const int64_t len = a.list_len[list];
    const int64_t blk0 = a.list_blk_off[list];
    const int64_t row_off = a.list_row_off[list];
    const int nchunk = a.nchunk;
    const int nstep = a.nstep;      // steps of 16 dims (4 chunks)
    const int ldq = nstep * 16 + 4; // floats per query row in LDS: an odd number of 16-byte quads

    float* sQ = reinterpret_cast<float*>(smem);          // [QT][ldq]
    float* sT = sQ + QT * ldq;                           // [QT] filter: accumulator threshold
    float* sC = sT + QT;                                 // [QT] pessimistic distance = c - 2 acc (L2) / acc - c (IP)
    int32_t* sPq = reinterpret_cast<int32_t*>(sC + QT);  // [QT] query of the pair (-1: none)
    int32_t* sPs = sPq + QT;                             // [QT] slot of the pair
    for (int j = wave; j < QT; j += MS_WAVES) { // wave per pair: the histogram row is read lane = bin
        float t = INFINITY, c = 0.f;
        int32_t q = -1, slot = 0;
        if (j < npair) {
            const KnPair p = a.pairs[it.pair0 + j];
            q = p.q;
            slot = p.slot;
            const float qn = a.qnorm[q];
            const float eps = a.eps_scale * (IS_L2 ? (qn + a.xnorm_max) : sqrtf(qn * a.xnorm_max)) + 1e-30f;
            c = IS_L2 ? qn + eps : eps;
            if (DUMP) {
                slot = a.sample_off[(int64_t)q * a.nslot + slot]; // (sPs then holds the pair's first dump column)
            } else {
                float tau = a.gthr[q];
                tau = tighter<IS_L2>(tau, ms_hist_bound<IS_L2>(a, q, a.k));
                if (tau == worst_dist<IS_L2>()) {
                    // no bound (fewer than k unfiltered rows in the sample): every row would pass -> exact fallback
                    if (lane == 0) {
                        a.overflow[q] = 1;
                        a.overflow[a.nq] = 1;
                    }
                } else {
                    t = IS_L2 ? (qn - tau - eps) * 0.5f : tau - eps;
                }
            }
        }
        if (lane == 0) {
            sT[j] = t;
            sC[j] = c;
            sPq[j] = q;
            sPs[j] = slot;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < QT * nstep * 4; t += MS_THREADS) {
        const int j = t / (nstep * 4), c = t % (nstep * 4);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (j < npair) {
            const float* src = a.queries + (int64_t)sPq[j] * a.d + c * 4;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (c * 4 + e < a.d) {
                    v[e] = src[e];
                }
            }
        }
        *reinterpret_cast<float4*>(sQ + j * ldq + c * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();

    const int hi = lane >> 5, lr = lane & 31;
    int64_t nblk = (len + 63) >> 6;
    if (DUMP) {
        nblk = min(nblk, (int64_t)((a.sample_cap + 63) / 64));
    }
    if (nblk <= 0) {
        return;
    }
    const float4* rows = reinterpret_cast<const float4*>(a.rows) + blk0 * (int64_t)nchunk * 64;
    // A operand of one step (16 dims): [row tile][8-dim slab]: chunk 4 s + 2 slab + hi of row tile * 32 + lr
    // Branch-free on purpose: a load inside a conditional makes the compiler wait for it at the join (vmcnt(0) right
    // behind the load: no prefetch at all).  A prefetch past this wave's last block re-reads that block (never used);
    // a chunk past the last one (odd chunk counts) re-reads the last chunk, whose query operand is zero-padded in LDS.
    auto load_step = [&](int64_t b, int s, float4 (&A)[2][2]) {
        const int64_t bb = min(b, nblk - 1);
#pragma unroll
        for (int sl = 0; sl < 2; sl++) {
            const int c = min(4 * s + 2 * sl + hi, nchunk - 1);
            const float4* p = rows + (bb * nchunk + c) * 64 + lr;
            A[0][sl] = p[0];
            A[1][sl] = p[32];
        }
    };
    // Two statically rotating row buffers: the loads of step g + 1 are issued (and pinned there: sched_barrier) before
    // the 32 MFMAs of step g.  The step loop is flattened over this wave's blocks and unrolled by two; both
    // sub-steps are unconditional (a load behind a branch is waited for at its join).
    float4 A[2][2][2];
    int64_t lb = wave; // load cursor
    int ls = 0;
    auto issue = [&](float4 (&dst)[2][2]) {
        load_step(lb, ls, dst);
        if (++ls == nstep) {
            ls = 0;
            lb += MS_WAVES;
        }
    };
    ms_f32x16 acc[2][NQT];
    auto init_acc = [&](int64_t b) {
        // L2: acc = -||x||^2 / 2 for the tile's rows (k = 0 carries the norm, k = 1 nothing)
        ms_f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            z[r] = 0.f;
        }
        ms_f32x16 i0 = z, i1 = z;
        if (IS_L2) {
            const int64_t bb = min(b, nblk - 1);
            float xn0 = a.xnorm[(blk0 + bb) * 64 + lr];
            float xn1 = a.xnorm[(blk0 + bb) * 64 + 32 + lr];
            xn0 = hi == 0 ? xn0 : 0.f;
            xn1 = hi == 0 ? xn1 : 0.f;
            const float mh = hi == 0 ? -0.5f : 0.f;
            i0 = __builtin_amdgcn_mfma_f32_32x32x2f32(xn0, mh, z, 0, 0, 0);
            i1 = __builtin_amdgcn_mfma_f32_32x32x2f32(xn1, mh, z, 0, 0, 0);
        }
#pragma unroll
        for (int qt = 0; qt < NQT; qt++) {
            acc[0][qt] = i0;
            acc[1][qt] = i1;
        }
    };
    auto compute = [&](const float4 (&Ac)[2][2], int s) {
#pragma unroll
        for (int sl = 0; sl < 2; sl++) {
            float4 B[NQT];
#pragma unroll
            for (int qt = 0; qt < NQT; qt++) {
                B[qt] = *reinterpret_cast<const float4*>(sQ + (qt * 32 + lr) * ldq + (2 * s + sl) * 8 + 4 * hi);
            }
#pragma unroll
            for (int qt = 0; qt < NQT; qt++) {
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    acc[t][qt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ac[t][sl].x, B[qt].x, acc[t][qt], 0, 0, 0);
                    acc[t][qt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ac[t][sl].y, B[qt].y, acc[t][qt], 0, 0, 0);
                    acc[t][qt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ac[t][sl].z, B[qt].z, acc[t][qt], 0, 0, 0);
                    acc[t][qt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ac[t][sl].w, B[qt].w, acc[t][qt], 0, 0, 0);
                }
            }
        }
    };
    auto epilogue = [&](int64_t b) {
        if (DUMP) {
            // ---- sample pass: the pessimistic distance of every (query, row), filtered rows as the neutral value ----
            const unsigned long long vmask = ms_valid_rows(a, b, len, row_off);
#pragma unroll
            for (int qt = 0; qt < NQT; qt++) {
                const float c = sC[qt * 32 + lr];
                const int32_t q = sPq[qt * 32 + lr];
                if (q >= 0) {
                    const int32_t off = sPs[qt * 32 + lr];
                    const int64_t lim = min(len, (int64_t)(a.sample_cap - off));
                    float* drow = a.dump + (int64_t)q * a.dump_stride + off + b * 64;
#pragma unroll
                    for (int t = 0; t < 2; t++) {
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int i = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                            if (b * 64 + i < lim) {
                                const float v = IS_L2 ? c - 2.0f * acc[t][qt][r] : acc[t][qt][r] - c;
                                drow[i] = ((vmask >> i) & 1ull) ? v : worst_dist<IS_L2>();
                            }
                        }
                    }
                }
            }
            return;
        }
        // ---- filter: one compare per (row, query); the slow path only where something passes ----------------------
#pragma unroll
        for (int qt = 0; qt < NQT; qt++) {
            const float thr = sT[qt * 32 + lr];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                float m = acc[t][qt][0];
#pragma unroll
                for (int r = 1; r < 16; r++) {
                    m = fmaxf(m, acc[t][qt][r]);
                }
                if (__ballot(m >= thr) != 0ull) {
                    if (m >= thr) {
                        const int32_t q = sPq[qt * 32 + lr], slot = sPs[qt * 32 + lr];
                        const float c = sC[qt * 32 + lr];
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int64_t pos = b * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                            if (acc[t][qt][r] >= thr && pos < len) {
                                ms_emit<IS_L2>(a, q, slot, row_off, pos,
                                               IS_L2 ? c - 2.0f * acc[t][qt][r] : acc[t][qt][r] - c);
                            }
                        }
                    }
                }
            }
        }
    };
    issue(A[0]);
    int64_t b = wave; // compute cursor
    int s = 0;
    const int64_t nbw = nblk > wave ? (nblk - wave + MS_WAVES - 1) / MS_WAVES : 0;
    const int64_t G = nbw * nstep;
    init_acc(b);
    for (int64_t g = 0; g < G; g += 2) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            issue(A[u ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
            compute(A[u], s);
            if (++s == nstep) {
                if (b < nblk) {
                    epilogue(b);
                }
                s = 0;
                b += MS_WAVES;
                init_acc(b);
            }
        }
    }
}

// One unit per workgroup in XCD-aware order; a.unit_loop: a fixed grid walks a unit table whose size only the device
// knows (the retry round of overflowed queries).  Every exit inside a unit is workgroup-uniform.
template <bool IS_L2, bool DUMP, int NQT, bool LOOP>
__global__ __launch_bounds__(MS_THREADS, 4) void mscan_flat_kernel(MScanArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int64_t nunits = *a.nunits_dev;
    if (LOOP) {
        for (int64_t u = blockIdx.x; u < nunits; u += gridDim.x) {
            mscan_flat_unit<IS_L2, DUMP, NQT>(a, u, smem);
            __syncthreads();
        }
    } else {
        if ((int64_t)blockIdx.x >= ((nunits + 7) / 8) * 8) {
            return;
        }
        const int64_t u = xcd_item(blockIdx.x, nunits);
        if (u >= nunits) {
            return;
        }
        mscan_flat_unit<IS_L2, DUMP, NQT>(a, u, smem);
    }
}

// ---- SQ8 codes -------------------------------------------------------------------------------------------------
// Decoded component (reference codecs.h:37-41, quantizers.h:139-145): x_i = vmin_i + vdiff_i (c_i + 0.5) / 255, so with
// y = q (IP) or q - centroid (L2, by_residual) and y'_i = y_i vdiff_i / 255:
//     <y, x> = A + sum_i y'_i c_i,        A = sum_i y_i (vmin_i + 0.5 vdiff_i / 255).
// The code bytes go to the matrix cores as f16 WITHOUT a convert: the bit pattern 0x6400 | c is the half 1024 + c
// exactly (one v_perm_b32 per two codes), y' is scaled by a power of two so that max |y'| lands in [2^9, 2^10) and is
// split into two halves hi + lo (22 significant bits), and v_mfma_f32_32x32x16_f16 accumulates
//     S = sum_i (hi_i + lo_i) (1024 + c_i)            (fp32 accumulate)
// so that  sum_i y'_i c_i = (S - 1024 sum_i (hi_i + lo_i)) / scale  up to the error bound eps below.
// One unit = (list, up to 32 pairs); a wave takes 64 rows = two 32-row tiles; step = 32 dims: lane l loads the 16 codes
// of chunk 2 s + (l >> 5) of row (l & 31) (one coalesced 16-byte load) = the A operands of two MFMAs.
// Acceptance on the accumulator (one compare per element):
//   IP: dis0 + A + (S - off) / sc >= tau - eps                  <=>  S >= sc (tau - eps - dis0 - A) + off
//   L2: ||y||^2 + ||x||^2 - 2 (A + (S - off) / sc) <= tau + eps <=>  S - sc ||x||^2 / 2 >= sc (||y||^2 - 2 A - tau - eps) / 2 + off
// (the -sc ||x||^2 / 2 term is the accumulator's start value: one fp32 MFMA with A = ||x||^2, B = -sc / 2).
// eps: see the bound spelled out where the thresholds are formed (worst-case rounding analysis, factor 2 on top).
constexpr int MQ_WAVES = 16;
constexpr int MQ_THREADS = MQ_WAVES * KN_WAVE;
constexpr int MQ_QT = 32;

__global__ void ms_sq8_norms_kernel(const uint4* __restrict__ rows, int64_t total_blk, int nchunk16, int d,
                                    const float* __restrict__ trained, float* __restrict__ out,
                                    float* __restrict__ out_max) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    const bool live = t < total_blk * 64;
    const int64_t b = live ? (t >> 6) : 0;
    const int r = (int)(t & 63);
    const uint4* p = rows + b * (int64_t)nchunk16 * 64 + r;
    for (int c = 0; live && c < nchunk16; c++) {
        const uint4 w = p[(int64_t)c * 64];
        const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int i = c * 16 + e;
            if (i < d) {
                const float code = (float)((ww[e >> 2] >> (8 * (e & 3))) & 0xffu);
                const float x = trained[i] + trained[d + i] * ((code + 0.5f) / 255.0f);
                acc += x * x;
            }
        }
    }
    if (live) {
        out[t] = acc;
    }
    float m = acc; // (norms are >= 0: their bit patterns order like ints)
    for (int off = 32; off > 0; off >>= 1) {
        m = fmaxf(m, __shfl_xor(m, off, KN_WAVE));
    }
    if (lane_id() == 0 && m > 0.f) {
        atomicMax(reinterpret_cast<int*>(out_max), __float_as_int(m));
    }
}

hipError_t launch_ms_sq8_norms(const uint4* rows, int64_t total_blk, int nchunk16, int d, const float* trained,
                               float* out, float* out_max, hipStream_t s) {
    hipError_t e = hipMemsetAsync(out_max, 0, sizeof(float), s);
    if (e != hipSuccess || total_blk <= 0) {
        return e;
    }
    hipLaunchKernelGGL(ms_sq8_norms_kernel, dim3((unsigned)((total_blk * 64 + 255) / 256)), dim3(256), 0, s, rows,
                       total_blk, nchunk16, d, trained, out, out_max);
    return hipGetLastError();
}

__device__ __forceinline__ float ms_wave_sum(float v) {
    for (int off = 32; off > 0; off >>= 1) {
        v += __shfl_xor(v, off, KN_WAVE);
    }
    return v;
}

// 16 code bytes -> two f16x8 operands (1024 + code)
__device__ __forceinline__ void ms_codes_to_f16(const uint4 w, ms_f16x8& lo8, ms_f16x8& hi8) {
    const uint32_t k64 = 0x64646464u;
    union {
        uint32_t u[4];
        ms_f16x8 v;
    } a, b;
    a.u[0] = __builtin_amdgcn_perm(k64, w.x, 0x04010400u);
    a.u[1] = __builtin_amdgcn_perm(k64, w.x, 0x04030402u);
    a.u[2] = __builtin_amdgcn_perm(k64, w.y, 0x04010400u);
    a.u[3] = __builtin_amdgcn_perm(k64, w.y, 0x04030402u);
    b.u[0] = __builtin_amdgcn_perm(k64, w.z, 0x04010400u);
    b.u[1] = __builtin_amdgcn_perm(k64, w.z, 0x04030402u);
    b.u[2] = __builtin_amdgcn_perm(k64, w.w, 0x04010400u);
    b.u[3] = __builtin_amdgcn_perm(k64, w.w, 0x04030402u);
    lo8 = a.v;
    hi8 = b.v;
}

// ---- inner product: the query operand does not depend on the list -> prepared once per batch -----------------------
// (by_residual IP: dis = coarse_dis + <q, x>; L2 needs q - centroid per (query, list) and keeps the in-kernel path.)
// One wave per query: y' = q vdiff / 255 scaled by sc = 2^ex (max |y'| sc in [2^9, 2^10)) and split into halves hi + lo
// -> qh / ql [nq][ldq] (zero padded to the step multiple); qs[q] = {sc, A, W, sum |y'|, sum (hi + lo), finite?, 0, 0}.
// The same arithmetic as the in-kernel prologue below.
__global__ __launch_bounds__(256) void ms_sq8_query_prep_kernel(const float* __restrict__ queries, int64_t nq, int d,
                                                                int ldq, const float* __restrict__ trained,
                                                                _Float16* __restrict__ qh, _Float16* __restrict__ ql,
                                                                float* __restrict__ qs) {
    const int lane = lane_id();
    const int64_t q = (int64_t)blockIdx.x * (blockDim.x / KN_WAVE) + threadIdx.x / KN_WAVE;
    if (q >= nq) {
        return;
    }
    const float* vmin = trained;
    const float* vdiff = trained + d;
    const float* qv = queries + q * d;
    float mx = 0.f;
    for (int i = lane; i < d; i += KN_WAVE) {
        mx = fmaxf(mx, fabsf(qv[i] * vdiff[i] * (1.0f / 255.0f)));
    }
    for (int off = 32; off > 0; off >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, off, KN_WAVE));
    }
    int ex = 0;
    if (mx > 0.f && mx < INFINITY) {
        ex = 9 - ilogbf(mx);
        ex = max(-60, min(60, ex));
    }
    const float sc = ldexpf(1.0f, ex);
    float sA = 0.f, sW = 0.f, sYp = 0.f, sHL = 0.f;
    for (int i = lane; i < ldq; i += KN_WAVE) {
        _Float16 h = (_Float16)0.f, l = (_Float16)0.f;
        if (i < d) {
            const float y = qv[i];
            const float yp = y * vdiff[i] * (1.0f / 255.0f) * sc;
            h = (_Float16)yp;
            l = (_Float16)(yp - (float)h);
            sA += y * (vmin[i] + 0.5f * vdiff[i] * (1.0f / 255.0f));
            sW += fabsf(y) * (fabsf(vmin[i]) + fabsf(vdiff[i]));
            sYp += fabsf(y * vdiff[i] * (1.0f / 255.0f));
            sHL += (float)h + (float)l;
        }
        qh[q * ldq + i] = h;
        ql[q * ldq + i] = l;
    }
    sA = ms_wave_sum(sA);
    sW = ms_wave_sum(sW);
    sYp = ms_wave_sum(sYp);
    sHL = ms_wave_sum(sHL);
    if (lane == 0) {
        float* o = qs + q * 8;
        o[0] = sc;
        o[1] = sA;
        o[2] = sW;
        o[3] = sYp;
        o[4] = sHL;
        o[5] = (mx < INFINITY) ? 1.f : 0.f;
        o[6] = 0.f;
        o[7] = 0.f;
    }
}

hipError_t launch_ms_sq8_query_prep(const float* queries, int64_t nq, int d, int ldq, const float* trained, void* qh,
                                    void* ql, float* qs, hipStream_t s) {
    if (nq <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(ms_sq8_query_prep_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, queries, nq, d, ldq,
                       trained, reinterpret_cast<_Float16*>(qh), reinterpret_cast<_Float16*>(ql), qs);
    return hipGetLastError();
}

// DUMP: as in mscan_flat_kernel; the pessimistic distance is u0 + v (acc - off) with the per-pair constants below.
template <bool IS_L2, bool DUMP>
__device__ __forceinline__ void mscan_sq8_unit(const MScanArgs a, const int64_t u, unsigned char* smem) {
    const int lane = lane_id();
    const int wave = threadIdx.x / KN_WAVE;
    const KnItem it = a.units[u];
    const int npair = it.npair;
    const int64_t list = it.list;
    const int64_t len = a.list_len[list];
    const int64_t blk0 = a.list_blk_off[list];
    const int64_t row_off = a.list_row_off[list];
    const int d = a.d;
    const int nchunk = a.nchunk; // 16-code chunks
    const int nstep = a.nstep;   // steps of 32 dims
    const int ldh = nstep * 32 + 8; // halves per query row: an odd number of 16-byte quads

    _Float16* sH = reinterpret_cast<_Float16*>(smem);      // [32][ldh]
    _Float16* sL = sH + MQ_QT * ldh;                       // [32][ldh]
    float* sT = reinterpret_cast<float*>(sL + MQ_QT * ldh); // [32] accumulator thresholds
    float* sSc = sT + MQ_QT;                               // [32] -scale / 2 (L2 start value)
    float* sU0 = sSc + MQ_QT;                              // [32] dump: value = u0 + v (acc - off)
    float* sV = sU0 + MQ_QT;
    float* sOff = sV + MQ_QT;
    int32_t* sPq = reinterpret_cast<int32_t*>(sOff + MQ_QT);
    int32_t* sPs = sPq + MQ_QT;
    const float* vmin = a.trained;
    const float* vdiff = a.trained + d;
    const float* cen = a.centroids + list * d;
    // ---- per pair: y' scaled + split into LDS, acceptance threshold -------------------------------------------------
    for (int j = wave; j < MQ_QT; j += MQ_WAVES) {
        float thr = INFINITY, nsc = 0.f, u0 = 0.f, vv = 0.f, offv = 0.f;
        int32_t q = -1, slot = 0;
        if (j < npair) {
            const KnPair p = a.pairs[it.pair0 + j];
            q = p.q;
            slot = p.slot;
            const int32_t slot_in = slot;
            if (DUMP) {
                slot = a.sample_off[(int64_t)q * a.nslot + slot]; // (sPs then holds the pair's first dump column)
            }
            const float* qv = a.queries + (int64_t)q * d;
            const bool prepared = !IS_L2 && a.qs != nullptr; // (IP: hi / lo rows and sums come from ms_sq8_query_prep)
            float mx = 0.f;
            for (int i = lane; !prepared && i < d; i += KN_WAVE) {
                const float y = IS_L2 ? (qv[i] - cen[i]) : qv[i];
                mx = fmaxf(mx, fabsf(y * vdiff[i] * (1.0f / 255.0f)));
            }
            for (int off = 32; off > 0; off >>= 1) {
                mx = fmaxf(mx, __shfl_xor(mx, off, KN_WAVE));
            }
            int ex = 0;
            if (mx > 0.f && mx < INFINITY) {
                ex = 9 - ilogbf(mx);
                ex = max(-60, min(60, ex));
            }
            float sc = ldexpf(1.0f, ex);
            float sA = 0.f, sW = 0.f, sYp = 0.f, sHL = 0.f, sR = 0.f;
            if (prepared) {
                const float* o = a.qs + (int64_t)q * 8;
                sc = o[0];
                sA = o[1];
                sW = o[2];
                sYp = o[3];
                sHL = o[4];
                mx = o[5] != 0.f ? 0.f : INFINITY;
            }
            for (int i = lane; !prepared && i < nstep * 32; i += KN_WAVE) {
                _Float16 h = (_Float16)0.f, l = (_Float16)0.f;
                if (i < d) {
                    const float y = IS_L2 ? (qv[i] - cen[i]) : qv[i];
                    const float yp = y * vdiff[i] * (1.0f / 255.0f) * sc;
                    h = (_Float16)yp;
                    l = (_Float16)(yp - (float)h);
                    sA += y * (vmin[i] + 0.5f * vdiff[i] * (1.0f / 255.0f));
                    sW += fabsf(y) * (fabsf(vmin[i]) + fabsf(vdiff[i])); // >= sum |y_i x_i|
                    sYp += fabsf(y * vdiff[i] * (1.0f / 255.0f));          // sum |y'_i| (unscaled)
                    sHL += (float)h + (float)l;
                    sR += y * y;
                }
                sH[j * ldh + i] = h;
                sL[j * ldh + i] = l;
            }
            if (!prepared) {
                sA = ms_wave_sum(sA);
                sW = ms_wave_sum(sW);
                sYp = ms_wave_sum(sYp);
                sHL = ms_wave_sum(sHL);
                sR = ms_wave_sum(sR);
            }
            const float dis0 = IS_L2 ? 0.f : a.coarse_dis[(int64_t)q * a.nslot + slot_in];
            // (the constants below are themselves rounded: a few ulp of the magnitudes they are formed from go on top)
            // Error bound, every term with a factor 2 on top of the standard worst-case analysis (u = 2^-24):
            //   matrix cores   (2 d + 64) u * 1279 sum |y'_i|   every product of S = sum (hi+lo)(1024+c) counted as one
            //                                                    rounded fp32 addition of magnitude <= 1279 |y'_i|
            //   split, y', A, off, tree sums, threshold forming   32 u (|A| + 1024 sum |y'| + |dis0| + ||y||^2)
            //   the exact sequence itself   IP: (d + 8) u sum |y_i| (|vmin_i| + |vdiff_i|) >= gamma_d sum |y_i x_i|
            //                               L2: (2 d + 16) u * 2 (||y||^2 + max ||x||^2) >= gamma sum (y_i - x_i)^2, and
            //                               the inner-product part enters twice
            const float u = 5.9604645e-8f;
            const float e_mfma = (2.0f * (float)d + 64.0f) * u * 1279.0f * sYp;
            const float e_misc = 32.0f * u * (fabsf(sA) + 1024.0f * sYp + fabsf(dis0) + sR);
            const float eps = 2.0f * (IS_L2 ? 2.0f * (e_mfma + e_misc) + (2.0f * (float)d + 16.0f) * u * 2.0f * (sR + a.xnorm_max)
                                            : e_mfma + e_misc + ((float)d + 8.0f) * u * sW) + 1e-30f;
            if (!DUMP && a.eps_max != nullptr && lane == 0) {
                // the finish prunes by pessimistic distances: it needs an eps that dominates every emission of the query (an
                // infinite or NaN eps orders above every finite one as a bit pattern: the finish then does not prune)
                atomicMax(a.eps_max + q, __float_as_uint(eps));
            }
            const float off = 1024.0f * sHL;
            nsc = -0.5f * sc;
            offv = off;
            u0 = IS_L2 ? (sR - 2.0f * sA + eps) : (dis0 + sA - eps);
            vv = IS_L2 ? -2.0f / sc : 1.0f / sc;
            if (!DUMP) {
                float tau = a.gthr[q];
                tau = tighter<IS_L2>(tau, ms_hist_bound<IS_L2>(a, q, a.k));
                if (tau == worst_dist<IS_L2>() || !(mx < INFINITY)) {
                    // no bound (fewer than k unfiltered rows in the sample): every row would pass -> exact fallback
                    if (lane == 0) {
                        a.overflow[q] = 1;
                        a.overflow[a.nq] = 1;
                    }
                } else {
                    if (IS_L2) {
                        thr = sc * ((sR - 2.0f * sA - tau - eps) * 0.5f) + off;
                    } else {
                        thr = sc * (tau - eps - dis0 - sA) + off;
                    }
                    thr -= 8.0f * 5.9604645e-8f * (fabsf(off) + fabsf(thr));
                }
            }
        } else if (IS_L2 || a.qs == nullptr) {
            for (int i = lane; i < nstep * 32; i += KN_WAVE) {
                sH[j * ldh + i] = (_Float16)0.f;
                sL[j * ldh + i] = (_Float16)0.f;
            }
        }
        if (lane == 0) {
            sT[j] = thr;
            sSc[j] = nsc;
            sU0[j] = u0;
            sV[j] = vv;
            sOff[j] = offv;
            sPq[j] = q;
            sPs[j] = slot;
        }
    }
    __syncthreads();
    if (!IS_L2 && a.qs != nullptr) {
        // prepared query operands: the 32 pairs' hi / lo rows are copied into LDS, 16 bytes per thread and turn
        const int nv = nstep * 4; // 16-byte pieces per row (nstep * 32 halves)
        const uint4* gh = reinterpret_cast<const uint4*>(a.qh);
        const uint4* gl = reinterpret_cast<const uint4*>(a.ql);
        for (int t = threadIdx.x; t < MQ_QT * nv; t += MQ_THREADS) {
            const int j = t / nv, v = t % nv;
            const int32_t q = sPq[j];
            uint4 h = make_uint4(0, 0, 0, 0), l = h;
            if (q >= 0) {
                h = gh[(int64_t)q * nv + v];
                l = gl[(int64_t)q * nv + v];
            }
            *reinterpret_cast<uint4*>(sH + j * ldh + v * 8) = h;
            *reinterpret_cast<uint4*>(sL + j * ldh + v * 8) = l;
        }
        __syncthreads();
    }

    const int hi = lane >> 5, lr = lane & 31;
    int64_t nblk = (len + 63) >> 6;
    if (DUMP) {
        nblk = min(nblk, (int64_t)((a.sample_cap + 63) / 64));
    }
    if (nblk <= 0) {
        return;
    }
    const uint4* rows = reinterpret_cast<const uint4*>(a.rows) + blk0 * (int64_t)nchunk * 64;
    // branch-free (see mscan_flat_kernel): clamped re-reads instead of conditionals around the loads
    auto load_step = [&](int64_t b, int s, uint4 (&A)[2]) {
        const int64_t bb = min(b, nblk - 1);
        const int c = min(2 * s + hi, nchunk - 1);
        const uint4* p = rows + (bb * nchunk + c) * 64 + lr;
        A[0] = p[0];
        A[1] = p[32];
    };
    // Code loads run THREE steps ahead of the MFMAs that consume them, in four statically rotating register sets
    // (the step loop is flattened over this wave's blocks and unrolled by four: a copy-rotation would make every
    // step wait for the newest load).  At d = 768 one workgroup fills a CU's LDS, so its 16 waves x 3 x 2 KiB in
    // flight are what covers the HBM latency.
    uint4 A[4][2];
    int64_t lb = wave; // load cursor
    int ls = 0;
    auto issue = [&](uint4 (&dst)[2]) {
        load_step(lb, ls, dst);
        if (++ls == nstep) {
            ls = 0;
            lb += MQ_WAVES;
        }
    };
    issue(A[0]);
    issue(A[1]);
    issue(A[2]);
    const float thr = sT[lr];
    ms_f32x16 acc[2];
    // L2: ||x||^2 of the NEXT block's rows is fetched a block ahead (a wait for it would drain the code prefetch)
    float xn_next0 = 0.f, xn_next1 = 0.f;
    auto fetch_xn = [&](int64_t b) {
        if (IS_L2) {
            const int64_t bb = min(b, nblk - 1);
            xn_next0 = a.xnorm[(blk0 + bb) * 64 + lr];
            xn_next1 = a.xnorm[(blk0 + bb) * 64 + 32 + lr];
        }
    };
    auto init_acc = [&]() {
        ms_f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            z[r] = 0.f;
        }
        acc[0] = z;
        acc[1] = z;
        if (IS_L2) {
            const float bs = hi == 0 ? sSc[lr] : 0.f;
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(hi == 0 ? xn_next0 : 0.f, bs, z, 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(hi == 0 ? xn_next1 : 0.f, bs, z, 0, 0, 0);
        }
    };
    auto compute = [&](const uint4 (&Ac)[2], int s) {
        ms_f16x8 Bh[2], Bl[2];
#pragma unroll
        for (int e8 = 0; e8 < 2; e8++) {
            Bh[e8] = *reinterpret_cast<const ms_f16x8*>(sH + lr * ldh + 32 * s + 16 * hi + 8 * e8);
            Bl[e8] = *reinterpret_cast<const ms_f16x8*>(sL + lr * ldh + 32 * s + 16 * hi + 8 * e8);
        }
#pragma unroll
        for (int t = 0; t < 2; t++) {
            ms_f16x8 Af[2];
            ms_codes_to_f16(Ac[t], Af[0], Af[1]);
#pragma unroll
            for (int e8 = 0; e8 < 2; e8++) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[e8], Bh[e8], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[e8], Bl[e8], acc[t], 0, 0, 0);
            }
        }
    };
    auto epilogue = [&](int64_t b) {
        if (DUMP) {
            const unsigned long long vmask = ms_valid_rows(a, b, len, row_off);
            const int32_t q = sPq[lr];
            if (q >= 0) {
                const float u0 = sU0[lr], vv = sV[lr], off = sOff[lr];
                const int32_t col0 = sPs[lr];
                const int64_t lim = min(len, (int64_t)(a.sample_cap - col0));
                float* drow = a.dump + (int64_t)q * a.dump_stride + col0 + b * 64;
#pragma unroll
                for (int t = 0; t < 2; t++) {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int i = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (b * 64 + i < lim) {
                            const float v = u0 + vv * (acc[t][r] - off);
                            drow[i] = ((vmask >> i) & 1ull) ? v : worst_dist<IS_L2>();
                        }
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int t = 0; t < 2; t++) {
            float m = acc[t][0];
#pragma unroll
            for (int r = 1; r < 16; r++) {
                m = fmaxf(m, acc[t][r]);
            }
            if (__ballot(m >= thr) != 0ull) {
                if (m >= thr) {
                    const int32_t q = sPq[lr], slot = sPs[lr];
                    const float u0 = sU0[lr], vv = sV[lr], off = sOff[lr];
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int64_t pos = b * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (acc[t][r] >= thr && pos < len) {
                            ms_emit<IS_L2>(a, q, slot, row_off, pos, u0 + vv * (acc[t][r] - off));
                        }
                    }
                }
            }
        }
    };
    int64_t b = wave; // compute cursor
    int s = 0;
    const int64_t nbw = nblk > wave ? (nblk - wave + MQ_WAVES - 1) / MQ_WAVES : 0;
    const int64_t G = nbw * nstep;
    fetch_xn(b);
    init_acc();
    fetch_xn(b + MQ_WAVES);
    // (the four sub-steps are unconditional -- loads behind a branch would be waited for at its join; up to three
    // steps past the end multiply garbage into an accumulator nobody reads)
    for (int64_t g = 0; g < G; g += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            issue(A[(u + 3) & 3]);
            __builtin_amdgcn_sched_barrier(0);
            compute(A[u], s);
            if (++s == nstep) {
                if (b < nblk) {
                    epilogue(b);
                }
                s = 0;
                b += MQ_WAVES;
                init_acc();
                fetch_xn(b + MQ_WAVES);
            }
        }
    }
}

template <bool IS_L2, bool DUMP, bool LOOP>
__global__ __launch_bounds__(MQ_THREADS) void mscan_sq8_kernel(MScanArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int64_t nunits = *a.nunits_dev;
    if (LOOP) { // (see mscan_flat_kernel)
        for (int64_t u = blockIdx.x; u < nunits; u += gridDim.x) {
            mscan_sq8_unit<IS_L2, DUMP>(a, u, smem);
            __syncthreads();
        }
    } else {
        if ((int64_t)blockIdx.x >= ((nunits + 7) / 8) * 8) {
            return;
        }
        const int64_t u = xcd_item(blockIdx.x, nunits);
        if (u >= nunits) {
            return;
        }
        mscan_sq8_unit<IS_L2, DUMP>(a, u, smem);
    }
}

// ---- sample plan ---------------------------------------------------------------------------------------------------
// Which (query, slot) pairs feed tau_q: the probes in coarse order until `smin` rows are covered (one list when the
// closest list is long enough, several when it is short or empty -- inner-product clusterings have many tiny lists),
// at most `cap` (<= MS_SAMPLE) rows in all -- a long list gives its first cap - cum rows.  sample_off[q][slot] = first dump column of the pair, -1 = not sampled;
// n_row[q] = columns used.
__global__ void ms_sample_plan_kernel(const int64_t* __restrict__ keys, int64_t nq, int nprobe, int64_t nlist,
                                      const int64_t* __restrict__ list_len, int smin, int cap,
                                      int32_t* __restrict__ sample_off, int32_t* __restrict__ n_row) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) {
        return;
    }
    int cum = 0;
    for (int slot = 0; slot < nprobe; slot++) {
        const int64_t key = keys[q * nprobe + slot];
        const int64_t len = (key >= 0 && key < nlist) ? list_len[key] : 0;
        int32_t off = -1;
        if (len > 0 && cum < smin && cum < cap) {
            off = cum;
            cum += (int)min(len, (int64_t)(cap - cum));
        }
        sample_off[q * nprobe + slot] = off;
    }
    n_row[q] = cum;
}

hipError_t launch_ms_sample_plan(const int64_t* keys, int64_t nq, int nprobe, int64_t nlist, const int64_t* list_len,
                                 int smin, int cap, int32_t* sample_off, int32_t* n_row, hipStream_t s) {
    if (nq <= 0) {
        return hipSuccess;
    }
    if (cap < 1 || cap > MS_SAMPLE) {
        return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(ms_sample_plan_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, keys, nq, nprobe, nlist,
                       list_len, smin, cap, sample_off, n_row);
    return hipGetLastError();
}

// ---- tau_q and the candidate histogram's range from the sample selection ---------------------------------------
template <bool IS_L2>
__global__ void ms_tau_kernel(const float* __restrict__ sel_d, int64_t nq, int k, float* __restrict__ gthr,
                              uint2* __restrict__ gmeta) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) {
        return;
    }
    const float kth = sel_d[q * k + k - 1]; // the neutral value when the sample holds fewer than k unfiltered rows
    gthr[q] = kth;
    if (gmeta != nullptr) {
        uint32_t lo = 0, shift = KN_HIST_OFF;
        if (kth != worst_dist<IS_L2>() && kth == kth) {
            lo = dist_key<IS_L2>(sel_d[q * k]);
            const uint32_t range = dist_key<IS_L2>(kth) - lo;
            shift = 0;
            while ((range >> shift) >= (uint32_t)(KN_HIST_BINS - 1)) {
                shift++;
            }
        }
        gmeta[q] = make_uint2(lo, shift);
    }
}

hipError_t launch_ms_tau(const float* sel_d, int64_t nq, int k, bool is_l2, float* gthr, uint2* gmeta, hipStream_t s) {
    if (nq <= 0) {
        return hipSuccess;
    }
    if (is_l2) {
        hipLaunchKernelGGL(ms_tau_kernel<true>, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, sel_d, nq, k, gthr,
                           gmeta);
    } else {
        hipLaunchKernelGGL(ms_tau_kernel<false>, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, sel_d, nq, k, gthr,
                           gmeta);
    }
    return hipGetLastError();
}

// ---- exact finish ---------------------------------------------------------------------------------------------
// One workgroup per query: exact distances of the candidates (thread per candidate, reference operation order) and a
// running top-k kept through chunks: LDS slots [0, k) hold the best so far, slots [k, P) the next chunk of candidates;
// bitonic sort by (distance key, id tie key); repeat.  Any number of candidates, canonical order throughout.
constexpr int MF_THREADS = 256;
constexpr int MF_MLP = 8; // row pieces requested at a time per candidate

template <bool IS_L2, int KIND> // KIND 1: fp32 rows, 2: PQ codes (M = 32, dsub = 4; pq_filter.hip), 3: SQ8
__global__ __launch_bounds__(MF_THREADS) void mscan_finish_kernel(MScanArgs a, const int64_t* __restrict__ keys,
                                                                  const float* __restrict__ coarse_dis, int nprobe,
                                                                  int k, int P_max, float* __restrict__ out_d,
                                                                  int64_t* __restrict__ out_i,
                                                                  unsigned long long* __restrict__ counters, int pass) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ float tab[256];
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x;
    // pass 1: queries that did not overflow are finished; an overflowed query (flag 1) with candidates gets a RETRY:
    //         the exact k-th of the candidates gathered so far is a tight bound (k unfiltered rows are at least that
    //         good) -> gthr, its list is emptied, flag 2; its pairs are then filtered again as one-query units.
    // pass 2: the retried queries (flag 2) are finished; one that overflowed again carries flag 1 (exact kernels).
    // (flag 3 = flag 1 that pass 1 has counted: no candidates to retry with)
    const int flag = a.overflow[q];
    const bool retry_prep = pass == 1 && flag == 1;
    if ((pass == 1 && flag != 0 && flag != 1) || (pass == 2 && flag != 2)) {
        if (pass == 2 && flag == 1 && tid == 0) {
            // flagged again in the retry round (its candidate list, or the decode form's record regions, filled up once
            // more): the exact kernels
            atomicAdd(counters + 1, 1ull);
        }
        return;
    }
    const int d = a.d;
    int n = min(a.cand_cnt[q], a.cap);
    const int n_filter = n;
    // ---- prune (PQ prefilter; fp32 rows since round 5: a loose tau from a short sample gives a few queries thousands of
    // candidates, and they were the stage's tail): every candidate came with its pessimistic distance (exact <= pess).  The k-th best of
    // those values bounds the final k-th distance, and a candidate whose OPTIMISTIC distance (>= pess - 2 eps_max) is
    // beyond it cannot enter: only the rest is recomputed exactly (32 dependent gathers each).  The k-th value comes from
    // a binary search over the order-preserving integer keys, the candidates sit in registers meanwhile.
    constexpr int MF_PRUNE_PER_THREAD = 16;
    // fp32 rows: any number of candidates -- the bound comes from the first MF_THREADS * MF_PRUNE_PER_THREAD of them (the
    // k-th best of ANY k candidates is a valid bound; the eps of this kind does not depend on it), the rest is streamed
    // through the same test.  (A query whose sample was unlucky can arrive with the full capacity of candidates: without
    // this it alone took 16 rounds of 1013 exact distances + a 1024-entry sort, the tail of the whole stage.)
    // (PQ codes: the same; the |tau| its eps carries is then bounded through the histogram's origin, see eps_max below.
    // SQ8 codes since round 5 as well: C5s spent 3.5 ms of a 3.6 ms finish on the few queries that arrived with the full
    // capacity of 32768 candidates -- 32 rounds of 1013 exact 768-dimensional distances each)
    const int n_head = min(n, MF_THREADS * MF_PRUNE_PER_THREAD);
    if ((KIND != 3 || a.eps_max != nullptr) && a.cand_pess != nullptr && !retry_prep && n >= 2 * k &&
        n_head <= MF_THREADS * MF_PRUNE_PER_THREAD) {
        __shared__ int s_cnt;
        __shared__ float s_red[MF_THREADS / KN_WAVE];
        // eps_max: the per-query part + the fp32 roundings at the largest |dis0| of the query's probes and its bound
        float cmax = 0.f;
        for (int sl = tid; KIND == 2 && sl < nprobe; sl += MF_THREADS) {
            cmax = fmaxf(cmax, fabsf(coarse_dis[q * nprobe + sl]));
        }
#pragma unroll
        for (int dlt = KN_WAVE / 2; dlt > 0; dlt >>= 1) {
            cmax = fmaxf(cmax, __shfl_xor(cmax, dlt, KN_WAVE));
        }
        if ((tid & (KN_WAVE - 1)) == 0) {
            s_red[tid / KN_WAVE] = cmax;
        }
        __syncthreads();
        cmax = s_red[0];
        for (int w = 1; w < MF_THREADS / KN_WAVE; w++) {
            cmax = fmaxf(cmax, s_red[w]);
        }
        // (completed below, once tau2 is known: the emission's eps used |tau| of a bound between gthr and tau2)
        float eps_q = 0.f, eps_mu = 0.f;
        if (KIND == 2) {
            eps_q = a.pq_qs[q * 4 + 2];
            eps_mu = a.pq_prune_mu ? fabsf(a.pq_qs[q * 4 + 1]) : 0.f;
        }
        const float gthr_abs = fabsf(a.gthr[q]);
        uint32_t pk[MF_PRUNE_PER_THREAD];
        int64_t pc[MF_PRUNE_PER_THREAD];
        float pp[MF_PRUNE_PER_THREAD];
#pragma unroll
        for (int u = 0; u < MF_PRUNE_PER_THREAD; u++) {
            const int ci = tid + u * MF_THREADS;
            pk[u] = 0xffffffffu;
            pc[u] = 0;
            pp[u] = worst_dist<IS_L2>();
            if (ci < n_head) {
                pc[u] = a.cand[q * (int64_t)a.cap + ci];
                pp[u] = a.cand_pess[q * (int64_t)a.cap + ci];
                pk[u] = dist_key<IS_L2>(pp[u]);
            }
        }
        // smallest key K with count(keys <= K) >= k: bisection, one counter and ONE barrier per step (the counters of all
        // 32 steps are zeroed up front; the loop runs exactly 32 times for every thread: the interval halves each step)
        __shared__ int s_step[32];
        if (tid < 32) {
            s_step[tid] = 0;
        }
        __syncthreads();
        // (the interval starts at the block's own key range: pessimistic distances of one query share their exponent, ~22
        // steps instead of 32; lo, hi are the same in every thread, so the trip count is uniform)
        __shared__ uint32_t s_kmin[MF_THREADS / KN_WAVE], s_kmax[MF_THREADS / KN_WAVE];
        uint32_t kmn = 0xffffffffu, kmx = 0u;
#pragma unroll
        for (int u = 0; u < MF_PRUNE_PER_THREAD; u++) {
            kmn = min(kmn, pk[u]);
            kmx = (tid + u * MF_THREADS < n_head) ? max(kmx, pk[u]) : kmx;
        }
#pragma unroll
        for (int dlt = KN_WAVE / 2; dlt > 0; dlt >>= 1) {
            kmn = min(kmn, (uint32_t)__shfl_xor((int)kmn, dlt, KN_WAVE));
            kmx = max(kmx, (uint32_t)__shfl_xor((int)kmx, dlt, KN_WAVE));
        }
        if ((tid & (KN_WAVE - 1)) == 0) {
            s_kmin[tid / KN_WAVE] = kmn;
            s_kmax[tid / KN_WAVE] = kmx;
        }
        __syncthreads();
        uint32_t lo = s_kmin[0], hi = s_kmax[0];
        for (int w = 1; w < MF_THREADS / KN_WAVE; w++) {
            lo = min(lo, s_kmin[w]);
            hi = max(hi, s_kmax[w]);
        }
        for (int it = 0; it < 32 && lo < hi; it++) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            int c = 0;
#pragma unroll
            for (int u = 0; u < MF_PRUNE_PER_THREAD; u++) {
                c += pk[u] <= mid ? 1 : 0;
            }
#pragma unroll
            for (int dlt = KN_WAVE / 2; dlt > 0; dlt >>= 1) {
                c += __shfl_xor(c, dlt, KN_WAVE);
            }
            if ((tid & (KN_WAVE - 1)) == 0 && c) {
                atomicAdd(&s_step[it], c);
            }
            __syncthreads();
            const int tot = s_step[it];
            if (lo < hi) {
                if (tot >= k) {
                    hi = mid;
                } else {
                    lo = mid + 1;
                }
            }
        }
        const float tau2 = dist_key_inv<IS_L2>(lo);
        // eps_max dominates the eps of EVERY emission of this query (pq_filter.hip: eps_base + 64 u (|dis0| + |tau| [+ |mu|])):
        // |dis0| <= cmax; the emission's tau lies between the sample bound gthr and the final k-th pessimistic distance
        // tau2 (a histogram edge is only read once k candidates sit below it), so |tau| <= max(|gthr|, |tau2|); the
        // integer form's per-query offset sum rides in [q][1] of its record
        float tau_abs = fmaxf(gthr_abs, fabsf(tau2));
        if (KIND == 2 && n_head < n && a.gmeta != nullptr) {
            // (tau2 then comes from the head only and may lie above the full list's: every threshold a unit used is gthr or
            // a histogram edge, and the edges start at the sample's best key)
            const uint2 mt = a.gmeta[q];
            if (mt.y != KN_HIST_OFF) {
                tau_abs = fmaxf(tau_abs, fabsf(dist_key_inv<IS_L2>(mt.x)));
            }
        }
        float eps_max = eps_q + 64.0f * 5.9604645e-8f * (cmax + tau_abs + eps_mu);
        if (KIND == 1) {
            // fp32 rows: every emission of this query used the one eps of mscan_flat*_unit (pess = approx widened by it)
            const float qn = a.qnorm[q];
            eps_max = a.eps_scale * (IS_L2 ? (qn + a.xnorm_max) : sqrtf(qn * a.xnorm_max)) + 1e-30f;
        }
        if (KIND == 3) {
            // SQ8 codes: the eps of an emission belongs to its (query, list) pair; the units publish the largest one
            // (mscan_sq8_unit, atomic max of the bit patterns of non-negative floats)
            eps_max = __uint_as_float(a.eps_max[q]);
        }
        if (eps_max < INFINITY && tau2 == tau2 && fabsf(tau2) < FLT_MAX) {
            if (tid == 0) {
                s_cnt = 0;
            }
            __syncthreads(); // (every candidate is in registers: the list is rewritten in place)
#pragma unroll
            for (int u = 0; u < MF_PRUNE_PER_THREAD; u++) {
                const int ci = tid + u * MF_THREADS;
                const bool keep = ci < n_head && (IS_L2 ? (pp[u] - 2.0f * eps_max <= tau2) : (pp[u] + 2.0f * eps_max >= tau2));
                if (keep) {
                    const int j = atomicAdd(&s_cnt, 1);
                    a.cand[q * (int64_t)a.cap + j] = pc[u];
                }
            }
            // the candidates past the head: kept ones land below everything still to be read (s_cnt <= candidates seen)
#pragma unroll 1
            for (int ci0 = n_head; ci0 < n; ci0 += MF_THREADS) {
                const int ci = ci0 + tid;
                int64_t c = 0;
                float p = worst_dist<IS_L2>();
                if (ci < n) {
                    c = a.cand[q * (int64_t)a.cap + ci];
                    p = a.cand_pess[q * (int64_t)a.cap + ci];
                }
                __syncthreads(); // (this row of candidates is in registers before any thread may overwrite part of it)
                if (ci < n && (IS_L2 ? (p - 2.0f * eps_max <= tau2) : (p + 2.0f * eps_max >= tau2))) {
                    const int j = atomicAdd(&s_cnt, 1);
                    a.cand[q * (int64_t)a.cap + j] = c;
                }
            }
            __syncthreads();
            n = s_cnt;
            __syncthreads();
        }
    }
    if (retry_prep && n < k) {
        if (tid == 0) {
            atomicAdd(counters + 1, 1ull);
            a.overflow[q] = 3; // (flag 1, counted)
        }
        return; // no bound and nothing gathered: the exact kernels
    }
    if (tid == 0 && !retry_prep) {
        atomicAdd(counters, 1ull);
        atomicAdd(counters + 2, (unsigned long long)n_filter);
        atomicAdd(counters + 3, (unsigned long long)n);
    }
    int P = 2;
    while ((P < n + k || P <= k) && P < P_max) { // (P > k always: a query without candidates and a power-of-two k left
        P <<= 1;                                   //  P == k, chunk 0 and the round loop below spinning for ever)
    }
    const int chunk = P - k; // candidates per round, >= 1 (P_max >= 2 k)
    // LDS: tie[P_max] (u64) | key[P_max] (u32) | query [dq] floats (| vmin, vdiff for SQ8)
    unsigned long long* tie = reinterpret_cast<unsigned long long*>(smem);
    uint32_t* key = reinterpret_cast<uint32_t*>(smem + (size_t)P_max * 8);
    float* sq = reinterpret_cast<float*>(smem + (size_t)P_max * 12);
    const int dq = KIND == 3 ? a.nchunk * 16 : a.nchunk * 4;
    float* svmin = sq + dq;
    float* svdiff = svmin + dq;
    for (int i = tid; i < dq; i += MF_THREADS) {
        sq[i] = (i < d) ? a.queries[q * d + i] : 0.f;
        if (KIND == 3) {
            svmin[i] = (i < d) ? a.trained[i] : 0.f;
            svdiff[i] = (i < d) ? a.trained[d + i] : 0.f;
        }
    }
    if (KIND == 3 && tid < 256) {
        tab[tid] = __fdiv_rn((float)tid + 0.5f, 255.0f); // Codec8bit::decode_component
    }
    for (int e = tid; e < k; e += MF_THREADS) { // the running top-k starts empty
        key[e] = 0xffffffffu;
        tie[e] = ~0ull;
    }
    __syncthreads();
    for (int base = 0; base < n || base == 0; base += chunk) {
        for (int e = k + tid; e < P; e += MF_THREADS) {
            uint32_t kk = 0xffffffffu;
            unsigned long long tt = ~0ull;
            const int ci = base + (e - k);
            if (ci < n) {
                const int64_t c = a.cand[q * (int64_t)a.cap + ci];
                const int slot = (int)(c >> 32);
                const int64_t pos = (int64_t)(uint32_t)c;
                const int64_t list = keys[q * nprobe + slot];
                const int64_t blk = KIND == 2 ? 0 : a.list_blk_off[list] + (pos >> 6);
                const int r = (int)(pos & 63);
                float acc = 0.f;
                // A candidate's row is gathered in 16-byte pieces 1 KiB apart (interleaved blocks): MF_MLP of them are
                // requested at a time -- one dependent load per piece made this stage latency-bound (C5: 25 ms).
                if (KIND == 1) {
                    const float4* p = reinterpret_cast<const float4*>(a.rows) + blk * (int64_t)a.nchunk * 64 + r;
                    for (int c0 = 0; c0 < a.nchunk; c0 += MF_MLP) {
                        float4 yy[MF_MLP];
#pragma unroll
                        for (int u = 0; u < MF_MLP; u++) {
                            yy[u] = p[(int64_t)min(c0 + u, a.nchunk - 1) * 64];
                        }
#pragma unroll
                        for (int u = 0; u < MF_MLP; u++) {
                            const int c4 = c0 + u;
                            if (c4 < a.nchunk) {
                                const float4 y = yy[u];
                                const float4 x = *reinterpret_cast<const float4*>(sq + c4 * 4);
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
                        }
                    }
                } else if (KIND == 2) {
                    // ADC in the reference's order (pq_scan_q4.hip header): table entry = term2 + (-2) <q_m, cb> (L2,
                    // precomputed table) / <q_m, cb> (IP) / ||(q - c_list)_m - cb||^2 (L2, residual tables), inner
                    // products and squared distances accumulated from 0 in dimension order, the entries summed from 0
                    // in m order, the coarse term added last
                    const uint4* cp = reinterpret_cast<const uint4*>(a.pq_codes + (a.list_row_off[list] + pos) * 32);
                    const uint4 cw[2] = {cp[0], cp[1]};
                    const uint32_t ww[8] = {cw[0].x, cw[0].y, cw[0].z, cw[0].w, cw[1].x, cw[1].y, cw[1].z, cw[1].w};
                    // The term-2 entry PT[list][code][m] = ||cb||^2 + 2 <c_list,m , cb> is recomputed here with the very
                    // operations of pq_precomp_table_kernel (pq_scan.hip) from the list's centroid -- 512 contiguous
                    // bytes per candidate -- instead of 32 four-byte gathers out of the list's 32 KB table (a cache
                    // line each: the finish was bound by them, round-3 profile).  Same bits as the stored table.
                    const float* cen = a.centroids + list * d;
#pragma unroll 16
                    for (int m = 0; m < 32; m++) {
                        const int code = (int)((ww[m >> 2] >> (8 * (m & 3))) & 0xffu);
                        const float4 y = a.pq_cb_t[code * 32 + m];
                        float4 x = *reinterpret_cast<const float4*>(sq + m * 4);
                        float t;
                        if (a.pq_lut_mode == PQ_LUT_RESIDUAL) {
                            const float4 cl = *reinterpret_cast<const float4*>(cen + m * 4);
                            x = make_float4(fsub_x(x.x, cl.x), fsub_x(x.y, cl.y), fsub_x(x.z, cl.z), fsub_x(x.w, cl.w));
                            t = l2_step(0.f, x.x, y.x);
                            t = l2_step(t, x.y, y.y);
                            t = l2_step(t, x.z, y.z);
                            t = l2_step(t, x.w, y.w);
                        } else {
                            t = ip_step(0.f, x.x, y.x);
                            t = ip_step(t, x.y, y.y);
                            t = ip_step(t, x.z, y.z);
                            t = ip_step(t, x.w, y.w);
                            if (a.pq_lut_mode == PQ_LUT_PRECOMP) {
                                const float4 cl = *reinterpret_cast<const float4*>(cen + m * 4);
                                float nrm = ip_step(0.f, y.x, y.x);
                                nrm = ip_step(nrm, y.y, y.y);
                                nrm = ip_step(nrm, y.z, y.z);
                                nrm = ip_step(nrm, y.w, y.w);
                                float ipc = ip_step(0.f, cl.x, y.x);
                                ipc = ip_step(ipc, cl.y, y.y);
                                ipc = ip_step(ipc, cl.z, y.z);
                                ipc = ip_step(ipc, cl.w, y.w);
                                const float t2 = fadd_x(nrm, fmul_x(2.0f, ipc));
                                t = fadd_x(t2, fmul_x(-2.0f, t));
                            }
                        }
                        acc = fadd_x(acc, t);
                    }
                    if (a.pq_lut_mode != PQ_LUT_RESIDUAL) {
                        acc = fadd_x(coarse_dis[q * nprobe + slot], acc);
                    }
                } else {
                    const uint4* p = reinterpret_cast<const uint4*>(a.rows) + blk * (int64_t)a.nchunk * 64 + r;
                    const float* cen = a.centroids + list * d;
                    for (int c0 = 0; c0 < a.nchunk; c0 += MF_MLP) {
                        uint4 wv[MF_MLP];
#pragma unroll
                        for (int u = 0; u < MF_MLP; u++) {
                            wv[u] = p[(int64_t)min(c0 + u, a.nchunk - 1) * 64];
                        }
#pragma unroll
                        for (int u = 0; u < MF_MLP; u++) {
                            const int c16 = c0 + u;
                            if (c16 < a.nchunk) {
                                const uint32_t ww[4] = {wv[u].x, wv[u].y, wv[u].z, wv[u].w};
#pragma unroll
                                for (int e2 = 0; e2 < 16; e2++) {
                                    const int i = c16 * 16 + e2;
                                    const uint32_t code = (ww[e2 >> 2] >> (8 * (e2 & 3))) & 0xffu;
                                    const float x = fadd_x(svmin[i], fmul_x(tab[code], svdiff[i]));
                                    if (IS_L2) {
                                        // padded dims: y = 0, x = 0: they add exactly +0
                                        const float y = (i < d) ? fsub_x(sq[i], cen[i]) : 0.f;
                                        acc = l2_step(acc, y, x);
                                    } else {
                                        acc = ip_step(acc, sq[i], x);
                                    }
                                }
                            }
                        }
                    }
                    if (!IS_L2) {
                        acc = fadd_x(coarse_dis[q * nprobe + slot], acc);
                    }
                }
                const int64_t id = a.ids[a.list_row_off[list] + pos];
                kk = dist_key<IS_L2>(acc);
                tt = IS_L2 ? (unsigned long long)id : ~(unsigned long long)id;
            }
            key[e] = kk;
            tie[e] = tt;
        }
        __syncthreads();
        for (int size = 2; size <= P; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int t = tid; t < P / 2; t += MF_THREADS) {
                    const int lo = (t / stride) * stride * 2 + (t % stride);
                    const int hi = lo + stride;
                    const bool up = ((lo & size) == 0);
                    const uint32_t ka = key[lo], kb = key[hi];
                    const unsigned long long ta = tie[lo], tb = tie[hi];
                    const bool gt = (ka > kb) || (ka == kb && ta > tb);
                    if (gt == up) {
                        key[lo] = kb;
                        key[hi] = ka;
                        tie[lo] = tb;
                        tie[hi] = ta;
                    }
                }
                __syncthreads();
            }
        }
    }
    if (retry_prep) {
        if (tid == 0) {
            if (!(key[k - 1] == 0xffffffffu && tie[k - 1] == ~0ull)) {
                a.gthr_rw[q] = dist_key_inv<IS_L2>(key[k - 1]);
                a.cand_cnt[q] = 0;
                a.overflow[q] = 2;
            } else {
                atomicAdd(counters + 1, 1ull); // (the exact kernels; 3 = counted)
                a.overflow[q] = 3;
            }
        }
        return;
    }
    for (int e = tid; e < k; e += MF_THREADS) {
        float dd = worst_dist<IS_L2>();
        int64_t ii = -1;
        if (!(key[e] == 0xffffffffu && tie[e] == ~0ull)) {
            dd = dist_key_inv<IS_L2>(key[e]);
            ii = (int64_t)(IS_L2 ? tie[e] : ~tie[e]);
        }
        out_d[q * k + e] = dd;
        out_i[q * k + e] = ii;
    }
    if (pass == 2 && tid == 0) {
        a.overflow[q] = 0; // finished: not part of the exact fallback's merge
    }
}

// ---- overflowed queries -> compact one-query work items for the exact kernels ------------------------------------
// thread per (query, slot): pairs of the queries whose flag == want become items {list, 1 pair} (= one-query units of
// the filter kernels for the retry round, want = 2; items of the exact kernels, want = 1); pairs of empty / invalid
// lists get their partial slot marked empty where asked (as the work table does).  *nitems is zeroed first.
__global__ void ms_flag_pairs_kernel(const int32_t* __restrict__ overflow, int want, const int64_t* __restrict__ keys,
                                     int64_t nq, int nprobe, int64_t nlist, const int64_t* __restrict__ list_len, int k,
                                     KnItem* __restrict__ items, KnPair* __restrict__ pairs, int64_t* __restrict__ nitems,
                                     int64_t* __restrict__ empty_mark) {
    if (overflow[nq] == 0) {
        return;
    }
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nq * nprobe) {
        return;
    }
    const int64_t q = t / nprobe;
    const int flag = overflow[q];
    if ((flag == 3 ? 1 : flag) != want) { // (3: flag 1 after the finish kernel's first pass counted it)
        return;
    }
    const int64_t key = keys[t];
    if (key < 0 || key >= nlist || list_len[key] == 0) {
        if (empty_mark != nullptr) {
            empty_mark[t * k] = -1;
        }
        return;
    }
    const int64_t pos = (int64_t)atomicAdd(reinterpret_cast<unsigned long long*>(nitems), 1ull);
    KnPair p;
    p.q = (int32_t)q;
    p.slot = (int32_t)(t % nprobe);
    pairs[pos] = p;
    KnItem it;
    it.list = (int32_t)key;
    it.npair = 1;
    it.pair0 = pos;
    items[pos] = it;
}

hipError_t launch_ms_flag_pairs(const int32_t* overflow, int want, const int64_t* keys, int64_t nq, int nprobe,
                                int64_t nlist, const int64_t* list_len, int k, KnItem* items, KnPair* pairs,
                                int64_t* nitems, int64_t* empty_mark, hipStream_t s) {
    hipError_t e = hipMemsetAsync(nitems, 0, sizeof(int64_t), s);
    if (e != hipSuccess || nq <= 0) {
        return e;
    }
    hipLaunchKernelGGL(ms_flag_pairs_kernel, dim3((unsigned)((nq * nprobe + 255) / 256)), dim3(256), 0, s, overflow, want,
                       keys, nq, nprobe, nlist, list_len, k, items, pairs, nitems, empty_mark);
    return hipGetLastError();
}

// ---- host launchers ---------------------------------------------------------------------------------------------
// queries per unit: filter pass / sample pass
int mscan_queries_per_unit(int kind, bool sample) {
    return kind == 1 ? (sample ? 32 : MS_QT) : kind == 2 ? 8 : MQ_QT; // (kind 2: pq_filter.hip, 8 queries per LUT)
}

size_t mscan_sq8_smem(int nstep) {
    return (size_t)2 * MQ_QT * (nstep * 32 + 8) * 2 + (size_t)MQ_QT * 28;
}

hipError_t launch_mscan_sq8(const MScanArgs& a, bool is_l2, int64_t units_bound, hipStream_t s) {
    if (units_bound <= 0) {
        return hipSuccess;
    }
    const size_t sm = mscan_sq8_smem(a.nstep);
    const bool dump = a.dump != nullptr;
    auto kern = is_l2 ? (dump ? mscan_sq8_kernel<true, true, false> : mscan_sq8_kernel<true, false, false>)
                      : (dump ? mscan_sq8_kernel<false, true, false> : mscan_sq8_kernel<false, false, false>);
    if (a.unit_loop && !dump) { // the retry round's one-query units
        kern = is_l2 ? mscan_sq8_kernel<true, false, true> : mscan_sq8_kernel<false, false, true>;
    }
    if (sm > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) {
            return e;
        }
    }
    const int64_t grid = a.unit_loop ? std::min<int64_t>(units_bound, 1024) : ((units_bound + 7) / 8) * 8;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(MQ_THREADS), sm, s, a);
    return hipGetLastError();
}

size_t mscan_flat_smem(int nstep) {
    return (size_t)MS_QT * (nstep * 16 + 4) * 4 + (size_t)MS_QT * 16;
}

int mscan_sample_rows() {
    return MS_SAMPLE;
}

hipError_t launch_mscan_flat(const MScanArgs& a, bool is_l2, int64_t units_bound, hipStream_t s) {
    if (units_bound <= 0) {
        return hipSuccess;
    }
    const bool dump = a.dump != nullptr;
    const size_t sm = dump ? (size_t)32 * (a.nstep * 16 + 4) * 4 + (size_t)32 * 16 : mscan_flat_smem(a.nstep);
    auto kern = is_l2 ? (dump ? mscan_flat_kernel<true, true, 1, false> : mscan_flat_kernel<true, false, MS_NQT, false>)
                      : (dump ? mscan_flat_kernel<false, true, 1, false> : mscan_flat_kernel<false, false, MS_NQT, false>);
    if (a.unit_loop && !dump) { // the retry round's one-query units
        kern = is_l2 ? mscan_flat_kernel<true, false, MS_NQT, true> : mscan_flat_kernel<false, false, MS_NQT, true>;
    }
    if (sm > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) {
            return e;
        }
    }
    const int64_t grid = a.unit_loop ? std::min<int64_t>(units_bound, 2048) : ((units_bound + 7) / 8) * 8;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(MS_THREADS), sm, s, a);
    return hipGetLastError();
}

// LDS entries of the finish kernel's sort: k running best + a chunk of candidates
int mscan_finish_pmax(int cap, int k) {
    int P = 1024;
    while (P < 2 * k) {
        P <<= 1;
    }
    (void)cap;
    return P;
}

hipError_t launch_mscan_finish(const MScanArgs& a, int kind, bool is_l2, const int64_t* keys, const float* coarse_dis,
                               int nprobe, int k, float* out_d, int64_t* out_i, unsigned long long* counters, int pass,
                               hipStream_t s) {
    if (a.nq <= 0) {
        return hipSuccess;
    }
    const int P_max = mscan_finish_pmax(a.cap, k);
    const int dq = kind == 3 ? a.nchunk * 16 : a.nchunk * 4;
    const size_t sm = (size_t)P_max * 12 + (size_t)dq * 4 * (kind == 3 ? 3 : 1);
#define MF_LAUNCH(L2_, KIND_)                                                                                   \
    do {                                                                                                        \
        auto kern = mscan_finish_kernel<L2_, KIND_>;                                                            \
        if (sm > 48 * 1024) {                                                                                   \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                             \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);            \
            if (e != hipSuccess) return e;                                                                      \
        }                                                                                                       \
        hipLaunchKernelGGL(kern, dim3((unsigned)a.nq), dim3(MF_THREADS), sm, s, a, keys, coarse_dis, nprobe, k, \
                           P_max, out_d, out_i, counters, pass);                                                \
    } while (0)
    if (kind == 1) {
        if (is_l2) MF_LAUNCH(true, 1); else MF_LAUNCH(false, 1);
    } else if (kind == 2) {
        if (is_l2) MF_LAUNCH(true, 2); else MF_LAUNCH(false, 2);
    } else {
        if (is_l2) MF_LAUNCH(true, 3); else MF_LAUNCH(false, 3);
    }
#undef MF_LAUNCH
    return hipGetLastError();
}

} // namespace knhip
