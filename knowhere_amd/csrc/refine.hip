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
// Quantised refine stores (Knowhere's `refine_type`, src/index/refine/refine_utils.cc:38-160: the refine index is an
// IndexScalarQuantizer instead of IndexFlat): rows kept as fp16 / bf16 / per-dimension 8-bit codes; the distance is the
// scalar quantizer's DistanceComputer (impl/scalar_quantizer/distance_computers.h, SIMDLevel::NONE): for i = 0 .. d - 1:
// x_i = reconstruct_component(code, i), L2: tmp = q_i - x_i, accu += tmp * tmp; IP: accu += q_i * x_i -- the same
// sequential sum with the decode in front (fp16: IEEE half -> float; bf16: bits << 16; 8 bit: vmin_i + vdiff_i *
// ((code_i + 0.5) / 255), quantizers.h:139-145, codecs.h:37-41).
// Distances use the reference's sequential fp32 order (fvec_L2sqr / fvec_inner_product), so they
// are bit-equal to the CPU refine.  288 GB of HBM3E holds the raw vectors of a 100M x 128 index
// (51 GB) next to its codes, so refine is a ~0.5 GB random gather per 10k-query batch: noise
// next to the scan, and what lifts PQ32 recall@10 past 0.95.

#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace knhip {

constexpr int RF_PITCH = 17 * 16; // bytes per staged row piece set (16 pieces + one of padding)
constexpr uint32_t REFINE_NOT_HERE = 0xffffffffu; // a NaN pattern no arithmetic produces: "this shard does not hold the row"

// ROWT: 0 fp32 rows, 1 fp16, 2 bf16, 3 per-dimension 8-bit codes (sq = vmin[d], vdiff[d]), 4 per-dimension 6-bit codes (four
// per three bytes: Codec6bit, codecs.h:63-118), 5 signed bytes stored + 128 (Quantizer8bitDirectSigned, quantizers.h:350-379),
// 6 4-bit codes with ONE range for all dimensions (sq = {vmin, vdiff}; Codec4bit, codecs.h:43-59, QuantizerTemplate UNIFORM)
template <int ROWT>
__device__ __forceinline__ float refine_row_value(const void* row, int i, const float* __restrict__ sq, int d) {
    if (ROWT == 1) {
        return (float)reinterpret_cast<const _Float16*>(row)[i]; // (exact: decode_fp16, utils/fp16-inl.h:88-108)
    }
    if (ROWT == 2) {
        return __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(row)[i] << 16);
    }
    if (ROWT == 3) {
        const float xi = __fdiv_rn(fadd_x((float)reinterpret_cast<const uint8_t*>(row)[i], 0.5f), 255.0f);
        return fadd_x(sq[i], fmul_x(xi, sq[d + i]));
    }
    if (ROWT == 4) {
        const uint8_t* g = reinterpret_cast<const uint8_t*>(row) + (i >> 2) * 3;
        uint32_t bits;
        switch (i & 3) {
            case 0: bits = g[0] & 0x3fu; break;
            case 1: bits = ((uint32_t)g[0] >> 6) | (((uint32_t)g[1] & 0xfu) << 2); break;
            case 2: bits = ((uint32_t)g[1] >> 4) | (((uint32_t)g[2] & 3u) << 4); break;
            default: bits = (uint32_t)g[2] >> 2; break;
        }
        const float xi = __fdiv_rn(fadd_x((float)bits, 0.5f), 63.0f);
        return fadd_x(sq[i], fmul_x(xi, sq[d + i]));
    }
    if (ROWT == 5) {
        return (float)((int)reinterpret_cast<const uint8_t*>(row)[i] - 128);
    }
    if (ROWT == 6) {
        const uint32_t bits = ((uint32_t)reinterpret_cast<const uint8_t*>(row)[i >> 1] >> ((i & 1) << 2)) & 0xfu;
        const float xi = __fdiv_rn(fadd_x((float)bits, 0.5f), 15.0f);
        return fadd_x(sq[0], fmul_x(xi, sq[1]));
    }
    return reinterpret_cast<const float*>(row)[i];
}

// bytes of one stored row
__host__ __device__ inline int64_t refine_row_bytes(int row_type, int d) {
    return row_type == 4   ? ((int64_t)d * 6 + 7) / 8
           : row_type == 6 ? ((int64_t)d * 4 + 7) / 8
           : (row_type == 3 || row_type == 5) ? (int64_t)d : row_type == 0 ? 4 * (int64_t)d : 2 * (int64_t)d;
}

template <bool IS_L2, int R, int ROWT>
__global__ __launch_bounds__(256) void refine_kernel(const float* __restrict__ base, int64_t nbase,
                                                     int64_t id_base, int d,
                                                     const float* __restrict__ queries, int64_t nq,
                                                     const int64_t* __restrict__ cand, int kbase, int k,
                                                     float* __restrict__ out_d,
                                                     int64_t* __restrict__ out_i, const float* __restrict__ sq_trained,
                                                     const float* __restrict__ dist_in, float* __restrict__ dist_out, int coop) {
    // Sharded refine (the raw rows cut into one id range per shard): the re-scored candidates pass the reference's heap in
    // CANDIDATE order whoever holds their rows, so the selection needs every candidate's distance.  dist_out != nullptr:
    // only the distances of the candidates held here are written ([nq][kbase]; REFINE_NOT_HERE elsewhere), nothing is
    // selected; dist_in != nullptr: the distances are given (the shards' arrays combined), only the selection runs.
    extern __shared__ __align__(16) float sq[]; // [4][d] queries, then [4][kbase] the candidates' distances
    const int lane = lane_id();
    const int wave = threadIdx.x / KN_WAVE;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    const bool live = q < nq;
    float* myq = sq + wave * d;
    float* mydis = sq + 4 * d + wave * kbase;
    if (live && dist_in == nullptr) {
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
        bool ok = lane < nvalid && id >= 0 && (id - id_base) >= 0 && (id - id_base) < nbase;
        float acc = 0.f;
        if (dist_in != nullptr) {
            acc = c < kbase ? dist_in[q * kbase + c] : 0.f;
            ok = lane < nvalid && id >= 0 && __float_as_uint(acc) != REFINE_NOT_HERE; // (held by no shard: skipped)
        }
        skipped = skipped || __ballot(lane < nvalid && !ok) != 0ull;
        if (ok && ROWT != 0 && dist_in == nullptr) {
            // quantised rows: decode + accumulate, element by element in the reference's order
            constexpr int ESZ = (ROWT == 3 || ROWT == 5) ? 1 : 2;
            constexpr int EPC = 16 / ESZ; // elements per 16-byte piece
            const unsigned char* y = reinterpret_cast<const unsigned char*>(base) + (id - id_base) * refine_row_bytes(ROWT, d);
            int i = 0;
            // (6-bit codes straddle bytes, 4-bit codes share them: the element loop below)
            if (ROWT != 4 && ROWT != 6 && ((d * ESZ) & 15) == 0 && (reinterpret_cast<uintptr_t>(base) & 15) == 0) {
                // 16-byte loads, eight in flight per lane (as the fp32 rows below); decoded and added in element order
                const uint4* y4 = reinterpret_cast<const uint4*>(y);
                const int n16 = (d * ESZ) >> 4;
                for (int j = 0; j < n16; j += 8) {
                    uint4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        v[u] = j + u < n16 ? y4[j + u] : make_uint4(0u, 0u, 0u, 0u);
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        if (j + u < n16) {
                            const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                            for (int e = 0; e < EPC; e++) {
                                const int col = (j + u) * EPC + e;
                                float x;
                                if (ROWT == 3) {
                                    const float xi = __fdiv_rn(fadd_x((float)((w[e >> 2] >> (8 * (e & 3))) & 0xffu), 0.5f), 255.0f);
                                    x = fadd_x(sq_trained[col], fmul_x(xi, sq_trained[d + col]));
                                } else if (ROWT == 5) {
                                    x = (float)((int)((w[e >> 2] >> (8 * (e & 3))) & 0xffu) - 128);
                                } else {
                                    const uint32_t hb = (w[e >> 1] >> (16 * (e & 1))) & 0xffffu;
                                    x = ROWT == 1 ? (float)__builtin_bit_cast(_Float16, (uint16_t)hb) : __uint_as_float(hb << 16);
                                }
                                acc = IS_L2 ? l2_step(acc, myq[col], x) : ip_step(acc, myq[col], x);
                            }
                        }
                    }
                }
                i = d;
            }
            for (; i < d; i++) {
                const float x = refine_row_value<ROWT>(y, i, sq_trained, d);
                acc = IS_L2 ? l2_step(acc, myq[i], x) : ip_step(acc, myq[i], x);
            }
        }
        if (ROWT == 0 && dist_in == nullptr && coop) {
            // fp32 rows, gathered by the wave TOGETHER: 16 lanes read 256 contiguous bytes of one row, an instruction covers
            // four rows (four pages) instead of 64 -- with lane = row every wave-level load touched 64 pages, and the random
            // 512-byte rows of a 51 GB array came in at 1.2 TB/s against 4.9 TB/s for whole-row requests
            // (tools/ubench/row_gather.hip, profiles/r05_ubench_row_gather.log): the translations, not the bytes.  The pieces
            // go through a per-wave LDS staging area (row pitch 17 x 16 bytes: conflict-free when lane = row reads them back)
            // and are accumulated by the candidate's own lane in dimension order, as before.
            const int64_t myrow = ok ? (id - id_base) : 0; // (a lane without a candidate names row 0: read, never used)
            const float4* b4 = reinterpret_cast<const float4*>(base);
            const float4* q4 = reinterpret_cast<const float4*>(myq);
            const int n4 = d >> 2;
            unsigned char* st = reinterpret_cast<unsigned char*>(sq + 4 * d + 4 * kbase) + wave * (KN_WAVE * RF_PITCH);
            const int sub = lane >> 4, pl = lane & 15;
            for (int p0 = 0; p0 < n4; p0 += 16) { // 16 pieces = 64 dimensions of every row
                float4 v[16];
                const int piece = min(p0 + pl, n4 - 1);
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const int64_t rr = __shfl(myrow, 4 * u + sub, KN_WAVE);
                    v[u] = 4 * u < nvalid ? b4[rr * n4 + piece] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    *reinterpret_cast<float4*>(st + (4 * u + sub) * RF_PITCH + pl * 16) = v[u];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const float4* mine = reinterpret_cast<const float4*>(st + lane * RF_PITCH);
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    if (p0 + e < n4) {
                        const float4 y = mine[e];
                        const float4 x = q4[p0 + e];
                        acc = IS_L2 ? l2_step(acc, x.x, y.x) : ip_step(acc, x.x, y.x);
                        acc = IS_L2 ? l2_step(acc, x.y, y.y) : ip_step(acc, x.y, y.y);
                        acc = IS_L2 ? l2_step(acc, x.z, y.z) : ip_step(acc, x.z, y.z);
                        acc = IS_L2 ? l2_step(acc, x.w, y.w) : ip_step(acc, x.w, y.w);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier(); // (the next chunk overwrites the staging area)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            acc = ok ? acc : 0.f;
        } else if (ok && ROWT == 0 && dist_in == nullptr) {
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
        if (dist_out != nullptr) {
            if (c < kbase) {
                dist_out[q * kbase + c] = ok ? acc : __uint_as_float(REFINE_NOT_HERE);
            }
            continue;
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
    if (dist_out != nullptr) {
        // (slots behind the end of the candidate list)
        for (int c = nend + lane; c < kbase; c += KN_WAVE) {
            dist_out[q * kbase + c] = __uint_as_float(REFINE_NOT_HERE);
        }
        return;
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
                         int64_t* out_i, hipStream_t s, int row_type, const float* sq_trained, const float* dist_in,
                         float* dist_out) {
    if (nq <= 0) {
        return hipSuccess;
    }
    if (row_type < 0 || row_type > 6 ||
        ((row_type == 3 || row_type == 4 || row_type == 6) && sq_trained == nullptr && dist_in == nullptr) ||
        (dist_in != nullptr && dist_out != nullptr)) {
        return hipErrorInvalidValue;
    }
    const unsigned grid = (unsigned)((nq + 3) / 4);
    // fp32 rows of 16-byte multiples are gathered cooperatively through a per-wave staging area (refine_kernel)
    // (nbase == 0: no row exists -- the cooperative gather names row 0 for lanes without a candidate and would read it;
    // a base pointer may legitimately be null then.  ADVICE round 5)
    int coop = (row_type == 0 && dist_in == nullptr && nbase > 0 && base != nullptr && (d & 3) == 0 &&
                (reinterpret_cast<uintptr_t>(base) & 15) == 0) ? 1 : 0;
    const size_t sm_plain = ((size_t)4 * d + (size_t)4 * kbase) * sizeof(float);
    size_t sm = sm_plain + (coop ? (size_t)4 * KN_WAVE * RF_PITCH : 0);
    if (sm > 160 * 1024 && coop) {
        // the staging area (69.6 KB) does not fit beside four long queries: the lane-per-row gather still does
        coop = 0;
        sm = sm_plain;
    }
    if (sm > 160 * 1024) {
        return hipErrorInvalidValue; // (four queries + their candidates' distances do not fit the CU's LDS)
    }
    // (above the default dynamic-LDS limit -- d >= ~3 k with k_base up to 1024 -- the limit is raised per instantiation)
#define KN_REFINE_ONE(L2_, ROWT_)                                                                                     \
    {                                                                                                                 \
        auto kern_ = refine_kernel<L2_, R_, ROWT_>;                                                                   \
        if (sm > 48 * 1024) {                                                                                         \
            const hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kern_),                           \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);           \
            if (e_ != hipSuccess) {                                                                                   \
                return e_;                                                                                            \
            }                                                                                                         \
        }                                                                                                             \
        hipLaunchKernelGGL(kern_, dim3(grid), dim3(256), sm, s, base, nbase, id_base, d, queries, nq, cand, kbase, k, \
                           out_d, out_i, sq_trained, dist_in, dist_out, coop);                                        \
    }
#define KN_REFINE_LAUNCH(ROWT_)                                                                                       \
    KN_DISPATCH_R(k, {                                                                                                \
        if (is_l2) {                                                                                                  \
            KN_REFINE_ONE(true, ROWT_)                                                                                \
        } else {                                                                                                      \
            KN_REFINE_ONE(false, ROWT_)                                                                               \
        }                                                                                                             \
    })
    switch (row_type) {
        case 1: KN_REFINE_LAUNCH(1); break;
        case 2: KN_REFINE_LAUNCH(2); break;
        case 3: KN_REFINE_LAUNCH(3); break;
        case 4: KN_REFINE_LAUNCH(4); break;
        case 5: KN_REFINE_LAUNCH(5); break;
        case 6: KN_REFINE_LAUNCH(6); break;
        default: KN_REFINE_LAUNCH(0); break;
    }
#undef KN_REFINE_LAUNCH
#undef KN_REFINE_ONE
    return hipGetLastError();
}

// ---- encoders of the quantised refine stores (ScalarQuantizer::compute_codes: QuantizerFP16 / QuantizerBF16) -------------
// bf16: utils/bf16.h:28-33 encode_bf16 = (bits + 0x8000) >> 16.
// fp16: the reference's scalar encode_fp16 (utils/fp16-inl.h:32-86, what SIMDLevel::NONE code is built with): the low 12 bits
// are masked off, the value is rescaled by 2^-112 in fp32 (half subnormals become fp32 subnormals), half a unit is added and
// bits 13.. are taken -- round HALF UP of the 11-bit truncation, not round-to-nearest-even (exact ties go up, so
// __float2half_rn would differ there).  Integer form of the same arithmetic (no dependence on the fp32 denormal mode);
// proven equal to the reference's on all 2^20 classes in tests/test_refine_rows.py.
__device__ __forceinline__ uint16_t encode_fp16_ref(float f) {
    const uint32_t bits = __float_as_uint(f), sign = (bits >> 16) & 0x8000u, fint = bits & 0x7fffffffu;
    if (fint >= 0x7f800000u) {
        return (uint16_t)(sign | (fint > 0x7f800000u ? 0x7e00u : 0x7c00u));
    }
    const uint32_t t = fint & ~0xfffu, E = t >> 23;
    uint32_t b;
    if (E >= 113u) {
        b = t - (112u << 23);
    } else {
        const uint32_t sh = 113u - E;
        b = sh <= 12u ? (((t & 0x7fffffu) | (E ? 0x800000u : 0u)) >> sh) : 0u; // (the shift is exact: 12 zero low bits)
    }
    b = min(b, (31u << 23) - 0x1000u);
    return (uint16_t)(sign | ((b + 0x1000u) >> 13));
}

__global__ void rows_encode16_kernel(const float* __restrict__ x, int64_t n, int bf16, uint16_t* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) {
        return;
    }
    out[t] = bf16 ? (uint16_t)((__float_as_uint(x[t]) + 0x8000u) >> 16) : encode_fp16_ref(x[t]);
}

hipError_t launch_rows_encode16(const float* x, int64_t n_elems, bool bf16, uint16_t* out, hipStream_t s) {
    if (n_elems <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(rows_encode16_kernel, dim3((unsigned)((n_elems + 255) / 256)), dim3(256), 0, s, x, n_elems, bf16 ? 1 : 0,
                       out);
    return hipGetLastError();
}

// QT_6bit (QuantizerTemplate<Codec6bit, NON_UNIFORM>::encode_vector, quantizers.h:124-137): xi = (x - vmin) / vdiff clamped to
// [0, 1] (0 where vdiff == 0), bits = (int)(xi * 63.0) -- a DOUBLE product in the reference, exact for a float in [0, 1], so
// the truncation never sees a rounded-up product --, four codes packed into three bytes.  Thread per group of four dimensions.
__global__ void rows_encode6_kernel(const float* __restrict__ x, int64_t n, int d, const float* __restrict__ trained,
                                    uint8_t* __restrict__ out) {
    const int ngrp = (d + 3) >> 2;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * ngrp) {
        return;
    }
    const int64_t r = t / ngrp;
    const int g = (int)(t % ngrp);
    const int64_t cs = ((int64_t)d * 6 + 7) / 8;
    uint32_t b[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int i = 4 * g + e;
        if (i < d) {
            float xi = 0.f;
            const float vd = trained[d + i];
            if (vd != 0.f) {
                xi = __fdiv_rn(fsub_x(x[r * d + i], trained[i]), vd);
                xi = xi < 0.f ? 0.f : xi;
                xi = xi > 1.0f ? 1.0f : xi;
            }
            b[e] = (uint32_t)(int)((double)xi * 63.0);
        }
    }
    const uint32_t w = b[0] | (b[1] << 6) | (b[2] << 12) | (b[3] << 18);
    uint8_t* o = out + r * cs + (int64_t)g * 3;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if ((int64_t)g * 3 + j < cs) { // (a ragged last group owns fewer than three bytes)
            o[j] = (uint8_t)(w >> (8 * j));
        }
    }
}

// QT_8bit_direct_signed (Quantizer8bitDirectSigned::encode_vector, quantizers.h:362-366): code = (uint8_t)(x + 128); defined
// for values in [-128, 127]
__global__ void rows_encode_i8_kernel(const float* __restrict__ x, int64_t n, uint8_t* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) {
        return;
    }
    out[t] = (uint8_t)(int)fadd_x(x[t], 128.0f);
}

// QT_4bit_uniform (QuantizerTemplate<Codec4bit, UNIFORM>::encode_vector, quantizers.h:76-90): xi = (x - vmin) / vdiff clamped
// to [0, 1] (0 when vdiff == 0), code = (int)(xi * 15.0) -- the DOUBLE product of the reference, codecs.h:51 --, dimension i
// in the low (even i) or high nibble of byte i / 2.  Thread per output byte.
__global__ void rows_encode4u_kernel(const float* __restrict__ x, int64_t n, int d, const float* __restrict__ trained,
                                     uint8_t* __restrict__ out) {
    const int cs = (d + 1) >> 1;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * cs) {
        return;
    }
    const int64_t r = t / cs;
    const int b = (int)(t % cs);
    const float vmin = trained[0], vdiff = trained[1];
    uint32_t byte = 0u;
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const int i = 2 * b + e;
        if (i < d) {
            float xi = 0.f;
            if (vdiff != 0.f) {
                xi = __fdiv_rn(fsub_x(x[r * d + i], vmin), vdiff);
                xi = xi < 0.f ? 0.f : xi;
                xi = xi > 1.0f ? 1.0f : xi;
            }
            byte |= (uint32_t)(int)((double)xi * 15.0) << (4 * e);
        }
    }
    out[t] = (uint8_t)byte;
}

// train_Uniform, RS_quantiles (training.cpp:247-261): order statistics of ALL values by a radix select over their
// order-preserving 32-bit keys, eight bits per pass.  One pass: the 256-bin histograms of the values whose key starts with
// prefix_lo / prefix_hi (the two ranks are walked together: the o-th smallest and the (n - 1 - o)-th).
__global__ __launch_bounds__(256) void rows_key_hist_kernel(const float* __restrict__ x, int64_t n, uint32_t mask,
                                                            uint32_t prefix_lo, uint32_t prefix_hi, int shift,
                                                            unsigned long long* __restrict__ hist) {
    __shared__ uint32_t s_h[2][256];
    s_h[0][threadIdx.x] = 0u;
    s_h[1][threadIdx.x] = 0u;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t b = __float_as_uint(x[i]);
        const uint32_t key = (b & 0x80000000u) ? ~b : (b | 0x80000000u); // ascending with the value
        if ((key & mask) == prefix_lo) {
            atomicAdd(&s_h[0][(key >> shift) & 255u], 1u);
        }
        if ((key & mask) == prefix_hi) {
            atomicAdd(&s_h[1][(key >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    if (s_h[0][threadIdx.x]) {
        atomicAdd(hist + threadIdx.x, (unsigned long long)s_h[0][threadIdx.x]);
    }
    if (s_h[1][threadIdx.x]) {
        atomicAdd(hist + 256 + threadIdx.x, (unsigned long long)s_h[1][threadIdx.x]);
    }
}

hipError_t launch_rows_key_hist(const float* x, int64_t n, uint32_t mask, uint32_t prefix_lo, uint32_t prefix_hi, int shift,
                                unsigned long long* hist, hipStream_t s) {
    if (n <= 0) {
        return hipSuccess;
    }
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(rows_key_hist_kernel, dim3(grid), dim3(256), 0, s, x, n, mask, prefix_lo, prefix_hi, shift, hist);
    return hipGetLastError();
}

hipError_t launch_rows_encode4u(const float* x, int64_t n, int d, const float* trained, uint8_t* out, hipStream_t s) {
    const int64_t nt = n * ((d + 1) >> 1);
    if (nt <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(rows_encode4u_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, x, n, d, trained, out);
    return hipGetLastError();
}

hipError_t launch_rows_encode6(const float* x, int64_t n, int d, const float* trained, uint8_t* out, hipStream_t s) {
    const int64_t nt = n * ((d + 3) >> 2);
    if (nt <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(rows_encode6_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, x, n, d, trained, out);
    return hipGetLastError();
}

hipError_t launch_rows_encode_i8(const float* x, int64_t n_elems, uint8_t* out, hipStream_t s) {
    if (n_elems <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(rows_encode_i8_kernel, dim3((unsigned)((n_elems + 255) / 256)), dim3(256), 0, s, x, n_elems, out);
    return hipGetLastError();
}

} // namespace knhip
