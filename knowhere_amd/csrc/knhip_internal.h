// knowhere_amd/csrc/knhip_internal.h -- what the translation units behind the C ABI (knhip_api*.hip) share: device buffers,
// the per-stream workspace, the index object, the quantised row store, and the internal entry points that cross files.
// Not installed, not part of the ABI (include/knhip.h is).
#pragma once
#include "../../include/knhip.h"
#include "common.h"
#include "kernels.h"
#include "knhip_env.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <random>
#include <string>
#include <vector>

using namespace knhip;

namespace knhip_host {

// sets the thread's last-error text (knhip_last_error) and returns `code` (knhip_api.hip)
int fail(int code, const std::string& msg);

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            return fail(e_ == hipErrorOutOfMemory ? KNHIP_ERR_OUT_OF_MEMORY : KNHIP_ERR_HIP_RUNTIME, \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                        \
        }                                                                                          \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) {
            (void)hipFree(p);
        }
        p = nullptr;
        bytes = 0;
    }
    hipError_t alloc(size_t n) {
        release();
        if (n == 0) {
            n = 16;
        }
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) {
            bytes = n;
        } else {
            p = nullptr;
        }
        return e;
    }
    // grow-only
    hipError_t reserve(size_t n) {
        if (n <= bytes) {
            return hipSuccess;
        }
        return alloc(n);
    }
    template <class T>
    T* as() const {
        return static_cast<T*>(p);
    }
};

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        (void)hipGetDevice(&prev);
        if (prev != dev) {
            (void)hipSetDevice(dev);
        }
    }
    ~DeviceGuard() {
        if (prev >= 0) {
            (void)hipSetDevice(prev);
        }
    }
};

// per-stream scratch; stream order makes reuse by consecutive searches on one stream safe
struct Workspace {
    DevBuf coarse_full;  // [qb][nlist] exact distances
    DevBuf keys;         // [qb][nprobe] int64
    DevBuf cdis;         // [qb][nprobe] float
    DevBuf t2t;          // [qb][256][M]
    DevBuf partial_d;    // [qb][nslot][k]
    DevBuf partial_i;
    DevBuf gthr;         // [qb] shared per-query thresholds
    DevBuf qnorm;        // [qb] ||q||^2 (MFMA coarse prefilter)
    DevBuf cand_keys;    // [qb][ncand]
    DevBuf cand_approx;  // [qb][ncand]
    DevBuf fail_flags;   // [qb] coarse certificate failed -> exact fallback
    DevBuf dump;         // [qb][max list len] rank-0 phase distances (pq_scan_v2 DUMP)
    DevBuf sel_keys;     // [qb][k]
    DevBuf sel_d;        // [qb][k]
    DevBuf rg_seg, rg_cnt, rg_off, rg_tot, rg_out_i, rg_out_d;  // range search scratch
    DevBuf cg_gmin, cg_bound, cg_cnt, cg_qs;                    // coarse prefilter on the bf16 pipe: group minima, bounds, counts, split queries
    DevBuf rg_state, rg_keys_w, rg_cdis_w;                       // rank waves: {empty run, stopped} per query, the wave's lists
    DevBuf recs4;        // [items] flat work records of the persistent 4-query scan (pq_scan_q4)
    DevBuf q4_ctr;       // [8 * 16] per-XCD item counters
    DevBuf ghist;        // [qb][64] per-query candidate histogram (pq_scan_v2 after a rank-0 phase)
    DevBuf gmeta;        // [qb] {first-bin key, shift}
    DevBuf list_count, list_pair_off, list_item_off, list_cursor, pairs, items, nitems;
    // a second work table (row-kind prefilters: the all-probes table is built on the side stream beside the sample pass)
    DevBuf list_count2, list_pair_off2, list_item_off2, list_cursor2, pairs2, items2, nitems2;
    // MFMA prefilter of the IVF-Flat / IVF-SQ8 scans (mfma_scan.hip)
    DevBuf ms_qi, ms_qis, ms_qmu, pq_recs16;                                 // integer form: tables, {step, mu sum, eps, A}, records
    DevBuf rs_ovf;                                                    // coarse stage: rows the two-pass selection left to the radix select
    DevBuf ms_eps_max;                                               // [qb] SQ8: largest emission eps per query (bit pattern)
    DevBuf ms_cand_pess;                                             // [qb][cap] pessimistic distances of the candidates (IVF-PQ)
    DevBuf ms_units, ms_unit_off, ms_nunits, ms_cand, ms_cand_cnt;  // ms_cand_cnt: [qb] counters + [qb + 1] overflow flags
    DevBuf ms_sample_off, ms_nrow;                                   // sample plan: [qb][nprobe] dump columns, [qb] rows
    DevBuf ms_qh, ms_ql, ms_qs;                                      // SQ8 IP: prepared query operands (halves) + sums
                                                                     // (IVF-PQ prefilter: ms_qh = half tables, ms_qs = scales)
    DevBuf pq_recs, pq_ctr;                                          // pq_filter.hip: unit records, per-XCD counters
    DevBuf bf_kth;                                                   // BRUTE_FORCE on the matrix cores: the running k-th best per query over the chunks searched
    DevBuf ms_qh16, ms_qd, pq_spill;                                 // pq_decode.hip: the queries as halves, their error records, parked lanes beyond LDS
    DevBuf rs_sort;                                                  // row selection of more than 16384 keys: sort scratch
    // host-boundary staging
    DevBuf h_queries, h_bitset, h_out_d, h_out_i, h_ref_d, h_ref_i;
    DevBuf tie_d, tie_i, tie_flag, tie_q, tie_r, tie_keys, tie_cdis; // search_batch_ties: k + 1 results, flagged queries
    DevBuf tie_arr_d, tie_arr_i, tie_arr_n;                          // ... their first k arrivals
    std::mutex mu;  // held while a *_device entry point enqueues on this (per-stream) workspace
    // side stream of the IVF-PQ prefilter: the grouping of the pairs by list (work table) runs beside the sample pass
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // A count the host must see in the MIDDLE of a search (the boundary rule's flag count): a one-thread kernel writes it
    // {value, sequence number} into coherent host memory and the host spins on the sequence number -- a
    // hipStreamSynchronize for the same word took ~0.2 ms of an otherwise back-to-back stream (C3: 0.24 of 6.9 ms)
    volatile int32_t* h_word = nullptr;
    int32_t word_seq = 0;
    ~Workspace() {
        if (h_word) (void)hipHostFree(const_cast<int32_t*>(h_word));
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (side) (void)hipStreamDestroy(side);
    }
};

// `s` waits for the side stream's work when the scope ends, on whatever path (the scratch of a workspace is only safe
// to reuse in the order of its own stream)
struct SideJoin {
    Workspace* ws;
    hipStream_t s;
    bool forked = false;
    int join() {
        if (forked) {
            forked = false;
            HIP_TRY(hipStreamWaitEvent(s, ws->ev_join, 0));
        }
        return 0;
    }
    ~SideJoin() {
        (void)join();
    }
};

struct PendingEvent {
    int stage;
    hipEvent_t e0, e1;
};

} // namespace knhip_host
using namespace knhip_host;

struct knhip_index {
    knhip_desc desc{};
    bool is_l2 = true;
    int64_t nlist = 0;
    int d = 0;
    // coarse quantizer
    bool has_coarse = false;
    DevBuf centroids;     // [nlist][d] row major
    DevBuf centroids_il;  // interleaved 64-row blocks
    DevBuf cnorm;         // [nlist] ||c||^2
    DevBuf centroids_bs;  // split bf16 operand rows of the coarse prefilter (coarse_gemm.hip: hi | lo per k slab of 32)
    float cnorm_max = 0.f;
    int coarse_gemm = 2;      // KNHIP_COARSE=exact: no MFMA prefilter; =fp32: the round-1 fp32 GEMM + select; default 2: bf16
    DevBuf coarse_fail_dev;   // unsigned long long: queries that took the exact fallback
    // PQ
    bool has_pq = false;
    DevBuf cb;            // [M][256][dsub]
    DevBuf precomp_t;     // [nlist][256][M]
    DevBuf cb_t;          // [256][M] float4, c-major codebook (pq_scan_q4: M = 32, dsub = 4)
    int use_precomp = 0;
    // SQ
    bool has_sq = false;
    DevBuf sq_trained;    // vmin[d], vdiff[d]
    // lists / base
    bool has_data = false;
    int64_t ntotal = 0;
    int64_t code_size = 0;
    int64_t id_offset = 0;
    std::vector<int64_t> h_list_len, h_list_row_off;
    DevBuf d_list_len, d_list_row_off, d_list_blk_off;
    DevBuf ids;
    DevBuf codes_aos;     // canonical list-sorted codes [ntotal][code_size] (faiss ArrayInvertedLists bytes): what a
                          // further Add merges into and Serialize reads back (BRUTE_FORCE: the raw rows)
    mutable bool aos_ready = false;  // IVF_FLAT / IVF_SQ8 drop the AoS copy of large lists (the interleaved `rows` hold the
                                     // same bytes); ensure_aos() rebuilds it when a further Add / Serialize needs it
    std::vector<int64_t> h_list_off;  // [nlist + 1]
    DevBuf rows;          // kind specific layout
    DevBuf rows2;         // IVF_PQ m=32: stream16 layout for the staggered scan (pq_scan_v2.hip)
    DevBuf d_list_blk_off2;
    bool pq_v2 = false;
    mutable bool skew_ready = false;  // IVF_PQ: `rows` holds the skewed layout of pq_scan.hip
    int pq_q4 = 2;             // KNHIP_Q4 = 0: never, 1: whenever the shape allows, 2 (default): when lists are shared by enough queries
    bool rank0_select = true;  // KNHIP_RANK0=0 switches the dump + radix-select phase off
    bool cand_hist = true;     // KNHIP_HIST=0 switches the per-query candidate histogram off
    bool pq_v1 = false;        // KNHIP_PQ_V1=1: m = 32 on the round-1 layout
    mutable bool rank0_phase_used = false;
    int64_t max_list_len = 0;
    // MFMA prefilter (mfma_scan.hip): KNHIP_MSCAN = 0 never, 1 whenever the shape allows, 2 (default) when the lists
    // are shared by enough queries of the batch
    int ksub = 256;              // IVF_PQ: codebook entries per sub-quantizer in use (2^nbits); the layouts and kernels are those of
                                 // 8-bit codes -- one byte per sub-quantizer, tables 256 wide, entries >= ksub copies of entry 0
                                 // that no code refers to
    int mscan = 2;
    bool flat_bf16 = true;    // KNHIP_MSCAN_FLAT=fp32: the IVF-Flat filter pass on the fp32 matrix instruction (round 2)
    int mscan_cap = 0;           // KNHIP_MSCAN_CAP: candidate capacity per query (0 = automatic; tests force the retry round)
    int pqd_spill_cap = 0;       // KNHIP_PQD_SPILL: parked records per workgroup in global memory (0 = automatic; tests force the overflow route of pq_decode.hip)
    // BRUTE_FORCE on the matrix cores (the coarse quantizer's bf16 prefilter + exact re-rank + certificate over the base rows):
    // split bf16 operand rows + ||x||^2 per row, built on first use.  KNHIP_BF=exact keeps the exact row scan
    mutable bool bf_split_ready = false;
    mutable DevBuf rows_bs, bf_norm;
    mutable float bf_norm_max = 0.f;
    bool bf_mfma = true;
    mutable int last_bf_mfma = 0;    // 1: the last BRUTE_FORCE search ran on the matrix cores
    mutable bool xnorm_ready = false;
    mutable DevBuf xnorm;        // [total blocks * 64] ||x||^2 per stored row position, built on first use
    mutable float xnorm_max = 0.f;
    int64_t total_blk = 0;       // 64-row blocks of the interleaved layout
    DevBuf row_scale;            // COSINE with stored norms: one float per stored row position (knhip_index_set_row_scale)
    int cos_mode = 0;            // 0 off, 1 ip / norm (IVF-Flat), 2 clamp(ip * inverse norm) (flat)
    std::vector<int64_t> h_list_blk_off; // [nlist + 1] first 64-row block of each list
    // IVF-PQ matrix-core ADC prefilter (pq_filter.hip): 1 = when the lists are shared by enough queries (default),
    // 2 = whenever the shape allows (KNHIP_PQF=1: tests), 0 = never (KNHIP_PQF=0).  Its layouts are built on first use.
    int pqf = 1;
    bool pqf_guard = true;           // KNHIP_PQF_GUARD=0 switches the selectivity guard off (tests of the overflow rounds)
    int pqf_form = 0;                // KNHIP_PQF_FORM: 0 = chosen per batch by the guard (decode form, else half), 1 = half
                                     // precision tables, 2 = int8 tables, 3 = decode form (pq_decode.hip)
    mutable bool psum_ready = false; // psum + its offsets (every form and the sample pass)
    mutable bool pqf_ready = false;  // ... + the half form's token stream
    mutable bool pqi_ready = false;  // ... + the integer form's
    mutable bool pqd_ready = false;  // ... + the decode form's half codebook, scales and start values
    // The selectivity guard of the IVF-PQ prefilter decides per (k, nprobe): synchronously the first time (and every 64th),
    // from the PREVIOUS batch's counters otherwise -- they arrive through a pinned buffer and an event, nothing waits.  The
    // decision only picks kernels; results do not depend on it.
    struct GuardEntry {
        int form = -1;            // -1 not decided yet, 0 exact kernels (abandon), 1 half form, 2 integer form
        int32_t* h_poor = nullptr; // pinned [2]
        hipEvent_t ev = nullptr;
        bool pending = false;
        int64_t pending_nq = 0;
        int age = 0;
    };
    mutable std::map<std::pair<int, int>, GuardEntry> guard_cache; // (under mu)
    mutable bool idmap_ready = false; // IVF-Flat direct map (knhip_index_get_vectors): built on first use
    mutable DevBuf idmap_ids, idmap_col;
    mutable int64_t last_range_ranks = 0; // coarse ranks the last range search scanned per query (rank waves)
    mutable int last_pq_form = 0;    // prefilter form of the last search: 0 none (exact kernels), 1 half precision, 2 int8
    mutable DevBuf rows_i;           // token stream of the integer form (stream16i)
    mutable DevBuf rows_r;           // rotated token stream (stream16r)
    mutable DevBuf d_list_blk_off_r; // [nlist + 1]
    mutable DevBuf psum;             // per stream position: sum_m term2 (L2)
    mutable DevBuf psum_s;           // decode form: -psum SC / 2 (the accumulators' start values)
    mutable DevBuf pqd_cb16;         // decode form: the codebook as halves (64 KB)
    mutable DevBuf pqd_st;           // decode form: [8] scales and constants (pq_decode.hip) + one word of scratch
    mutable float pabs_max = 0.f;    // max over vectors of sum_m |term2|
    // scratch
    mutable std::mutex mu;
    std::mutex add_mu;     // serialises Add / Train
    mutable std::map<void*, std::unique_ptr<Workspace>> ws_by_stream;
    mutable std::vector<std::unique_ptr<Workspace>> ws_free;
    // profiling
    bool prof = false;
    mutable std::vector<PendingEvent> pending;
    mutable knhip_stage_times times{};
    DevBuf scan_bytes_dev; // double accumulator
    mutable double coarse_flops = 0;
    mutable int64_t tie_queries = 0;  // queries resolved by the reference's admission rule (search_batch_ties)
    mutable DevBuf rg_seg_dev;        // range_segments(): the segment table of the current lists, built on first use
    mutable int64_t rg_seg_nseg = -1, rg_seg_ncol = 0;
    mutable int64_t last_items_bound = 0;

    int64_t device_bytes() const {
        const DevBuf* all[] = {&centroids, &centroids_il, &centroids_bs, &cb, &precomp_t, &sq_trained, &d_list_len,
                               &d_list_row_off, &d_list_blk_off, &ids, &rows, &rows2, &d_list_blk_off2, &cb_t, &codes_aos,
                               &rows_r, &d_list_blk_off_r, &psum, &rows_i, &idmap_ids, &idmap_col, &psum_s, &pqd_cb16, &pqd_st, &rows_bs, &bf_norm};
        int64_t t = 0;
        for (auto* b : all) {
            t += (int64_t)b->bytes;
        }
        return t;
    }
};

namespace knhip_host {

struct StageTimer {
    const knhip_index* idx;
    hipStream_t s;
    int stage;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    StageTimer(const knhip_index* i, hipStream_t st, int stg) : idx(i), s(st), stage(stg) {
        if (idx->prof) {
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, s);
        }
    }
    ~StageTimer() {
        if (idx->prof) {
            (void)hipEventRecord(e1, s);
            std::lock_guard<std::mutex> lk(idx->mu);
            idx->pending.push_back({stage, e0, e1});
        }
    }
};

inline int check_index(const knhip_index* idx) {
    if (!idx) {
        return fail(KNHIP_ERR_INVALID_ARGS, "null index");
    }
    return KNHIP_OK;
}

inline int64_t round_up(int64_t a, int64_t b) {
    return (a + b - 1) / b * b;
}

// upload helpers ---------------------------------------------------------------------------------
inline int upload(DevBuf& dst, const void* src, size_t bytes) {
    HIP_TRY(dst.alloc(bytes));
    if (bytes) {
        HIP_TRY(hipMemcpy(dst.p, src, bytes, hipMemcpyHostToDevice));
    }
    return KNHIP_OK;
}

// ---- entry points that cross translation units (defined in knhip_api.hip unless noted) ----------------------------------
int coarse_stage(const knhip_index* idx, Workspace* ws, const float* d_q, int64_t nq, int nprobe, int64_t* keys, float* cdis,
                 hipStream_t s);
int maybe_build_precomp(knhip_index* idx);
int build_list_layout(knhip_index* idx, const std::vector<int64_t>& list_off, const uint8_t* d_codes, const int64_t* d_ids);
int ensure_aos(const knhip_index* cidx);
Workspace* acquire_ws(const knhip_index* idx, void* stream_key, bool pooled_by_stream);
void release_ws(const knhip_index* idx, Workspace* w);
// one batch of queries, everything on the device (pre_keys / pre_cdis non-null: the coarse assignment is given)
int search_batch(const knhip_index* idx, Workspace* ws, const float* d_q, int64_t nq, int k, int nprobe, const uint8_t* d_bitset,
                 int64_t nbits, int64_t* d_out_i, float* d_out_d, hipStream_t s, const int64_t* pre_keys = nullptr,
                 const float* pre_cdis = nullptr);
int validate_search(const knhip_index* idx, int64_t nq, int32_t k, int32_t& nprobe);
int64_t query_batch(const knhip_index* idx, int64_t nq, int k, int nprobe);
// search_batch + the reference's first-come admission at the k-th boundary (knhip_api_range.hip)
int search_batch_ties(const knhip_index* idx, Workspace* ws, const float* d_q, int64_t nq, int k, int nprobe,
                      const uint8_t* d_bitset, int64_t nbits, int64_t* d_out_i, float* d_out_d, hipStream_t s,
                      const int64_t* pre_keys, const float* pre_cdis);
// BRUTE_FORCE rows -> the interleaved blocks (knhip_index_add_vectors*; the GPU build appends through it)
int add_vectors_common(knhip_index* idx, int64_t n, const float* d_x, const int64_t* d_ids, int64_t id_offset);
} // namespace knhip_host

struct knhip_rows {
    int device = 0, d = 0, row_type = KNHIP_ROWS_FP16;
    int64_t n = 0;
    bool trained = false;
    DevBuf codes;      // [n][code_size]
    DevBuf sq;         // vmin[d], vdiff[d] (sq8)
    std::mutex mu;
    // trained ranges: per-dimension vmin / vdiff (sq8, sq6: 2 d floats) or one for all dimensions (sq4u: 2 floats)
    bool ranged() const { return row_type == KNHIP_ROWS_SQ8 || row_type == KNHIP_ROWS_SQ6 || row_type == KNHIP_ROWS_SQ4U; }
    int nrange() const { return row_type == KNHIP_ROWS_SQ4U ? 1 : d; } // floats per half of the trained vector
    int64_t code_size() const {
        if (row_type == KNHIP_ROWS_SQ4U) {
            return ((int64_t)d * 4 + 7) / 8;
        }
        return row_type == KNHIP_ROWS_SQ6 ? ((int64_t)d * 6 + 7) / 8
             : (row_type == KNHIP_ROWS_SQ8 || row_type == KNHIP_ROWS_INT8) ? (int64_t)d : 2 * (int64_t)d;
    }
};

