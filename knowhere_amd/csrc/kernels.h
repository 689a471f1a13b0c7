// knowhere_amd/csrc/kernels.h -- host-side launch interface between knhip_api.hip and the
// kernel translation units.  Everything here is internal to libknhip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

namespace knhip {

// ---- work item table ------------------------------------------------------------------------------
struct KnItem {
    int32_t list;   // inverted list (or base chunk) to scan
    int32_t npair;  // number of (query, slot) pairs in this item, 1..QG
    int64_t pair0;  // first entry in the sorted pair array
};
// sorted pair entry: query index and slot (probe rank) packed
struct KnPair {
    int32_t q;
    int32_t slot;
};

// ---- stream16 code layout of the staggered PQ scans (pq_scan_v2.hip, pq_scan_q4.hip) -----------------
// Lane L of a wave starts its vector pq_stream_phase(L) steps late.  16 phases, chosen so that every lane
// group an LDS gather is serviced in sees each phase equally often:
//   ds_read_b128 (pq_scan_q4): four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32)  -> each
//     phase once per group; with LUT[code][m][4 queries] (16-B entries, bank quad = m mod 16) the 16 lanes
//     of a group sit on 16 consecutive m: no bank conflict for any code values.  The map is also
//     conflict-free should the groups be the contiguous sixteenths {0-15}, {16-31}.
//   ds_read_b64 (pq_scan_v2): two 32-lane groups -> each phase twice: the 2-way conflict that kernel accepts.
__host__ __device__ constexpr int pq_stream_phase(int lane) {
    const int l = lane & 31;
    return l < 4 ? l : l < 12 ? l + 4 : l < 16 ? l - 8 : l < 20 ? l - 16 : l < 28 ? l - 12 : l - 24;
}
constexpr int PQ_STREAM_PHASES = 16;
// lanes (of one 32-lane half; both halves are alike) that are already on the window's NEW vector at step j
__host__ __device__ constexpr uint32_t pq_stream_mask(int j) {
    uint32_t m = 0;
    for (int l = 0; l < 32; l++) {
        if (pq_stream_phase(l) <= j) {
            m |= 1u << l;
        }
    }
    return m;
}

// flat work record of the persistent 4-query scan (pq_scan_q4.hip), one per work item
struct P4Rec {
    int32_t list;
    int32_t npair;
    int32_t q[4];
    int32_t slot[4];
    float dis0[4];
    int64_t len;
    int64_t sblk0;
    int64_t row_off;
    int64_t pad[2];
};
static_assert(sizeof(P4Rec) == 96, "P4Rec layout");

// flat work record of the half-precision ADC prefilter (pq_filter.hip), one per unit of (list, <= 8 pairs)
struct P8Rec {
    int32_t list;
    int32_t npair;
    int32_t q[8];
    int32_t slot[8];  // filter: probe slot; sample pass: first dump column of the pair
    float dis0[8];
    int64_t len;      // rows to scan (sample pass: capped at the dump width)
    int64_t sblk0;    // first block of the list in the rotated token stream
    int64_t row_off;
};
static_assert(sizeof(P8Rec) == 128, "P8Rec layout");

// ... and of its 16-query integer form (pq_filter.hip, pqi_kernel): one unit of (list, <= 16 pairs)
struct P16Rec {
    int32_t list;
    int32_t npair;
    int32_t q[16];
    int32_t slot[16];
    float dis0[16];
    int64_t len;
    int64_t sblk0;    // first block of the list in the token stream (same block offsets as the half-precision form)
    int64_t row_off;
    int64_t pad[4];
};
static_assert(sizeof(P16Rec) == 256, "P16Rec layout");

struct FlatScanArgs {
    // rows
    const float4* rows;          // interleaved blocks
    const int64_t* list_blk_off; // [nlist] first block of each list (TABLE) / nullptr (DENSE)
    const int64_t* list_len;     // [nlist] rows in each list (TABLE)
    const int64_t* list_row_off; // [nlist] first entry in ids[] of each list (TABLE)
    const int64_t* ids;          // [ntotal] (TABLE) or nullptr (DENSE: id = row + id_offset)
    int64_t nrows;               // DENSE: total rows
    int64_t chunk_rows;          // DENSE: rows per chunk (multiple of 64)
    int64_t id_offset;           // DENSE
    int32_t d;
    int32_t nchunk;              // ceil(d/4)
    // queries
    const float* queries;        // [nq][d]
    int64_t nq;
    // work
    const KnItem* items;         // TABLE
    const KnPair* pairs;         // TABLE
    const int64_t* nitems_dev;   // TABLE: device scalar
    int64_t nitems_dense;        // DENSE: nchunks * ngroups
    int64_t ngroups;             // DENSE
    // filter
    const uint8_t* bitset;
    int64_t bitset_nbits;
    // output
    float* partial_d;            // [nq][nslot][k]
    int64_t* partial_i;
    float* gthr;                 // [nq] shared per-query threshold (see common.h gthr_*)
    int32_t nslot;
    int32_t k;
    int32_t item_loop;           // TABLE: 1 = the (fixed) grid walks items blockIdx.x, + gridDim.x, ... < *nitems_dev
    // COSINE with stored norms (inner product only): one float per stored row position, applied to the finished product
    // sum.  cos_mode 1: dis = ip / scale (IVF-Flat: cppcontrib/knowhere/IndexIVFFlat.cpp:199-210);
    // 2: dis = clamp(ip * scale, -1, 1) (flat: cppcontrib/knowhere/utils/distances.cpp:367-409); 0: off
    const float* row_scale;
    int32_t cos_mode;
};

enum PqLutMode { PQ_LUT_PRECOMP = 0, PQ_LUT_IP = 1, PQ_LUT_RESIDUAL = 2 };

struct PqScanArgs {
    // index
    const uint4* codes_skew;       // skewed code blocks
    const int64_t* list_sblk_off;  // [nlist+1] first skew block of each list
    const int64_t* list_len;       // [nlist]
    const int64_t* list_row_off;   // [nlist] offset into ids[]
    const int64_t* ids;            // [ntotal] sorted by list, ascending inside a list
    const float* precomp_t;        // [nlist][256][M]   (PRECOMP)
    const float* cb;               // [M][256][dsub]    (RESIDUAL)
    const float* centroids;        // [nlist][d]        (RESIDUAL)
    int32_t d;
    int32_t lut_mode;
    // per search
    const float* queries;          // [nq][d]           (RESIDUAL)
    const float* t2t;              // [nq][256][M]      (PRECOMP, IP)
    const float* coarse_dis;       // [nq][nprobe]
    const KnItem* items;
    const KnPair* pairs;
    const int64_t* nitems_dev;
    const uint8_t* bitset;
    int64_t bitset_nbits;
    float* partial_d;              // [nq][nslot][k]
    int64_t* partial_i;
    float* gthr;                   // [nq] shared per-query threshold
    int32_t nslot;                 // = nprobe
    int32_t k;
    // work-item range [*item_lo, *item_hi) of this launch (device scalars; item_lo may be null = 0)
    const int64_t* item_lo;
    const int64_t* item_hi;
    // rank-0 "dump" phase (pq_scan_v2.hip): every finished distance goes to dump[q * dump_stride + offset]
    float* dump;
    int64_t dump_stride;
    int32_t dump_by_row;           // 1: column = storage position of the vector (list_row_off + offset), range search
    // per-query candidate histogram (pq_scan_v2 after a rank-0 phase; null = off): ghist[q][64] counts the
    // vectors seen so far per distance bin, gmeta[q] = {key of the first bin, bin shift | KN_HIST_OFF}
    uint32_t* ghist;
    const uint2* gmeta;
    // pq_scan_q4 (persistent, 4 queries per item): flat records, per-XCD item counters, c-major codebook
    P4Rec* recs4;                  // [item bound]
    int32_t* q4_ctr;               // [8 * 16] one counter per XCD, 64 B apart (zeroed by the launcher)
    const float4* cb_t;            // [256][M] float4: cb_t[c][m] = codebook entry (m, c), dsub = 4
};


struct SqScanArgs {
    const uint4* rows;           // interleaved blocks: [nchunk16][64 rows][16 code bytes]
    const int64_t* list_blk_off; // [nlist]
    const int64_t* list_len;     // [nlist]
    const int64_t* list_row_off; // [nlist]
    const int64_t* ids;          // [ntotal]
    const float* trained;        // vmin[d], vdiff[d]
    const float* centroids;      // [nlist][d]  (L2: query residual)
    int32_t d;
    int32_t nchunk16;            // ceil(d/16)
    const float* queries;        // [nq][d]
    const float* coarse_dis;     // [nq][nprobe] (IP: accu0)
    const KnItem* items;
    const KnPair* pairs;
    const int64_t* nitems_dev;
    const uint8_t* bitset;
    int64_t bitset_nbits;
    float* partial_d;            // [nq][nslot][k]
    int64_t* partial_i;
    float* gthr;                 // [nq] shared per-query threshold (see common.h gthr_*)
    int32_t nslot;
    int32_t k;
    // range search: non-null = write every distance to dump[q * dump_stride + storage position], no top-k
    float* dump;
    int64_t dump_stride;
    int32_t item_loop;           // 1 = the (fixed) grid walks items blockIdx.x, + gridDim.x, ... < *nitems_dev
};

// ---- mfma_scan.hip: MFMA prefilter + exact finish for the IVF-Flat / IVF-SQ8 list scans ----
struct MScanArgs {
    const void* rows;            // interleaved blocks (float4 / uint4), as FlatScanArgs / SqScanArgs
    const float* xnorm;          // [total blocks * 64] ||x||^2 per stored row position (L2)
    float xnorm_max;             // fp32 rows: max ||x||^2 (error bound)
    const int64_t* list_blk_off;
    const int64_t* list_len;
    const int64_t* list_row_off;
    const int64_t* ids;
    const float* trained;        // SQ8: vmin[d], vdiff[d]
    const float* centroids;      // SQ8 L2: query residual
    int32_t d;
    int32_t nchunk;              // fp32: ceil(d / 4) float4 chunks; SQ8: ceil(d / 16) code chunks
    int32_t nstep;               // fp32: ceil(nchunk / 4) steps of 16 dims; SQ8: ceil(nchunk / 2) steps of 32 dims
    const float* queries;
    const float* qnorm;          // [nq] ||q||^2 (fp32 rows)
    const float* coarse_dis;     // [nq][nslot] (SQ8 IP: accu0)
    int64_t nq;
    int32_t nslot;
    const KnItem* units;         // (list, pair range) units of the bulk probes
    const KnPair* pairs;
    const int64_t* nunits_dev;
    const float* gthr;           // [nq] exact k-th distance of the rank-0 list (or the neutral value)
    float eps_scale;
    const uint8_t* bitset;
    int64_t bitset_nbits;
    int32_t* cand_cnt;           // [nq]
    int64_t* cand;               // [nq][cap]: slot << 32 | row position in the list
    uint32_t* eps_max;           // [nq] SQ8: bit pattern of the largest eps any unit of the query emitted with (null = off)
    float* cand_pess;            // [nq][cap] pessimistic distance of each candidate (null = not kept): the finish kernel
                                 // drops the candidates that cannot beat the k-th best of them before it recomputes any
    int32_t cap;
    int32_t* overflow;           // [nq + 1]: per-query flag, [nq] = any
    // sample pass (non-null = DUMP mode): pessimistic distance of (query, row) -> dump[q * dump_stride + row]
    float* dump;
    int64_t dump_stride;
    int32_t sample_cap;          // rows of a query's sample, at most (<= dump_stride; the plan's `cap`)
    // per-query candidate histogram (null = off): ghist[q][64], gmeta[q] = {key of the first bin, bin shift | KN_HIST_OFF}
    uint32_t* ghist;
    const uint2* gmeta;
    int32_t k;
    // sample plan: first dump column of pair (q, slot), or -1 when the pair is not part of the sample
    const int32_t* sample_off;
    // SQ8 inner product: query operands prepared once per batch (ms_sq8_query_prep; null = computed per unit):
    // qh / ql [nq][nstep * 32] halves, qs [nq][8] = {scale, A, W, sum |y'|, sum (hi + lo), finite, -, -}
    const void* qh;
    const void* ql;
    const float* qs;
    int32_t unit_loop;           // 1 = the (fixed) grid walks units blockIdx.x, + gridDim.x, ... < *nunits_dev
    float* gthr_rw;              // = gthr, written by the finish kernel's retry preparation
    // IVF-PQ half-precision prefilter (pq_filter.hip; M = 32, dsub = 4)
    const uint4* pq_codes_r;       // rotated token stream (pq_stream16r_kernel)
    const int64_t* pq_sblk_off_r;  // [nlist + 1] first block of each list in it
    const float* pq_psum;          // per stream position: sum_m term2[list][m][code_m] (L2; unused for IP)
    const void* pq_qh;             // [nq][64][16][4][2] halves: per-query tables (pqf_query_table_kernel)
    const float* pq_qs;            // [nq][4] = {scale, 1 / scale, eps_base, A}
    const float4* pq_cb_t;         // [256][32] c-major codebook (exact finish)
    const float* pq_precomp_t;     // [nlist][256][32] (exact finish, PQ_LUT_PRECOMP)
    const uint8_t* pq_codes;       // canonical AoS codes [ntotal][32] (exact finish)
    int32_t pq_lut_mode;           // PqLutMode
    P8Rec* pq_recs;                // [unit bound]
    // ... its 16-query integer form (pqi_kernel): int8 tables, units of (list, <= 16 pairs)
    const uint4* pq_codes_i;       // token stream of the integer form (pq_stream16i_kernel), block offsets = pq_sblk_off_r
    const void* pq_qi;             // [nq][64][16][4][2] int8: per-query tables (pqi_query_table_kernel)
    const float* pq_qis;           // [nq][4] = {step, sum of the per-m offsets, eps_base, A}
    P16Rec* pq_recs16;             // [unit bound]
    int32_t* pq_ctr;               // [8 * 16] one unit counter per XCD, 64 B apart
    int32_t pq_prune_mu;           // finish kernel: pq_qs is the integer form's record ([q][1] = sum of the per-m offsets)
    // ... its DECODE form (pq_decode.hip, pqd_kernel): rows decoded once per (list, <= 128 queries), dense f16 contraction
    const void* pq_cb16;           // [32][256][4] halves: the codebook scaled by a power of two (per index)
    const void* pq_qh16;           // [nq][128] halves: the queries scaled by the batch's power of two
    const float* pq_qd;            // [nq][4] = {||q||_1, B_q, eps_base, 2 B_q}
    const float* pq_sc;            // [8] = {sc_y, 1 / sc_y, max |cb|, Ysum, sc_q, 1 / sc_q, SC, 1 / SC}: the index's scales (device)
    const float* pq_psum_s;        // -psum SC / 2 per stream position (same offsets as pq_psum): the accumulators' start values
    unsigned char* pq_spill;       // [pq_spill_wgs][pq_spill_cap] records of 80 bytes: where a workgroup parks passing lanes once its
    int32_t pq_spill_cap;          // LDS regions are full (a unit that is hot for many of its queries at once)
    int32_t pq_spill_wgs;          // workgroups the buffer serves (the launch takes no more)
};

// ---- pq_filter.hip ----
int64_t pq_stream16r_blocks(int64_t len);
hipError_t launch_pq_stream16r(const uint8_t* codes, const int64_t* list_row_off, const int64_t* list_len,
                               const int64_t* list_sblk_off, int64_t nlist, uint4* out, hipStream_t s);
hipError_t launch_pq_psum(const uint8_t* codes, const int64_t* list_row_off, const int64_t* list_len,
                          const int64_t* list_sblk_off, int64_t nlist, const float* precomp_t, float* psum,
                          uint32_t* pabs_max_bits, hipStream_t s);
hipError_t launch_pqf_query_table(const float* queries, const float4* cb_m, int d, int64_t nq, bool is_l2, float pabs_max,
                                  void* qh, float* qs, hipStream_t s);
// selectivity guard: *poor = queries whose predicted candidate count exceeds half the capacity
hipError_t launch_pqf_predict(const float* dump, int64_t stride, const int32_t* n_row, const float* gthr, const float* qs,
                              const float* qs2, const int64_t* keys, int nprobe, int64_t nlist, const int64_t* list_len,
                              int64_t nq, int cap, int k, bool is_l2, int32_t* poor, hipStream_t s);
size_t pqf_smem();
// integer form: 16 queries per unit, int8 tables, v_mfma_i32_16x16x64_i8
hipError_t launch_pq_stream16i(const uint8_t* codes, const int64_t* list_row_off, const int64_t* list_len,
                               const int64_t* list_sblk_off, int64_t nlist, uint4* out, hipStream_t s);
// cb_m: the codebook in FAISS order [m][256][4]; qis: nq * 4 + 4 floats; qmu: nq * 32 floats of scratch
hipError_t launch_pqi_query_table(const float* queries, const float4* cb_m, int d, int64_t nq, bool is_l2, float pabs_max,
                                  void* qi, float* qis, float* qmu, bool stats_done, hipStream_t s);
// the sample pass of the IVF-PQ prefilter, one workgroup per query (no units, no work table): dump + n_row, the half
// form's qs records, and (qis / qmu non-null) pass 1 of the integer form
// (smin: rows the plan wants at least; scap <= 4096: rows it takes at most)
// gthr_out / gmeta_out non-null: tau_q (the ksel-th best sampled value) and the histogram range are selected in the kernel
hipError_t launch_pq_sample(const MScanArgs& a, const int64_t* keys, const float4* cb_m, int64_t nlist, int smin,
                            int scap, float pabs_max, bool is_l2, int32_t* n_row, float* qs, float* qis, float* qmu,
                            hipStream_t s, float* gthr_out = nullptr, uint2* gmeta_out = nullptr, int ksel = 0);
hipError_t launch_pqi(const MScanArgs& a, bool is_l2, int64_t units_bound, hipStream_t s);

// ---- pq_decode.hip: the decode form of the IVF-PQ prefilter ----
constexpr int PD_QT = 128; // queries per unit
size_t pqd_smem();
bool pqd_supports(int M, int d);
// per index: cb (FAISS order [m][256][4]) -> cb16 (64 KB of halves) + st[8] (scales, constants); centroids: ncent floats (their
// largest magnitude sets the query scale); psum -> psum_s (npsum floats; L2 only); scratch: one uint32
hipError_t launch_pqd_index_prep(const float4* cb, const float* centroids, int64_t ncent, void* cb16, float* st,
                                 uint32_t* scratch, const float* psum, int64_t npsum, float* psum_s, hipStream_t s);
// per batch: qh16: nq * 128 halves; qd: nq * 4 floats = {||q||_1, B_q, eps_base, 2 B_q}
hipError_t launch_pqd_query_prep(const float* queries, const float4* cb, const float* cst, int64_t nq, bool is_l2,
                                 float pabs_max, void* qh16, float* qd, hipStream_t s);
hipError_t launch_pqd(const MScanArgs& a, bool is_l2, int64_t units_bound, hipStream_t s);

bool pqf_supports(int M, int d);
hipError_t launch_pqf(const MScanArgs& a, bool is_l2, int64_t units_bound, hipStream_t s);

// ---- flat_scan.hip ----
int flat_scan_qg(int k);
int sq_scan_qg(int k);
hipError_t launch_flat_scan(const FlatScanArgs& a, bool is_l2, bool dense, int64_t grid, hipStream_t s,
                            int qg_override = 0);
hipError_t launch_flat_full(const FlatScanArgs& a, bool is_l2, float* out, const int32_t* q_subset,
                            int64_t nq_subset, const int32_t* row_flags, hipStream_t s);
hipError_t launch_interleave_rows(const float* src, int64_t n, int d, float4* dst, int64_t dst_blk0,
                                  hipStream_t s);
hipError_t launch_interleave_lists(const float* src, const int64_t* list_row_off,
                                   const int64_t* list_len, const int64_t* list_blk_off, int64_t nlist,
                                   int d, float4* dst, hipStream_t s);

// ---- pq_scan.hip ----
int pq_scan_supported_m(int M);
int pq_scan_qg(int M);
int64_t pq_skew_blocks(int64_t len, int M);
hipError_t launch_pq_scan(const PqScanArgs& a, bool is_l2, int M, int64_t grid, hipStream_t s);
hipError_t launch_pq_query_table(const float* queries, const float* cb, int d, int M, int64_t nq,
                                 float* t2t, hipStream_t s);
hipError_t launch_pq_precomp_table(const float* centroids, const float* cb, int d, int M,
                                   int64_t nlist, float* pt, hipStream_t s);
hipError_t launch_pq_skew_codes(const uint8_t* codes, const int64_t* list_row_off,
                                const int64_t* list_len, const int64_t* list_sblk_off, int64_t nlist,
                                int M, uint4* out, hipStream_t s);

// ---- pq_scan_v2.hip (M = 32, k <= 192: lane-stationary staggered ADC on the stream16 layout) ----
bool pq_scan_v2_supports(int M, int k);
int64_t pq_stream16_blocks(int64_t len);
hipError_t launch_pq_scan_v2(const PqScanArgs& a, bool is_l2, bool dump, int64_t grid, hipStream_t s);
// rank-0 phase epilogue: per query, top-k of its dumped closest list -> partial slot 0 + shared threshold
hipError_t launch_rank0_select(const float* dump, int64_t dump_stride, const int64_t* keys, int nprobe,
                               const int64_t* list_len, const int64_t* list_row_off, const int64_t* ids,
                               int64_t nq, int k, bool is_l2, float* partial_d, int64_t* partial_i, float* gthr,
                               int64_t* tmp_keys, float* tmp_d, uint32_t* ghist, uint2* gmeta, hipStream_t s);
hipError_t launch_pq_stream16(const uint8_t* codes, const int64_t* list_row_off, const int64_t* list_len,
                              const int64_t* list_sblk_off, int64_t nlist, uint4* out, hipStream_t s);

// ---- pq_scan_q4.hip (M = 32, dsub = 4, k <= 192: persistent 4-query staggered ADC, LUT built in-kernel) ----
bool pq_scan_q4_supports(int M, int d, int k);
hipError_t launch_pq_scan_q4(const PqScanArgs& a, bool is_l2, int64_t items_bound, hipStream_t s);
hipError_t launch_pq_cb_transpose(const float* cb, int M, int dsub, float4* cb_t, hipStream_t s);

// ---- mfma_scan.hip ----
int mscan_queries_per_unit(int kind, bool sample);
size_t mscan_flat_smem(int nstep);
size_t mscan_sq8_smem(int nstep);
int mscan_finish_pmax(int cap, int k);
int mscan_sample_rows();
hipError_t launch_ms_block_norms(const float4* rows, int64_t total_blk, int nchunk, float* out, float* out_max,
                                 hipStream_t s);
hipError_t launch_ms_sq8_norms(const uint4* rows, int64_t total_blk, int nchunk16, int d, const float* trained,
                               float* out, float* out_max, hipStream_t s);
// units from one virtual-list range of the work table (`*_v` = the table's arrays offset to that range)
hipError_t launch_ms_units(const int32_t* list_count_v, const int64_t* list_pair_off_v, int64_t nlist, int qt,
                           int64_t* unit_off, int64_t* nunits, KnItem* units, const int64_t* list_len,
                           int64_t code_size, double* unit_bytes, hipStream_t s);
hipError_t launch_mscan_flat(const MScanArgs& a, bool is_l2, int64_t units_bound, hipStream_t s);
// ... its filter pass on the bf16 matrix pipe (mfma_scan_bf16.hip): queries per unit for this shape (0 = not served), LDS
// bytes per workgroup, launch (a.dump must be null; units cut for mscan_flat_bf16_qt(a.nstep) queries)
int mscan_flat_bf16_qt(int nstep);
size_t mscan_flat_bf16_smem(int nstep);
hipError_t launch_mscan_flat_bf16(const MScanArgs& a, bool is_l2, int64_t units_bound, hipStream_t s);
hipError_t launch_mscan_sq8(const MScanArgs& a, bool is_l2, int64_t units_bound, hipStream_t s);
hipError_t launch_ms_sample_plan(const int64_t* keys, int64_t nq, int nprobe, int64_t nlist, const int64_t* list_len,
                                 int smin, int cap, int32_t* sample_off, int32_t* n_row, hipStream_t s);
hipError_t launch_ms_sq8_query_prep(const float* queries, int64_t nq, int d, int ldq, const float* trained, void* qh,
                                    void* ql, float* qs, hipStream_t s);
hipError_t launch_ms_tau(const float* sel_d, int64_t nq, int k, bool is_l2, float* gthr, uint2* gmeta, hipStream_t s);
hipError_t launch_mscan_finish(const MScanArgs& a, int kind, bool is_l2, const int64_t* keys, const float* coarse_dis,
                               int nprobe, int k, float* out_d, int64_t* out_i, unsigned long long* counters, int pass,
                               hipStream_t s);
hipError_t launch_ms_flag_pairs(const int32_t* overflow, int want, const int64_t* keys, int64_t nq, int nprobe,
                                int64_t nlist, const int64_t* list_len, int k, KnItem* items, KnPair* pairs,
                                int64_t* nitems, int64_t* empty_mark, hipStream_t s);

// ---- sq_scan.hip ----
hipError_t launch_sq_scan(const SqScanArgs& a, bool is_l2, int64_t grid, hipStream_t s, int qg_override = 0);
hipError_t launch_sq_interleave(const uint8_t* codes, const int64_t* list_row_off,
                                const int64_t* list_len, const int64_t* list_blk_off, int64_t nlist,
                                int d, uint4* out, hipStream_t s);

// ---- worktable.hip: (query, probe) pairs -> per-list work items ----
struct WorkTable {
    // device buffers sized by the caller; "virtual lists" = 2 * nlist (rank-0 probes first)
    int32_t* list_count;   // [2*nlist]
    int64_t* list_pair_off;// [2*nlist+1]
    int64_t* list_item_off;// [2*nlist+1]
    int32_t* list_cursor;  // [2*nlist]
    KnPair* pairs;         // [nq*nprobe]
    KnItem* items;         // [nq*nprobe/qg + 2*nlist]
    int64_t* nitems;       // [1]
    double* scan_bytes;    // [1] sum over pairs of len(list)*code_size
    // pairs of empty lists get no work item; if non-null, entry 0 of their partial list (partial_i + t * k,
    // t = q * nprobe + slot) is set to -1 instead
    int64_t* empty_mark;
    int32_t k;
};
hipError_t launch_direct_items(const int64_t* keys, int64_t nq, int nprobe, int64_t nlist, const int64_t* list_len,
                               const WorkTable& wt, hipStream_t s);
hipError_t launch_build_worktable(const int64_t* keys, int64_t nq, int nprobe, int64_t nlist, int qg0, int qg1,
                                  const int64_t* list_len, int64_t code_size, const WorkTable& wt,
                                  hipStream_t s, int rank0_slot = 0, const int32_t* cls = nullptr);
hipError_t launch_fill_f32(float* p, int64_t n, float v, hipStream_t s);

// ---- range.hip: range search epilogue over a dumped distance matrix ----
struct RangeArgs {
    const float* dist;        // [nq][ncol] every distance of every probed list (scan kernels, dump mode)
    int64_t ncol;
    const int64_t* seg_col;   // [nseg] first column of each list / row segment
    const int64_t* seg_idpos; // [nseg] first entry in ids[] (or first row number when ids == nullptr)
    const int64_t* seg_len;   // [nseg]
    const int64_t* ids;       // storage-order ids, or nullptr: id = position + id_offset
    int64_t id_offset;
    const int64_t* order;     // [nq][nprobe] segment visited at each rank (coarse order), nullptr: rank == segment
    int32_t nprobe;
    float radius;
    const uint8_t* bitset;
    int64_t bitset_nbits;
    // tie resolution of Search() (knhip_api.hip, search_batch_ties): one radius per query and the bound itself is a hit
    const float* radius_q;    // [nq], nullptr: `radius` for every query
    int32_t inclusive;        // 1: dist <= radius (L2) / dist >= radius (IP)
};
// k-th-boundary ties (ResultHandler.h:258-279): rows of kk = k + 1 canonical results -> the first k to (out_d, out_i); a
// query whose (k + 1)-th entry exists with the k-th distance is appended to flagged[] (count in *nflag)
hipError_t launch_tie_detect(const float* d, const int64_t* i, int64_t nq, int k, float* out_d, int64_t* out_i,
                             int32_t* flagged, int32_t* nflag, hipStream_t s);
// ... and the resolve itself: the flagged queries' rows gathered, then (after the dump pass over their probed lists) the
// reference's rule applied per query, the result written over its row of (out_d, out_i) -- no host round trip
hipError_t launch_tie_gather(const int32_t* flagged, int nflag, const float* q, int d, const int64_t* keys, const float* cdis,
                             int nprobe, float* q_out, int64_t* keys_out, float* cdis_out, const float* can_d, int k,
                             float* rad_out, hipStream_t s);
hipError_t launch_tie_sort_flags(const int32_t* in, int n, int32_t* out, hipStream_t s);
hipError_t launch_refine_combine(const float* parts, int nsh, int64_t n, float* out, hipStream_t s);
// (nsh shards' arrival lists [nsh][nflag][k] with their scan-order keys; nsh = 1: the index's own, keys optional)
hipError_t launch_tie_resolve(const int32_t* flagged, int nflag, int nsh, const float* can_d, const int64_t* can_i, int k,
                              bool is_l2, const float* arr_d, const int64_t* arr_i, const int64_t* arr_key,
                              const int64_t* arr_n, int64_t arr_n_stride, float* out_d, int64_t* out_i, int32_t* anomalies,
                              hipStream_t s);
// every exact ADC distance of every probed list of an IVF-PQ index with any M x 8 bit codes (range.hip)
struct PqDumpArgs {
    float* dist;                 // [nq][ncol], column = list_row_off[list] + position
    int64_t ncol;
    const int64_t* keys;         // [nq][nprobe] probed lists (coarse order)
    const float* coarse_dis;     // [nq][nprobe]
    int32_t nprobe;
    int64_t nlist;
    const int64_t* list_len;
    const int64_t* list_row_off;
    const uint8_t* codes;        // canonical AoS codes [ntotal][M]
    int32_t M;
    int32_t d;
    int32_t lut_mode;            // PqLutMode
    const float* t2t;            // [nq][256][M] <q_m, cb>      (PRECOMP, IP)
    const float* precomp_t;      // [nlist][256][M]             (PRECOMP)
    const float* cb;             // [M][256][dsub]              (RESIDUAL)
    const float* centroids;      // [nlist][d]                  (RESIDUAL)
    const float* queries;        // [nq][d]                     (RESIDUAL)
};
hipError_t launch_pq_adc_dump(const PqDumpArgs& a, int64_t nq, bool is_l2, hipStream_t s);

// ---- pq_scan_any.hip: exact IVF-PQ top-k scan for any number of 8-bit sub-quantizers (the widths the fast kernels do not
// take).  One workgroup per (query, probe) writing pq_scan_any_parts(k) sorted partial lists (k <= 64: one per wave,
// partial slot = 4 * probe rank + wave; larger k: one), so merge_partials runs over parts * nprobe slots
struct PqAnyArgs {
    const int64_t* keys;         // [nq][nprobe] probed lists (coarse order; < 0: none)
    const float* coarse_dis;     // [nq][nprobe]
    int32_t nprobe;
    int64_t nlist;
    const int64_t* list_len;
    const int64_t* list_row_off;
    const uint8_t* codes;        // canonical list-sorted AoS codes [ntotal][M]
    const int64_t* ids;          // [ntotal]
    int32_t M;
    int32_t d;
    int32_t lut_mode;            // PqLutMode
    const float* t2t;            // [nq][256][M]     (PRECOMP, IP)
    const float* precomp_t;      // [nlist][256][M]  (PRECOMP)
    const float* cb;             // [M][256][dsub]   (RESIDUAL)
    const float* centroids;      // [nlist][d]       (RESIDUAL)
    const float* queries;        // [nq][d]          (RESIDUAL)
    const uint8_t* bitset;
    int64_t bitset_nbits;
    float* partial_d;            // [nq][nprobe * parts][k]
    int64_t* partial_i;
    int32_t k;
};
int pq_scan_any_parts(int k);    // partial lists per (query, probe): 4 (one per wave, k <= 64) or 1 (block selection)
int pq_scan_any_supports(int M, int d);
hipError_t launch_pq_scan_any(const PqAnyArgs& a, int64_t nq, bool is_l2, hipStream_t s);
// (rank0, nrank >= 0: one wave of ranks; qstate: queries with qstate[q][1] != 0 are skipped)
hipError_t launch_range_count(const RangeArgs& a, int64_t nq, bool is_l2, int32_t* cnt, hipStream_t s, int rank0 = 0,
                              int nrank = -1, const int32_t* qstate = nullptr);
// rank waves of the range search (range.hip): the wave's lists of the running queries, the early-stop state, and the
// IVF-Flat dump of a wave
hipError_t launch_range_wave_gather(const int64_t* keys, const float* cdis, int64_t nq, int nprobe, int r0, int W,
                                    const int32_t* qstate, int64_t* keys_w, float* cdis_w, hipStream_t s);
hipError_t launch_range_wave_state(const int32_t* cnt, int64_t nq, int nprobe, int r0, int r1, int max_empty,
                                   int32_t* qstate, int32_t* alive, hipStream_t s);
hipError_t launch_range_flat_dump(const FlatScanArgs& a, const int64_t* keys_w, int64_t nq, int W, int64_t nlist,
                                  const int64_t* seg_col, const int64_t* seg_len, float* dist, int64_t ncol, bool is_l2,
                                  hipStream_t s);
hipError_t launch_range_plan(const int32_t* cnt, int64_t nq, int nprobe, int max_empty, int64_t* off, int64_t* total,
                             hipStream_t s);
hipError_t launch_range_emit(const RangeArgs& a, int64_t nq, bool is_l2, const int64_t* off, const int64_t* qbase,
                             int64_t* out_ids, float* out_dis, hipStream_t s, int64_t cap = 0, int64_t* out_key = nullptr,
                             int64_t key_base = 0);

// ---- topk.hip: selection kernels ----
// per query: k best of nslot sorted partial lists -> out (canonical order, sentinel padded);
// list (q, slot) starts at q * q_stride + slot * slot_stride (elements)
hipError_t launch_merge_partials(const float* partial_d, const int64_t* partial_i, int64_t nq,
                                 int nslot, int k, int64_t q_stride, int64_t slot_stride, bool is_l2,
                                 float* out_d, int64_t* out_i, hipStream_t s, const int32_t* q_only = nullptr);
// per row: the k best of n values (index = column), canonical order; out_keys int64, out_d float
// k above row_select_lds_max_k() (up to row_select_max_k()): the selected keys are sorted in `sort_scratch`
// ([nrows][next power of two >= k] u64, caller's memory) instead of the LDS
// two-pass selection for k << n (topk.hip: group minima -> bound -> candidates); rows it cannot take are flagged in
// ovf_flags for launch_row_select
bool row_select_thr_supports(int64_t n, int k);
hipError_t launch_row_select_thr(const float* vals, int64_t nrows, int64_t n, int k, bool is_l2, int64_t* out_keys,
                                 float* out_d, int32_t* ovf_flags, hipStream_t s);
hipError_t launch_row_select(const float* vals, int64_t nrows, int64_t n, int k, bool is_l2,
                             int64_t* out_keys, float* out_d, const int32_t* row_flags, hipStream_t s,
                             unsigned long long* sort_scratch = nullptr);
size_t row_select_lds_max_k();
// rows of different length: row r has n = list_len[keys[r * key_stride]] values at vals + r * stride
hipError_t launch_row_select_var(const float* vals, int64_t stride, const int64_t* keys, int key_stride,
                                 const int64_t* list_len, int64_t nrows, int k, bool is_l2, int64_t* out_keys,
                                 float* out_d, hipStream_t s, int64_t n_cap = 0, const int32_t* n_row = nullptr);
size_t row_select_max_k();

// ---- coarse_gemm.hip: fp32 MFMA prefilter for the coarse quantizer ----
hipError_t launch_row_norms(const float* x, int64_t n, int d, float* out, hipStream_t s);
hipError_t launch_coarse_gemm(const float* q, const float* qnorm, const float* c, const float* cnorm,
                              int64_t nq, int64_t nlist, int d, bool is_l2, float* out, hipStream_t s);
hipError_t launch_coarse_rerank(const float* queries, const float* centroids, int d, int64_t nq,
                                int64_t nlist, int ncand, const int64_t* cand_keys,
                                const float* cand_approx, int nprobe, bool is_l2, const float* qnorm,
                                float cnorm_max, int64_t* out_keys, float* out_d, int32_t* fail_flags,
                                unsigned long long* nfail, hipStream_t s, const int32_t* cand_cnt = nullptr,
                                const float* bound = nullptr,
                                bool need_kth = true);
// the coarse prefilter on the bf16 matrix pipe with the selection fused (two GEMM passes, no distance matrix): per query
// bound[q] and cand_cnt[q] unordered candidates (capacity cap) = every centroid with approx <= bound[q]
bool coarse_bf16_supports(int64_t nlist, int ncand);
int64_t coarse_bf16_groups(int64_t nlist);
int coarse_bf16_slabs(int d);
hipError_t launch_coarse_bf16_split(const float* x, int64_t n, int d, void* out, hipStream_t s);
hipError_t launch_coarse_bf16(const void* q_split, const float* qnorm, const void* c_split, const float* cnorm, int64_t nq,
                              int64_t nlist, int d, bool is_l2, int ncand, int cap, float* gmin, float* bound,
                              int32_t* cand_cnt, int64_t* cand, hipStream_t s, bool bound_given = false);
hipError_t launch_coarse_ext_bound(float* kth, const float* chunk_d, int k, int64_t nq, const float* qnorm, float cnorm_max, int d,
                                   bool is_l2, float* bound_out, hipStream_t s);

// ---- refine.hip ----
// row_type: 0 fp32 rows (base), 1 fp16, 2 bf16, 3 per-dimension 8-bit codes with sq_trained = vmin[d], vdiff[d] (base is
// then the byte / half-word array, row r = id id_base + r)
hipError_t launch_refine(const float* base, int64_t nbase, int64_t id_base, int d, const float* queries,
                         int64_t nq, const int64_t* cand, int kbase, int k, bool is_l2, float* out_d,
                         int64_t* out_i, hipStream_t s, int row_type = 0, const float* sq_trained = nullptr,
                         const float* dist_in = nullptr, float* dist_out = nullptr);
hipError_t launch_rows_encode16(const float* x, int64_t n_elems, bool bf16, uint16_t* out, hipStream_t s);
hipError_t launch_rows_encode6(const float* x, int64_t n, int d, const float* trained, uint8_t* out, hipStream_t s);
hipError_t launch_rows_encode_i8(const float* x, int64_t n_elems, uint8_t* out, hipStream_t s);
hipError_t launch_rows_encode4u(const float* x, int64_t n, int d, const float* trained, uint8_t* out, hipStream_t s);
// one pass of the radix select behind the quantile range of an sq4u store: hist[2][256] (64-bit counters, accumulated)
hipError_t launch_rows_key_hist(const float* x, int64_t n, uint32_t mask, uint32_t prefix_lo, uint32_t prefix_hi, int shift,
                                unsigned long long* hist, hipStream_t s);

// ---- build.hip: Train / Add on the device ----
// direct map of an IVF-Flat index (GetVectorByIds): ids sorted with the columns of their rows in the interleaved store
size_t idmap_sort_tmp_bytes(int64_t n);
hipError_t launch_idmap_build(const int64_t* ids, const int64_t* list_row_off, const int64_t* list_blk_off, int64_t nlist,
                              int64_t ntotal, int64_t* col_tmp, int64_t* ids_sorted, int64_t* col_sorted, void* tmp,
                              size_t tmp_bytes, hipStream_t s);
hipError_t launch_idmap_gather(const int64_t* want, int64_t n, const int64_t* ids_sorted, const int64_t* col_sorted,
                               int64_t ntotal, const float4* rows, int d, float* out, int32_t* missing, hipStream_t s,
                               uint8_t* found = nullptr); // found [n] (optional): 1 = the id is stored here, its row was written
hipError_t launch_gather_rows(const float* x, const int64_t* rows, int64_t n, int d, float* out, hipStream_t s);
hipError_t launch_residual(const float* x, const float* cen, const int64_t* assign, int64_t n, int d, float* out,
                           hipStream_t s);
hipError_t launch_nearest_small(const float* x, int64_t n, int64_t ld, int off, int dsub, const float* cb, int ksub,
                                int32_t* out_idx, hipStream_t s);
hipError_t launch_pq_encode(const float* resid, int64_t n, int d, int M, const float* cb, uint8_t* codes, hipStream_t s);
hipError_t launch_sq8_encode(const float* resid, int64_t n, int d, const float* trained, uint8_t* codes, hipStream_t s);
hipError_t launch_col_minmax(const float* x, int64_t n, int d, float* vmin, float* vmax, hipStream_t s);
size_t group_rows_tmp_bytes(int64_t n, int64_t k);
hipError_t group_rows_by_key(const int64_t* keys64, const int32_t* keys32, int64_t n, int64_t k, int32_t* sorted_rows,
                             int64_t* seg_off, void* tmp, size_t tmp_bytes, hipStream_t s);
hipError_t launch_centroid_update(const float* x, int64_t ld, int off, int dsub, const int32_t* sorted_rows,
                                  const int64_t* seg_off, int64_t k, float* centroids, float* hassign, hipStream_t s);
hipError_t launch_merge_lists(const uint8_t* old_codes, const int64_t* old_ids, const int64_t* old_off,
                              const uint8_t* new_codes, const int64_t* new_ids, const int32_t* new_rows,
                              const int64_t* new_seg, const int64_t* out_off, int64_t nlist, int64_t code_size,
                              uint8_t* out_codes, int64_t* out_ids, hipStream_t s);
hipError_t launch_iota_i64(int64_t* out, int64_t n, int64_t base, hipStream_t s);
// k = 1 assignment under the inner product with the reference's first-maximum tie rule (see build.hip)
hipError_t launch_assign_first_max_ip(const float* x, int64_t n, int d, const float* cen, int64_t nlist,
                                      const int64_t* top2_keys, const float* top2_dis, int64_t* out, hipStream_t s);
// fvec_renorm_L2 of k rows in place; inv_tmp: k floats of scratch
hipError_t launch_renorm_rows(float* x, int64_t k, int d, float* inv_tmp, hipStream_t s);

// ---- prims.hip ----
hipError_t launch_deinterleave_lists(const uint4* rows, const int64_t* list_row_off, const int64_t* list_len,
                                     const int64_t* list_blk_off, int64_t nlist, int64_t code_size, uint8_t* dst,
                                     hipStream_t s);
hipError_t launch_fvec_ny(float* out, const float* x, const float* y, int64_t d, int64_t ny,
                          bool is_l2, hipStream_t s);
hipError_t launch_fvec_norms(float* out, const float* x, int64_t d, int64_t n, hipStream_t s);
hipError_t launch_fvec_madd(int64_t n, const float* a, float bf, const float* b, float* c,
                            hipStream_t s);
hipError_t launch_int8_ny(float* out, const int8_t* x, const int8_t* y, int64_t d, int64_t ny,
                          bool is_l2, hipStream_t s);
// op: 0 L2sqr, 1 inner product, 2 norm (faiss float accumulator), 3 L1, 4 Linf, 5 norm (_ref: double accumulator)
hipError_t launch_fvec_rows(int op, float* out, const float* x, const float* y, int64_t d, int64_t ny, hipStream_t s);
// dtype 0 fp16, 1 bf16, 2 int8; op 0 L2sqr, 1 inner product, 2 norm_L2sqr
hipError_t launch_typed_rows(int dtype, int op, float* out, const void* x, const void* y, int64_t d, int64_t ny,
                             hipStream_t s);
hipError_t launch_ivec_ny(int32_t* out, const int8_t* x, const int8_t* y, int64_t d, int64_t ny, bool is_l2,
                          hipStream_t s);
hipError_t launch_argmin(const float* v, int64_t n, float limit, int64_t none_value, int64_t* d_idx, hipStream_t s);
hipError_t launch_fvec_madd_and_argmin(int64_t n, const float* a, float bf, const float* b, float* c, int64_t* d_imin,
                                       hipStream_t s);
hipError_t launch_l2_transposed(float* dis, const float* x, const float* y, const float* y_sqlen, int64_t d,
                                int64_t d_offset, int64_t ny, hipStream_t s);
// dtype -1 fp32, 0 fp16, 1 bf16, 2 int8
hipError_t launch_batch4(int dtype, bool is_l2, const void* x, const void* y0, const void* y1, const void* y2,
                         const void* y3, int64_t d, float* out, hipStream_t s);

} // namespace knhip
