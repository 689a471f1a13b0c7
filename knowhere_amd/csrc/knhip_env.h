// knowhere_amd/csrc/knhip_env.h -- every KNHIP_* environment switch of libknhip.so, read in ONE place (DESIGN 4.6 lists what
// each one is for).  None of them changes a result bit except KNHIP_TIES (the licensed canonical answer, include/knhip.h);
// they pick between kernels that return the same bits, size scratch, or switch experiments on.  The library never reads the
// environment anywhere else.
//
// Two moments of reading, by what the tests and tools rely on:
//   * LAYOUT switches: read when an index lays out its lists / takes its rows / takes its coarse quantizer -- the index keeps
//     what it read, a later change of the variable does not touch an existing index;
//   * SEARCH switches: read by every search (tests flip them between two searches of one index: KNHIP_TIES,
//     KNHIP_RANGE_NO_WAVES) -- six lookups per call.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace knhip_host {

struct EnvLayout {
    bool coarse_given;     // KNHIP_COARSE set at all (BRUTE_FORCE then keeps the row scan: the stage's machinery is under test)
    int coarse_gemm;       // KNHIP_COARSE: "exact" -> 0 (all-pairs kernel), "fp32" -> 1 (fp32 GEMM prefilter), else 2 (split bf16)
    size_t aos_keep_bytes; // KNHIP_AOS_KEEP_MB: flat / SQ8 indexes above it keep only the interleaved layout (default 8 GiB)
    bool rank0_select;     // KNHIP_RANK0=0: no rank-0 phase (dump + select of the nearest list)
    bool cand_hist;        // KNHIP_HIST=0: no per-query candidate histogram
    int pq_q4;             // KNHIP_Q4=0|1|2: the 4-query exact ADC kernel off / on / where it wins (default)
    int mscan;             // KNHIP_MSCAN=0|1|2: the matrix-core prefilter off / forced / where it wins (default)
    bool flat_bf16;        // KNHIP_MSCAN_FLAT=fp32: the IVF-Flat filter on the fp32 kernel
    int mscan_cap;         // KNHIP_MSCAN_CAP=n: candidate slots per query (0 = the library's choice)
    int pqd_spill_cap;     // KNHIP_PQD_SPILL=n: parked records per wave in global memory (0 = the library's choice)
    int pqf;               // KNHIP_PQF=0|1: the IVF-PQ prefilter off / forced (default: where it wins) -> 0 / 2 / 1
    bool pqf_guard;        // KNHIP_PQF_GUARD=0: no selectivity guard
    int pqf_form;          // KNHIP_PQF_FORM=half|int8|decode -> 1 / 2 / 3 (0 = the guard's choice)
    bool pq_v1;            // KNHIP_PQ_V1=1: m = 32 on the round-1 layout
    bool bf_exact;         // KNHIP_BF=exact: BRUTE_FORCE keeps the row scan
    int pq_waves;          // KNHIP_PQ_WAVES=4|8|16: waves per workgroup of the round-1 ADC kernel (tuning)
};

struct EnvSearch {
    int ms_sample_rows;    // KNHIP_MS_SAMPLE_ROWS=n: rows of the row kinds' sample pass (0 = the library's choice)
    int pq_sample_rows;    // KNHIP_PQ_SAMPLE_ROWS=n: rows of the IVF-PQ sample pass, at most (0 = the library's choice)
    bool no_side_stream;   // KNHIP_NO_SIDE_STREAM: the work table is built on the search stream
    bool guard_sync;       // KNHIP_PQF_GUARD_SYNC: the guard reads its counters in every batch
    bool range_no_waves;   // KNHIP_RANGE_NO_WAVES: range search probes all lists in one pass
    bool ties_canonical;   // KNHIP_TIES=canonical|0: the canonical k without the boundary rule
    bool ties_trace;       // KNHIP_TIES_TRACE: one line per flagged batch on stderr
};

inline bool env_is(const char* v, const char* what) { return v != nullptr && std::strcmp(v, what) == 0; }
inline int env_digit(const char* v, int lo, int hi, int dflt) { return (v && v[0] >= '0' + lo && v[0] <= '0' + hi) ? v[0] - '0' : dflt; }
inline int env_count(const char* v) { return (v && *v) ? std::max(0, std::atoi(v)) : 0; }

inline EnvLayout env_layout() {
    EnvLayout e{};
    const char* c = std::getenv("KNHIP_COARSE");
    e.coarse_given = c != nullptr;
    e.coarse_gemm = env_is(c, "exact") ? 0 : env_is(c, "fp32") ? 1 : 2;
    const char* keep = std::getenv("KNHIP_AOS_KEEP_MB");
    e.aos_keep_bytes = (keep && *keep) ? (size_t)std::max<long long>(0, std::atoll(keep)) << 20 : (size_t)8 << 30;
    const char* r0 = std::getenv("KNHIP_RANK0");
    e.rank0_select = !(r0 && r0[0] == '0');
    const char* h = std::getenv("KNHIP_HIST");
    e.cand_hist = !(h && h[0] == '0');
    e.pq_q4 = env_digit(std::getenv("KNHIP_Q4"), 0, 2, 2);
    e.mscan = env_digit(std::getenv("KNHIP_MSCAN"), 0, 2, 2);
    e.flat_bf16 = !env_is(std::getenv("KNHIP_MSCAN_FLAT"), "fp32");
    e.mscan_cap = env_count(std::getenv("KNHIP_MSCAN_CAP"));
    e.pqd_spill_cap = env_count(std::getenv("KNHIP_PQD_SPILL"));
    const char* pf = std::getenv("KNHIP_PQF");
    e.pqf = (pf && pf[0] == '1') ? 2 : (pf && pf[0] == '0') ? 0 : 1;
    const char* pg = std::getenv("KNHIP_PQF_GUARD");
    e.pqf_guard = !(pg && pg[0] == '0');
    const char* pm = std::getenv("KNHIP_PQF_FORM");
    e.pqf_form = (pm && pm[0] == 'h') ? 1 : (pm && pm[0] == 'i') ? 2 : (pm && pm[0] == 'd') ? 3 : 0;
    const char* v1 = std::getenv("KNHIP_PQ_V1");
    e.pq_v1 = v1 && v1[0] == '1';
    e.bf_exact = env_is(std::getenv("KNHIP_BF"), "exact");
    const char* w = std::getenv("KNHIP_PQ_WAVES");
    e.pq_waves = w ? std::atoi(w) : 8;
    return e;
}

inline EnvSearch env_search() {
    EnvSearch e{};
    const char* a = std::getenv("KNHIP_MS_SAMPLE_ROWS");
    e.ms_sample_rows = a ? std::max(1, std::atoi(a)) : 0;
    const char* b = std::getenv("KNHIP_PQ_SAMPLE_ROWS");
    e.pq_sample_rows = b ? std::max(1, std::atoi(b)) : 0;
    e.no_side_stream = std::getenv("KNHIP_NO_SIDE_STREAM") != nullptr;
    e.guard_sync = std::getenv("KNHIP_PQF_GUARD_SYNC") != nullptr;
    e.range_no_waves = std::getenv("KNHIP_RANGE_NO_WAVES") != nullptr;
    const char* t = std::getenv("KNHIP_TIES");
    e.ties_canonical = t && (t[0] == 'c' || t[0] == 'C' || t[0] == '0');
    e.ties_trace = std::getenv("KNHIP_TIES_TRACE") != nullptr;
    return e;
}

} // namespace knhip_host
