// knowhere_amd/csrc/ms_common.h -- device helpers shared by the prefilter kernels (mfma_scan.hip: fp32 rows and SQ8 codes
// on the matrix cores; pq_filter.hip: half-precision ADC): candidate append, candidate-histogram bound, valid-row mask.
#pragma once
#include "common.h"
#include "kernels.h"

namespace knhip {

// ---- candidate append (slow path of the epilogue) ---------------------------------------------------------------
// `pess` is the candidate's pessimistic distance (approx widened by eps: the exact distance is at least as good).
// It feeds the query's candidate histogram (64 bins over the key range [best, k-th] of the sample): a unit that
// starts later and finds cum(bins <= b) >= k knows that k unfiltered rows are at least as good as the upper edge of
// bin b -- a valid and ever tightening bound on the final k-th distance (the scheme of pq_scan_v2.hip).
template <bool IS_L2>
__device__ __forceinline__ void ms_emit(const MScanArgs& a, int32_t q, int32_t slot, int64_t row_off, int64_t pos,
                                        float pess) {
    if (a.bitset != nullptr && bitset_filtered(a.bitset, a.bitset_nbits, a.ids[row_off + pos])) {
        return;
    }
    const int n = atomicAdd(a.cand_cnt + q, 1);
    if (n < a.cap) {
        a.cand[(int64_t)q * a.cap + n] = ((int64_t)slot << 32) | (int64_t)(uint32_t)pos;
        if (a.cand_pess != nullptr) {
            a.cand_pess[(int64_t)q * a.cap + n] = pess;
        }
    } else {
        a.overflow[q] = 1;
        a.overflow[a.nq] = 1; // "some query overflowed": the fallback kernels return at once while this stays 0
    }
    if (a.ghist != nullptr) {
        const uint2 mt = a.gmeta[q];
        if (mt.y != KN_HIST_OFF) {
            atomicAdd(a.ghist + (int64_t)q * KN_HIST_BINS + hist_bin(dist_key<IS_L2>(pess), mt.x, mt.y), 1u);
        }
    }
}

// bound on the query's final k-th distance from its candidate histogram (full wave: lane = bin); the neutral value
// when the histogram is off or k candidates have not been seen yet.  Two halves, so that a caller can issue the loads
// (ms_hist_load) together with its other loads and evaluate (ms_hist_eval) when they are back.
struct MsHist {
    bool on;      // (wave-uniform) the other fields are loaded; they stay unset otherwise (no merge, no early wait)
    uint2 mt;     // the query's histogram origin and shift (shift == KN_HIST_OFF: no histogram)
    uint32_t cum; // this lane's bin
};

__device__ __forceinline__ void ms_hist_load(const MScanArgs& a, int32_t q, MsHist& h) {
    h.on = a.ghist != nullptr && q >= 0;
    if (h.on) {
        h.mt = a.gmeta[q];
        h.cum = __hip_atomic_load(a.ghist + (int64_t)q * KN_HIST_BINS + lane_id(), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <bool IS_L2>
__device__ __forceinline__ float ms_hist_eval(const MsHist& h, int k) {
    float bound = worst_dist<IS_L2>();
    if (!h.on) {
        return bound;
    }
    const uint2 mt = h.mt;
    if (mt.y == KN_HIST_OFF) {
        return bound;
    }
    const int lane = lane_id();
    uint32_t cum = h.cum;
#pragma unroll
    for (int dlt = 1; dlt < KN_WAVE; dlt <<= 1) {
        const uint32_t up = __shfl_up(cum, dlt, KN_WAVE);
        cum += lane >= dlt ? up : 0u;
    }
    const unsigned long long reach = __ballot(cum >= (uint32_t)k);
    const int b = reach ? __ffsll((long long)reach) - 1 : KN_HIST_BINS;
    if (b < KN_HIST_BINS - 1) { // (the last bin also collects everything beyond the range)
        const unsigned long long edge = (unsigned long long)mt.x + (((unsigned long long)b + 1ull) << mt.y) - 1ull;
        if (edge < 0xffffffffull) {
            const float e = dist_key_inv<IS_L2>((uint32_t)edge);
            if (e == e && fabsf(e) < FLT_MAX) {
                bound = e;
            }
        }
    }
    return bound;
}

template <bool IS_L2>
__device__ __forceinline__ float ms_hist_bound(const MScanArgs& a, int32_t q, int k) {
    MsHist h;
    ms_hist_load(a, q, h);
    return ms_hist_eval<IS_L2>(h, k);
}

// The same bound evaluated by ONE lane for its own query (a unit's prologue: one thread per pair, every pair's loads in
// flight together, instead of one wave per pair and a memory round trip per pair one after the other).  64-bit relaxed
// atomic loads at agent scope, like ms_hist_load: counts that other CUs bumped a moment ago are seen (a stale row would
// only loosen the bound).
template <bool IS_L2>
__device__ __forceinline__ float ms_hist_bound_lane(const MScanArgs& a, int32_t q, int k) {
    float bound = worst_dist<IS_L2>();
    if (a.ghist == nullptr || q < 0) {
        return bound;
    }
    const uint2 mt = a.gmeta[q];
    if (mt.y == KN_HIST_OFF) {
        return bound;
    }
    const unsigned long long* hp = reinterpret_cast<const unsigned long long*>(a.ghist + (int64_t)q * KN_HIST_BINS);
    unsigned long long w[KN_HIST_BINS / 2];
#pragma unroll
    for (int i = 0; i < KN_HIST_BINS / 2; i++) {
        w[i] = __hip_atomic_load(hp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint32_t cum = 0;
    int b = KN_HIST_BINS;
#pragma unroll
    for (int i = 0; i < KN_HIST_BINS / 2; i++) {
        cum += (uint32_t)w[i];
        b = (b == KN_HIST_BINS && cum >= (uint32_t)k) ? 2 * i : b;
        cum += (uint32_t)(w[i] >> 32);
        b = (b == KN_HIST_BINS && cum >= (uint32_t)k) ? 2 * i + 1 : b;
    }
    if (b < KN_HIST_BINS - 1) { // (the last bin also collects everything beyond the range)
        const unsigned long long edge = (unsigned long long)mt.x + (((unsigned long long)b + 1ull) << mt.y) - 1ull;
        if (edge < 0xffffffffull) {
            const float e = dist_key_inv<IS_L2>((uint32_t)edge);
            if (e == e && fabsf(e) < FLT_MAX) {
                bound = e;
            }
        }
    }
    return bound;
}

// 64-bit mask of the block's rows that take part (inside the list, not filtered): lane = row
__device__ __forceinline__ unsigned long long ms_valid_rows(const MScanArgs& a, int64_t b, int64_t len, int64_t row_off) {
    const int64_t row = b * 64 + lane_id();
    bool v = row < len;
    if (v && a.bitset != nullptr) {
        v = !bitset_filtered(a.bitset, a.bitset_nbits, a.ids[row_off + row]);
    }
    return __ballot(v);
}

} // namespace knhip
