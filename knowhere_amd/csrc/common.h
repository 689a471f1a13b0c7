// knowhere_amd/csrc/common.h -- shared device/host helpers for the gfx950 kernels.
//
// Conventions used by every kernel in this directory:
//  * wave = 64 lanes, hard-coded (CDNA4); block sizes are multiples of 64.
//  * "exact" arithmetic: the reference's scalar operation order with one IEEE rounding per
//    operation.  The translation units are built with -ffp-contract=off and the helpers
//    below use the _rn intrinsics, so no mul+add is ever fused behind our back.
//  * canonical result order (what heap_reorder produces in the reference,
//    thirdparty/faiss/faiss/utils/Heap.h:427-457 with the comparators of
//    utils/ordered_key_value.h:43-83):  L2: (dist asc, id asc);  IP: (dist desc, id desc).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include "kernels.h"

#define KN_WAVE 64

// ---- exact (never contracted) fp32 ops -----------------------------------------------------
__device__ __forceinline__ float fadd_x(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float fsub_x(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float fmul_x(float a, float b) { return __fmul_rn(a, b); }

// res += (x - y)^2 with the reference's three roundings (src/simd/distances_ref.cc:30-37)
__device__ __forceinline__ float l2_step(float res, float x, float y) {
    const float t = fsub_x(x, y);
    return fadd_x(res, fmul_x(t, t));
}
// res += x * y (src/simd/distances_ref.cc:21-28)
__device__ __forceinline__ float ip_step(float res, float x, float y) {
    return fadd_x(res, fmul_x(x, y));
}

// COSINE with stored norms: mode 1 = <q, y> / norm (one IEEE division), mode 2 = clamp(<q, y> * inverse norm, -1, 1)
__device__ __forceinline__ float cosine_finish(float ip, float scale, int mode) {
    if (mode == 1) {
        return __fdiv_rn(ip, scale);
    }
    const float v = __fmul_rn(ip, scale);
    return v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);
}

// ---- canonical ordering ----------------------------------------------------------------------
// IS_L2: "a is better than b"  <=>  (da < db) || (da == db && ia < ib)
// IP   :                         (da > db) || (da == db && ia > ib)
template <bool IS_L2>
__device__ __forceinline__ bool better(float da, int64_t ia, float db, int64_t ib) {
    if (IS_L2) {
        return (da < db) || (da == db && ia < ib);
    }
    return (da > db) || (da == db && ia > ib);
}
template <bool IS_L2>
__device__ __forceinline__ bool better_or_equal_dist(float da, float db) {
    return IS_L2 ? (da <= db) : (da >= db);
}
template <bool IS_L2>
__device__ __forceinline__ float worst_dist() {
    // C::neutral(): +FLT_MAX for the max-heap (L2), lowest() for the min-heap (IP)
    return IS_L2 ? FLT_MAX : -FLT_MAX;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (KN_WAVE - 1); }

// 64-bit shuffle helpers (ds_bpermute based; used only on slow paths)
__device__ __forceinline__ int64_t shfl_i64(int64_t v, int src) {
    int lo = __shfl((int)(v & 0xffffffffll), src, KN_WAVE);
    int hi = __shfl((int)(v >> 32), src, KN_WAVE);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}

// ---- cross-lane helpers that stay in the VALU (no LDS round trip) -------------------------------
__device__ __forceinline__ float readlane_f(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ int64_t readlane_i64(int64_t v, int l) {
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), l);
    const int hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}
// value of lane-1 (lane 0 keeps its own value): DPP wave_shr:1
__device__ __forceinline__ int wave_shr1(int v) {
    return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);
}

// ---- WaveTopK: a wave-resident sorted list of the K best (dist, idx) ---------------------------
// Element e (0 = best) lives in lane e % 64, register e / 64.  R = registers per lane, capacity
// 64*R >= k.  All methods must be called by the full wave with wave-uniform arguments unless
// noted.  `idx` is whatever the caller orders ties by (list offset, row number or id) -- it is
// compared as a signed 64-bit integer in the canonical direction.  Insertion is one compare +
// ballot per register to find the position, then a DPP shift of the tail: no LDS traffic.
template <class T>
__device__ __forceinline__ T readlane_idx(T v, int l);
template <>
__device__ __forceinline__ int64_t readlane_idx<int64_t>(int64_t v, int l) {
    return readlane_i64(v, l);
}
template <>
__device__ __forceinline__ int32_t readlane_idx<int32_t>(int32_t v, int l) {
    return __builtin_amdgcn_readlane(v, l);
}
template <class T>
__device__ __forceinline__ T wave_shr1_idx(T v);
template <>
__device__ __forceinline__ int64_t wave_shr1_idx<int64_t>(int64_t v) {
    const int lo = wave_shr1((int)(v & 0xffffffffll));
    const int hi = wave_shr1((int)(v >> 32));
    return ((int64_t)hi << 32) | (uint32_t)lo;
}
template <>
__device__ __forceinline__ int32_t wave_shr1_idx<int32_t>(int32_t v) {
    return wave_shr1(v);
}

// IdxT = int32_t where the tie-break index is a list offset (saves a VGPR per register)
template <bool IS_L2, int R, class IdxT = int64_t>
struct WaveTopK {
    float d[R];
    IdxT i[R];
    int k;

    __device__ __forceinline__ void init(int k_) {
        k = k_;
#pragma unroll
        for (int r = 0; r < R; r++) {
            d[r] = worst_dist<IS_L2>();
            i[r] = -1;
        }
    }

    // distance / idx of the current k-th element (wave-uniform)
    __device__ __forceinline__ float kth_dist() const {
        const int e = k - 1;
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (r == e / KN_WAVE) {
                v = readlane_f(d[r], e % KN_WAVE);
            }
        }
        return v;
    }
    __device__ __forceinline__ IdxT kth_idx() const {
        const int e = k - 1;
        IdxT v = -1;
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (r == e / KN_WAVE) {
                v = readlane_idx<IdxT>(i[r], e % KN_WAVE);
            }
        }
        return v;
    }

    // would (dist, idx) enter the list?  Empty slots have idx -1 and the neutral distance: a real
    // candidate whose distance equals the neutral value is rejected, as the reference's strict
    // heap admission does (thirdparty/faiss/faiss/impl/ResultHandler.h:271-278).
    __device__ __forceinline__ bool admits(float dist, IdxT idx, float kd, IdxT ki) const {
        if (ki < 0) {
            return IS_L2 ? (dist < kd) : (dist > kd);
        }
        return better<IS_L2>(dist, idx, kd, ki);
    }

    // insert a wave-uniform candidate that is known to be admissible
    __device__ __forceinline__ void insert(float dist, IdxT idx) {
        const int lane = lane_id();
        // position = number of stored elements strictly better than the candidate
        int pos = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const bool b = (i[r] >= 0) && better<IS_L2>(d[r], i[r], dist, idx);
            pos += __popcll(__ballot(b));
        }
        // shift elements [pos, k-2] one place towards the tail
        float carry_d = 0.f;
        IdxT carry_i = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            // pre-shift value of this register's last lane feeds lane 0 of the next register
            const float last_d = (R > 1) ? readlane_f(d[r], KN_WAVE - 1) : 0.f;
            const IdxT last_i = (R > 1) ? readlane_idx<IdxT>(i[r], KN_WAVE - 1) : 0;
            float up_d = __builtin_bit_cast(float, wave_shr1(__builtin_bit_cast(int, d[r])));
            IdxT up_i = wave_shr1_idx<IdxT>(i[r]);
            if (R > 1 && lane == 0) {
                up_d = carry_d;
                up_i = carry_i;
            }
            const int e = r * KN_WAVE + lane;
            if (e > pos) {
                d[r] = up_d;
                i[r] = up_i;
            } else if (e == pos) {
                d[r] = dist;
                i[r] = idx;
            }
            carry_d = last_d;
            carry_i = last_i;
        }
        // elements beyond k-1 are garbage by construction; re-neutralise them so kth/merge
        // never see them
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int e = r * KN_WAVE + lane;
            if (e >= k) {
                d[r] = worst_dist<IS_L2>();
                i[r] = -1;
            }
        }
    }

    // write the k elements (best first) to dst arrays (global or LDS), one per lane-slot
    template <class TD, class TI>
    __device__ __forceinline__ void store(TD* dd, TI* ii) const {
        const int lane = lane_id();
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int e = r * KN_WAVE + lane;
            if (e < k) {
                dd[e] = d[r];
                ii[e] = i[r];
            }
        }
    }
};

// ---- per-query global threshold, shared by every wave that scans for the query -------------------
// Any wave whose local list is full holds k real candidates, so its k-th distance bounds the final
// k-th distance from the losing side; publishing the best such bound lets every other wave drop
// candidates that cannot make the final top-k.  Candidates EQUAL to the bound are kept (they may
// win the canonical id tie-break).  Stale reads only make the bound looser: never incorrect.
template <bool IS_L2>
__device__ __forceinline__ float gthr_load(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool IS_L2>
__device__ __forceinline__ void gthr_publish(float* p, float v) {
    // called by one lane.  min (L2) / max (IP) of floats as ONE fire-and-forget integer atomic -- no value comes
    // back, the wave never waits for the round trip to the L2: for v >= 0 the bit pattern orders like a signed
    // int, for v < 0 it orders inversely as an unsigned int.  (-0.0f takes the signed branch as INT_MIN: the bound
    // is then simply not tightened, which is always valid.)  NaN never reaches here.
    int* ip = reinterpret_cast<int*>(p);
    unsigned int* up = reinterpret_cast<unsigned int*>(p);
    if (IS_L2) {
        if (v >= 0.f) {
            (void)__hip_atomic_fetch_min(ip, __float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            (void)__hip_atomic_fetch_max(up, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        if (v >= 0.f) {
            (void)__hip_atomic_fetch_max(ip, __float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            (void)__hip_atomic_fetch_min(up, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
template <bool IS_L2>
__device__ __forceinline__ bool within_gthr(float dist, float g) {
    return IS_L2 ? (dist <= g) : (dist >= g);
}
template <bool IS_L2>
__device__ __forceinline__ float tighter(float a, float b) {
    return IS_L2 ? fminf(a, b) : fmaxf(a, b);
}

// ---- order-preserving integer keys for distances (smaller key = better candidate) ------------------
// used by the per-query candidate histogram (pq_scan_v2.hip): binning in the integer key domain has no
// rounding, so "every distance in bins <= b is <= bound(b)" holds exactly
__device__ __forceinline__ uint32_t okey_f32(float d) {
    const uint32_t b = __float_as_uint(d);
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float okey_inv_f32(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}
template <bool IS_L2>
__device__ __forceinline__ uint32_t dist_key(float d) {
    return IS_L2 ? okey_f32(d) : ~okey_f32(d);
}
template <bool IS_L2>
__device__ __forceinline__ float dist_key_inv(uint32_t k) {
    return okey_inv_f32(IS_L2 ? k : ~k);
}
constexpr int KN_HIST_BINS = 64;
constexpr uint32_t KN_HIST_OFF = 0xffu; // meta.y: histogram disabled for this query
__device__ __forceinline__ uint32_t hist_bin(uint32_t key, uint32_t lo, uint32_t shift) {
    if (key <= lo) {
        return 0u;
    }
    const uint32_t b = (key - lo) >> shift;
    return b < (uint32_t)(KN_HIST_BINS - 1) ? b : (uint32_t)(KN_HIST_BINS - 1);
}

// runtime k -> compile-time R dispatch (k <= 1024)
#define KN_MAX_K 1024
#define KN_DISPATCH_R(k, ...)                \
    do {                                     \
        if ((k) <= 64) {                     \
            constexpr int R_ = 1;            \
            __VA_ARGS__                      \
        } else if ((k) <= 128) {             \
            constexpr int R_ = 2;            \
            __VA_ARGS__                      \
        } else if ((k) <= 256) {             \
            constexpr int R_ = 4;            \
            __VA_ARGS__                      \
        } else if ((k) <= 512) {             \
            constexpr int R_ = 8;            \
            __VA_ARGS__                      \
        } else {                             \
            constexpr int R_ = 16;           \
            __VA_ARGS__                      \
        }                                    \
    } while (0)

// ---- bitset (bit set => filtered out, LSB first; include/knowhere/bitsetview_idselector.h) ---
__device__ __forceinline__ bool bitset_filtered(const uint8_t* bitset, int64_t nbits, int64_t id) {
    if (bitset == nullptr || id < 0 || id >= nbits) {
        return false;
    }
    return (bitset[id >> 3] >> (id & 7)) & 1;
}

// ---- XCD-aware work mapping --------------------------------------------------------------------
// Consecutive work items share an inverted list; the dispatcher places block b on XCD b % 8
// (speed only, never correctness), so give each XCD a contiguous run of items and its private
// L2 sees every list once.
__device__ __forceinline__ int64_t xcd_item(int64_t b, int64_t nitems) {
    const int64_t per = (nitems + 7) / 8;
    return (b % 8) * per + (b / 8);
}

// ---- per-(query,slot) partial result layout -----------------------------------------------------
// partial_d/partial_i : [nq][nslot][k], slot = probe rank (IVF) or base chunk (brute force)

