// knowhere_amd/csrc/build.hip -- Train / Add on the device: the kernels behind knhip_index_train* / knhip_index_add*.
//
// What they restate (reference, all under thirdparty/faiss/faiss unless noted):
//   k-means            Clustering::train_encoded (Clustering.cpp:150-380): assignment by the index's own exact search
//                      (k = 1), detail::compute_centroids (impl/ClusteringHelpers.cpp:101-171: members summed in
//                      index order, then scaled by 1 / count), detail::split_clusters (:177-240, on the host),
//                      post_process_centroids (Clustering.cpp:35-38: spherical renormalisation, what IndexIVF turns
//                      on for the inner product, IndexIVF.cpp:178-181).
//   residual           Index::compute_residual (Index.cpp): x - centroid, element-wise.
//   PQ encode          ProductQuantizer::compute_code (impl/ProductQuantizer.cpp:230-299) ->
//                      fvec_L2sqr_ny_nearest (utils/distances_simd.cpp:106-125): exact squared L2 to the 256 entries of
//                      sub-quantizer m in scalar order, FIRST minimum wins.
//   SQ8 encode         QuantizerTemplate<Codec8bit, NON_UNIFORM>::encode_vector (impl/scalar_quantizer/quantizers.h:
//                      108-146) + Codec8bit::encode_component (codecs.h:26-35): xi = (x - vmin) / vdiff clamped to
//                      [0, 1] (0 when vdiff == 0), code = (int)(255 * xi).
//   SQ8 training       train_NonUniform / train_Uniform with RS_minmax, rangestat_arg = 0 (impl/ScalarQuantizer.cpp,
//                      trained = vmin[d], vdiff[d]).
// Every distance is the scalar sequential form with one rounding per operation (common.h), so codes and assignments are
// bit-equal to the scalar reference given the same codebooks / centroids (tests/test_gpu_build.py vs oracle.c).
#include "common.h"
#include "kernels.h"

#include <hipcub/hipcub.hpp>

namespace knhip {

__global__ void gather_rows_kernel(const float* __restrict__ x, const int64_t* __restrict__ rows, int64_t n, int d,
                                   float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * d) {
        return;
    }
    const int64_t i = t / d;
    const int j = (int)(t % d);
    out[t] = x[rows[i] * d + j];
}

hipError_t launch_gather_rows(const float* x, const int64_t* rows, int64_t n, int d, float* out, hipStream_t s) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n * d + 255) / 256)), dim3(256), 0, s, x, rows, n, d, out);
    return hipGetLastError();
}

// ---- direct map of an IVF-Flat index: id -> column of the interleaved store (GetVectorByIds) ------------------------
// col[p] for storage position p (list-sorted order): 64 * first block of its list + offset inside the list
__global__ void idmap_cols_kernel(const int64_t* __restrict__ list_row_off, const int64_t* __restrict__ list_blk_off,
                                  int64_t nlist, int64_t ntotal, int64_t* __restrict__ col) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= ntotal) {
        return;
    }
    int64_t lo = 0, hi = nlist - 1; // the last list whose first row is <= p (empty lists share their successor's offset)
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (list_row_off[mid] <= p) {
            lo = mid;
        } else {
            hi = mid - 1;
        }
    }
    col[p] = list_blk_off[lo] * 64 + (p - list_row_off[lo]);
}

// one thread per (requested id, chunk of 4 dims): binary search in the sorted ids, then the 16-byte piece of the row
__global__ void idmap_gather_kernel(const int64_t* __restrict__ want, int64_t n, const int64_t* __restrict__ ids_sorted,
                                    const int64_t* __restrict__ col_sorted, int64_t ntotal,
                                    const float4* __restrict__ rows, int d, int nchunk, float* __restrict__ out,
                                    int32_t* __restrict__ missing, uint8_t* __restrict__ found) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * nchunk) {
        return;
    }
    const int64_t i = t / nchunk;
    const int c = (int)(t % nchunk);
    const int64_t id = want[i];
    int64_t lo = 0, hi = ntotal;
    while (lo < hi) { // first entry >= id
        const int64_t mid = (lo + hi) >> 1;
        if (ids_sorted[mid] < id) {
            lo = mid + 1;
        } else {
            hi = mid;
        }
    }
    if (lo >= ntotal || ids_sorted[lo] != id) {
        if (c == 0) {
            atomicAdd(missing, 1);
            if (found) {
                found[i] = 0;
            }
        }
        return;
    }
    if (c == 0 && found) {
        found[i] = 1;
    }
    const int64_t col = col_sorted[lo];
    const float4 v = rows[(col >> 6) * (int64_t)nchunk * 64 + (int64_t)c * 64 + (col & 63)];
    const float e[4] = {v.x, v.y, v.z, v.w};
    for (int j = 0; j < 4 && c * 4 + j < d; j++) {
        out[i * d + c * 4 + j] = e[j];
    }
}

size_t idmap_sort_tmp_bytes(int64_t n) {
    size_t b = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, (const int64_t*)nullptr, (int64_t*)nullptr, (const int64_t*)nullptr,
                                             (int64_t*)nullptr, (int)n);
    return b + 256;
}

// ids [ntotal] (storage order) -> ids_sorted / col_sorted; col_tmp [ntotal] and tmp are scratch
hipError_t launch_idmap_build(const int64_t* ids, const int64_t* list_row_off, const int64_t* list_blk_off, int64_t nlist,
                              int64_t ntotal, int64_t* col_tmp, int64_t* ids_sorted, int64_t* col_sorted, void* tmp,
                              size_t tmp_bytes, hipStream_t s) {
    if (ntotal <= 0) {
        return hipSuccess;
    }
    if (ntotal >= ((int64_t)1 << 31)) {
        return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(idmap_cols_kernel, dim3((unsigned)((ntotal + 255) / 256)), dim3(256), 0, s, list_row_off,
                       list_blk_off, nlist, ntotal, col_tmp);
    size_t b = tmp_bytes;
    return hipcub::DeviceRadixSort::SortPairs(tmp, b, ids, ids_sorted, col_tmp, col_sorted, (int)ntotal, 0, 64, s);
}

hipError_t launch_idmap_gather(const int64_t* want, int64_t n, const int64_t* ids_sorted, const int64_t* col_sorted,
                               int64_t ntotal, const float4* rows, int d, float* out, int32_t* missing, hipStream_t s,
                               uint8_t* found) {
    if (n <= 0) {
        return hipSuccess;
    }
    const int nchunk = (d + 3) / 4;
    hipLaunchKernelGGL(idmap_gather_kernel, dim3((unsigned)((n * nchunk + 255) / 256)), dim3(256), 0, s, want, n,
                       ids_sorted, col_sorted, ntotal, rows, d, nchunk, out, missing, found);
    return hipGetLastError();
}

__global__ void residual_kernel(const float* __restrict__ x, const float* __restrict__ cen,
                                const int64_t* __restrict__ assign, int64_t n, int d, float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * d) {
        return;
    }
    const int64_t i = t / d;
    const int j = (int)(t % d);
    const int64_t a = assign[i];
    out[t] = a >= 0 ? fsub_x(x[t], cen[a * d + j]) : x[t];
}

hipError_t launch_residual(const float* x, const float* cen, const int64_t* assign, int64_t n, int d, float* out,
                           hipStream_t s) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(residual_kernel, dim3((unsigned)((n * d + 255) / 256)), dim3(256), 0, s, x, cen, assign, n, d, out);
    return hipGetLastError();
}

// ---- nearest of a small codebook (<= 1024 entries x dsub dims, in LDS): one thread per vector ----------------------
// x rows have leading dimension ld, the sub-vector starts at column off.  grid.y walks `nbook` codebooks laid out
// [nbook][ksub][dsub] with sub-vector offsets off + y * dsub and output column y (PQ encode); nbook = 1 otherwise.
template <class OutT>
__global__ __launch_bounds__(256) void nearest_small_kernel(const float* __restrict__ x, int64_t n, int64_t ld, int off,
                                                            int dsub, const float* __restrict__ cb, int ksub,
                                                            OutT* __restrict__ out, int64_t out_ld) {
    extern __shared__ float s_cb[]; // [ksub][dsub]
    const int book = blockIdx.y;
    const float* cbm = cb + (int64_t)book * ksub * dsub;
    for (int e = threadIdx.x; e < ksub * dsub; e += blockDim.x) {
        s_cb[e] = cbm[e];
    }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const float* xi = x + i * ld + off + (int64_t)book * dsub;
    int best = 0;
    float best_d = HUGE_VALF;
    if (dsub <= 8) {
        float xr[8];
#pragma unroll
        for (int t = 0; t < 8; t++) {
            xr[t] = t < dsub ? xi[t] : 0.f;
        }
        for (int c = 0; c < ksub; c++) {
            const float* y = s_cb + c * dsub;
            float res = 0.f;
#pragma unroll
            for (int t = 0; t < 8; t++) {
                if (t < dsub) {
                    res = l2_step(res, xr[t], y[t]);
                }
            }
            if (res < best_d) {
                best_d = res;
                best = c;
            }
        }
    } else {
        for (int c = 0; c < ksub; c++) {
            const float* y = s_cb + c * dsub;
            float res = 0.f;
            for (int t = 0; t < dsub; t++) {
                res = l2_step(res, xi[t], y[t]);
            }
            if (res < best_d) {
                best_d = res;
                best = c;
            }
        }
    }
    out[i * out_ld + book] = (OutT)best;
}

hipError_t launch_nearest_small(const float* x, int64_t n, int64_t ld, int off, int dsub, const float* cb, int ksub,
                                int32_t* out_idx, hipStream_t s) {
    if (n <= 0) {
        return hipSuccess;
    }
    const size_t sm = (size_t)ksub * dsub * sizeof(float);
    auto kern = nearest_small_kernel<int32_t>;
    if (sm > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)sm);
        if (e != hipSuccess) {
            return e;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)((n + 255) / 256), 1), dim3(256), sm, s, x, n, ld, off, dsub, cb, ksub, out_idx,
                       (int64_t)1);
    return hipGetLastError();
}

hipError_t launch_pq_encode(const float* resid, int64_t n, int d, int M, const float* cb, uint8_t* codes, hipStream_t s) {
    if (n <= 0) {
        return hipSuccess;
    }
    const int dsub = d / M;
    const size_t sm = (size_t)256 * dsub * sizeof(float);
    auto kern = nearest_small_kernel<uint8_t>;
    if (sm > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)sm);
        if (e != hipSuccess) {
            return e;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)((n + 255) / 256), (unsigned)M), dim3(256), sm, s, resid, n, (int64_t)d, 0, dsub,
                       cb, 256, codes, (int64_t)M);
    return hipGetLastError();
}

__global__ void sq8_encode_kernel(const float* __restrict__ r, int64_t n, int d, const float* __restrict__ trained,
                                  uint8_t* __restrict__ codes) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * d) {
        return;
    }
    const int j = (int)(t % d);
    const float vmin = trained[j], vdiff = trained[d + j];
    float xi = 0.f;
    if (vdiff != 0.f) {
        xi = __fdiv_rn(fsub_x(r[t], vmin), vdiff);
        if (xi < 0.f) {
            xi = 0.f;
        }
        if (xi > 1.0f) {
            xi = 1.0f;
        }
    }
    codes[t] = (uint8_t)(int)fmul_x(255.f, xi);
}

hipError_t launch_sq8_encode(const float* resid, int64_t n, int d, const float* trained, uint8_t* codes, hipStream_t s) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(sq8_encode_kernel, dim3((unsigned)((n * d + 255) / 256)), dim3(256), 0, s, resid, n, d, trained, codes);
    return hipGetLastError();
}

// per column min / max over n rows: one workgroup per 64 columns, rows strided over waves (order independent)
__global__ __launch_bounds__(256) void col_minmax_kernel(const float* __restrict__ x, int64_t n, int d, float* vmin,
                                                         float* vmax) {
    __shared__ float s_lo[4][64], s_hi[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int w = threadIdx.x >> 6;
    float lo = HUGE_VALF, hi = -HUGE_VALF;
    if (col < d) {
        for (int64_t i = w; i < n; i += 4) {
            const float v = x[i * d + col];
            lo = fminf(lo, v);
            hi = fmaxf(hi, v);
        }
    }
    s_lo[w][threadIdx.x & 63] = lo;
    s_hi[w][threadIdx.x & 63] = hi;
    __syncthreads();
    if (w == 0 && col < d) {
        for (int u = 1; u < 4; u++) {
            lo = fminf(lo, s_lo[u][threadIdx.x]);
            hi = fmaxf(hi, s_hi[u][threadIdx.x]);
        }
        vmin[col] = lo;
        vmax[col] = hi;
    }
}

hipError_t launch_col_minmax(const float* x, int64_t n, int d, float* vmin, float* vmax, hipStream_t s) {
    hipLaunchKernelGGL(col_minmax_kernel, dim3((unsigned)((d + 63) / 64)), dim3(256), 0, s, x, n, d, vmin, vmax);
    return hipGetLastError();
}

// ---- k-means update ------------------------------------------------------------------------------------------------
__global__ void keys_to_i32_kernel(const int64_t* __restrict__ keys, int64_t n, int32_t* __restrict__ out,
                                   int32_t* __restrict__ iota, int32_t* counts) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) {
        return;
    }
    const int32_t k = (int32_t)keys[t];
    out[t] = k;
    iota[t] = (int32_t)t;
    if (counts != nullptr && k >= 0) {
        atomicAdd(&counts[k], 1);
    }
}

__global__ void iota_count_kernel(const int32_t* __restrict__ keys, int64_t n, int32_t* __restrict__ iota, int32_t* counts) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) {
        return;
    }
    iota[t] = (int32_t)t;
    const int32_t k = keys[t];
    if (k >= 0) {
        atomicAdd(&counts[k], 1);
    }
}

size_t group_rows_tmp_bytes(int64_t n, int64_t k) {
    size_t sort_b = 0, scan_b = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_b, (const int32_t*)nullptr, (int32_t*)nullptr,
                                             (const int32_t*)nullptr, (int32_t*)nullptr, (int)n);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_b, (const int32_t*)nullptr, (int64_t*)nullptr, (int)(k + 1));
    // layout: [keys32 n][iota n][keys_sorted n][counts k+1] + cub scratch
    return (size_t)n * 12 + (size_t)(k + 1) * 4 + 256 + std::max(sort_b, scan_b) + 256;
}

// keys (int64 [n], or int32 if keys32 != nullptr) in [0, k) -> rows grouped by key, ascending row number inside a
// group (stable sort), seg_off[k + 1]
hipError_t group_rows_by_key(const int64_t* keys64, const int32_t* keys32_in, int64_t n, int64_t k, int32_t* sorted_rows,
                             int64_t* seg_off, void* tmp, size_t tmp_bytes, hipStream_t s) {
    if (n >= (int64_t)1 << 31 || k >= (int64_t)1 << 31) {
        return hipErrorInvalidValue;
    }
    char* p = static_cast<char*>(tmp);
    int32_t* keys32 = reinterpret_cast<int32_t*>(p);
    int32_t* iota = keys32 + n;
    int32_t* keys_sorted = iota + n;
    int32_t* counts = keys_sorted + n;
    char* cub_tmp = reinterpret_cast<char*>(((uintptr_t)(counts + k + 1) + 255) & ~(uintptr_t)255);
    size_t cub_bytes = tmp_bytes - (size_t)(cub_tmp - p);
    hipError_t e = hipMemsetAsync(counts, 0, (size_t)(k + 1) * 4, s);
    if (e != hipSuccess) {
        return e;
    }
    const unsigned g = (unsigned)((n + 255) / 256);
    if (n > 0) {
        if (keys32_in != nullptr) {
            hipLaunchKernelGGL(iota_count_kernel, dim3(g), dim3(256), 0, s, keys32_in, n, iota, counts);
        } else {
            hipLaunchKernelGGL(keys_to_i32_kernel, dim3(g), dim3(256), 0, s, keys64, n, keys32, iota, counts);
        }
    }
    int bits = 1;
    while (((int64_t)1 << bits) < k) {
        bits++;
    }
    size_t b = cub_bytes;
    e = hipcub::DeviceRadixSort::SortPairs(cub_tmp, b, keys32_in != nullptr ? keys32_in : keys32, keys_sorted, iota,
                                           sorted_rows, (int)n, 0, bits, s);
    if (e != hipSuccess) {
        return e;
    }
    b = cub_bytes;
    return hipcub::DeviceScan::ExclusiveSum(cub_tmp, b, counts, seg_off, (int)(k + 1), s);
}

// one thread per (centroid, dim): members summed in row order, then scaled by 1 / count (compute_centroids)
__global__ void centroid_update_kernel(const float* __restrict__ x, int64_t ld, int off, int dsub,
                                       const int32_t* __restrict__ sorted_rows, const int64_t* __restrict__ seg_off,
                                       int64_t k, float* __restrict__ centroids, float* __restrict__ hassign) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= k * dsub) {
        return;
    }
    const int64_t c = t / dsub;
    const int j = (int)(t % dsub);
    const int64_t b = seg_off[c], e = seg_off[c + 1];
    float acc = 0.f;
    for (int64_t u = b; u < e; u++) {
        acc = fadd_x(acc, x[(int64_t)sorted_rows[u] * ld + off + j]);
    }
    const float cnt = (float)(e - b);
    if (e > b) {
        acc = fmul_x(acc, __fdiv_rn(1.0f, cnt));
    }
    centroids[t] = acc;
    if (j == 0) {
        hassign[c] = cnt;
    }
}

hipError_t launch_centroid_update(const float* x, int64_t ld, int off, int dsub, const int32_t* sorted_rows,
                                  const int64_t* seg_off, int64_t k, float* centroids, float* hassign, hipStream_t s) {
    hipLaunchKernelGGL(centroid_update_kernel, dim3((unsigned)((k * dsub + 255) / 256)), dim3(256), 0, s, x, ld, off, dsub,
                       sorted_rows, seg_off, k, centroids, hassign);
    return hipGetLastError();
}

// ---- IndexFlat::assign under the inner product: exact ties ---------------------------------------------------------
// The reference's k = 1 search keeps the FIRST centroid that reaches the maximum (strict improvement, IndexFlat.cpp ->
// exhaustive_inner_product_seq / the heap's strict admission); the library's coarse search returns ties in canonical
// order, which for the inner product is the HIGHEST id first.  Exact ties are not rare in training: split_clusters
// (impl/ClusteringHelpers.cpp:177-240) makes two centroids that differ by a factor 1 +- 1/1024 per dimension, and after the
// spherical renormalisation their products with a row can round to the same float.  One wave per row: rows whose two
// best candidates tie bit for bit are rescanned over all centroids (sequential exact products, lowest index of the
// maximum); every other row keeps its best candidate.
__global__ __launch_bounds__(256) void assign_first_max_ip_kernel(const float* __restrict__ x, int64_t n, int d,
                                                                  const float* __restrict__ cen, int64_t nlist,
                                                                  const int64_t* __restrict__ top2_keys,
                                                                  const float* __restrict__ top2_dis,
                                                                  int64_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    if (i >= n) {
        return;
    }
    const int64_t k0 = top2_keys[2 * i];
    if (!(top2_dis[2 * i] == top2_dis[2 * i + 1]) || top2_keys[2 * i + 1] < 0) {
        if (lane == 0) {
            out[i] = k0;
        }
        return;
    }
    const float* xi = x + i * d;
    float best = -INFINITY;
    int64_t bi = INT64_MAX;
    for (int64_t c = lane; c < nlist; c += 64) {
        const float* cc = cen + c * d;
        float t = 0.f;
        for (int j = 0; j < d; j++) {
            t = ip_step(t, xi[j], cc[j]);
        }
        if (t > best) { // (c ascends within a lane: the first maximum stays)
            best = t;
            bi = c;
        }
    }
#pragma unroll
    for (int dlt = 32; dlt > 0; dlt >>= 1) {
        const float ob = __shfl_xor(best, dlt, 64);
        const int64_t oi = __shfl_xor(bi, dlt, 64);
        if (ob > best || (ob == best && oi < bi)) {
            best = ob;
            bi = oi;
        }
    }
    if (lane == 0) {
        out[i] = bi == INT64_MAX ? k0 : bi;
    }
}

hipError_t launch_assign_first_max_ip(const float* x, int64_t n, int d, const float* cen, int64_t nlist,
                                      const int64_t* top2_keys, const float* top2_dis, int64_t* out, hipStream_t s) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(assign_first_max_ip_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, x, n, d, cen, nlist,
                       top2_keys, top2_dis, out);
    return hipGetLastError();
}

// spherical k-means (Clustering::post_process_centroids, Clustering.cpp:35-38 -> fvec_renorm_L2, utils/distances.cpp:
// 238-275): every row of non-zero squared norm (sequential float sum) is scaled by (float)(1.0 / sqrtf(norm2)).
// One thread per row for the norm (k <= 65536 rows), then one thread per element.
__global__ void row_inv_norm_kernel(const float* __restrict__ x, int64_t k, int d, float* __restrict__ inv) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= k) {
        return;
    }
    const float* xi = x + c * d;
    float nr = 0.f;
    for (int j = 0; j < d; j++) {
        nr = fadd_x(nr, fmul_x(xi[j], xi[j]));
    }
    // sqrtf, not __fsqrt_rn: the intrinsic compiles to the bare v_sqrt_f32 (1 ulp), sqrtf to the correctly rounded
    // sequence -- what the host's sqrtf returns.  (float)(1.0 / (double)s) == 1.0f / s: a double quotient rounded to
    // float is the correctly rounded float quotient
    inv[c] = nr > 0.f ? (float)(1.0 / (double)sqrtf(nr)) : 1.0f;
}

__global__ void row_scale_kernel(float* __restrict__ x, int64_t k, int d, const float* __restrict__ inv) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= k * d) {
        return;
    }
    const float f = inv[t / d];
    if (f != 1.0f) {  // (a zero row keeps its bits; x * 1.0f is the identity anyway)
        x[t] = fmul_x(x[t], f);
    }
}

hipError_t launch_renorm_rows(float* x, int64_t k, int d, float* inv_tmp, hipStream_t s) {
    hipLaunchKernelGGL(row_inv_norm_kernel, dim3((unsigned)((k + 63) / 64)), dim3(64), 0, s, x, k, d, inv_tmp);
    hipLaunchKernelGGL(row_scale_kernel, dim3((unsigned)((k * d + 255) / 256)), dim3(256), 0, s, x, k, d, inv_tmp);
    return hipGetLastError();
}

// ---- list append: old list-sorted (codes, ids) + a batch grouped by list -> new list-sorted arrays ------------------
// one thread per output entry; entry e of list l comes from the old list when e < old_len[l]
__global__ void merge_lists_kernel(const uint8_t* __restrict__ old_codes, const int64_t* __restrict__ old_ids,
                                   const int64_t* __restrict__ old_off, const uint8_t* __restrict__ new_codes,
                                   const int64_t* __restrict__ new_ids, const int32_t* __restrict__ new_rows,
                                   const int64_t* __restrict__ new_seg, const int64_t* __restrict__ out_off, int64_t nlist,
                                   int64_t code_size, uint8_t* __restrict__ out_codes, int64_t* __restrict__ out_ids) {
    const int64_t l = blockIdx.x;
    const int64_t o0 = out_off[l], o1 = out_off[l + 1];
    const int64_t nold = old_off != nullptr ? old_off[l + 1] - old_off[l] : 0;
    for (int64_t e = threadIdx.x; e < o1 - o0; e += blockDim.x) {
        const uint8_t* src;
        int64_t id;
        if (e < nold) {
            src = old_codes + (old_off[l] + e) * code_size;
            id = old_ids[old_off[l] + e];
        } else {
            const int64_t r = new_rows[new_seg[l] + (e - nold)];
            src = new_codes + r * code_size;
            id = new_ids[r];
        }
        uint8_t* dst = out_codes + (o0 + e) * code_size;
        for (int64_t b = 0; b < code_size; b++) {
            dst[b] = src[b];
        }
        out_ids[o0 + e] = id;
    }
}

hipError_t launch_merge_lists(const uint8_t* old_codes, const int64_t* old_ids, const int64_t* old_off,
                              const uint8_t* new_codes, const int64_t* new_ids, const int32_t* new_rows,
                              const int64_t* new_seg, const int64_t* out_off, int64_t nlist, int64_t code_size,
                              uint8_t* out_codes, int64_t* out_ids, hipStream_t s) {
    if (nlist <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(merge_lists_kernel, dim3((unsigned)nlist), dim3(256), 0, s, old_codes, old_ids, old_off, new_codes,
                       new_ids, new_rows, new_seg, out_off, nlist, code_size, out_codes, out_ids);
    return hipGetLastError();
}

__global__ void iota_i64_kernel(int64_t* out, int64_t n, int64_t base) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        out[t] = base + t;
    }
}

hipError_t launch_iota_i64(int64_t* out, int64_t n, int64_t base, hipStream_t s) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(iota_i64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, n, base);
    return hipGetLastError();
}

// ---- interleaved list blocks (flat: float4, SQ8: uint4 -- both are "16 bytes of row r, chunk c" at
// blk[(blk_off + b) * nchunk + c][r]) -> canonical list-sorted AoS rows [ntotal][code_size].  The inverse of
// interleave_lists_kernel / sq_interleave_kernel: lets the index drop its AoS copy of large flat / SQ8 lists (a further
// Add, Serialize or GetVectorByIds rebuilds it on demand).
__global__ void deinterleave_lists_kernel(const uint4* __restrict__ rows, const int64_t* __restrict__ list_row_off,
                                          const int64_t* __restrict__ list_len,
                                          const int64_t* __restrict__ list_blk_off, int64_t nlist, int64_t code_size,
                                          int nchunk, uint8_t* __restrict__ dst) {
    const int64_t l = blockIdx.y + (int64_t)blockIdx.z * gridDim.y;
    if (l >= nlist) {
        return;
    }
    const int64_t len = list_len[l];
    const int64_t nblk = (len + 63) / 64;
    const int64_t row_off = list_row_off[l];
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nblk * nchunk * 64;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / ((int64_t)nchunk * 64);
        const int rem = (int)(t % ((int64_t)nchunk * 64));
        const int c = rem / 64, r = rem % 64;
        const int64_t row = b * 64 + r;
        if (row >= len) {
            continue;
        }
        const uint4 w = rows[(list_blk_off[l] + b) * (int64_t)nchunk * 64 + rem];
        uint8_t* o = dst + (row_off + row) * code_size + (int64_t)c * 16;
        const int64_t nbytes = min((int64_t)16, code_size - (int64_t)c * 16);
        if (nbytes == 16 && (code_size & 15) == 0) {
            *reinterpret_cast<uint4*>(o) = w;
        } else {
            const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
            for (int e = 0; e < (int)nbytes; e++) {
                o[e] = (uint8_t)(ww[e >> 2] >> (8 * (e & 3)));
            }
        }
    }
}

hipError_t launch_deinterleave_lists(const uint4* rows, const int64_t* list_row_off, const int64_t* list_len,
                                     const int64_t* list_blk_off, int64_t nlist, int64_t code_size, uint8_t* dst,
                                     hipStream_t s) {
    if (nlist <= 0) {
        return hipSuccess;
    }
    const int nchunk = (int)((code_size + 15) / 16);
    const unsigned gy = (unsigned)std::min<int64_t>(nlist, 32768);
    const unsigned gz = (unsigned)((nlist + gy - 1) / gy);
    hipLaunchKernelGGL(deinterleave_lists_kernel, dim3(8, gy, gz), dim3(256), 0, s, rows, list_row_off, list_len,
                       list_blk_off, nlist, code_size, nchunk, dst);
    return hipGetLastError();
}

} // namespace knhip
