// knowhere_amd/csrc/knhip_api_prims.hip -- C ABI of the src/simd hook table on the device (include/knhip.h; kernels: prims.hip).
#include "knhip_internal.h"

extern "C" {

// ---- primitives ------------------------------------------------------------------------------------
static int prim_args_ok(const void* out, const void* a, const void* b, int64_t d, int64_t n) {
    if (d < 0 || n < 0 || (n > 0 && (!out || !a || !b))) {
        return fail(KNHIP_ERR_INVALID_ARGS, "primitive: null pointer or negative size");
    }
    if (n >= ((int64_t)1 << 32)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "primitive: more than 2^32 - 1 rows");
    }
    return KNHIP_OK;
}
int knhip_fvec_L1_ny(float* d_dis, const float* d_x, const float* d_y, int64_t d, int64_t ny, void* stream) {
    if (int rc = prim_args_ok(d_dis, d_x, d_y, d, ny)) return rc;
    HIP_TRY(launch_fvec_rows(3, d_dis, d_x, d_y, d, ny, static_cast<hipStream_t>(stream)));
    return KNHIP_OK;
}
int knhip_fvec_Linf_ny(float* d_dis, const float* d_x, const float* d_y, int64_t d, int64_t ny, void* stream) {
    if (int rc = prim_args_ok(d_dis, d_x, d_y, d, ny)) return rc;
    HIP_TRY(launch_fvec_rows(4, d_dis, d_x, d_y, d, ny, static_cast<hipStream_t>(stream)));
    return KNHIP_OK;
}
int knhip_fvec_norms_L2sqr_ref(float* d_out, const float* d_x, int64_t d, int64_t n, void* stream) {
    if (int rc = prim_args_ok(d_out, d_x, d_x, d, n)) return rc;
    HIP_TRY(launch_fvec_rows(5, d_out, nullptr, d_x, d, n, static_cast<hipStream_t>(stream)));
    return KNHIP_OK;
}
int knhip_fvec_L2sqr_ny_transposed(float* d_dis, const float* d_x, const float* d_y, const float* d_y_sqlen, int64_t d,
                                   int64_t d_offset, int64_t ny, void* stream) {
    if (int rc = prim_args_ok(d_dis, d_x, d_y, d, ny)) return rc;
    if (ny > 0 && (!d_y_sqlen || d_offset < ny)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "L2sqr_ny_transposed: need y_sqlen and d_offset >= ny");
    }
    HIP_TRY(launch_l2_transposed(d_dis, d_x, d_y, d_y_sqlen, d, d_offset, ny, static_cast<hipStream_t>(stream)));
    return KNHIP_OK;
}
int knhip_fvec_L2sqr_ny_nearest(float* d_dis_tmp, const float* d_x, const float* d_y, int64_t d, int64_t ny,
                                int64_t* d_nearest, void* stream) {
    if (int rc = prim_args_ok(d_dis_tmp, d_x, d_y, d, ny)) return rc;
    if (!d_nearest) return fail(KNHIP_ERR_INVALID_ARGS, "L2sqr_ny_nearest: null output");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(launch_fvec_ny(d_dis_tmp, d_x, d_y, d, ny, true, s));
    HIP_TRY(launch_argmin(d_dis_tmp, ny, HUGE_VALF, 0, d_nearest, s));
    return KNHIP_OK;
}
int knhip_fvec_L2sqr_ny_nearest_y_transposed(float* d_dis_tmp, const float* d_x, const float* d_y,
                                             const float* d_y_sqlen, int64_t d, int64_t d_offset, int64_t ny,
                                             int64_t* d_nearest, void* stream) {
    if (!d_nearest) return fail(KNHIP_ERR_INVALID_ARGS, "L2sqr_ny_nearest_y_transposed: null output");
    if (int rc = knhip_fvec_L2sqr_ny_transposed(d_dis_tmp, d_x, d_y, d_y_sqlen, d, d_offset, ny, stream)) return rc;
    HIP_TRY(launch_argmin(d_dis_tmp, ny, HUGE_VALF, 0, d_nearest, static_cast<hipStream_t>(stream)));
    return KNHIP_OK;
}
int knhip_fvec_madd_and_argmin(int64_t n, const float* d_a, float bf, const float* d_b, float* d_c, int64_t* d_imin,
                               void* stream) {
    if (int rc = prim_args_ok(d_c, d_a, d_b, 0, n)) return rc;
    if (!d_imin) return fail(KNHIP_ERR_INVALID_ARGS, "madd_and_argmin: null output");
    HIP_TRY(launch_fvec_madd_and_argmin(n, d_a, bf, d_b, d_c, d_imin, static_cast<hipStream_t>(stream)));
    return KNHIP_OK;
}
int knhip_fvec_batch_4(int32_t metric, const float* d_x, const float* d_y0, const float* d_y1, const float* d_y2,
                       const float* d_y3, int64_t d, float* d_out4, void* stream) {
    if (d < 0 || !d_out4 || (d > 0 && (!d_x || !d_y0 || !d_y1 || !d_y2 || !d_y3)) ||
        (metric != KNHIP_L2 && metric != KNHIP_IP)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "batch_4: bad arguments");
    }
    HIP_TRY(launch_batch4(-1, metric == KNHIP_L2, d_x, d_y0, d_y1, d_y2, d_y3, d, d_out4, static_cast<hipStream_t>(stream)));
    return KNHIP_OK;
}
int knhip_typed_vec_ny(int32_t dtype, int32_t op, float* d_out, const void* d_x, const void* d_y, int64_t d,
                       int64_t ny, void* stream) {
    if (dtype < KNHIP_DT_FP16 || dtype > KNHIP_DT_INT8 || op < 0 || op > 2) {
        return fail(KNHIP_ERR_INVALID_ARGS, "typed_vec_ny: dtype in {fp16, bf16, int8}, op in {L2sqr, ip, norm}");
    }
    if (int rc = prim_args_ok(d_out, op == 2 ? d_y : d_x, d_y, d, ny)) return rc;
    HIP_TRY(launch_typed_rows(dtype, op, d_out, d_x, d_y, d, ny, static_cast<hipStream_t>(stream)));
    return KNHIP_OK;
}
int knhip_typed_vec_batch_4(int32_t dtype, int32_t metric, const void* d_x, const void* d_y0, const void* d_y1,
                            const void* d_y2, const void* d_y3, int64_t d, float* d_out4, void* stream) {
    if (dtype < KNHIP_DT_FP16 || dtype > KNHIP_DT_INT8 || d < 0 || !d_out4 ||
        (d > 0 && (!d_x || !d_y0 || !d_y1 || !d_y2 || !d_y3)) || (metric != KNHIP_L2 && metric != KNHIP_IP)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "typed batch_4: bad arguments");
    }
    HIP_TRY(launch_batch4(dtype, metric == KNHIP_L2, d_x, d_y0, d_y1, d_y2, d_y3, d, d_out4,
                          static_cast<hipStream_t>(stream)));
    return KNHIP_OK;
}
int knhip_ivec_ny(int32_t metric, int32_t* d_out, const int8_t* d_x, const int8_t* d_y, int64_t d, int64_t ny,
                  void* stream) {
    if (metric != KNHIP_L2 && metric != KNHIP_IP) return fail(KNHIP_ERR_INVALID_ARGS, "ivec_ny: metric");
    if (int rc = prim_args_ok(d_out, d_x, d_y, d, ny)) return rc;
    HIP_TRY(launch_ivec_ny(d_out, d_x, d_y, d, ny, metric == KNHIP_L2, static_cast<hipStream_t>(stream)));
    return KNHIP_OK;
}
int knhip_fvec_L2sqr_ny(float* d_dis, const float* d_x, const float* d_y, int64_t d, int64_t ny, void* stream) {
    HIP_TRY(launch_fvec_ny(d_dis, d_x, d_y, d, ny, true, static_cast<hipStream_t>(stream)));
    return KNHIP_OK;
}
int knhip_fvec_inner_products_ny(float* d_ip, const float* d_x, const float* d_y, int64_t d, int64_t ny,
                                 void* stream) {
    HIP_TRY(launch_fvec_ny(d_ip, d_x, d_y, d, ny, false, static_cast<hipStream_t>(stream)));
    return KNHIP_OK;
}
int knhip_fvec_norms_L2sqr(float* d_out, const float* d_x, int64_t d, int64_t n, void* stream) {
    HIP_TRY(launch_fvec_norms(d_out, d_x, d, n, static_cast<hipStream_t>(stream)));
    return KNHIP_OK;
}
int knhip_fvec_madd(int64_t n, const float* d_a, float bf, const float* d_b, float* d_c, void* stream) {
    HIP_TRY(launch_fvec_madd(n, d_a, bf, d_b, d_c, static_cast<hipStream_t>(stream)));
    return KNHIP_OK;
}
int knhip_int8_vec_L2sqr_ny(float* d_dis, const int8_t* d_x, const int8_t* d_y, int64_t d, int64_t ny,
                            void* stream) {
    HIP_TRY(launch_int8_ny(d_dis, d_x, d_y, d, ny, true, static_cast<hipStream_t>(stream)));
    return KNHIP_OK;
}
int knhip_int8_vec_inner_products_ny(float* d_ip, const int8_t* d_x, const int8_t* d_y, int64_t d, int64_t ny,
                                     void* stream) {
    HIP_TRY(launch_int8_ny(d_ip, d_x, d_y, d, ny, false, static_cast<hipStream_t>(stream)));
    return KNHIP_OK;
}

// ---- profiling -------------------------------------------------------------------------------------
// =================================================================================================
// Train / Add on the device (include/knhip.h "GPU build")
// =================================================================================================
} // extern "C"
