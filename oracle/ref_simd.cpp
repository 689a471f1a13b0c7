// oracle/ref_simd.cpp -- TEST INFRASTRUCTURE ONLY.
//
// C-ABI over the reference's scalar distance definitions, src/simd/distances_ref.cc (the known-answer every SIMD level
// of the hook table is compared against in tests/ut/test_simd.cc:259-568), compiled from where it lies into
// oracle/_ref/libknowhere_ref.so by oracle/Makefile.  Nothing here computes a distance: every entry forwards to the
// reference's own *_ref function.  fp16 / bf16 operands cross the ABI as raw uint16 bit patterns
// (include/knowhere/operands.h:53-160: both types are a single uint16_t).
#include <cstdint>
#include <cstdlib>

#include "knowhere/operands.h"
#include "simd/distances_ref.h"

namespace R = faiss::cppcontrib::knowhere;
using knowhere::bf16;
using knowhere::fp16;

static_assert(sizeof(fp16) == 2 && sizeof(bf16) == 2, "operand types are bare 16-bit patterns");

extern "C" {

// not on the search path (calculate_hash_ref); see stubs/xxhash.h
uint64_t XXH3_64bits(const void*, size_t) { std::abort(); }

float ref_simd_fvec_inner_product(const float* x, const float* y, int64_t d) { return R::fvec_inner_product_ref(x, y, d); }
float ref_simd_fvec_L2sqr(const float* x, const float* y, int64_t d) { return R::fvec_L2sqr_ref(x, y, d); }
float ref_simd_fvec_L1(const float* x, const float* y, int64_t d) { return R::fvec_L1_ref(x, y, d); }
float ref_simd_fvec_Linf(const float* x, const float* y, int64_t d) { return R::fvec_Linf_ref(x, y, d); }
float ref_simd_fvec_norm_L2sqr(const float* x, int64_t d) { return R::fvec_norm_L2sqr_ref(x, d); }
void ref_simd_fvec_L2sqr_ny(float* dis, const float* x, const float* y, int64_t d, int64_t ny) {
    R::fvec_L2sqr_ny_ref(dis, x, y, d, ny);
}
void ref_simd_fvec_inner_products_ny(float* ip, const float* x, const float* y, int64_t d, int64_t ny) {
    R::fvec_inner_products_ny_ref(ip, x, y, d, ny);
}
void ref_simd_fvec_L2sqr_ny_transposed(float* dis, const float* x, const float* y, const float* y_sqlen, int64_t d,
                                       int64_t d_offset, int64_t ny) {
    R::fvec_L2sqr_ny_transposed_ref(dis, x, y, y_sqlen, d, d_offset, ny);
}
int64_t ref_simd_fvec_L2sqr_ny_nearest(float* tmp, const float* x, const float* y, int64_t d, int64_t ny) {
    return (int64_t)R::fvec_L2sqr_ny_nearest_ref(tmp, x, y, d, ny);
}
int64_t ref_simd_fvec_L2sqr_ny_nearest_y_transposed(float* tmp, const float* x, const float* y, const float* y_sqlen,
                                                    int64_t d, int64_t d_offset, int64_t ny) {
    return (int64_t)R::fvec_L2sqr_ny_nearest_y_transposed_ref(tmp, x, y, y_sqlen, d, d_offset, ny);
}
void ref_simd_fvec_madd(int64_t n, const float* a, float bf, const float* b, float* c) { R::fvec_madd_ref(n, a, bf, b, c); }
int ref_simd_fvec_madd_and_argmin(int64_t n, const float* a, float bf, const float* b, float* c) {
    return R::fvec_madd_and_argmin_ref(n, a, bf, b, c);
}
void ref_simd_fvec_batch_4(int is_l2, const float* x, const float* y0, const float* y1, const float* y2, const float* y3,
                           int64_t d, float* out4) {
    if (is_l2) R::fvec_L2sqr_batch_4_ref(x, y0, y1, y2, y3, d, out4[0], out4[1], out4[2], out4[3]);
    else R::fvec_inner_product_batch_4_ref(x, y0, y1, y2, y3, d, out4[0], out4[1], out4[2], out4[3]);
}
int32_t ref_simd_ivec_inner_product(const int8_t* x, const int8_t* y, int64_t d) { return R::ivec_inner_product_ref(x, y, d); }
int32_t ref_simd_ivec_L2sqr(const int8_t* x, const int8_t* y, int64_t d) { return R::ivec_L2sqr_ref(x, y, d); }

// typed operands: op 0 = L2sqr, 1 = inner product, 2 = norm_L2sqr (y ignored); type 0 = fp16, 1 = bf16, 2 = int8
float ref_simd_typed(int type, int op, const void* x, const void* y, int64_t d) {
    switch (type * 3 + op) {
        case 0: return R::fp16_vec_L2sqr_ref((const fp16*)x, (const fp16*)y, d);
        case 1: return R::fp16_vec_inner_product_ref((const fp16*)x, (const fp16*)y, d);
        case 2: return R::fp16_vec_norm_L2sqr_ref((const fp16*)x, d);
        case 3: return R::bf16_vec_L2sqr_ref((const bf16*)x, (const bf16*)y, d);
        case 4: return R::bf16_vec_inner_product_ref((const bf16*)x, (const bf16*)y, d);
        case 5: return R::bf16_vec_norm_L2sqr_ref((const bf16*)x, d);
        case 6: return R::int8_vec_L2sqr_ref((const int8_t*)x, (const int8_t*)y, d);
        case 7: return R::int8_vec_inner_product_ref((const int8_t*)x, (const int8_t*)y, d);
        case 8: return R::int8_vec_norm_L2sqr_ref((const int8_t*)x, d);
    }
    std::abort();
}
void ref_simd_typed_batch_4(int type, int is_l2, const void* x, const void* y0, const void* y1, const void* y2,
                            const void* y3, int64_t d, float* o) {
    if (type == 0) {
        auto X = (const fp16*)x; auto A = (const fp16*)y0; auto B = (const fp16*)y1; auto C = (const fp16*)y2; auto D = (const fp16*)y3;
        if (is_l2) R::fp16_vec_L2sqr_batch_4_ref(X, A, B, C, D, d, o[0], o[1], o[2], o[3]);
        else R::fp16_vec_inner_product_batch_4_ref(X, A, B, C, D, d, o[0], o[1], o[2], o[3]);
    } else if (type == 1) {
        auto X = (const bf16*)x; auto A = (const bf16*)y0; auto B = (const bf16*)y1; auto C = (const bf16*)y2; auto D = (const bf16*)y3;
        if (is_l2) R::bf16_vec_L2sqr_batch_4_ref(X, A, B, C, D, d, o[0], o[1], o[2], o[3]);
        else R::bf16_vec_inner_product_batch_4_ref(X, A, B, C, D, d, o[0], o[1], o[2], o[3]);
    } else {
        auto X = (const int8_t*)x; auto A = (const int8_t*)y0; auto B = (const int8_t*)y1; auto C = (const int8_t*)y2; auto D = (const int8_t*)y3;
        if (is_l2) R::int8_vec_L2sqr_batch_4_ref(X, A, B, C, D, d, o[0], o[1], o[2], o[3]);
        else R::int8_vec_inner_product_batch_4_ref(X, A, B, C, D, d, o[0], o[1], o[2], o[3]);
    }
}

}  // extern "C"
