// oracle/ref_hooks.cpp -- TEST INFRASTRUCTURE ONLY.
//
// The reference's src/simd/hook.cc fills its function-pointer table at start-up from the CPU's ISA (hook.cc:163-382);
// with no SIMD level selected every pointer is the scalar *_ref definition of src/simd/distances_ref.cc.  This file is
// that state, stated directly: each hook the Knowhere FAISS fork (thirdparty/faiss/faiss/cppcontrib/knowhere) reads is
// defined and bound to the reference's own scalar function.  hook.cc itself is not compiled (it needs every per-ISA
// translation unit); nothing here computes anything.
#include "knowhere/operands.h"
#include "simd/distances_ref.h"
#include "simd/hook.h"

namespace faiss::cppcontrib::knowhere {

#define BIND(name) decltype(name) name = name##_ref
BIND(fvec_inner_product);
BIND(fvec_L2sqr);
BIND(fvec_L1);
BIND(fvec_Linf);
BIND(fvec_norm_L2sqr);
BIND(fvec_L2sqr_ny);
BIND(fvec_inner_products_ny);
BIND(fvec_L2sqr_ny_transposed);
BIND(fvec_L2sqr_ny_nearest);
BIND(fvec_L2sqr_ny_nearest_y_transposed);
BIND(fvec_madd);
BIND(fvec_madd_and_argmin);
BIND(fvec_inner_product_batch_4);
BIND(fvec_L2sqr_batch_4);
BIND(ivec_inner_product);
BIND(ivec_L2sqr);
BIND(fp16_vec_inner_product);
BIND(fp16_vec_L2sqr);
BIND(fp16_vec_norm_L2sqr);
BIND(fp16_vec_inner_product_batch_4);
BIND(fp16_vec_L2sqr_batch_4);
BIND(bf16_vec_inner_product);
BIND(bf16_vec_L2sqr);
BIND(bf16_vec_norm_L2sqr);
BIND(bf16_vec_inner_product_batch_4);
BIND(bf16_vec_L2sqr_batch_4);
BIND(int8_vec_inner_product);
BIND(int8_vec_L2sqr);
BIND(int8_vec_norm_L2sqr);
BIND(int8_vec_inner_product_batch_4);
BIND(int8_vec_L2sqr_batch_4);
#undef BIND

}  // namespace faiss::cppcontrib::knowhere
