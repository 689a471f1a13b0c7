// knowhere_amd/csrc/coarse_gemm.hip -- fp32 MFMA prefilter for the coarse quantizer (see below).
#include "common.cuh"
#include "kernels.h"

namespace knhip {
// filled in by the next milestone (MFMA GEMM + exact re-rank + certificate)
} // namespace knhip
