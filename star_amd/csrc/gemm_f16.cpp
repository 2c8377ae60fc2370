// gemm_f16.cpp -- the fp16 instantiations of the GEMM / implicit-conv kernels (gemm_impl.h)
#include "gemm_impl.h"

namespace star {
int launch_gemm_f16(Ctx* ctx, const GemmArgs& a) { return launch_gemm<f16>(ctx, a); }
}  // namespace star
