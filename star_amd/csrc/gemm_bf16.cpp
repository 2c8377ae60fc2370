// gemm_bf16.cpp -- the bf16 instantiations of the GEMM / implicit-conv kernels (gemm_impl.h)
#include "gemm_impl.h"

namespace star {
int launch_gemm_bf16(Ctx* ctx, const GemmArgs& a) { return launch_gemm<bf16>(ctx, a); }
}  // namespace star
