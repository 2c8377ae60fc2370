// ops.h -- host-side launchers of the HIP kernels (one per hot-path operator).
// Every function enqueues on ctx->stream and returns 0 on success.
#pragma once
#include "ctx.h"
#include "gemm.h"

namespace star {

struct GemmArgs {
  const void* A = nullptr; const void* W = nullptr; void* C = nullptr;
  const float* bias = nullptr; const void* res = nullptr;
  int M = 0, N = 0, K = 0;
  int lda = 0, ldc = 0, ldr = 0;
  int mode = A_PLAIN;
  int H = 0, Wd = 0, Cin = 0, Ho = 0, Wo = 0, stride = 1, pad_t = 1, pad_l = 1;
  int HW = 0, F = 0;
  int epi = 0;
  int force_tile = 0;  // 0 = auto; 1 = 256x256, 2 = 256x320, 3 = 128x128, 4 = 256x128 (tests)
};
int op_gemm(Ctx* ctx, const GemmArgs& a);

}  // namespace star
