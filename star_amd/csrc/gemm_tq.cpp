// gemm_tq.cpp -- launcher of gemm_tq_kernel (gemm_tq.h): q | k | v projection (folded LayerNorm) + temporal attention in one kernel,
// level-0 width.  Built like gemm_as.cpp with -mllvm -amdgpu-mfma-vgpr-form (one wave per SIMD, the A panel in AGPRs).
#include "ops.h"
#include "gemm_tq.h"

namespace star {

bool temporal_qkv_attn_covers(int C, int heads, int F) { return C == 320 && heads == 5 && F >= 1 && F <= 32; }

template <class T>
static int launch_tq(Ctx* ctx, const TqArgs& a) {
  TqParams p{a.A, a.W, a.O, a.bias, a.colsum, a.rowab, a.lda, a.ldo, a.HW, a.F, a.scale * 1.4426950408889634f};
  const size_t smem = 2 * (size_t)40960 + 4 * (size_t)8192 + 2 * (size_t)960 * sizeof(float);
  const dim3 grid((unsigned)((a.HW + 7) / 8)), block(256);
  STAR_LAUNCH((gemm_tq_kernel<T>), grid, block, smem, ctx->stream, p);
  return 0;
}

int op_temporal_qkv_attn(Ctx* ctx, const TqArgs& a) {
  if (!temporal_qkv_attn_covers(a.C, a.heads, a.F)) return ctx->fail("temporal_qkv_attn: C = 320, 5 heads, 1..32 frames only");
  if (!a.A || !a.W || !a.O || !a.bias || !a.colsum || !a.rowab) return ctx->fail("temporal_qkv_attn: null operand");
  if (a.lda % 8 || a.ldo % 8 || a.lda < 320 || a.ldo < 320) return ctx->fail("temporal_qkv_attn: row strides must be multiples of 8 and >= 320");
  if ((size_t)a.F * a.HW * a.ldo * 2 >= 0xFFFF0000ull) return ctx->fail("temporal_qkv_attn: output too large for a 32-bit buffer range");
  if (a.HW <= 0) return 0;
  const double M = (double)a.F * a.HW;
  ProfScope ps(ctx, PK_TATTN, 2.0 * M * 960.0 * 320.0 + 4.0 * M * a.F * 320.0, M * (320.0 + 320.0) * 2.0, (int)M, 960, 320, a.F);
  if (ctx->dtype == DT_F16) return launch_tq<f16>(ctx, a);
  if (ctx->dtype == DT_BF16) return launch_tq<bf16>(ctx, a);
  return ctx->fail("temporal_qkv_attn: unsupported dtype");
}

}  // namespace star
