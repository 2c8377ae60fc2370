// gemm_p.cpp -- launcher of gemm_persist_kernel (gemm_p.h): the persistent one-wave-per-SIMD GEMM for plain-A layers ("tile 18").
#include <cstdlib>
#include "ops.h"
#include "gemm_p.h"

namespace star {

// what the kernel can compute: plain A, 16-bit output, bias / residual / GEGLU / folded-LayerNorm epilogues
bool gemm_persist_covers(const GemmArgs& a) {
  if (a.mode != A_PLAIN || a.K % 64 || a.K < 64 || a.N % 8 || a.lda % 8 || a.ldc % 8) return false;
  if (a.epi & (EPI_OUT_F32 | EPI_GELU_TANH)) return false;
  if ((a.epi & EPI_GEGLU) && ((a.epi & EPI_RES) || a.N % 64)) return false;
  if ((a.epi & EPI_RES) && a.ldr % 8) return false;
  if (a.epi & EPI_ROWAFF) return !(a.epi & EPI_RES) && (a.epi & EPI_BIAS) && a.rowab && a.colsum && a.bias;
  return true;
}

template <class T>
static int launch_persist_t(Ctx* ctx, const GemmArgs& a) {
  GemmParams p{};
  p.A = a.A; p.W = a.W; p.C = a.C; p.bias = a.bias; p.res = a.res;
  p.M = a.M; p.N = a.N; p.K = a.K; p.lda = a.lda; p.ldc = a.ldc; p.ldr = a.ldr; p.epi = a.epi;
  p.rowab = a.rowab; p.colsum = a.colsum;
  p.m_off = a.m_off;
  p.tiles_m = ((a.m_end > 0 ? a.m_end : a.M) - a.m_off + 255) / 256;
  p.tiles_n = (a.N + 255) / 256;
  p.group_m = a.group_m != -1 ? a.group_m : (p.tiles_n >= 12 ? 8 : 1);   // (< -1: column strips, bench A/B)
  const int nblk = p.tiles_m * p.tiles_n;
  if (nblk <= 0) return 0;
  const int cus = a.assume_cus > 0 ? a.assume_cus : (ctx->num_cus > 0 ? ctx->num_cus : 256);
  int G = nblk < cus ? nblk : cus;          // one resident workgroup per CU; a multiple of 8 keeps a workgroup's tiles on its XCD's walk
  if (G >= 8) G &= ~7;
  constexpr size_t smem = 2 * (size_t)(256 + 256) * 128 + 4 * (size_t)4096 + 2 * (size_t)2048;
  const dim3 grid((unsigned)G), block(256);
#ifdef STAR_BENCH_VARIANTS   // round-6 A/B reference (bit-identical): 60 = the round-5 kernel's plain output stores
  if (a.force_tile == 60 || std::getenv("STAR_PERSIST_PLAIN")) {   // (the environment switch: in-situ A/B of a whole forward on the bench build)
    if ((a.epi & EPI_ROWAFF) && (a.epi & EPI_GEGLU)) STAR_LAUNCH((gemm_persist_kernel<T, 10, 0>), grid, block, smem, ctx->stream, p);
    else if (a.epi & EPI_GEGLU) STAR_LAUNCH((gemm_persist_kernel<T, 2, 0>), grid, block, smem, ctx->stream, p);
    else if (a.epi & EPI_ROWAFF) STAR_LAUNCH((gemm_persist_kernel<T, 8, 0>), grid, block, smem, ctx->stream, p);
    else if (a.epi & EPI_RES) STAR_LAUNCH((gemm_persist_kernel<T, 1, 0>), grid, block, smem, ctx->stream, p);
    else STAR_LAUNCH((gemm_persist_kernel<T, 0, 0>), grid, block, smem, ctx->stream, p);
    return 0;
  }
#endif
  // the product form stores NON-TEMPORALLY (STPOL 2): the 0.4-1.1 GB output streams past the L2 instead of evicting the operand panels the
  // resident workgroups of an XCD share.  cbench A/B on one box, bit-identical (profiles/r06_cbench_persist_walk.txt): GEGLU level 1
  // +1.8 %, level 2 +2.2 %, q | k | v level 2 +4.5 %, stem 4096 x 512 +1.4 %.  (The TILED kernels' epilogues gain nothing from it:
  // profiles/r05_cbench_gemm_nt.txt.)
  if ((a.epi & EPI_ROWAFF) && (a.epi & EPI_GEGLU)) STAR_LAUNCH((gemm_persist_kernel<T, 10, 2>), grid, block, smem, ctx->stream, p);
  else if (a.epi & EPI_GEGLU) STAR_LAUNCH((gemm_persist_kernel<T, 2, 2>), grid, block, smem, ctx->stream, p);
  else if (a.epi & EPI_ROWAFF) STAR_LAUNCH((gemm_persist_kernel<T, 8, 2>), grid, block, smem, ctx->stream, p);
  else if (a.epi & EPI_RES) STAR_LAUNCH((gemm_persist_kernel<T, 1, 2>), grid, block, smem, ctx->stream, p);
  else STAR_LAUNCH((gemm_persist_kernel<T, 0, 2>), grid, block, smem, ctx->stream, p);
  return 0;
}

int launch_gemm_persist(Ctx* ctx, const GemmArgs& a) {
  if (!gemm_persist_covers(a)) return ctx->fail("gemm (persistent tile 18): plain-A layers with the bias / residual / GEGLU / folded-LayerNorm 16-bit epilogues only");
  if ((size_t)256 * a.lda * 2 >= ((size_t)1 << 32) || (size_t)256 * a.ldc * 2 >= ((size_t)1 << 32) || (size_t)256 * a.K * 2 >= ((size_t)1 << 32))
    return ctx->fail("gemm (persistent tile 18): rows too long for 32-bit buffer ranges");
  if (ctx->dtype == DT_F16) return launch_persist_t<f16>(ctx, a);
  if (ctx->dtype == DT_BF16) return launch_persist_t<bf16>(ctx, a);
  return ctx->fail("gemm (persistent tile 18): unsupported dtype");
}

}  // namespace star
