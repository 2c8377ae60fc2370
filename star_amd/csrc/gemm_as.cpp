// gemm_as.cpp -- launcher of gemm_astat_kernel (gemm_as.h); its own translation unit because it is built with
// -mllvm -amdgpu-mfma-vgpr-form (Makefile): one wave per SIMD has a 512-register budget, the A panel lives in AGPRs and the
// accumulators must stay in architectural VGPRs.
#include "ops.h"
#include "gemm_as.h"

namespace star {

// what the kernel can compute at all ...
static bool covered(const GemmArgs& a) {
  return a.mode == A_PLAIN && a.K == 320 && a.N % 64 == 0 && a.N <= 4096 && (a.epi & EPI_ROWAFF) && (a.epi & EPI_BIAS) &&
         !(a.epi & (EPI_RES | EPI_OUT_F32 | EPI_GELU_TANH)) && a.rowab && a.colsum && a.bias && a.lda % 8 == 0 && a.ldc % 8 == 0;
}
// ... and where it is the automatic choice: wide outputs on many rows (the level-0 q | k | v and GEGLU projections)
bool gemm_astat_applies(const GemmArgs& a) { return covered(a) && a.N >= 640 && a.M >= 65536; }

template <class T>
static int launch_astat(Ctx* ctx, const GemmArgs& a) {
  GemmParams p{};
  p.A = a.A; p.W = a.W; p.C = a.C; p.bias = a.bias; p.M = a.M; p.N = a.N; p.K = a.K; p.lda = a.lda; p.ldc = a.ldc; p.epi = a.epi;
  p.rowab = a.rowab; p.colsum = a.colsum;
  const size_t smem = 2 * (size_t)40960 + 4 * (size_t)8192 + 2 * (size_t)a.N * sizeof(float);   // W ring, per-wave staging blocks, bias + colsum
  const dim3 grid((unsigned)((a.M + 255) / 256)), block(256);
  {   // staggered first round (gemm_as.h): one workgroup per CU is resident, so the first round is the first num_cus workgroups
    const int cus = a.assume_cus > 0 ? a.assume_cus : (ctx->num_cus > 0 ? ctx->num_cus : 256);
    p.tiles_m = (long long)grid.x >= 4LL * cus ? cus : 0;
  }
#ifdef STAR_BENCH_VARIANTS   // timing ablations (wrong results): no epilogue / no W staging / W fragments not re-read
  if (a.force_tile == 31) { STAR_LAUNCH((gemm_astat_kernel<T, 0, 1>), grid, block, smem, ctx->stream, p); return 0; }
  if (a.force_tile == 32) { STAR_LAUNCH((gemm_astat_kernel<T, 0, 2>), grid, block, smem, ctx->stream, p); return 0; }
  if (a.force_tile == 33) { STAR_LAUNCH((gemm_astat_kernel<T, 0, 3>), grid, block, smem, ctx->stream, p); return 0; }
  if (a.force_tile == 34) { STAR_LAUNCH((gemm_astat_kernel<T, 0, 4>), grid, block, smem, ctx->stream, p); return 0; }
  if (a.force_tile == 35) { STAR_LAUNCH((gemm_astat_kernel<T, 0, 5>), grid, block, smem, ctx->stream, p); return 0; }
  if (a.force_tile == 42) {   // ILV 6: the two row blocks of a column quad share one fetch of its column sums / biases (correct results, bit-identical)
    if (a.epi & EPI_GEGLU) STAR_LAUNCH((gemm_astat_kernel<T, 1, 0, 6>), grid, block, smem, ctx->stream, p); else STAR_LAUNCH((gemm_astat_kernel<T, 0, 0, 6>), grid, block, smem, ctx->stream, p);
    return 0;
  }
  if (a.force_tile == 41) {   // no global stores (timing only): are the flush's stores what the per-tile vmcnt wait waits for?
    if (a.epi & EPI_GEGLU) STAR_LAUNCH((gemm_astat_kernel<T, 1, 6>), grid, block, smem, ctx->stream, p); else STAR_LAUNCH((gemm_astat_kernel<T, 0, 6>), grid, block, smem, ctx->stream, p);
    return 0;
  }
  if (a.force_tile == 29) {   // the round-4 kernel: plain stores, lockstep start (A/B reference)
    if (a.epi & EPI_GEGLU) STAR_LAUNCH((gemm_astat_kernel<T, 1, 12>), grid, block, smem, ctx->stream, p); else STAR_LAUNCH((gemm_astat_kernel<T, 0, 12>), grid, block, smem, ctx->stream, p);
    return 0;
  }
  if (a.force_tile >= 47 && a.force_tile <= 50) {   // round 5 (correct results): staggered first round, 1 / 2 / 4 / 8 sleep units (127 x 64 cycles) per phase
    p.group_m = 1 << (a.force_tile - 47);
    if (a.epi & EPI_GEGLU) STAR_LAUNCH((gemm_astat_kernel<T, 1, 11>), grid, block, smem, ctx->stream, p); else STAR_LAUNCH((gemm_astat_kernel<T, 0, 11>), grid, block, smem, ctx->stream, p);
    return 0;
  }
  if (a.force_tile >= 43 && a.force_tile <= 46) {   // round 5 (correct results): 43 / 44 / 45 = the flush's stores plain / write-through / write-through + nt (the product's are nt); 46 = every per-tile wait is vmcnt(0)
    const bool g = (a.epi & EPI_GEGLU) != 0;
    switch (a.force_tile) {
      case 43: if (g) STAR_LAUNCH((gemm_astat_kernel<T, 1, 7>), grid, block, smem, ctx->stream, p); else STAR_LAUNCH((gemm_astat_kernel<T, 0, 7>), grid, block, smem, ctx->stream, p); break;
      case 44: if (g) STAR_LAUNCH((gemm_astat_kernel<T, 1, 8>), grid, block, smem, ctx->stream, p); else STAR_LAUNCH((gemm_astat_kernel<T, 0, 8>), grid, block, smem, ctx->stream, p); break;
      case 45: if (g) STAR_LAUNCH((gemm_astat_kernel<T, 1, 9>), grid, block, smem, ctx->stream, p); else STAR_LAUNCH((gemm_astat_kernel<T, 0, 9>), grid, block, smem, ctx->stream, p); break;
      default: if (g) STAR_LAUNCH((gemm_astat_kernel<T, 1, 10>), grid, block, smem, ctx->stream, p); else STAR_LAUNCH((gemm_astat_kernel<T, 0, 10>), grid, block, smem, ctx->stream, p); break;
    }
    return 0;
  }
  // scheduling variants (correct results, bit-identical): 36 / 37 = ILV 1 / 2 (sched_group_barrier interleave of the epilogue with the
  // MFMAs: half units per step / whole units over a pair of steps), 38 / 39 / 40 = the same placements 1 / 0 / 2 with the GELU
  // polynomial as scalar v_fma_f32 (which overlap a wave's own MFMAs; v_pk_fma_f32 do not)
  if (a.force_tile >= 36 && a.force_tile <= 40) {
    const bool g = (a.epi & EPI_GEGLU) != 0;
    switch (a.force_tile) {
      case 36: if (g) STAR_LAUNCH((gemm_astat_kernel<T, 1, 0, 1>), grid, block, smem, ctx->stream, p); else STAR_LAUNCH((gemm_astat_kernel<T, 0, 0, 1>), grid, block, smem, ctx->stream, p); break;
      case 37: if (g) STAR_LAUNCH((gemm_astat_kernel<T, 1, 0, 2>), grid, block, smem, ctx->stream, p); else STAR_LAUNCH((gemm_astat_kernel<T, 0, 0, 2>), grid, block, smem, ctx->stream, p); break;
      case 38: if (g) STAR_LAUNCH((gemm_astat_kernel<T, 1, 0, 3>), grid, block, smem, ctx->stream, p); else return ctx->fail("gemm (A-stationary): tiles 38-40 are GEGLU variants"); break;
      case 39: if (g) STAR_LAUNCH((gemm_astat_kernel<T, 1, 0, 4>), grid, block, smem, ctx->stream, p); else return ctx->fail("gemm (A-stationary): tiles 38-40 are GEGLU variants"); break;
      default: if (g) STAR_LAUNCH((gemm_astat_kernel<T, 1, 0, 5>), grid, block, smem, ctx->stream, p); else return ctx->fail("gemm (A-stationary): tiles 38-40 are GEGLU variants"); break;
    }
    return 0;
  }
#else
  if (a.force_tile > 30 || a.force_tile == 29) return ctx->fail("gemm (A-stationary): ablation ids exist only in the bench build");
#endif
  if (a.epi & EPI_GEGLU) STAR_LAUNCH((gemm_astat_kernel<T, 1>), grid, block, smem, ctx->stream, p);
  else STAR_LAUNCH((gemm_astat_kernel<T, 0>), grid, block, smem, ctx->stream, p);
  return 0;
}

int launch_gemm_astat(Ctx* ctx, const GemmArgs& a) {
  if (!covered(a)) return ctx->fail("gemm (A-stationary): K = 320 plain-A layers with the folded-LayerNorm epilogue only");
  if ((size_t)(a.M < 256 ? a.M : 256) * a.ldc * 2 >= ((size_t)1 << 32)) return ctx->fail("gemm (A-stationary): output rows too long for a 32-bit buffer range");
  if (ctx->dtype == DT_F16) return launch_astat<f16>(ctx, a);
  if (ctx->dtype == DT_BF16) return launch_astat<bf16>(ctx, a);
  return ctx->fail("gemm (A-stationary): unsupported dtype");
}

}  // namespace star
