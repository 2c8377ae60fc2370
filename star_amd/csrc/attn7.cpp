// attn7.cpp -- launcher of flash_attn_v7_kernel (attn7.h).  Its own translation unit because it is built with
// -mllvm -amdgpu-mfma-vgpr-form (Makefile): one wave per SIMD has a 512-register budget, and without that flag hipcc selects
// the AGPR form for every MFMA result, which vector instructions cannot read.
// Bench build only (-DSTAR_BENCH_VARIANTS): measured against the shipped kernel in profiles/r03_attn7_*.txt and not faster --
// both are pinned to the socket power cap -- so the product library does not carry it.
#ifdef STAR_BENCH_VARIANTS
#include <cstdio>
#include <cstdlib>
#include "ops.h"
#include "attn7.h"

namespace star {

template <class T, int NQ>
static int launch_v7(Ctx* ctx, AttnParams p, int pksum) {
  p.nqb = (p.Nq + 128 * NQ - 1) / (128 * NQ);
  const int BH = p.batch * p.heads;
  const long long nblk = 8LL * p.nqb * ((BH + 7) / 8);
  constexpr size_t LDS = 4 * 16384;
  if constexpr (__is_same(T, f16)) {
    if (pksum) { STAR_LAUNCH((flash_attn_v7_kernel<T, NQ, 1>), dim3((unsigned)nblk), dim3(256), LDS, ctx->stream, p); return 0; }
  }
  STAR_LAUNCH((flash_attn_v7_kernel<T, NQ, 0>), dim3((unsigned)nblk), dim3(256), LDS, ctx->stream, p);
  return 0;
}

template <class T, int ABL>
static int launch_v7_abl(Ctx* ctx, AttnParams p) {
  p.nqb = (p.Nq + 128 * 3 - 1) / (128 * 3);
  const long long nblk = 8LL * p.nqb * ((p.batch * p.heads + 7) / 8);
  STAR_LAUNCH((flash_attn_v7_kernel<T, 3, 1, ABL>), dim3((unsigned)nblk), dim3(256), (size_t)(4 * 16384), ctx->stream, p);
  return 0;
}
// timing ablations of the NQ = 3 kernel (f16): wrong results by construction
int launch_flash_v7_abl(Ctx* ctx, const AttnParams& p, int abl) {
  if (ctx->dtype != DT_F16) return ctx->fail("flash_attn v7 ablations: f16 only");
  switch (abl) {
    case 1: return launch_v7_abl<f16, 1>(ctx, p);
    case 2: return launch_v7_abl<f16, 2>(ctx, p);
    case 4: return launch_v7_abl<f16, 4>(ctx, p);
    case 6: return launch_v7_abl<f16, 6>(ctx, p);
    case 8: return launch_v7_abl<f16, 8>(ctx, p);
    case 14: return launch_v7_abl<f16, 14>(ctx, p);
    default: return ctx->fail("flash_attn v7 ablations: unknown mask");
  }
}

// nq: 32-row query blocks per wave (2 or 3); pksum: packed 16-bit row sums (f16, long key ranges)
int launch_flash_v7(Ctx* ctx, const AttnParams& p, int nq, int pksum) {
  if (ctx->dtype == DT_F16) return nq == 3 ? launch_v7<f16, 3>(ctx, p, pksum) : launch_v7<f16, 2>(ctx, p, pksum);
  if (ctx->dtype == DT_BF16) return nq == 3 ? launch_v7<bf16, 3>(ctx, p, 0) : launch_v7<bf16, 2>(ctx, p, 0);
  return ctx->fail("flash_attn v7: unsupported dtype");
}

}  // namespace star
#endif  // STAR_BENCH_VARIANTS
