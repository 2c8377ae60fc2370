// attn.cpp -- launchers for the attention kernels (attn.h)
#include <cstdio>
#include <cstdlib>
#include "ops.h"
#include "attn.h"
#include "attn5.h"

namespace star {

int launch_flash_v7(Ctx* ctx, const AttnParams& p, int nq, int pksum);   // attn7.cpp
int launch_flash_v7_abl(Ctx* ctx, const AttnParams& p, int abl);

template <class T>
static int launch_flash(Ctx* ctx, const AttnArgs& a) {
  AttnParams p{};
  p.Q = a.Q; p.K = a.K; p.V = a.V; p.O = a.O;
  p.ldq = a.ldq; p.ldk = a.ldk; p.ldv = a.ldv; p.ldo = a.ldo;
  p.bsq = a.bsq; p.bsk = a.bsk; p.bsv = a.bsv; p.bso = a.bso;
  p.Nq = a.Nq; p.Nk = a.Nk; p.heads = a.heads; p.batch = a.batch;
  p.scale_log2e = a.scale * 1.4426950408889634f;
  p.nqb = (a.Nq + 255) / 256;
  const int BH = a.batch * a.heads;
  const long long nblk = 8LL * p.nqb * ((BH + 7) / 8);
  p.variant = a.variant;
#ifndef STAR_HOSTEMU
  {   // resident workgroups per CU of the product kernel (debug aid; the environment is read once)
    static bool once = std::getenv("STAR_DEBUG_OCC") == nullptr;
    if (!once) {
      once = true;
      int nb = -1;
      hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, flash_attn_v5_kernel<T, 0, 1>, 256, 32768);
      fprintf(stderr, "[star] flash_attn_v5<0>: %d workgroups of 256 threads per CU (err %d)\n", nb, (int)e);
    }
  }
#endif
  if (a.causal) {   // text tower: 77 tokens, causal mask (embedder.py:59)
    if (a.variant != 9) return ctx->fail("flash_attn: the causal mask exists in the product kernel only");
    if (a.Nq != a.Nk) return ctx->fail("flash_attn: the causal mask needs Nq == Nk");
    STAR_LAUNCH((flash_attn_v5_kernel<T, 0, 1, 1>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p);
    return 0;
  }
  if (a.variant == 9) {   // the shipped kernel (attn5.h).  Packed 16-bit row sums only where hundreds of key tiles average their
                          // rounding out (spatial self-attention); the 77-token cross-attention keeps fp32 row sums
    if constexpr (__is_same(T, f16)) {
      if (a.Nk >= 1024) { STAR_LAUNCH((flash_attn_v5_kernel<T, 1, 1>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p); return 0; }
    }
    STAR_LAUNCH((flash_attn_v5_kernel<T, 0, 1>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p);
    return 0;
  }
#ifdef STAR_BENCH_VARIANTS
  if (a.variant == 40 || a.variant == 41) return launch_flash_v7(ctx, p, a.variant == 41 ? 3 : 2, a.Nk >= 1024 ? 1 : 0);   // attn7.h: one wave per SIMD, query-block pipeline
  if (a.variant >= 50 && a.variant < 70) return launch_flash_v7_abl(ctx, p, a.variant - 50);
  if (a.variant == 33) {   // v5 with the augmented k-step as a full-depth 32x32x16 MFMA (the shipped kernel uses the half-depth 32x32x8 form)
    if constexpr (__is_same(T, f16)) { STAR_LAUNCH((flash_attn_v5_kernel<T, 1, 0>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p); return 0; }
    else { STAR_LAUNCH((flash_attn_v5_kernel<T, 0, 0>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p); return 0; }
  }
  if (a.variant == 34) {   // round 6: ONE 512-thread workgroup per CU, eight waves share a K / V stage (attn5.h NW = 8); bit-identical to the product kernel
    p.nqb = (a.Nq + 511) / 512;
    const long long nblk8 = 8LL * p.nqb * ((BH + 7) / 8);
    if constexpr (__is_same(T, f16)) {
      if (a.Nk >= 1024) { STAR_LAUNCH((flash_attn_v5_kernel<T, 1, 1, 0, 8>), dim3((unsigned)nblk8), dim3(512), (size_t)32768, ctx->stream, p); return 0; }
    }
    STAR_LAUNCH((flash_attn_v5_kernel<T, 0, 1, 0, 8>), dim3((unsigned)nblk8), dim3(512), (size_t)32768, ctx->stream, p);
    return 0;
  }
  if (a.variant == 35) {   // round 6: the product kernel with the probabilities packed round-toward-zero (v_cvt_pkrtz_f16_f32), f16 / long key ranges
    if constexpr (__is_same(T, f16)) {
      if (a.Nk >= 1024) { STAR_LAUNCH((flash_attn_v5_kernel<T, 1, 1, 0, 4, 1>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p); return 0; }
    }
    return ctx->fail("flash_attn: variant 35 (round-toward-zero pack) is f16 with Nk >= 1024 only");
  }
  if (a.variant == 32) { STAR_LAUNCH((flash_attn_v3_kernel<T, 2, 1, 0, 1>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p); return 0; }   // the product kernel of rounds 1-2
  if (a.variant == 30) { STAR_LAUNCH((flash_attn_v5_kernel<T, 0>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p); return 0; }
  if (a.variant == 31) {
    if constexpr (__is_same(T, f16)) { STAR_LAUNCH((flash_attn_v5_kernel<T, 1>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p); return 0; }
    else return ctx->fail("flash_attn: variant 31 (packed row sums) is f16 only");
  }
  // measured-and-lost A/B variants and ablation probes (several compute deliberately wrong results): bench build / emulator only
  if (a.variant == 2) STAR_LAUNCH((flash_attn_v3_kernel<T, 2>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p);
  else if (a.variant == 6) STAR_LAUNCH((flash_attn_v3_kernel<T, 2, 1>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p);
  else if (a.variant == 7) STAR_LAUNCH((flash_attn_v3_kernel<T, 2, 2>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p);
  else if (a.variant == 8) STAR_LAUNCH((flash_attn_v3_kernel<T, 2, 3>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p);
  else if (a.variant == 10) STAR_LAUNCH((flash_attn_v3_kernel<T, 2, 4>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p);
  else if (a.variant == 15) STAR_LAUNCH((flash_attn_v3_kernel<T, 2, 5>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p);
  else if (a.variant == 11) STAR_LAUNCH((flash_attn_v3_kernel<T, 2, 1, 1>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p);   // ablation probes of variant 6:
  else if (a.variant == 12) STAR_LAUNCH((flash_attn_v3_kernel<T, 2, 1, 2>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p);   // no exp / no PV MFMA /
  else if (a.variant == 13) STAR_LAUNCH((flash_attn_v3_kernel<T, 2, 1, 3>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p);   // no K/V staging /
  else if (a.variant == 14) STAR_LAUNCH((flash_attn_v3_kernel<T, 2, 1, 4>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p);   // no staging, no barriers
  else if (a.variant == 16) STAR_LAUNCH((flash_attn_v3_kernel<T, 2, 1, 5, 1>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p);   // no softmax VALU
  else if (a.variant == 17) STAR_LAUNCH((flash_attn_v3_kernel<T, 2, 1, 3, 1>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p);   // variant 9 without staging
  else if (a.variant == 23) STAR_LAUNCH((flash_attn_v3_kernel<T, 2, 1, 0, 1, 0, 0, 1>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p);   // s_setprio 1 around the MFMA clusters
  else if (a.variant == 24) STAR_LAUNCH((flash_attn_v3_kernel<T, 2, 1, 0, 1, 0, 0, 2>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p);   // static priority for odd workgroups
  else if (a.variant == 25) STAR_LAUNCH((flash_attn_v3_kernel<T, 2, 1, 0, 1, 0, 0, 3>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p);   // ... by bit 8 of the block id
  else if (a.variant == 26) STAR_LAUNCH((flash_attn_v3_kernel<T, 2, 1, 0, 1, 0, 0, 4>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p);   // ... by bit 3
  else if (a.variant == 27) STAR_LAUNCH((flash_attn_v3_kernel<T, 2, 1, 0, 1, 0, 2>), dim3((unsigned)nblk), dim3(256), (size_t)65536, ctx->stream, p);   // key tiles in pairs: one barrier per 128 keys
  else if (a.variant == 21) STAR_LAUNCH((flash_attn_v3_kernel<T, 2, 1, 0, 1, 1>), dim3((unsigned)nblk), dim3(256), (size_t)32768, ctx->stream, p);
  else if (a.variant == 22) STAR_LAUNCH((flash_attn_v3_kernel<T, 2, 1, 0, 1, 0, 1>), dim3((unsigned)nblk), dim3(256), (size_t)(3 * 16384), ctx->stream, p);
  else if (a.variant == 3) {   // 128-row workgroups, 4 waves per SIMD
    p.nqb = (a.Nq + 127) / 128;
    const long long nblk4 = 8LL * p.nqb * ((BH + 7) / 8);
    STAR_LAUNCH((flash_attn_v3_kernel<T, 1>), dim3((unsigned)nblk4), dim3(256), (size_t)32768, ctx->stream, p);
  } else return ctx->fail("flash_attn: unknown variant id (0 / 1 / 4 / 5 / 20, the first losers of round 1, were removed in round 3: git history)");
  return 0;
#else
  return ctx->fail("flash_attn: variant ids other than 9 exist only in the bench build (make bench)");
#endif
}

int op_flash_attn(Ctx* ctx, const AttnArgs& a) {
  if (a.Nq <= 0 || a.Nk <= 0 || a.batch * a.heads <= 0) return 0;
  if ((a.ldq | a.ldk | a.ldv | a.ldo) & 7) return ctx->fail("flash_attn: row strides must be multiples of 8 elements");
  ProfScope ps(ctx, a.bsk == 0 && a.batch > 1 ? PK_ATTN_CROSS : PK_ATTN_SELF, 4.0 * a.batch * a.heads * (double)a.Nq * a.Nk * 64.0,
               2.0 * a.batch * a.heads * 64.0 * (2.0 * a.Nq + 2.0 * (a.bsk == 0 ? a.Nk / (double)a.batch : a.Nk)),
               a.batch, a.heads, a.Nq, a.Nk);
  if (ctx->dtype == DT_F16) return launch_flash<f16>(ctx, a);
  if (ctx->dtype == DT_BF16) return launch_flash<bf16>(ctx, a);
  return ctx->fail("flash_attn: unsupported dtype");
}

template <class T>
static int launch_tattn(Ctx* ctx, const TAttnArgs& a) {
  TAttnParams p{};
  p.Q = a.Q; p.K = a.K; p.V = a.V; p.O = a.O;
  p.ldq = a.ldq; p.ldk = a.ldk; p.ldv = a.ldv; p.ldo = a.ldo;
  p.F = a.F; p.HW = a.HW; p.heads = a.heads;
  p.scale_log2e = a.scale * 1.4426950408889634f;
  const long long items = (long long)a.HW * a.heads;
  const unsigned grid = (unsigned)((items + 3) / 4);
  if (a.F <= 32) STAR_LAUNCH((temporal_attn_kernel<T, 1>), dim3(grid), dim3(256), (size_t)(4 * 4096), ctx->stream, p);
  else if (a.F <= 64) STAR_LAUNCH((temporal_attn_kernel<T, 2>), dim3(grid), dim3(256), (size_t)(4 * 8192), ctx->stream, p);
  else if (a.F <= 96) STAR_LAUNCH((temporal_attn_kernel<T, 3>), dim3(grid), dim3(256), (size_t)(4 * 12288), ctx->stream, p);
  else STAR_LAUNCH((temporal_attn_kernel<T, 4>), dim3(grid), dim3(256), (size_t)(4 * 16384), ctx->stream, p);
  return 0;
}

int op_temporal_attn(Ctx* ctx, const TAttnArgs& a) {
  if (a.F <= 0 || a.HW <= 0) return 0;
  if (a.F > 128) return ctx->fail("temporal_attn: at most 128 frames per chunk");
  if ((a.ldq | a.ldk | a.ldv | a.ldo) & 7) return ctx->fail("temporal_attn: row strides must be multiples of 8 elements");
  ProfScope ps(ctx, PK_TATTN, 4.0 * a.HW * a.heads * (double)a.F * a.F * 64.0, 2.0 * 4.0 * a.F * (double)a.HW * a.heads * 64.0);
  if (ctx->dtype == DT_F16) return launch_tattn<f16>(ctx, a);
  if (ctx->dtype == DT_BF16) return launch_tattn<bf16>(ctx, a);
  return ctx->fail("temporal_attn: unsupported dtype");
}

}  // namespace star
