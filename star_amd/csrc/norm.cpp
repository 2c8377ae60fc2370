// norm.cpp -- launchers for norm.h
#include <cstdint>
#include <cstdlib>
#include "ops.h"
#include "norm.h"

namespace star {

static inline unsigned ew_grid(long long total, int block = 256) {
  long long g = (total + block - 1) / block;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (unsigned)g;
}

// ab_out != nullptr: statistics + finalize only -- the per-(stat, channel) affine pairs go to ab_out and nothing is applied
template <class T>
static int gn_t(Ctx* ctx, const void* x, int ldx, void* y, int ldy, const float* gamma, const float* beta,
                int rows, int C, int rows_per_stat, float eps, bool silu, float* ab_out = nullptr, float* mu_out = nullptr) {
  const int nstat = rows / rows_per_stat;
  const int CC8 = C / 8;
  int RL = 256 / CC8; if (RL < 1) RL = 1;
  const int nthreads = CC8 * RL;
  // slab: enough rows per block to amortise the reduction, enough blocks to fill the chip
  int slab = 256;
  while ((long long)((rows_per_stat + slab - 1) / slab) * nstat < 1024 && slab > 32) slab >>= 1;
  const int nslab = (rows_per_stat + slab - 1) / slab;
  Buf partial(ctx, (size_t)nstat * nslab * 64 * sizeof(double));
  Buf ab(ctx, ab_out ? 0 : (size_t)nstat * C * 2 * sizeof(float));
  float* abp = ab_out ? ab_out : ab.as<float>();
  if (!partial.p || !abp) return ctx->fail("group_norm: out of device memory");
  // (Round 4 folded the finalize step into the statistics kernel -- its last block per stat, elected by an arrival counter, partials
  // laid out [stat][group][slab] -- and measured it SLOWER: 2x for the family with every norm folded, +28 % with only the per-frame
  // norms folded (profiles/r04_gn_fold_ab.txt): the agent-scope release every block needs before it arrives writes back / invalidates
  // the XCD's L2, and a whole-chunk norm's single electing block reduces 3294 slabs alone.  Three launches it stays.)
  const double count = (double)rows_per_stat * (C / 32);
  GnStatsParams sp{x, ldx, C, rows_per_stat, slab, partial.as<double>()};
  dim3 grid((unsigned)nslab, (unsigned)nstat);
  STAR_LAUNCH((gn_stats_kernel<T>), grid, dim3(nthreads), (size_t)nthreads * 64, ctx->stream, sp);
  GnFinalizeParams fp{partial.as<double>(), gamma, beta, abp, C, nstat, nslab, count, eps, mu_out};
  STAR_LAUNCH(gn_finalize_kernel, dim3((unsigned)((nstat * 32 + 3) / 4)), dim3(256), (size_t)0, ctx->stream, fp);
  if (ab_out) return 0;
  GnApplyParams ap{x, y, abp, ldx, ldy, C, rows_per_stat, slab, silu ? 1 : 0};
  if (silu) STAR_LAUNCH((gn_apply_kernel<T, true>), grid, dim3(nthreads), (size_t)0, ctx->stream, ap);
  else STAR_LAUNCH((gn_apply_kernel<T, false>), grid, dim3(nthreads), (size_t)0, ctx->stream, ap);
  return 0;
}

// GroupNorm whose statistics come from the producer's epilogue (gn_partial: gemm.h EPIF bit 4): finalize from the partials (+ the rows
// of the slots a stat boundary cuts), then the apply pass; y == nullptr: finalize only (ab_out / mu_out, for the weight fold)
template <class T>
static int gn_fused_t(Ctx* ctx, const void* x, int ldx, void* y, int ldy, const float* gamma, const float* beta, int rows, int C,
                      int rows_per_stat, float eps, bool silu, const float* partial, float* ab_out, float* mu_out) {
  const int nstat = rows / rows_per_stat;
  Buf ab(ctx, ab_out ? 0 : (size_t)nstat * C * 2 * sizeof(float));
  float* abp = ab_out ? ab_out : ab.as<float>();
  if (!abp) return ctx->fail("group_norm: out of device memory");
  const double count = (double)rows_per_stat * (C / 32);
  // enough workgroups to fill the chip: a whole-chunk norm (nstat = 1, 26 352 slots at level 0) is split into parts along the slots,
  // reduced by the stand-alone path's finalize kernel (fixed order)
  int split = 1;
  while (nstat * 32 * split < 1024 && (rows_per_stat >> 5) / (split * 2) >= 64) split *= 2;
  Buf parts(ctx, split > 1 ? (size_t)nstat * 32 * split * 2 * sizeof(double) : 0);
  if (split > 1 && !parts.p) return ctx->fail("group_norm: out of device memory");
  GnFinalizeFusedParams fp{partial, x, ldx, gamma, beta, abp, mu_out, C, nstat, rows_per_stat, count, eps, split, parts.as<double>()};
  const int nthr = 256;
  STAR_LAUNCH((gn_finalize_fused_kernel<T>), dim3((unsigned)(nstat * 32 * split)), dim3((unsigned)nthr), (size_t)nthr * 16, ctx->stream, fp);
  if (split > 1) {
    GnFinalizeParams f2{parts.as<double>(), gamma, beta, abp, C, nstat, split, count, eps, mu_out};
    STAR_LAUNCH(gn_finalize_kernel, dim3((unsigned)((nstat * 32 + 3) / 4)), dim3(256), (size_t)0, ctx->stream, f2);
  }
  if (!y) return 0;
  const int CC8 = C / 8;
  int RL = 256 / CC8; if (RL < 1) RL = 1;
  int slab = 256;
  while ((long long)((rows_per_stat + slab - 1) / slab) * nstat < 1024 && slab > 32) slab >>= 1;
  const int nslab = (rows_per_stat + slab - 1) / slab;
  GnApplyParams ap{x, y, abp, ldx, ldy, C, rows_per_stat, slab, silu ? 1 : 0};
  if (silu) STAR_LAUNCH((gn_apply_kernel<T, true>), dim3((unsigned)nslab, (unsigned)nstat), dim3((unsigned)(CC8 * RL)), (size_t)0, ctx->stream, ap);
  else STAR_LAUNCH((gn_apply_kernel<T, false>), dim3((unsigned)nslab, (unsigned)nstat), dim3((unsigned)(CC8 * RL)), (size_t)0, ctx->stream, ap);
  return 0;
}

int op_group_norm_fused(Ctx* ctx, const void* x, int ldx, void* y, int ldy, const float* gamma, const float* beta, int rows, int C,
                        int rows_per_stat, float eps, bool silu, const float* partial, float* ab_out, float* mu_out) {
  if (C % 64) return ctx->fail("group_norm (fused statistics): C must be a multiple of 64 (an even number of channels per group)");
  if (rows % rows_per_stat) return ctx->fail("group_norm: rows not a multiple of rows_per_stat");
  if ((ldx | (y ? ldy : 8)) & 7) return ctx->fail("group_norm: row strides must be multiples of 8");
  if (C / 8 > 1024) return ctx->fail("group_norm: C too large");
  if (!partial || (!y && !ab_out)) return ctx->fail("group_norm (fused statistics): null argument");
  ++ctx->gn_fused;
  ProfScope ps(ctx, PK_GN, 0.0, (y ? 2.0 : 0.0) * rows * (double)C * 2.0 + (double)((rows + 31) / 32) * C * 4.0);
  if (ctx->dtype == DT_F16) return gn_fused_t<f16>(ctx, x, ldx, y, ldy, gamma, beta, rows, C, rows_per_stat, eps, silu, partial, ab_out, mu_out);
  return gn_fused_t<bf16>(ctx, x, ldx, y, ldy, gamma, beta, rows, C, rows_per_stat, eps, silu, partial, ab_out, mu_out);
}

int op_group_norm(Ctx* ctx, const void* x, int ldx, void* y, int ldy, const float* gamma, const float* beta,
                  int rows, int C, int rows_per_stat, float eps, bool silu) {
  if (C % 32 || C % 8) return ctx->fail("group_norm: C must be a multiple of 32");
  if (rows % rows_per_stat) return ctx->fail("group_norm: rows not a multiple of rows_per_stat");
  if ((ldx | ldy) & 7) return ctx->fail("group_norm: row strides must be multiples of 8");
  if (C / 8 > 1024) return ctx->fail("group_norm: C too large");
  ProfScope ps(ctx, PK_GN, 0.0, 3.0 * rows * (double)C * 2.0);
  if (ctx->dtype == DT_F16) return gn_t<f16>(ctx, x, ldx, y, ldy, gamma, beta, rows, C, rows_per_stat, eps, silu);
  return gn_t<bf16>(ctx, x, ldx, y, ldy, gamma, beta, rows, C, rows_per_stat, eps, silu);
}

int op_group_norm_stats(Ctx* ctx, const void* x, int ldx, const float* gamma, const float* beta, int rows, int C, int rows_per_stat,
                        float eps, float* ab, float* mu) {
  if (C % 32 || C % 8) return ctx->fail("group_norm: C must be a multiple of 32");
  if (rows % rows_per_stat) return ctx->fail("group_norm: rows not a multiple of rows_per_stat");
  if (ldx & 7) return ctx->fail("group_norm: row strides must be multiples of 8");
  if (C / 8 > 1024) return ctx->fail("group_norm: C too large");
  if (!ab) return ctx->fail("group_norm_stats: null output");
  ProfScope ps(ctx, PK_GN, 0.0, 1.0 * rows * (double)C * 2.0);
  if (ctx->dtype == DT_F16) return gn_t<f16>(ctx, x, ldx, nullptr, 8, gamma, beta, rows, C, rows_per_stat, eps, false, ab, mu);
  return gn_t<bf16>(ctx, x, ldx, nullptr, 8, gamma, beta, rows, C, rows_per_stat, eps, false, ab, mu);
}

int op_gn_fold_weights(Ctx* ctx, const void* W, const float* bias, const float* ab, void* Wout, float* bias_out, int N, int K, const float* mu) {
  if (N <= 0 || K <= 0) return 0;
  ProfScope ps(ctx, PK_GN, 0.0, 2.0 * N * (double)K * 2.0);
  GnFoldParams p{W, bias, ab, Wout, bias_out, N, K, mu};
  if (ctx->dtype == DT_F16) STAR_LAUNCH((gn_fold_weights_kernel<f16>), dim3((unsigned)N), dim3(256), (size_t)256 * 4, ctx->stream, p);
  else STAR_LAUNCH((gn_fold_weights_kernel<bf16>), dim3((unsigned)N), dim3(256), (size_t)256 * 4, ctx->stream, p);
  return 0;
}

template <class T>
static int ln_t(Ctx* ctx, LnParams p) {
  const int CC8 = p.C / 8;
  int lpr = 8;
  while (lpr * 5 < CC8) lpr <<= 1;
  const bool wide = lpr > 64;       // 2560 < C <= 5120: one row per wavefront, ten 16-B chunks per lane
  if (wide) lpr = 64;
  if (wide && 64 * 10 < CC8) return ctx->fail("layer_norm: C too large (max 5120)");
  p.lpr = lpr;
  const int rows_per_block = 4 * (64 / lpr);
  dim3 grid((unsigned)((p.rows + rows_per_block - 1) / rows_per_block)), block(256);
  if (wide) STAR_LAUNCH((ln_kernel<T, 10>), grid, block, (size_t)0, ctx->stream, p);
  else STAR_LAUNCH((ln_kernel<T, 5>), grid, block, (size_t)0, ctx->stream, p);
  return 0;
}

int op_layer_norm(Ctx* ctx, const void* x, int ldx, void* y, int ldy, const float* gamma, const float* beta,
                  int rows, int C, float eps, int mode, const float* gate_w, float* maps, int H, int W, float* rowab) {
  if (C % 8) return ctx->fail("layer_norm: C must be a multiple of 8");
  if ((ldx | ldy) & 7) return ctx->fail("layer_norm: row strides must be multiples of 8");
  if (rows <= 0) return 0;
  ProfScope ps(ctx, PK_LN, 0.0, ((mode == LN_STATS_ONLY || rowab) ? 1.0 : 2.0) * rows * (double)C * 2.0);
  LnParams p{x, y, gamma, beta, gate_w, maps, ldx, ldy, C, rows, H, W, eps, mode, 64, rowab};
  if (ctx->dtype == DT_F16) return ln_t<f16>(ctx, p);
  return ln_t<bf16>(ctx, p);
}

// the row coefficients / LIEM maps of a LayerNorm from the row statistics its input's producer wrote (GemmArgs::ln_partial)
int op_layer_norm_from_partials(Ctx* ctx, const float* partial, int parts, int rows, int C, float eps, int mode, const float* gate_w,
                                float* maps, int H, int W, float* rowab) {
  if (rows <= 0) return 0;
  if (!partial || parts <= 0) return ctx->fail("layer_norm_from_partials: no partials");
  if (mode == LN_STATS_ONLY ? !maps : !rowab) return ctx->fail("layer_norm_from_partials: null output");
  if ((mode == LN_GATE_LINEAR && !gate_w) || (mode == LN_GATE_MAP && (!gate_w || !maps || H * W <= 0))) return ctx->fail("layer_norm_from_partials: gate operands missing");
  ProfScope ps(ctx, PK_LN, 0.0, (double)rows * parts * 16.0);
  ++ctx->ln_fused;
  LnPartParams p{partial, parts, gate_w, maps, rowab, C, rows, H, W, eps, mode};
  STAR_LAUNCH(ln_from_partials_kernel, dim3((unsigned)((rows + 31) / 32)), dim3(256), (size_t)0, ctx->stream, p);
  return 0;
}

int op_concat_add(Ctx* ctx, const void* a, const void* b, const void* c, void* out, int rows, int C1, int C2, float* gn_partial) {
  ProfScope ps(ctx, PK_MISC, 0.0, 2.0 * rows * (double)(C1 + C2 + (c ? C2 : 0)) * 2.0);
  if ((C1 | C2) & 7) return ctx->fail("concat_add: channel counts must be multiples of 8");
  if (gn_partial) {   // + the GroupNorm partial statistics of the output (one workgroup per 32-row slot)
    const int CT8 = (C1 + C2) / 8;
    if (CT8 > 1024) return ctx->fail("concat_add: too many channels for the statistics form");
    int RL = 256 / CT8; if (RL < 1) RL = 1;
    if (RL > 32) RL = 32;
    ConcatStatsParams sp{a, b, c, out, C1, C2, rows, gn_partial};
    const dim3 grid((unsigned)((rows + 31) / 32)), block((unsigned)(CT8 * RL));
    const size_t smem = (size_t)CT8 * RL * 8 * sizeof(float);
    if (ctx->dtype == DT_F16) STAR_LAUNCH((concat_add_stats_kernel<f16>), grid, block, smem, ctx->stream, sp);
    else STAR_LAUNCH((concat_add_stats_kernel<bf16>), grid, block, smem, ctx->stream, sp);
    return 0;
  }
  ConcatParams p{a, b, c, out, C1, C2, rows};
  const unsigned g = ew_grid((long long)rows * ((C1 + C2) / 8));
  if (ctx->dtype == DT_F16) STAR_LAUNCH((concat_add_kernel<f16>), dim3(g), dim3(256), (size_t)0, ctx->stream, p);
  else STAR_LAUNCH((concat_add_kernel<bf16>), dim3(g), dim3(256), (size_t)0, ctx->stream, p);
  return 0;
}

int op_add(Ctx* ctx, const void* a, const void* b, void* out, long long n) {
  ProfScope ps(ctx, PK_MISC, 0.0, 3.0 * n * 2.0);
  if (n & 7) return ctx->fail("add: n must be a multiple of 8");
  AddParams p{a, b, out, n / 8};
  const unsigned g = ew_grid(n / 8);
  if (ctx->dtype == DT_F16) STAR_LAUNCH((add_kernel<f16>), dim3(g), dim3(256), (size_t)0, ctx->stream, p);
  else STAR_LAUNCH((add_kernel<bf16>), dim3(g), dim3(256), (size_t)0, ctx->stream, p);
  return 0;
}

int op_stem_im2col(Ctx* ctx, const float* latent, void* out, int Cl, int F, int H, int W, bool frame_major) {
  if (9 * Cl > 64) return ctx->fail("stem_im2col: at most 7 latent channels");
  const long long hw = (long long)H * W;
  StemIm2colParams p{latent, out, Cl, F, H, W, frame_major ? (long long)Cl * hw : hw, frame_major ? hw : (long long)F * hw};
  const unsigned g = ew_grid((long long)F * H * W * 8);
  if (ctx->dtype == DT_F16) STAR_LAUNCH((stem_im2col_kernel<f16>), dim3(g), dim3(256), (size_t)0, ctx->stream, p);
  else STAR_LAUNCH((stem_im2col_kernel<bf16>), dim3(g), dim3(256), (size_t)0, ctx->stream, p);
  return 0;
}

int op_rows_to_latent(Ctx* ctx, const float* rows, float* out, int Cl, int ld, long long ntok) {
  RowsToLatentParams p{rows, out, Cl, ld, ntok};
  STAR_LAUNCH(rows_to_latent_kernel, dim3(ew_grid(ntok * Cl)), dim3(256), (size_t)0, ctx->stream, p);
  return 0;
}

int op_gemv(Ctx* ctx, const float* x, const void* W, const float* b, float* y, int N, int K, bool silu_in, bool silu_out) {
  if (K & 7) return ctx->fail("gemv: K must be a multiple of 8");
  GemvParams p{x, W, b, y, N, K, silu_in ? 1 : 0, silu_out ? 1 : 0};
  dim3 grid((unsigned)((N + 3) / 4)), block(256);
  if (ctx->dtype == DT_F16) STAR_LAUNCH((gemv_kernel<f16>), grid, block, (size_t)0, ctx->stream, p);
  else STAR_LAUNCH((gemv_kernel<bf16>), grid, block, (size_t)0, ctx->stream, p);
  return 0;
}

int op_softmax_rows(Ctx* ctx, const float* s, int lds, void* pout, int ldp, int rows, int n, float scale) {
  ProfScope ps(ctx, PK_MISC, 0.0, (double)rows * n * 10.0);
  SoftmaxParams p{s, pout, rows, n, lds, ldp, scale * 1.4426950408889634f};
  // one-read form (norm.h: softmax_rows_vec_kernel) where the layout allows whole 8-element chunks: the VAE's logits (ld = Np, a multiple of 64)
  if (!(lds & 3) && !(ldp & 7) && ldp >= 8 && n <= ldp && n <= lds && !((uintptr_t)s & 15) && !((uintptr_t)pout & 15)) {
    if (ctx->dtype == DT_F16) STAR_LAUNCH((softmax_rows_vec_kernel<f16, 13>), dim3((unsigned)rows), dim3(256), (size_t)64, ctx->stream, p);
    else STAR_LAUNCH((softmax_rows_vec_kernel<bf16, 13>), dim3((unsigned)rows), dim3(256), (size_t)64, ctx->stream, p);
    return 0;
  }
  if (ctx->dtype == DT_F16) STAR_LAUNCH((softmax_rows_kernel<f16>), dim3((unsigned)rows), dim3(256), (size_t)64, ctx->stream, p);
  else STAR_LAUNCH((softmax_rows_kernel<bf16>), dim3((unsigned)rows), dim3(256), (size_t)64, ctx->stream, p);
  return 0;
}

int op_time_conv_out(Ctx* ctx, const float* rows, int ld, float* out, const float* w, const float* b, int F, int HW, int C) {
  TimeConvOutParams p{rows, out, w, b, F, HW, ld, C};
  STAR_LAUNCH(time_conv_out_kernel, dim3(ew_grid((long long)F * HW)), dim3(256), (size_t)0, ctx->stream, p);
  return 0;
}

int op_vec_add_f32(Ctx* ctx, float* a, const float* b, int n) {
  VecAddParams p{a, b, n};
  STAR_LAUNCH(vec_add_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), (size_t)0, ctx->stream, p);
  return 0;
}

int op_cast(Ctx* ctx, const float* x, void* y, long long n) {
  if (n & 7) return ctx->fail("cast: n must be a multiple of 8");
  CastParams p{x, y, n / 8};
  const unsigned g = ew_grid(n / 8);
  if (ctx->dtype == DT_F16) STAR_LAUNCH((cast_kernel<f16>), dim3(g), dim3(256), (size_t)0, ctx->stream, p);
  else STAR_LAUNCH((cast_kernel<bf16>), dim3(g), dim3(256), (size_t)0, ctx->stream, p);
  return 0;
}

}  // namespace star
