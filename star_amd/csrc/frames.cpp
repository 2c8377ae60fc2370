// frames.cpp -- launchers for frames.h
#include "ops.h"
#include "frames.h"

namespace star {

int op_resize_pad(Ctx* ctx, const float* src, float* dst, int planes, int h, int w, int th, int tw,
                  int pad_l, int pad_r, int pad_t, int pad_b, float pad_value) {
  if (planes <= 0 || h <= 0 || w <= 0 || th <= 0 || tw <= 0) return ctx->fail("resize_pad: empty input");
  if (pad_l < 0 || pad_r < 0 || pad_t < 0 || pad_b < 0) return ctx->fail("resize_pad: negative padding");
  if (planes > 65535) return ctx->fail("resize_pad: too many planes");
  ResizePadParams p{};
  p.src = src; p.dst = dst; p.planes = planes; p.h = h; p.w = w; p.th = th; p.tw = tw;
  p.pad_l = pad_l; p.pad_t = pad_t; p.oh = th + pad_t + pad_b; p.ow = tw + pad_l + pad_r;
  p.sy = (float)h / (float)th; p.sx = (float)w / (float)tw; p.pad_value = pad_value;
  const long long total = (long long)p.oh * p.ow;
  if (total > 0x7fffffffLL) return ctx->fail("resize_pad: plane too large");
  ProfScope ps(ctx, PK_MISC, 0.0, ((double)planes * h * w + (double)planes * total) * 4.0);
  long long gx = (total + 255) / 256; if (gx > 1024) gx = 1024;
  STAR_LAUNCH(resize_pad_kernel, dim3((unsigned)gx, (unsigned)planes), dim3(256), (size_t)0, ctx->stream, p);
  return 0;
}

// Per-plane (mean, std) pairs.  Plane pl starts at element (pl / pa) * pb_stride + (pl % pa) * pa_stride with its n
// values es elements apart; its statistics land at index (pl % perm_a) * perm_b + pl / perm_a.
struct PlaneLayout { int pa; long long pa_stride, pb_stride, es; int perm_a, perm_b; };
static int plane_stats_impl(Ctx* ctx, const float* x, float* stats, int planes, long long n, float scale, float shift,
                            bool clamp01, bool div255, float eps, const PlaneLayout& L) {
  if (planes <= 0 || n <= 0) return ctx->fail("plane_stats: empty input");
  if (planes > 65535) return ctx->fail("plane_stats: too many planes");
  if (L.perm_a <= 0 || planes % L.perm_a != 0 || L.pa <= 0) return ctx->fail("plane_stats: bad plane layout");
  // <= 64 values per lane and slab, at least one slab
  long long nslab = (n + 256 * 64 - 1) / (256 * 64);
  if (nslab < 1) nslab = 1;
  if (nslab > 4096) nslab = 4096;
  Buf partial(ctx, (size_t)planes * nslab * 2 * sizeof(float));
  if (!partial.p) return ctx->fail("plane_stats: out of device memory");
  ProfScope ps(ctx, PK_MISC, 0.0, (double)planes * n * 4.0);
  PlaneStatsParams sp{x, partial.as<float>(), n, (int)nslab, scale, shift, clamp01 ? 1 : 0, L.pa, L.pa_stride, L.pb_stride, L.es,
                      div255 ? 1 : 0};
  STAR_LAUNCH(plane_stats_kernel, dim3((unsigned)nslab, (unsigned)planes), dim3(256), (size_t)64, ctx->stream, sp);
  PlaneStatsFinalParams fp{partial.as<float>(), stats, n, (int)nslab, eps, L.perm_a, L.perm_b};
  STAR_LAUNCH(plane_stats_final_kernel, dim3((unsigned)planes), dim3(64), (size_t)0, ctx->stream, fp);
  return 0;
}

int op_plane_stats(Ctx* ctx, const float* x, float* stats, int planes, long long n, float scale, float shift,
                   bool clamp01, float eps) {
  return plane_stats_impl(ctx, x, stats, planes, n, scale, shift, clamp01, false, eps, PlaneLayout{planes > 0 ? planes : 1, n, 0, 1, planes > 0 ? planes : 1, 1});
}

// from_model: x = [1, C, F, H, W] pipeline output (tensor2vid fused); else x = [F, H, W, C] in 0..255
int op_color_fix(Ctx* ctx, const float* x, bool from_model, const float* src, float* out, int F, int C, int H, int W, int h, int w, unsigned char* out_u8) {
  if (F <= 0 || C <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0) return ctx->fail("color_fix: empty input");
  if (F > 65535 || (long long)F * C > 65535) return ctx->fail("color_fix: too many frames");
  const long long HW = (long long)H * W;
  Buf stats(ctx, (size_t)F * C * 4 * sizeof(float));
  if (!stats.p) return ctx->fail("color_fix: out of device memory");
  float* content = stats.as<float>();
  float* style = content + (size_t)F * C * 2;
  if (from_model) {
    // planes of x are ordered [c][f]; the statistics are wanted as [f][c].  tensor2vid's clamp comes first.
    if (int rc = plane_stats_impl(ctx, x, content, C * F, HW, 0.5f, 0.5f, true, false, 1e-5f, PlaneLayout{C * F, HW, 0, 1, F, C})) return rc;
  } else {
    // plane (f, c) of the [F][HW][C] tensor: base f*HW*C + c, elements C apart; target / 255 (color_fix.py:17)
    if (int rc = plane_stats_impl(ctx, x, content, F * C, HW, 0.f, 0.f, false, true, 1e-5f, PlaneLayout{C, 1, HW * C, C, F * C, 1})) return rc;
  }
  // style: the low-resolution clip [F][C][h][w] in [-1, 1] -> (s + 1) / 2   (color_fix.py:18)
  if (int rc = op_plane_stats(ctx, src, style, F * C, (long long)h * w, 0.5f, 0.5f, false, 1e-5f)) return rc;
  ProfScope ps(ctx, PK_MISC, 0.0, (double)F * C * HW * (out_u8 ? 5.0 : 8.0));
  ColorFixParams p{x, out, content, style, C, F, HW, from_model ? (long long)F * HW : 1, from_model ? HW : HW * C, from_model ? 1 : C,
                   from_model ? 1 : 0, out_u8};
  long long gx = (HW + 255) / 256; if (gx > 2048) gx = 2048;
  STAR_LAUNCH(color_fix_kernel, dim3((unsigned)gx, (unsigned)F), dim3(256), (size_t)0, ctx->stream, p);
  return 0;
}

}  // namespace star
