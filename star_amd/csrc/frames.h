// frames.h -- full-resolution frame kernels either side of the diffusion path:
//   * bilinear resize to the target resolution + constant pad      (VideoToVideo_sr.test, video_to_video_model.py:81-87)
//   * tensor2vid + AdaIN colour fix against the low-resolution clip (inference_utils.py:16-23, color_fix.py:15-29,47-74)
// All three are pure HBM traffic on fp32 planes; every pixel is read/written exactly once per pass, 16 B per lane where
// the layout allows, no atomics (plane statistics are reduced slab -> plane in a fixed order, so they are reproducible).
#pragma once
#include "prim.h"

namespace star {

// ---- bilinear resize (align_corners = False, no antialias: torch F.interpolate(mode='bilinear')) + constant pad
// src: planes x [h][w] fp32; dst: planes x [th + pad_t + pad_b][tw + pad_l + pad_r] fp32
struct ResizePadParams {
  const float* src; float* dst;
  int planes, h, w, th, tw, pad_l, pad_t, oh, ow;
  float sy, sx, pad_value;
};
STAR_GLOBAL void resize_pad_kernel(const ResizePadParams p) {
  const int plane = blockIdx.y;
  const float* __restrict__ s = p.src + (size_t)plane * p.h * p.w;
  float* __restrict__ d = p.dst + (size_t)plane * p.oh * p.ow;
  const int total = p.oh * p.ow;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
    const int yo = q / p.ow, xo = q - yo * p.ow;
    const int y = yo - p.pad_t, x = xo - p.pad_l;
    float v = p.pad_value;
    if (y >= 0 && y < p.th && x >= 0 && x < p.tw) {
      // source index = (dst + 0.5) * (in / out) - 0.5, clamped at 0 (ATen area_pixel_compute_source_index)
      float fy = ((float)y + 0.5f) * p.sy - 0.5f; fy = fy < 0.f ? 0.f : fy;
      float fx = ((float)x + 0.5f) * p.sx - 0.5f; fx = fx < 0.f ? 0.f : fx;
      const int y0 = (int)fy, x0 = (int)fx;
      const int y1 = y0 + (y0 < p.h - 1 ? 1 : 0), x1 = x0 + (x0 < p.w - 1 ? 1 : 0);
      const float ly = fy - (float)y0, lx = fx - (float)x0;
      const float hy = 1.f - ly, hx = 1.f - lx;
      const float* r0 = s + (size_t)y0 * p.w;
      const float* r1 = s + (size_t)y1 * p.w;
      v = hy * (hx * r0[x0] + lx * r0[x1]) + ly * (hx * r1[x0] + lx * r1[x1]);
    }
    d[q] = v;
  }
}

// ---- per-plane mean / unbiased variance of  v = x * scale + shift  (optionally clamped to [0, 1]), or v = x / 255
// plane pl starts at element (pl / pa) * pb_stride + (pl % pa) * pa_stride, its n values are es elements apart
// pass 1: slab partial sums of (v - 0.5), (v - 0.5)^2 in fp32 (<= 64 values per lane, then a fixed-order tree);
// pass 2: one wave per plane adds the slab partials in fp64.
struct PlaneStatsParams {
  const float* x; float* partial;   // partial[plane][nslab][2]
  long long n; int nslab; float scale, shift; int clamp01;
  int pa; long long pa_stride, pb_stride, es; int div255;
};
STAR_GLOBAL void plane_stats_kernel(const PlaneStatsParams p) {
  float* red = reinterpret_cast<float*>(dyn_smem());   // [2][4]
  const int plane = blockIdx.y, slab = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* __restrict__ x = p.x + (size_t)(plane / p.pa) * p.pb_stride + (size_t)(plane % p.pa) * p.pa_stride;
  const long long per = (p.n + p.nslab - 1) / p.nslab;
  const long long lo = (long long)slab * per;
  long long hi = lo + per; if (hi > p.n) hi = p.n;
  float s1 = 0.f, s2 = 0.f;
  for (long long i = lo + t; i < hi; i += blockDim.x) {
    float v = x[i * p.es];
    v = p.div255 ? v / 255.0f : v * p.scale + p.shift;
    if (p.clamp01) v = (fminf(fmaxf(v, 0.f), 1.f) * 255.0f) / 255.0f;   // tensor2vid's clamp and its x255 / 255 round trip
    v -= 0.5f;
    s1 += v; s2 += v * v;
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane == 0) { red[wave] = s1; red[4 + wave] = s2; }
  block_sync();
  if (t == 0) {
    float a = 0.f, b = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { a += red[w]; b += red[4 + w]; }
    float* o = p.partial + ((size_t)plane * p.nslab + slab) * 2;
    o[0] = a; o[1] = b;
  }
}
struct PlaneStatsFinalParams { const float* partial; float* stats; long long n; int nslab; float eps; int perm_a, perm_b; };
// stats[(plane % perm_a) * perm_b + plane / perm_a] = (mean, sqrt(unbiased var + eps))   (color_fix.py:62-74 calc_mean_std)
STAR_GLOBAL void plane_stats_final_kernel(const PlaneStatsFinalParams p) {
  const int plane = blockIdx.x;
  if (threadIdx.x != 0) return;
  const float* q = p.partial + (size_t)plane * p.nslab * 2;
  double a = 0.0, b = 0.0;
  for (int s = 0; s < p.nslab; ++s) { a += (double)q[2 * s]; b += (double)q[2 * s + 1]; }
  const double n = (double)p.n;
  const double m = a / n;                                  // mean of (v - 0.5)
  double var = (b - n * m * m) / (n > 1.0 ? n - 1.0 : 1.0);
  if (var < 0.0) var = 0.0;
  const int o = (plane % p.perm_a) * p.perm_b + plane / p.perm_a;
  p.stats[2 * o] = (float)(m + 0.5);
  p.stats[2 * o + 1] = (float)__builtin_sqrt(var + (double)p.eps);
}

// ---- AdaIN apply -> out fp32 [F][H*W][C] in 0..255.  x(c, f, q) = x[c*sc + f*sf + q*sq]:
//   from_model = 1: x is the pipeline's [1, C, F, H, W] output in ~[-1, 1] and tensor2vid is applied on the fly;
//   from_model = 0: x is a tensor2vid result [F, H, W, C] in 0..255 (the reference's stand-alone adain_color_fix).
// content stats index: f*C + c ; style stats index: f*C + c   (both as (mean, std) pairs)
struct ColorFixParams {
  const float* x; float* out; const float* content; const float* style;
  int C, F; long long HW; long long sc, sf, sq; int from_model;
  unsigned char* out_u8;   // non-null: write [F][H*W][C] BYTES, truncated like the reference's `.astype('uint8')` in save_video
                           // (inference_utils.py:92) -- a quarter of the bytes leave the GPU and the host does no conversion pass
};
STAR_GLOBAL void color_fix_kernel(const ColorFixParams p) {
  const int f = blockIdx.y;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < p.HW; q += (long long)gridDim.x * blockDim.x) {
    for (int c = 0; c < p.C; ++c) {
      float v = p.x[(size_t)c * p.sc + (size_t)f * p.sf + (size_t)q * p.sq];
      if (p.from_model) {
        v = v * 0.5f + 0.5f;                                              // tensor2vid: mul_(std).add_(mean)
        v = fminf(fmaxf(v, 0.f), 1.f);                                    //             clamp_(0, 1)
        v = v * 255.0f;                                                   //             * 255
      }
      v = v / 255.0f;                                                     // adain_color_fix: target / 255
      const float cm = p.content[2 * (f * p.C + c)], cs = p.content[2 * (f * p.C + c) + 1];
      const float sm = p.style[2 * (f * p.C + c)], ss = p.style[2 * (f * p.C + c) + 1];
      float r = (v - cm) / cs * ss + sm;                                  // adaptive_instance_normalization
      r = fminf(fmaxf(r, 0.f), 1.f);
      if (p.out_u8) p.out_u8[((size_t)f * p.HW + q) * p.C + c] = (unsigned char)(r * 255.0f);
      else p.out[((size_t)f * p.HW + q) * p.C + c] = r * 255.0f;
    }
  }
}

}  // namespace star
