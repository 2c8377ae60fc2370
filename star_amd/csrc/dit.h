// dit.h -- the kernels a CogVideoX-5B DiT block needs beyond the UNet's GEMM / flash-attention core (SURVEY.md §8(f) rank 4;
// reference cogvideox-based/sat/dit_video_concat.py:482-563 AdaLNMixin.layer_forward, :254-346 3-D rotary, :571-598 QK
// LayerNorm, cogvideox-based/transformer.py:316-348 LIEM gates, :485-486).  All bandwidth-bound, channels-last token rows.
#pragma once
#include "prim.h"

namespace star {

STAR_DEV float dit_sigmoid(float x) { return fast_rcp(1.0f + fast_exp2(-1.4426950408889634f * x)); }

// ---------------------------------------------------------------------------------------------------------------------------
// adaLN LayerNorm (+ LIEM gates behind it).  y = ((x - mean) rstd gamma + beta) (1 + scale) + shift   (layer.input_layernorm /
// post_attention_layernorm followed by modulate(), dit_video_concat.py:518-521, 545-548, 349-350).  One wavefront per token row.
//   DLN_PLAIN      write y
//   DLN_STATS      write maps[row] = (max_c y, mean_c y) of the 16-bit-rounded y, no y          (first LIEM pass)
//   DLN_GATE       y <- g_t g_s y with g_s = sigmoid(conv7x7([max_c, mean_c]))(token) over the (H, W) grid of the token's frame
//                  (SpatialAttention, :523-527) and g_t = sigmoid(w0 max_c(g_s y) + w1 mean_c(g_s y)) (TemporalLocalAttention,
//                  :529-531; max and mean scale with g_s > 0, so both gates come from the same two statistics of y)
enum DitLnMode : int { DLN_PLAIN = 0, DLN_STATS = 1, DLN_GATE = 2 };
struct DitLnParams {
  const void* x; void* y; const float* gamma; const float* beta; const float* scale; const float* shift;   // fp32 [C]
  const float* w_spa;   // [2][7][7]
  const float* w_tmp;   // [2]
  float* maps;          // [rows][2]
  int ldx, ldy, C, rows, H, W; float eps; int mode;
};
template <class T>
STAR_GLOBAL void dit_ln_kernel(const DitLnParams p) {
  constexpr int CPL = 8;     // 16-B chunks per lane: C <= 64 * 8 * 8 = 4096
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const int CC8 = p.C >> 3;
  float v[CPL][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int cc = lane + 64 * i;
    if (cc < CC8) {
      const vec<T, 8> t = *reinterpret_cast<const vec<T, 8>*>((const T*)p.x + (size_t)row * p.ldx + cc * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[i][e] = to_f32<T>(t[e]); sum += v[i][e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  const float inv_c = 1.0f / (float)p.C;
  const float mean = wave_sum(sum) * inv_c;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int cc = lane + 64 * i;
    if (cc < CC8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; sq += d * d; }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_c + p.eps);
  float mx = -3.0e38f, ysum = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int cc = lane + 64 * i;
    if (cc < CC8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = cc * 8 + e;
        float y = (v[i][e] - mean) * rstd * p.gamma[c] + p.beta[c];
        y = y * (1.0f + p.scale[c]) + p.shift[c];
        y = to_f32<T>(from_f32<T>(y));      // the reference takes the LIEM statistics of the 16-bit tensor
        v[i][e] = y;
        mx = fmaxf(mx, y);
        ysum += y;
      }
    }
  }
  if (p.mode == DLN_STATS) {
    mx = wave_max(mx);
    ysum = wave_sum(ysum);
    if (lane == 0) { p.maps[2 * (size_t)row] = mx; p.maps[2 * (size_t)row + 1] = ysum * inv_c; }
    return;
  }
  float gate = 1.0f;
  if (p.mode == DLN_GATE) {
    const int hw = p.H * p.W;
    const int f = row / hw, rem = row - f * hw;
    const int yy0 = rem / p.W, xx0 = rem - yy0 * p.W;
    float acc = 0.f;
    for (int tap = lane; tap < 98; tap += 64) {
      const int ch = tap / 49, k = tap - ch * 49;
      const int yy = yy0 + k / 7 - 3, xx = xx0 + k % 7 - 3;
      if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) acc += p.w_spa[tap] * p.maps[2 * ((size_t)f * hw + (size_t)yy * p.W + xx) + ch];
    }
    const float gs = dit_sigmoid(wave_sum(acc));
    const float m0 = p.maps[2 * (size_t)row], m1 = p.maps[2 * (size_t)row + 1];
    const float gt = dit_sigmoid(p.w_tmp[0] * (gs * m0) + p.w_tmp[1] * (gs * m1));
    gate = gs * gt;
  }
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int cc = lane + 64 * i;
    if (cc < CC8) {
      vec<T, 8> o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(v[i][e] * gate);
      *reinterpret_cast<vec<T, 8>*>((T*)p.y + (size_t)row * p.ldy + cc * 8) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// per-head LayerNorm(64) of q and k (AdaLNMixin.attention_fn, :571-598) + 3-D rotary embedding of the video tokens
// (Rotary3DPositionEmbeddingMixin, :254-346: pairs (2i, 2i+1) rotate by the angle of channel 2i; 16 channels carry the frame
// index, 24 the row, 24 the column; text tokens are not rotated), in place on the fused [S][3 D] q|k|v rows.
// One wavefront per (token, head): lane = channel.
struct QkNormRopeParams {
  void* qkv; int ld, D, heads, S, text_len;
  const float* qg; const float* qb; const float* kg; const float* kb;   // [64]
  const float* cosb; const float* sinb;                                  // [S - text_len][64]
  float eps;
};
template <class T>
STAR_GLOBAL void qk_norm_rope_kernel(const QkNormRopeParams p) {
  const int lane = threadIdx.x & 63;
  const long long item = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (item >= (long long)p.S * p.heads) return;
  const int tok = (int)(item / p.heads), hd = (int)(item - (long long)tok * p.heads);
  T* q = (T*)p.qkv + (size_t)tok * p.ld + hd * 64 + lane;
  T* k = q + p.D;
  float c = 1.f, s = 0.f;
  if (tok >= p.text_len) { c = p.cosb[(size_t)(tok - p.text_len) * 64 + lane]; s = p.sinb[(size_t)(tok - p.text_len) * 64 + lane]; }
  auto one = [&](T* ptr, const float* g, const float* b) {
    const float x = to_f32<T>(*ptr);
    const float mean = wave_sum(x) * (1.0f / 64.0f);
    const float d = x - mean;
    const float rstd = 1.0f / sqrtf(wave_sum(d * d) * (1.0f / 64.0f) + p.eps);
    const float y = to_f32<T>(from_f32<T>(d * rstd * g[lane] + b[lane]));   // the reference rotates the 16-bit LayerNorm output
    const float other = shfl_xor(y, 1);
    const float rot = (lane & 1) ? other : -other;                           // rotate_half: (x0, x1) -> (-x1, x0)
    *ptr = from_f32<T>(y * c + rot * s);
  };
  one(q, p.qg, p.qb);
  one(k, p.kg, p.kb);
}

// out[m][:] = res[m][:] + gate[:] * d[m][:]   (gated residual of the adaLN block, :541-542, 561-562; fp32 gate vector)
struct GatedAddParams { const void* res; const void* d; void* out; const float* gate; int C; long long rows; };
template <class T>
STAR_GLOBAL void gated_add_kernel(const GatedAddParams p) {
  const int CC8 = p.C >> 3;
  const long long total = p.rows * CC8;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(q % CC8);
    const vec<T, 8> r = reinterpret_cast<const vec<T, 8>*>(p.res)[q], d = reinterpret_cast<const vec<T, 8>*>(p.d)[q];
    vec<T, 8> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(to_f32<T>(r[e]) + p.gate[cc * 8 + e] * to_f32<T>(d[e]));
    reinterpret_cast<vec<T, 8>*>(p.out)[q] = o;
  }
}

}  // namespace star
