// norm.h -- bandwidth-bound kernels of the hot path: GroupNorm (4-D per-frame and
// 5-D whole-chunk statistics), LayerNorm with the LIEM gates fused in
// (K8, K9), SiLU, residual / concat plumbing (K10), the tiny time-embedding GEMVs
// (K11) and the latent layout conversions at the UNet boundary.
// All activations are channels-last [tokens, C]; loads/stores are 16 B per lane.
#pragma once
#include "prim.h"
#include "optypes.h"

namespace star {

STAR_DEV float silu_f(float x) { return x * fast_rcp(1.0f + fast_exp2(-1.4426950408889634f * x)); }
STAR_DEV float sigmoid_f(float x) { return fast_rcp(1.0f + fast_exp2(-1.4426950408889634f * x)); }

// ------------------------------------------------------------------ GroupNorm statistics
// reference: nn.GroupNorm(32, C) on (b f) c h w  [stats per frame]  unet_v2v.py:610,635,268
//            nn.GroupNorm(32, C) on  b c f h w   [stats over the whole chunk] unet_v2v.py:1210-1219,1002
struct GnStatsParams {
  const void* x; int ld; int C; int rows_per_stat; int slab; double* partial;  // partial[nstat][32][nslab][2]: a group's slab partials are contiguous
};
// partial sums of one (stat, group) -> the affine pairs of the group's channels: one wavefront, lanes stride over the slabs in
// order, then a fixed shuffle tree (shared by the folded and the separate finalize: bit-identical results)
STAR_DEV void gn_finalize_group(const double* partial, int nslab, int stat, int g, int C, double count, float eps, const float* gamma,
                                const float* beta, float* ab, int lane, float* mu = nullptr) {
  double a = 0.0, b = 0.0;
  const double* q0 = partial + ((size_t)stat * 32 + g) * (size_t)nslab * 2;
  for (int sl = lane; sl < nslab; sl += 64) {
    a += q0[2 * sl];
    b += q0[2 * sl + 1];
  }
  // fixed-order tree over the 64 lanes (two floats carry one double: hi/lo split keeps it exact enough and deterministic)
  for (int m = 32; m >= 1; m >>= 1) {
    const float ah = (float)a, al = (float)(a - (double)ah), bh = (float)b, bl = (float)(b - (double)bh);
    a += (double)shfl_xor(ah, m) + (double)shfl_xor(al, m);
    b += (double)shfl_xor(bh, m) + (double)shfl_xor(bl, m);
  }
  const double mean = a / count;
  double var = b / count - mean * mean;
  if (var < 0) var = 0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const int cg = C >> 5;
  for (int c = g * cg + lane; c < (g + 1) * cg; c += 64) {
    const float sc = gamma[c] * rstd;
    ab[2 * ((size_t)stat * C + c)] = sc;
    ab[2 * ((size_t)stat * C + c) + 1] = beta[c] - (float)mean * sc;
    if (mu) mu[(size_t)stat * C + c] = (float)mean;   // the group mean per channel, for the weight fold (gn_fold_weights_kernel)
  }
}
// Deterministic by construction (fixed reduction order, no atomics): every launch gives bit-identical statistics, hence a
// bit-reproducible forward.  Per block: threads own a fixed 8-channel chunk and stride over the slab's rows; the per-thread
// sums go through LDS and are reduced in index order by one thread per group.
template <class T>
STAR_GLOBAL void gn_stats_kernel(const GnStatsParams p) {
  float* ts = reinterpret_cast<float*>(dyn_smem());  // [nthreads][16]: s[8] | ss[8]
  const int t = threadIdx.x;
  const int CC8 = p.C >> 3;
  const int RL = blockDim.x / CC8;
  const int cc = t % CC8, rl = t / CC8;
  // The statistics pass walks the tensor BACKWARDS (the first workgroups take the last rows): the producer wrote x front to
  // back, so its tail is what the 256 MB Infinity Cache still holds, and the apply pass, which runs front to back, then finds
  // the head this pass read last.  Slab indices (and with them the reduction order and the results) are unchanged.
  const int stat = (int)gridDim.y - 1 - (int)blockIdx.y;
  const int slab_id = (int)gridDim.x - 1 - (int)blockIdx.x;
  const int r0 = slab_id * p.slab;
  int r1 = r0 + p.slab;
  if (r1 > p.rows_per_stat) r1 = p.rows_per_stat;
  float s[8], ss[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
  const T* __restrict__ base = (const T*)p.x + ((size_t)stat * p.rows_per_stat) * p.ld + cc * 8;
  if (rl < RL) {
    int r = r0 + rl;
    for (; r + 3 * RL < r1; r += 4 * RL) {   // four independent 16-B loads in flight per lane
      vec<T, 8> v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const vec<T, 8>*>(base + (size_t)(r + u * RL) * p.ld);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = to_f32<T>(v[u][e]); s[e] += f; ss[e] += f * f; }
    }
    for (; r < r1; r += RL) {
      const vec<T, 8> v = *reinterpret_cast<const vec<T, 8>*>(base + (size_t)r * p.ld);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float f = to_f32<T>(v[e]); s[e] += f; ss[e] += f * f; }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { ts[t * 16 + e] = s[e]; ts[t * 16 + 8 + e] = ss[e]; }
  block_sync();
  if (t < 32) {   // group t: channels [t*cg, (t+1)*cg), all row lanes, fixed order
    const int cg = p.C >> 5;
    double a = 0.0, b = 0.0;
    for (int c = t * cg; c < (t + 1) * cg; ++c) {
      const int ccx = c >> 3, e = c & 7;
      for (int l = 0; l < RL; ++l) {
        const float* q = ts + (l * CC8 + ccx) * 16;
        a += (double)q[e];
        b += (double)q[8 + e];
      }
    }
    double* out = p.partial + (((size_t)stat * 32 + t) * gridDim.x + slab_id) * 2;
    out[0] = a;
    out[1] = b;
  }
}

// partial sums -> per-(stat, channel) affine (a, b): y = x*a + b.  One wavefront per (stat, group): lanes stride over the
// slabs in order, then a fixed shuffle tree.
struct GnFinalizeParams {
  const double* partial; const float* gamma; const float* beta; float* ab;  // ab[nstat][C][2]
  int C; int nstat; int nslab; double count; float eps;
  float* mu;   // optional [nstat][C]: the channel's group mean
};
STAR_GLOBAL void gn_finalize_kernel(const GnFinalizeParams p) {
  const int lane = threadIdx.x & 63;
  const int wg = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // (stat, group) index
  if (wg >= p.nstat * 32) return;
  gn_finalize_group(p.partial, p.nslab, wg >> 5, wg & 31, p.C, p.count, p.eps, p.gamma, p.beta, p.ab, lane, p.mu);
}

// ------------------------------------------------------------------ GroupNorm finalize from the PRODUCER's partial statistics
// The GEMM-family kernel that wrote x also wrote, per 32-row slot and channel pair, (sum, sum of squares) of the stored values
// (gemm.h EPIF bit 4: gn_partial[ceil(rows / 32)][C / 2][2]) -- the statistics pass over x (a third of a GroupNorm's traffic) is
// gone.  One workgroup per (stat, group): the slots that lie wholly inside the stat's rows are summed from the partials, the at most
// 2 x 31 rows of the two slots a stat boundary cuts through (a frame of 26 352 rows is 823.5 slots) are read from x itself.
// Fixed assignment of items to threads and a fixed tree over the threads: bit-identical from launch to launch.
struct GnFinalizeFusedParams {
  const float* partial; const void* x; int ldx;
  const float* gamma; const float* beta; float* ab; float* mu;   // ab[nstat][C][2]; mu optional [nstat][C]
  int C; int nstat; int rows_per_stat; double count; float eps;
  // split > 1 (a whole-chunk norm: one stat of 26 352 slots at level 0 would otherwise be summed by 32 workgroups): workgroup
  // (stat, group, part) sums a contiguous 1 / split of the whole slots and writes its two doubles to part_out[(stat * 32 + group) * split + part]
  // (= the layout gn_finalize_kernel reduces: nslab = split); part 0 also takes the boundary rows
  int split; double* part_out;
};
template <class T>
STAR_GLOBAL void gn_finalize_fused_kernel(const GnFinalizeFusedParams p) {
  double* red = reinterpret_cast<double*>(dyn_smem());   // [blockDim.x][2]
  const int t = threadIdx.x, nt = blockDim.x;
  const int part = (int)(blockIdx.x % (unsigned)p.split), sg = (int)(blockIdx.x / (unsigned)p.split);
  const int stat = sg >> 5, g = sg & 31;
  const int cg = p.C >> 5, hp = cg >> 1;                  // channels / channel pairs per group
  const long long b0 = (long long)stat * p.rows_per_stat, b1 = b0 + p.rows_per_stat;
  const long long j0 = (b0 + 31) >> 5, j1 = b1 >> 5;      // whole slots [j0, j1)
  long long head_end = j0 << 5; if (head_end > b1) head_end = b1;
  long long tail_begin = j1 << 5; if (tail_begin < head_end) tail_begin = head_end;
  double a = 0.0, b = 0.0;
  // this part's share [ja, jb) of the whole slots
  long long ja = j0, jb = j1;
  if (p.split > 1 && j1 > j0) {
    const long long per = (j1 - j0 + p.split - 1) / p.split;
    ja = j0 + per * part; jb = ja + per;
    if (ja > j1) ja = j1;
    if (jb > j1) jb = j1;
  }
  if (jb > ja) {
    const long long items = (jb - ja) * hp;
    const float* base = p.partial + ((size_t)ja * (p.C >> 1) + (size_t)g * hp) * 2;
    for (long long it = t; it < items; it += nt) {
      const long long sl = it / hp; const int pr = (int)(it - sl * hp);
      const vec<float, 2> v = *reinterpret_cast<const vec<float, 2>*>(base + ((size_t)sl * (p.C >> 1) + pr) * 2);
      a += (double)v[0]; b += (double)v[1];
    }
  }
  if (part == 0) {   // boundary rows [b0, head_end) and [tail_begin, b1), straight from the tensor
    const long long nh = head_end - b0, ntl = b1 - tail_begin;
    const T* __restrict__ xg = (const T*)p.x + (size_t)g * cg;
    for (long long it = t; it < (nh + ntl) * cg; it += nt) {
      const long long r = it / cg; const int c = (int)(it - r * cg);
      const long long row = r < nh ? b0 + r : tail_begin + (r - nh);
      const float f = to_f32<T>(xg[(size_t)row * p.ldx + c]);
      a += (double)f; b += (double)f * (double)f;
    }
  }
  red[2 * t] = a; red[2 * t + 1] = b;
  block_sync();
  for (int st = 1; st < nt; st <<= 1) {   // fixed tree
    if ((t & (2 * st - 1)) == 0 && t + st < nt) { red[2 * t] += red[2 * (t + st)]; red[2 * t + 1] += red[2 * (t + st) + 1]; }
    block_sync();
  }
  if (p.split > 1) {   // the second stage (gn_finalize_kernel) reduces the parts in order
    if (t == 0) { p.part_out[2 * (size_t)blockIdx.x] = red[0]; p.part_out[2 * (size_t)blockIdx.x + 1] = red[1]; }
    return;
  }
  const double mean = red[0] / p.count;
  double var = red[1] / p.count - mean * mean;
  if (var < 0) var = 0;
  const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
  for (int c = g * cg + t; c < (g + 1) * cg; c += nt) {
    const float sc = p.gamma[c] * rstd;
    p.ab[2 * ((size_t)stat * p.C + c)] = sc;
    p.ab[2 * ((size_t)stat * p.C + c) + 1] = p.beta[c] - (float)mean * sc;
    if (p.mu) p.mu[(size_t)stat * p.C + c] = (float)mean;
  }
}

struct GnApplyParams {
  const void* x; void* y; const float* ab; int ldx, ldy, C; int rows_per_stat; int slab; int silu;
};
// grid (slab, stat), CC8 x RL threads like gn_stats_kernel: a thread owns one 8-channel chunk, keeps its 16 affine
// coefficients in registers and walks the rows of the slab (no index divisions, no coefficient re-loads in the loop)
// SILU: compile-time (round 6: the run-time flag cost one v_cndmask per element).  x and y may be the SAME tensor (in place: a thread
// reads a 16-byte chunk and writes the same chunk; Fwd::res_block normalises the tensors nobody else reads in place -- an in-place
// read-modify-write stream runs 6.1 TB/s on this part against 5.2 for a copy, profiles/r01_hbm_rw_probe.txt).
template <class T, bool SILU>
STAR_GLOBAL void gn_apply_kernel(const GnApplyParams p) {
  const int CC8 = p.C >> 3;
  const int t = threadIdx.x;
  const int RL = blockDim.x / CC8;
  const int cc = t % CC8, rl = t / CC8;
  if (rl >= RL) return;
  const int stat = blockIdx.y;
  const int r0 = blockIdx.x * p.slab;
  int r1 = r0 + p.slab;
  if (r1 > p.rows_per_stat) r1 = p.rows_per_stat;
  const float* ab = p.ab + 2 * ((size_t)stat * p.C + cc * 8);
  const f32x4 ab0 = *reinterpret_cast<const f32x4*>(ab), ab1 = *reinterpret_cast<const f32x4*>(ab + 4),
              ab2 = *reinterpret_cast<const f32x4*>(ab + 8), ab3 = *reinterpret_cast<const f32x4*>(ab + 12);
  const float av[8] = {ab0[0], ab0[2], ab1[0], ab1[2], ab2[0], ab2[2], ab3[0], ab3[2]};
  const float bv[8] = {ab0[1], ab0[3], ab1[1], ab1[3], ab2[1], ab2[3], ab3[1], ab3[3]};
  const T* xb = (const T*)p.x + ((size_t)stat * p.rows_per_stat) * p.ldx + cc * 8;
  T* yb = (T*)p.y + ((size_t)stat * p.rows_per_stat) * p.ldy + cc * 8;
  auto one = [&](int r, const vec<T, 8>& v) {
    vec<T, 8> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float f = to_f32<T>(v[e]) * av[e] + bv[e];
      if constexpr (SILU) f = silu_f(f);
      o[e] = from_f32<T>(f);
    }
    *reinterpret_cast<vec<T, 8>*>(yb + (size_t)r * p.ldy) = o;
  };
  int r = r0 + rl;
  for (; r + 3 * RL < r1; r += 4 * RL) {   // four independent 16-B loads in flight per lane
    vec<T, 8> v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const vec<T, 8>*>(xb + (size_t)(r + u * RL) * p.ldx);
#pragma unroll
    for (int u = 0; u < 4; ++u) one(r + u * RL, v[u]);
  }
  for (; r < r1; r += RL) one(r, *reinterpret_cast<const vec<T, 8>*>(xb + (size_t)r * p.ldx));
}

// ------------------------------------------------------------------ GroupNorm folded into the Linear behind it
// A whole-chunk (5-D) GroupNorm WITHOUT activation that feeds only a Linear / Conv1d(k=1) -- TemporalTransformer.norm -> proj_in
// (unet_v2v.py:1002-1005, 1052-1060) -- is an affine map per CHANNEL with one (a_c, b_c) for all rows (batch 1):
//   proj_in(GN(x))[m][n] = sum_c W[n][c] (a_c x[m][c] + b_c) + bias[n] = (x W'^T)[m][n] + bias'[n],
//   W'[n][c] = round_T(W[n][c] a_c),  bias'[n] = bias[n] + sum_c (W[n][c] b_c + (W[n][c] a_c - W'[n][c]) mu_c).
// The last term is what keeps the fold as accurate as the unfolded path when a group's |mean| is large against its spread:
// b_c = beta_c - a_c mu_c, so bias' = bias + sum_c W beta_c - sum_c W' mu_c -- the mean is subtracted with the ROUNDED weights the
// GEMM really multiplies x by, i.e. y = sum_c W'[n][c] (x_c - mu_c) + const: the rounding error of W' scales with |x - mu|, not with
// |x|  (without it groups at |mean| = 10-40 x spread were 5x worse than the unfolded pair: 6.4e-3 against 1.2e-3 in f16,
// tests/test_unet.py::test_group_norm_fold_with_large_group_means).
// The statistics pass stays; the apply pass (one read + one write of the activation) and the normalised tensor disappear: the
// projection reads x itself.  One block per output row n; ab / mu = the finalize kernel's per-channel pairs and group means.
struct GnFoldParams {
  const void* W; const float* bias; const float* ab; void* Wout; float* bias_out; int N, K;
  const float* mu;   // [K] group mean per channel (nullptr: the plain fold)
};
template <class T>
STAR_GLOBAL void gn_fold_weights_kernel(const GnFoldParams p) {
  float* red = reinterpret_cast<float*>(dyn_smem());   // [blockDim.x]
  const int n = blockIdx.x, t = threadIdx.x;
  const T* __restrict__ w = (const T*)p.W + (size_t)n * p.K;
  T* __restrict__ wo = (T*)p.Wout + (size_t)n * p.K;
  float acc = 0.f;
  for (int k = t; k < p.K; k += blockDim.x) {
    const float wv = to_f32<T>(w[k]);
    const float wa = wv * p.ab[2 * k];
    const T wr = from_f32<T>(wa);
    wo[k] = wr;
    acc += wv * p.ab[2 * k + 1];
    if (p.mu) acc += (wa - to_f32<T>(wr)) * p.mu[k];
  }
  red[t] = acc;
  block_sync();
  if (t == 0) {   // fixed order: reproducible
    float sum = 0.f;
    for (int i = 0; i < (int)blockDim.x; ++i) sum += red[i];
    p.bias_out[n] = sum + (p.bias ? p.bias[n] : 0.f);
  }
}

// ------------------------------------------------------------------ LayerNorm rows (+ LIEM gates)
// reference: BasicTransformerBlock.norm1/2/3 (unet_v2v.py:448-450), LIEM SpatialAttention (:380-394)
//            and TemporalLocalAttention (:396-411) applied in front of norm1 / norm2 (:466-490).
struct LnParams {
  const void* x; void* y; const float* gamma; const float* beta;
  const float* gate_w;   // LINEAR: [2]; MAP: [2][7][7]
  float* maps;           // [tokens][2] (written by STATS_ONLY, read by GATE_MAP)
  int ldx, ldy, C, rows; int H, W;  // H,W: image size for GATE_MAP (token = (f*H + y)*W + x)
  float eps; int mode; int lpr;
  float* rowab;          // non-null: do not write y; write rowab[token] = (a, b) with LN(gate x) = (a x + b) gamma + beta, for the
                         // GEMM that has this LayerNorm folded into its weights and epilogue (gemm.h EPI_ROWAFF)
};
// A row is shared by LPR = 8 / 16 / 32 / 64 lanes (<= 5 16-B chunks per lane), so a wavefront normalises 64/LPR rows at
// once and every lane is busy at every layer width (C = 320: 8 lanes x 5 chunks, 8 rows per wave; C = 2560: one row).
// CPL = 10 chunks per lane covers rows up to 5120 wide (the final LayerNorms of CogVideoX-5B at hidden 3072, modules/dit.py).
template <class T, int CPL = 5>
STAR_GLOBAL void ln_kernel(const LnParams p) {
  const int lane = threadIdx.x & 63;
  const int LPR = p.lpr;                       // lanes per row (power of two)
  const int sub = lane & (LPR - 1), rowi = lane / LPR, rpw = 64 / LPR;
  const int row = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * rpw + rowi;
  const bool active = row < p.rows;
  const int rr = active ? row : p.rows - 1;
  const int CC8 = p.C >> 3;
  auto group_sum = [&](float v) { for (int m = LPR >> 1; m >= 1; m >>= 1) v += shfl_xor(v, m); return v; };
  auto group_max = [&](float v) { for (int m = LPR >> 1; m >= 1; m >>= 1) v = fmaxf(v, shfl_xor(v, m)); return v; };
  float v[CPL][8];
  float mx = -3.0e38f, sum = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int cc = sub + LPR * i;
    if (cc < CC8) {
      const vec<T, 8> t = *reinterpret_cast<const vec<T, 8>*>((const T*)p.x + (size_t)rr * p.ldx + cc * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[i][e] = to_f32<T>(t[e]); mx = fmaxf(mx, v[i][e]); sum += v[i][e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  sum = group_sum(sum);
  const float inv_c = 1.0f / (float)p.C;
  float mean = sum * inv_c;
  if (p.mode != LN_PLAIN) mx = group_max(mx);
  if (p.mode == LN_STATS_ONLY) {
    if (active && sub == 0) { p.maps[2 * (size_t)row] = mx; p.maps[2 * (size_t)row + 1] = mean; }
    return;
  }
  float gate = 1.0f;
  if (p.mode == LN_GATE_LINEAR) {
    gate = sigmoid_f(p.gate_w[0] * mx + p.gate_w[1] * mean);
  } else if (p.mode == LN_GATE_MAP) {
    const int hw = p.H * p.W;
    const int f = rr / hw, rem = rr - f * hw;
    const int y = rem / p.W, x = rem - y * p.W;
    float acc = 0.f;
    for (int tap = sub; tap < 98; tap += LPR) {
      const int ch = tap / 49, k = tap - ch * 49;
      const int dy = k / 7 - 3, dx = k - (k / 7) * 7 - 3;
      const int yy = y + dy, xx = x + dx;
      if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W)
        acc += p.gate_w[tap] * p.maps[2 * ((size_t)f * hw + (size_t)yy * p.W + xx) + ch];
    }
    gate = sigmoid_f(group_sum(acc));
  }
  if (p.mode != LN_PLAIN) {
    mean *= gate;
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] *= gate;
  }
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int cc = sub + LPR * i;
    if (cc < CC8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; sq += d * d; }
    }
  }
  sq = group_sum(sq);
  const float rstd = 1.0f / sqrtf(sq * inv_c + p.eps);
  if (!active) return;
  if (p.rowab) {     // (gate x - gate mean) rstd = a x + b
    if (sub == 0) { p.rowab[2 * (size_t)row] = gate * rstd; p.rowab[2 * (size_t)row + 1] = -rstd * mean; }
    return;
  }
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int cc = sub + LPR * i;
    if (cc < CC8) {
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gamma + cc * 8), g1 = *reinterpret_cast<const f32x4*>(p.gamma + cc * 8 + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.beta + cc * 8), b1 = *reinterpret_cast<const f32x4*>(p.beta + cc * 8 + 4);
      vec<T, 8> o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = from_f32<T>((v[i][e] - mean) * rstd * g0[e] + b0[e]);
        o[4 + e] = from_f32<T>((v[i][4 + e] - mean) * rstd * g1[e] + b1[e]);
      }
      *reinterpret_cast<vec<T, 8>*>((T*)p.y + (size_t)row * p.ldy + cc * 8) = o;
    }
  }
}

// ------------------------------------------------------------------ LayerNorm row coefficients from the PRODUCER's row statistics
// The GEMM that wrote x also wrote, per row and column part, (sum, sum of squares, max) of the stored values (gemm.h EPIF bit 5): the
// LayerNorms that are folded into the next projection (rowab: LN(gate x) = (a x + b) gamma + beta) and the LIEM maps need nothing
// else of x -- 16 x parts bytes per row instead of the row itself (2 parts at C = 320: 32 of 640 bytes).  Same modes and outputs as
// ln_kernel: LN_STATS_ONLY -> maps[token] = (max_c, mean_c); LN_PLAIN / LN_GATE_LINEAR / LN_GATE_MAP -> rowab[token] = (a, b).
// The variance is E[x^2] - mean^2 from fp32 sums of exact 16-bit squares, finished in double.  Eight lanes per row (the 7x7 taps of
// the map gate are shared among them as in ln_kernel).
struct LnPartParams {
  const float* partial; int parts;
  const float* gate_w; float* maps; float* rowab;
  int C, rows, H, W; float eps; int mode;
};
STAR_GLOBAL void ln_from_partials_kernel(const LnPartParams p) {
  constexpr int LPR = 8;
  const int lane = threadIdx.x & 63;
  const int sub = lane & (LPR - 1), rowi = lane / LPR, rpw = 64 / LPR;
  const int row = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * rpw + rowi;
  const bool active = row < p.rows;
  const int rr = active ? row : p.rows - 1;
  auto group_sum = [&](float v) { for (int m = LPR >> 1; m >= 1; m >>= 1) v += shfl_xor(v, m); return v; };
  auto group_max = [&](float v) { for (int m = LPR >> 1; m >= 1; m >>= 1) v = fmaxf(v, shfl_xor(v, m)); return v; };
  float s = 0.f, q = 0.f, mx = -3.0e38f;
  for (int pt = sub; pt < p.parts; pt += LPR) {
    const f32x4 rec = *reinterpret_cast<const f32x4*>(p.partial + ((size_t)rr * p.parts + pt) * 4);
    s += rec[0]; q += rec[1]; mx = fmaxf(mx, rec[2]);
  }
  s = group_sum(s); q = group_sum(q); mx = group_max(mx);
  const float inv_c = 1.0f / (float)p.C;
  float mean = s * inv_c;
  if (p.mode == LN_STATS_ONLY) {
    if (active && sub == 0) { p.maps[2 * (size_t)row] = mx; p.maps[2 * (size_t)row + 1] = mean; }
    return;
  }
  double var = (double)q * (double)inv_c - (double)mean * (double)mean;
  if (var < 0.0) var = 0.0;
  float gate = 1.0f;
  if (p.mode == LN_GATE_LINEAR) {
    gate = sigmoid_f(p.gate_w[0] * mx + p.gate_w[1] * mean);
  } else if (p.mode == LN_GATE_MAP) {
    const int hw = p.H * p.W;
    const int f = rr / hw, rem = rr - f * hw;
    const int y = rem / p.W, x = rem - y * p.W;
    float acc = 0.f;
    for (int tap = sub; tap < 98; tap += LPR) {
      const int ch = tap / 49, k = tap - ch * 49;
      const int dy = k / 7 - 3, dx = k - (k / 7) * 7 - 3;
      const int yy = y + dy, xx = x + dx;
      if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W)
        acc += p.gate_w[tap] * p.maps[2 * ((size_t)f * hw + (size_t)yy * p.W + xx) + ch];
    }
    gate = sigmoid_f(group_sum(acc));
  }
  // the gated row g x has mean g mean and variance g^2 var
  const float gm = gate * mean;
  const float rstd = (float)(1.0 / sqrt((double)gate * (double)gate * var + (double)p.eps));
  if (active && sub == 0) { p.rowab[2 * (size_t)row] = gate * rstd; p.rowab[2 * (size_t)row + 1] = -rstd * gm; }
}

// ------------------------------------------------------------------ elementwise plumbing
// out[row] = concat(a[row][0:C1], b[row][0:C2] (+ c[row][0:C2]))   (unet_v2v.py:1792 torch.cat([x, xs.pop() + control.pop()]))
struct ConcatParams { const void* a; const void* b; const void* c; void* out; int C1, C2, rows; };
template <class T>
STAR_GLOBAL void concat_add_kernel(const ConcatParams p) {
  const int CT8 = (p.C1 + p.C2) >> 3, C18 = p.C1 >> 3;
  const long long total = (long long)p.rows * CT8;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(q / CT8), cc = (int)(q - (long long)row * CT8);
    vec<T, 8> o;
    if (cc < C18) {
      o = *reinterpret_cast<const vec<T, 8>*>((const T*)p.a + (size_t)row * p.C1 + cc * 8);
    } else {
      const size_t off = (size_t)row * p.C2 + (cc - C18) * 8;
      o = *reinterpret_cast<const vec<T, 8>*>((const T*)p.b + off);
      if (p.c) {
        const vec<T, 8> c = *reinterpret_cast<const vec<T, 8>*>((const T*)p.c + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(to_f32<T>(o[e]) + to_f32<T>(c[e]));
      }
    }
    *reinterpret_cast<vec<T, 8>*>((T*)p.out + (size_t)row * (p.C1 + p.C2) + cc * 8) = o;
  }
}

// The same concat, ALSO writing the GroupNorm partial statistics of its output (the decoder's in_layers.0 reads it next,
// unet_v2v.py:1792,609-612) in the layout of the GEMM epilogues (gemm.h EPIF bit 4: [ceil(rows / 32)][C / 2][2], sum and sum of squares
// of the stored values per 32-row slot and channel pair).  One workgroup per slot: thread (rl, cc) owns the 8-channel chunk column cc and
// walks the rows rl, rl + RL, ...; the RL row lanes of a column meet in LDS and are added in order (deterministic).
struct ConcatStatsParams { const void* a; const void* b; const void* c; void* out; int C1, C2, rows; float* partial; };
template <class T>
STAR_GLOBAL void concat_add_stats_kernel(const ConcatStatsParams p) {
  float* red = reinterpret_cast<float*>(dyn_smem());   // [RL][CT8][8]
  const int CT = p.C1 + p.C2, CT8 = CT >> 3, C18 = p.C1 >> 3;
  const int t = threadIdx.x;
  const int RL = blockDim.x / CT8;
  const int cc = t % CT8, rl = t / CT8;
  const int r0 = blockIdx.x * 32;
  int r1 = r0 + 32;
  if (r1 > p.rows) r1 = p.rows;
  float ps[4], pq[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { ps[e] = 0.f; pq[e] = 0.f; }
  if (rl < RL) {
    for (int row = r0 + rl; row < r1; row += RL) {
      vec<T, 8> o;
      if (cc < C18) {
        o = *reinterpret_cast<const vec<T, 8>*>((const T*)p.a + (size_t)row * p.C1 + cc * 8);
      } else {
        const size_t off = (size_t)row * p.C2 + (cc - C18) * 8;
        o = *reinterpret_cast<const vec<T, 8>*>((const T*)p.b + off);
        if (p.c) {
          const vec<T, 8> c = *reinterpret_cast<const vec<T, 8>*>((const T*)p.c + off);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(to_f32<T>(o[e]) + to_f32<T>(c[e]));
        }
      }
      *reinterpret_cast<vec<T, 8>*>((T*)p.out + (size_t)row * CT + cc * 8) = o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        vec<T, 2> pr;
        pr[0] = o[2 * e]; pr[1] = o[2 * e + 1];
        ps[e] = dot2_one<T>(pr, ps[e]);
        pq[e] = dot2_acc<T>(pr, pr, pq[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[(rl * CT8 + cc) * 8 + 2 * e] = ps[e]; red[(rl * CT8 + cc) * 8 + 2 * e + 1] = pq[e]; }
  }
  block_sync();
  if (rl == 0) {
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < RL; ++l) {
      const float* q = red + (l * CT8 + cc) * 8;
      lo += *reinterpret_cast<const f32x4*>(q);
      hi += *reinterpret_cast<const f32x4*>(q + 4);
    }
    float* dst = p.partial + (size_t)blockIdx.x * CT + cc * 8;
    *reinterpret_cast<f32x4*>(dst) = lo;
    *reinterpret_cast<f32x4*>(dst + 4) = hi;
  }
}

// out = a + b (same shape, n multiple of 8)
struct AddParams { const void* a; const void* b; void* out; long long n8; };
template <class T>
STAR_GLOBAL void add_kernel(const AddParams p) {
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < p.n8; q += (long long)gridDim.x * blockDim.x) {
    const vec<T, 8> a = reinterpret_cast<const vec<T, 8>*>(p.a)[q], b = reinterpret_cast<const vec<T, 8>*>(p.b)[q];
    vec<T, 8> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(to_f32<T>(a[e]) + to_f32<T>(b[e]));
    reinterpret_cast<vec<T, 8>*>(p.out)[q] = o;
  }
}

// latent [1, Cl, F, H, W] (fp32) -> im2col rows [F*H*W][64] of the 3x3 pad-1 stem conv: column tap*Cl + c, rest 0
// (unet_v2v.py:1353 nn.Conv2d(in_dim, dim, 3, padding=1) and :2128 input_hint_block)
struct StemIm2colParams { const float* x; void* out; int Cl, F, H, W; long long fs, cs; };  // x[f*fs + c*cs + y*W + x]
template <class T>
STAR_GLOBAL void stem_im2col_kernel(const StemIm2colParams p) {
  const long long rows = (long long)p.F * p.H * p.W;
  const long long total = rows * 8;  // 8 chunks of 8 columns
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
    const long long row = q >> 3; const int ch = (int)(q & 7);
    const int hw = p.H * p.W;
    const int f = (int)(row / hw), rem = (int)(row - (long long)f * hw);
    const int y = rem / p.W, x = rem - y * p.W;
    vec<T, 8> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int col = ch * 8 + e;
      float v = 0.f;
      if (col < 9 * p.Cl) {
        const int tap = col / p.Cl, c = col - tap * p.Cl;
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) v = p.x[(size_t)f * p.fs + (size_t)c * p.cs + (size_t)yy * p.W + xx];
      }
      o[e] = from_f32<T>(v);
    }
    *reinterpret_cast<vec<T, 8>*>((T*)p.out + row * 64 + ch * 8) = o;
  }
}

// rows [F*H*W][ld] fp32 (first Cl columns) -> latent [1, Cl, F, H, W] fp32
struct RowsToLatentParams { const float* rows; float* out; int Cl, ld; long long ntok; };
STAR_GLOBAL void rows_to_latent_kernel(const RowsToLatentParams p) {
  const long long total = p.ntok * p.Cl;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(q / p.ntok); const long long tok = q - (long long)c * p.ntok;
    p.out[q] = p.rows[tok * p.ld + c];
  }
}

// y[n] = act_out( sum_k W[n][k] * act_in(x[k]) + b[n] ), fp32 vectors, T weights: one wavefront per output
// (time_embed MLP unet_v2v.py:1340-1342 and ResBlock.emb_layers :626-633; M = 1 because batch = 1)
struct GemvParams { const float* x; const void* W; const float* b; float* y; int N, K; int silu_in, silu_out; };
template <class T>
STAR_GLOBAL void gemv_kernel(const GemvParams p) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (n >= p.N) return;
  const T* __restrict__ w = (const T*)p.W + (size_t)n * p.K;
  float acc = 0.f;
  for (int k = lane * 8; k < p.K; k += 512) {
    const vec<T, 8> wv = *reinterpret_cast<const vec<T, 8>*>(w + k);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float xv = p.x[k + e];
      if (p.silu_in) xv = silu_f(xv);
      acc += to_f32<T>(wv[e]) * xv;
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    float r = acc + (p.b ? p.b[n] : 0.f);
    if (p.silu_out) r = silu_f(r);
    p.y[n] = r;
  }
}

// ------------------------------------------------------------------ VAE-only kernels
// row softmax of fp32 logits (scaled) -> T probabilities, zero-filling the padded tail [n, ld)
// (SVD VAE mid-block attention: one head of 512 channels over all H*W tokens; logits are materialised in fp32)
struct SoftmaxParams { const float* s; void* p; int rows, n, lds, ldp; float scale_log2e; };
template <class T>
STAR_GLOBAL void softmax_rows_kernel(const SoftmaxParams p) {
  float* red = reinterpret_cast<float*>(dyn_smem());  // [8]
  const int row = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = blockDim.x >> 6;
  const float* __restrict__ s = p.s + (size_t)row * p.lds;
  float mx = -3.0e38f;
  for (int i = t; i < p.n; i += blockDim.x) mx = fmaxf(mx, s[i]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  block_sync();
  mx = red[0];
  for (int w = 1; w < nw; ++w) mx = fmaxf(mx, red[w]);
  block_sync();
  const float c = p.scale_log2e, mc = mx * c;
  float sum = 0.f;
  for (int i = t; i < p.n; i += blockDim.x) sum += fast_exp2(s[i] * c - mc);
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  block_sync();
  sum = 0.f;
  for (int w = 0; w < nw; ++w) sum += red[w];
  const float inv = 1.0f / sum;
  T* __restrict__ o = (T*)p.p + (size_t)row * p.ldp;
  for (int i = t; i < p.ldp; i += blockDim.x) o[i] = from_f32<T>(i < p.n ? fast_exp2(s[i] * c - mc) * inv : 0.f);
}

// The same rows with ONE read of the logits (round 6; the kernel above reads a row three times with 4-byte loads: 1.8 TB/s effective on the
// VAE's 26352-wide rows, 2.3 ms per frame and attention).  256 threads own 8-element chunks (two 16-byte loads, one 16-byte store each):
// rows of up to 256 x V8 chunks stay in registers between the maximum, the sum and the store; longer rows (the 133 712-token frames of the
// 2160p configuration) take an online maximum / sum in the first pass and are read a second time for the store.  Needs lds % 4 == 0,
// ldp % 8 == 0 and 16-byte aligned bases (the launcher checks; anything else runs the kernel above).
template <class T, int V8>
STAR_GLOBAL void STAR_LAUNCH_BOUNDS(256, 1) softmax_rows_vec_kernel(const SoftmaxParams p) {
  float* red = reinterpret_cast<float*>(dyn_smem());  // [8]
  const int row = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* __restrict__ s = p.s + (size_t)row * p.lds;
  T* __restrict__ o = (T*)p.p + (size_t)row * p.ldp;
  const int nch = p.ldp >> 3;
  const float c = p.scale_log2e;
  constexpr float NEG = -3.0e38f;
  auto load8 = [&](int ch, float (&v)[8]) STAR_ALWAYS_INLINE {
    const f32x4 a = *reinterpret_cast<const f32x4*>(s + (size_t)ch * 8), b = *reinterpret_cast<const f32x4*>(s + (size_t)ch * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
    if (ch * 8 + 8 > p.n) {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (ch * 8 + e >= p.n) v[e] = NEG;
    }
  };
  auto block_max = [&](float m) STAR_ALWAYS_INLINE {
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    block_sync();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    block_sync();
    return m;
  };
  auto block_sum = [&](float x) STAR_ALWAYS_INLINE {
    x = wave_sum(x);
    if (lane == 0) red[wave] = x;
    block_sync();
    x = (red[0] + red[1]) + (red[2] + red[3]);
    block_sync();
    return x;
  };
  auto store8 = [&](int ch, const float (&v)[8]) STAR_ALWAYS_INLINE {
    vec<T, 8> w;
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = from_f32<T>(v[e]);
    *reinterpret_cast<vec<T, 8>*>(o + (size_t)ch * 8) = w;
  };
  if (nch <= 256 * V8) {                       // the row lives in registers
    float v[V8][8];
    float mx = NEG;
#pragma unroll
    for (int j = 0; j < V8; ++j) {
      const int ch = t + j * 256;
      if (ch < nch) load8(ch, v[j]);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e] = NEG;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) mx = fmaxf(mx, v[j][e]);
    }
    mx = block_max(mx);
    const float mc = mx * c;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < V8; ++j) {
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[j][e] = fast_exp2(v[j][e] * c - mc); q += v[j][e]; }
      sum += q;
    }
    const float inv = 1.0f / block_sum(sum);
#pragma unroll
    for (int j = 0; j < V8; ++j) {
      const int ch = t + j * 256;
      if (ch < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e] *= inv;
        store8(ch, v[j]);
      }
    }
    return;
  }
  // long rows: online maximum and sum in one pass, then a second read for the store
  float m = NEG, sum = 0.f;
  for (int ch = t; ch < nch; ch += 256) {
    float v[8];
    load8(ch, v);
    float cm = v[0];
#pragma unroll
    for (int e = 1; e < 8; ++e) cm = fmaxf(cm, v[e]);
    if (!(cm > NEG)) continue;                                    // a chunk of padding columns only
    if (cm > m) { sum *= fast_exp2((m - cm) * c); m = cm; }      // (first chunk: sum is 0)
    const float mc = m * c;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) q += fast_exp2(v[e] * c - mc);
    sum += q;
  }
  const float M = block_max(m);
  const float Mc = M * c;
  const float inv = 1.0f / block_sum(m > NEG ? sum * fast_exp2((m - M) * c) : 0.f);
  for (int ch = t; ch < nch; ch += 256) {
    float v[8];
    load8(ch, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fast_exp2(v[e] * c - Mc) * inv;
    store8(ch, v);
  }
}

// time_conv_out (Conv3d 3->3, kernel (3,1,1), zero pad over frames) + rows -> NCHW:
// rows[f*HW + p][ld] fp32 -> out[f][c][p] fp32  (diffusers TemporalDecoder.time_conv_out)
struct TimeConvOutParams { const float* rows; float* out; const float* w; const float* b; int F, HW, ld, C; };
STAR_GLOBAL void time_conv_out_kernel(const TimeConvOutParams p) {
  const long long total = (long long)p.F * p.HW;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
    const int f = (int)(q / p.HW); const int px = (int)(q - (long long)f * p.HW);
    for (int co = 0; co < p.C; ++co) {
      float acc = p.b[co];
      for (int tap = 0; tap < 3; ++tap) {
        const int ff = f + tap - 1;
        if (ff < 0 || ff >= p.F) continue;
        const float* r = p.rows + ((size_t)ff * p.HW + px) * p.ld;
        for (int ci = 0; ci < p.C; ++ci) acc += p.w[(co * p.C + ci) * 3 + tap] * r[ci];
      }
      p.out[((size_t)f * p.C + co) * p.HW + px] = acc;
    }
  }
}

// a[i] += b[i] (fp32, small vectors: conv bias + time-embedding projection)
struct VecAddParams { float* a; const float* b; int n; };
STAR_GLOBAL void vec_add_f32_kernel(const VecAddParams p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < p.n) p.a[i] += p.b[i];
}

// fp32 -> T conversion of a flat array (n multiple of 8), e.g. the text context y[77,1024]
struct CastParams { const float* x; void* y; long long n8; };
template <class T>
STAR_GLOBAL void cast_kernel(const CastParams p) {
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < p.n8; q += (long long)gridDim.x * blockDim.x) {
    vec<T, 8> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(p.x[q * 8 + e]);
    reinterpret_cast<vec<T, 8>*>(p.y)[q] = o;
  }
}

}  // namespace star
