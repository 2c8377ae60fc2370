// ops.h -- host-side launchers of the HIP kernels (one per hot-path operator).
// Every function enqueues on ctx->stream and returns 0 on success.
#pragma once
#include "ctx.h"
#include "optypes.h"

namespace star {

// RAII: brackets one kernel launch with HIP events when ctx->profiling is on
struct ProfScope {
  Ctx* ctx; int idx = -1;
  ProfScope(Ctx* c, int kind, double flops, double bytes, int d0 = 0, int d1 = 0, int d2 = 0, int d3 = 0) : ctx(c) {
    if (!c->profiling || !((c->prof_mask >> kind) & 1u)) return;
    ProfRec r{kind, flops, bytes, rt::event_record(c->stream), nullptr, d0, d1, d2, d3};
    c->prof.push_back(r);
    idx = (int)c->prof.size() - 1;
  }
  ~ProfScope() { if (idx >= 0) ctx->prof[idx].e1 = rt::event_record(ctx->stream); }
};

struct GemmArgs {
  const void* A = nullptr; const void* W = nullptr; void* C = nullptr;
  const float* bias = nullptr; const void* res = nullptr;
  int M = 0, N = 0, K = 0;
  int lda = 0, ldc = 0, ldr = 0;
  int mode = A_PLAIN;
  int H = 0, Wd = 0, Cin = 0, Ho = 0, Wo = 0, stride = 1, pad_t = 1, pad_l = 1;
  int HW = 0, F = 0;
  int up_crop = 1;
  int epi = 0;
  int force_tile = 0;  // 0 = auto; 1 = 256x256, 2 = 256x320, 3 = 128x128, 4 = 256x128 (tests)
  const float* rowab = nullptr; const float* colsum = nullptr;   // EPI_ROWAFF operands
  int group_m = -1;    // tile walk: -1 auto, 0 / 1 row-major, n column-major inside groups of n tile rows (gemm.h)
  int persist = 0;     // > 0: at most this many workgroups walk the output tiles (multiple of 8); 0 = one workgroup per tile
  int m_off = 0, m_end = 0;   // internal (tail split, gemm_impl.h): this launch covers output rows [m_off, m_end) of the M rows (m_end 0 = M)
  int assume_cus = 0;         // > 0: balance the tile rounds for this many compute units instead of the device's (tests)
  // GroupNorm statistics in this layer's epilogue (gemm.h EPIF bit 4): gn_partial[ceil(M / 32)][N / 2][2] fp32, written when the tile the
  // launcher picks has the flavour (256 x 320, 128 x 128 and the scheduled 256 x 256 tile; plain / 3x3 / temporal conv, 16-bit output,
  // bias (+ residual) epilogue) -- *gn_done says whether it was; otherwise the consumer runs its own statistics pass
  float* gn_partial = nullptr;
  bool* gn_done = nullptr;
  // LayerNorm row statistics in this layer's epilogue (gemm.h EPIF bit 5): ln_partial[M][parts][4] fp32 = (sum, sum of squares, max, 0) of
  // the stored outputs per row and column part; the launcher writes the number of parts (<= ln_parts_cap, which sized the buffer) to
  // *ln_parts and sets *ln_done when the launched tile has the flavour (256 x 320 and 128 x 128 tiles, plain-A layers, bias (+ residual)
  // 16-bit epilogue); otherwise the consumer runs its own pass over the rows
  float* ln_partial = nullptr;
  int ln_parts_cap = 0;
  int* ln_parts = nullptr;
  bool* ln_done = nullptr;
};
int op_gemm(Ctx* ctx, const GemmArgs& a);

}  // namespace star

namespace star {

struct AttnArgs {
  const void* Q = nullptr; const void* K = nullptr; const void* V = nullptr; void* O = nullptr;
  int ldq = 0, ldk = 0, ldv = 0, ldo = 0;
  long long bsq = 0, bsk = 0, bsv = 0, bso = 0;
  int Nq = 0, Nk = 0, heads = 0, batch = 0;
  float scale = 0.125f;
  int causal = 0;    // 1: key j visible to query i only for j <= i (Nq == Nk; the text tower)
  int variant = 9;   // 0 baseline, 1 v2, 2 v3, 3 v3 with 4 waves/SIMD, 4/5 software-pipelined, 6 v3 + lazy maxima, 7 lazy maxima probed per
                     // query block, 8 key-half pipeline, 9 = 6 with fp32-add row sums (default: fastest in the A/Bs, profiles/r01_attn_ab.txt)
};
int op_flash_attn(Ctx* ctx, const AttnArgs& a);

struct TAttnArgs {
  const void* Q = nullptr; const void* K = nullptr; const void* V = nullptr; void* O = nullptr;
  int ldq = 0, ldk = 0, ldv = 0, ldo = 0;
  int F = 0, HW = 0, heads = 0;
  float scale = 0.125f;
};
int op_temporal_attn(Ctx* ctx, const TAttnArgs& a);

// q | k | v projection (LayerNorm folded: rowab / colsum / bias) + attention over the frame axis in one kernel (gemm_tq.h): level-0 width
// (C = 320, 5 heads), F <= 32 frames.  W: [960][320] with its 64-row tiles ordered (q_h, k_h, v_h) per head; bias / colsum in that order.
struct TqArgs {
  const void* A = nullptr; const void* W = nullptr; void* O = nullptr;
  const float* bias = nullptr; const float* colsum = nullptr; const float* rowab = nullptr;
  int lda = 0, ldo = 0, HW = 0, F = 0, C = 320, heads = 5;
  float scale = 0.125f;
};
bool temporal_qkv_attn_covers(int C, int heads, int F);
int op_temporal_qkv_attn(Ctx* ctx, const TqArgs& a);

// GroupNorm(32 groups) over channels-last rows; rows_per_stat = H*W (per frame) or F*H*W (whole chunk)
int op_group_norm(Ctx* ctx, const void* x, int ldx, void* y, int ldy, const float* gamma, const float* beta,
                  int rows, int C, int rows_per_stat, float eps, bool silu);
// GroupNorm of a tensor whose PRODUCER wrote the partial statistics (GemmArgs::gn_partial): no statistics pass; y == nullptr: the
// affine pairs only (ab_out [nstat][C][2], mu_out optional [nstat][C])
int op_group_norm_fused(Ctx* ctx, const void* x, int ldx, void* y, int ldy, const float* gamma, const float* beta, int rows, int C,
                        int rows_per_stat, float eps, bool silu, const float* partial, float* ab_out = nullptr, float* mu_out = nullptr);
// statistics + finalize of a GroupNorm only: ab[nstat][C][2] = the per-channel affine pairs (y = x * a + b); nothing is applied;
// mu (optional) [nstat][C] = the channel's group mean
int op_group_norm_stats(Ctx* ctx, const void* x, int ldx, const float* gamma, const float* beta, int rows, int C, int rows_per_stat,
                        float eps, float* ab, float* mu = nullptr);
// a whole-chunk GroupNorm folded into the Linear behind it (norm.h): Wout[n][k] = round(W[n][k] * ab[k][0]),
// bias_out[n] = bias[n] + sum_k W[n][k] * ab[k][1] (+ the rounding residual of Wout times mu[k]: the mean is subtracted with the rounded weights)
int op_gn_fold_weights(Ctx* ctx, const void* W, const float* bias, const float* ab, void* Wout, float* bias_out, int N, int K,
                       const float* mu = nullptr);
int op_layer_norm(Ctx* ctx, const void* x, int ldx, void* y, int ldy, const float* gamma, const float* beta,
                  int rows, int C, float eps, int mode, const float* gate_w, float* maps, int H, int W, float* rowab = nullptr);
// gn_partial (optional): also write the output's GroupNorm partial statistics [ceil(rows / 32)][(C1 + C2) / 2][2] (GemmArgs::gn_partial layout)
// LayerNorm row coefficients (rowab) / LIEM maps from the row statistics the input's producer wrote (GemmArgs::ln_partial): same modes as
// op_layer_norm with rowab / STATS_ONLY outputs, 16 x parts bytes read per row
int op_layer_norm_from_partials(Ctx* ctx, const float* partial, int parts, int rows, int C, float eps, int mode, const float* gate_w,
                                float* maps, int H, int W, float* rowab);
int op_concat_add(Ctx* ctx, const void* a, const void* b, const void* c, void* out, int rows, int C1, int C2, float* gn_partial = nullptr);
int op_add(Ctx* ctx, const void* a, const void* b, void* out, long long n);
int op_stem_im2col(Ctx* ctx, const float* latent, void* out, int Cl, int F, int H, int W, bool frame_major = false);
int op_softmax_rows(Ctx* ctx, const float* s, int lds, void* pout, int ldp, int rows, int n, float scale);
// frames.cpp: full-resolution frame kernels either side of the diffusion path
int op_resize_pad(Ctx* ctx, const float* src, float* dst, int planes, int h, int w, int th, int tw,
                  int pad_l, int pad_r, int pad_t, int pad_b, float pad_value);
int op_plane_stats(Ctx* ctx, const float* x, float* stats, int planes, long long n, float scale, float shift,
                   bool clamp01, float eps);
int op_color_fix(Ctx* ctx, const float* x, bool from_model, const float* src, float* out, int F, int C, int H, int W, int h, int w,
                 unsigned char* out_u8 = nullptr);
int op_time_conv_out(Ctx* ctx, const float* rows, int ld, float* out, const float* w, const float* b, int F, int HW, int C);
int op_rows_to_latent(Ctx* ctx, const float* rows, float* out, int Cl, int ld, long long ntok);
int op_gemv(Ctx* ctx, const float* x, const void* W, const float* b, float* y, int N, int K, bool silu_in, bool silu_out);
int op_cast(Ctx* ctx, const float* x, void* y, long long n);
int op_vec_add_f32(Ctx* ctx, float* a, const float* b, int n);

}  // namespace star
