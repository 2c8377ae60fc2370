// dit.cpp -- one CogVideoX-5B DiT block (STAR's CogVideoX variant, BASELINE config #5) on the UNet's
// kernels: AdaLNMixin.layer_forward (cogvideox-based/sat/dit_video_concat.py:482-563) with the LIEM gates of
// cogvideox-based/transformer.py:316-348,485-486, 3-D rotary embedding (:254-346) and QK LayerNorm (:571-598).  The dense
// layers, LayerNorms, attention core and MLP the mixin calls live in the un-vendored SwissArmyTransformer==0.4.12
// (sat/requirements.txt:1); their arithmetic is restated from the published package (oracle/dit_oracle.py, parity unpinned).
//
// Tokens are rows [S = text_len + T*H*W, D] in the storage dtype for the whole block; the fused QKV projection, the d = 64
// flash-attention kernel (48 heads x 9676 tokens at full size), the bias / tanh-GELU GEMM epilogues are the UNet's.
#include "dit.h"
#include "graph.h"
#include <cmath>
#include <mutex>
#include <unordered_map>

namespace star {

struct DitLayerW {
  NormW ln1, ln2;            // input_layernorm / post_attention_layernorm (affine)
  LinW qkv, dense, fc1, fc2;
  LinW ada;                  // adaLN_modulations[i][1]: [12 D][E]
  NormW qn, kn;              // query / key LayerNorm(64)
  DevW w_spa, w_tmp;         // LIEM gates: [1,2,7,7] and [1,2]
};
struct DitModel {
  int D = 0, heads = 0, E = 0, n_layers = 0;
  float ln_eps = 1e-5f;
  std::vector<DitLayerW> layers;
  std::vector<void*> owned;
  // rotary tables for the last (T, H, W) seen
  int rt_T = 0, rt_H = 0, rt_W = 0;
  void* cosb = nullptr; void* sinb = nullptr;
  ~DitModel() { for (void* p : owned) rt::dev_free(p); if (cosb) rt::dev_free(cosb); if (sinb) rt::dev_free(sinb); }
};

// DiT models by context.  Contexts may be created and destroyed on different threads: the registry is guarded, lookups return a
// shared_ptr copy (a forward keeps its model alive), and dit_release ERASES the key (it used to leave an empty entry behind for every
// context ever destroyed, UNet and VAE contexts included).
static std::mutex& dit_mutex() { static std::mutex m; return m; }
static std::unordered_map<Ctx*, std::shared_ptr<DitModel>>& dit_models() {
  static std::unordered_map<Ctx*, std::shared_ptr<DitModel>> models;
  return models;
}
static std::shared_ptr<DitModel> dit_of(Ctx* ctx) {
  std::lock_guard<std::mutex> g(dit_mutex());
  auto it = dit_models().find(ctx);
  return it == dit_models().end() ? nullptr : it->second;
}
static void dit_set(Ctx* ctx, std::shared_ptr<DitModel> m) {
  std::lock_guard<std::mutex> g(dit_mutex());
  dit_models()[ctx] = std::move(m);
}
void dit_release(Ctx* ctx) {
  std::lock_guard<std::mutex> g(dit_mutex());
  dit_models().erase(ctx);
}

int dit_build(Ctx* ctx, int D, int heads, int E, int n_layers, float ln_eps) {
  if (D % 64 || heads * 64 != D) return ctx->fail("dit_build: hidden size must be heads x 64");
  if (D > 4096) return ctx->fail("dit_build: hidden size above 4096 is not supported");
  if (E % 8) return ctx->fail("dit_build: time_embed_dim must be a multiple of 8");
  auto m = std::make_shared<DitModel>();
  m->D = D; m->heads = heads; m->E = E; m->n_layers = n_layers; m->ln_eps = ln_eps;
  Builder b{ctx, &m->owned, ""};
  for (int i = 0; i < n_layers; ++i) {
    const std::string L = "transformer.layers." + std::to_string(i) + ".";
    const std::string A = "mixins.adaln_layer.";
    DitLayerW w;
    w.ln1 = b.norm(L + "input_layernorm");
    w.ln2 = b.norm(L + "post_attention_layernorm");
    w.qkv = b.linear(L + "attention.query_key_value");
    w.dense = b.linear(L + "attention.dense");
    w.fc1 = b.linear(L + "mlp.dense_h_to_4h");
    w.fc2 = b.linear(L + "mlp.dense_4h_to_h");
    w.ada = b.linear(A + "adaLN_modulations." + std::to_string(i) + ".1");
    w.qn = b.norm(A + "query_layernorm_list." + std::to_string(i));
    w.kn = b.norm(A + "key_layernorm_list." + std::to_string(i));
    w.w_spa = b.raw_f32(L + "spa_local.conv1.weight");
    w.w_tmp = b.raw_f32(L + "temp_local.conv1.weight");
    if (!b.err.empty()) return ctx->fail("dit_build: " + b.err);
    if (w.qkv.N != 3 * D || w.qkv.K != D || w.ada.N != 12 * D || w.ada.K != E || w.qn.C != 64 || w.kn.C != 64)
      return ctx->fail("dit_build: tensor shapes do not match the configuration");
    m->layers.push_back(w);
  }
  dit_set(ctx, m);
  ctx->host_tensors.clear();
  return 0;
}

// cos / sin tables of Rotary3DPositionEmbeddingMixin.__init__ (dit_video_concat.py:267-295): per video token (t, h, w) the
// 64 angles [16: frame | 24: row | 24: column], each frequency repeated for the pair (2i, 2i + 1)
static int rotary_tables(Ctx* ctx, DitModel& m, int T, int H, int W) {
  if (m.rt_T == T && m.rt_H == H && m.rt_W == W && m.cosb) return 0;
  const int n = T * H * W;
  std::vector<float> c((size_t)n * 64), s((size_t)n * 64);
  const int dim_t = 16, dim_h = 24, dim_w = 24;
  auto freq = [](int i2, int dim) { return 1.0f / powf(10000.0f, (float)i2 / (float)dim); };
  for (int t = 0; t < T; ++t)
    for (int h = 0; h < H; ++h)
      for (int w = 0; w < W; ++w) {
        float* cc = &c[((size_t)(t * H + h) * W + w) * 64];
        float* ss = &s[((size_t)(t * H + h) * W + w) * 64];
        int o = 0;
        for (int i = 0; i < dim_t; ++i, ++o) { const float a = (float)t * freq((i / 2) * 2, dim_t); cc[o] = cosf(a); ss[o] = sinf(a); }
        for (int i = 0; i < dim_h; ++i, ++o) { const float a = (float)h * freq((i / 2) * 2, dim_h); cc[o] = cosf(a); ss[o] = sinf(a); }
        for (int i = 0; i < dim_w; ++i, ++o) { const float a = (float)w * freq((i / 2) * 2, dim_w); cc[o] = cosf(a); ss[o] = sinf(a); }
      }
  if (m.cosb) { rt::dev_free(m.cosb); rt::dev_free(m.sinb); m.cosb = m.sinb = nullptr; }
  if (rt::dev_malloc(&m.cosb, c.size() * 4) || rt::dev_malloc(&m.sinb, s.size() * 4)) return ctx->fail("dit: out of device memory (rotary tables)");
  rt::memcpy_h2d(m.cosb, c.data(), c.size() * 4, ctx->stream);
  rt::memcpy_h2d(m.sinb, s.data(), s.size() * 4, ctx->stream);
  rt::stream_sync(ctx->stream);
  m.rt_T = T; m.rt_H = H; m.rt_W = W;
  return 0;
}

template <class T>
static void launch_dit_ln(Ctx* ctx, const DitLnParams& p) {
  STAR_LAUNCH((dit_ln_kernel<T>), dim3((unsigned)((p.rows + 3) / 4)), dim3(256), (size_t)0, ctx->stream, p);
}
static int op_dit_ln(Ctx* ctx, DitLnParams p) {
  if (p.rows <= 0) return 0;
  ProfScope ps(ctx, PK_LN, 0.0, (p.mode == DLN_STATS ? 1.0 : 2.0) * p.rows * (double)p.C * 2.0);
  if (ctx->dtype == DT_F16) launch_dit_ln<f16>(ctx, p); else launch_dit_ln<bf16>(ctx, p);
  return 0;
}

int dit_block_forward(Ctx* ctx, int layer, const void* x_in, const float* emb, void* x_out, int text_len, int T, int H, int W) {
  auto mp = dit_of(ctx);
  if (!mp) return ctx->fail("dit_block_forward: no model built (star_dit_build)");
  DitModel& m = *mp;
  if (layer < 0 || layer >= m.n_layers) return ctx->fail("dit_block_forward: bad layer index");
  if (text_len < 0 || T < 1 || H < 1 || W < 1) return ctx->fail("dit_block_forward: bad token geometry");
  const DitLayerW& w = m.layers[layer];
  const int D = m.D, S = text_len + T * H * W, NV = T * H * W;
  const size_t es = ctx->esize();
  if (rotary_tables(ctx, m, T, H, W)) return 1;
  int rc = 0;
  auto ok = [&](int r) { if (r && !rc) rc = r; };

  // adaLN_modulation(emb) = Linear(SiLU(emb)) -> 12 chunks of D (dit_video_concat.py:497-510): shift / scale / gate of the
  // attention and MLP halves for the video tokens, then the same six for the text tokens
  Buf mod(ctx, (size_t)12 * D * 4);
  ok(op_gemv(ctx, emb, w.ada.w.p, (const float*)w.ada.b.p, mod.as<float>(), 12 * D, m.E, true, false));
  const float* M = mod.as<float>();
  const float *shift_msa = M, *scale_msa = M + D, *gate_msa = M + 2 * D, *shift_mlp = M + 3 * D, *scale_mlp = M + 4 * D, *gate_mlp = M + 5 * D;
  const float *t_shift_msa = M + 6 * D, *t_scale_msa = M + 7 * D, *t_gate_msa = M + 8 * D, *t_shift_mlp = M + 9 * D, *t_scale_mlp = M + 10 * D,
              *t_gate_mlp = M + 11 * D;
  const char* xin = (const char*)x_in;
  const size_t row_b = (size_t)D * es;

  auto ln = [&](const void* x, void* y, int rows, const NormW& n, const float* scale, const float* shift, int mode, float* maps) {
    DitLnParams p{x, y, (const float*)n.g.p, (const float*)n.b.p, scale, shift, (const float*)w.w_spa.p, (const float*)w.w_tmp.p, maps,
                  D, D, D, rows, H, W, m.ln_eps, mode};
    ok(op_dit_ln(ctx, p));
  };
  auto gemm = [&](const void* A, int lda, const LinW& lw, void* C, int ldc, int extra) {
    GemmArgs g;
    g.A = A; g.W = lw.w.p; g.C = C; g.M = S; g.N = lw.N; g.K = lw.K; g.lda = lda; g.ldc = ldc;
    g.bias = (const float*)lw.b.p; g.epi = (g.bias ? EPI_BIAS : 0) | extra;
    ok(op_gemm(ctx, g));
  };
  auto gated_add = [&](const void* res, const void* d, void* out, const float* gate, long long rows) {
    if (rows <= 0) return;
    ProfScope ps(ctx, PK_MISC, 0.0, 3.0 * rows * (double)D * 2.0);
    GatedAddParams p{res, d, out, gate, D, rows};
    long long blocks = (rows * (D / 8) + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    if (ctx->dtype == DT_F16) STAR_LAUNCH((gated_add_kernel<f16>), dim3((unsigned)blocks), dim3(256), (size_t)0, ctx->stream, p);
    else STAR_LAUNCH((gated_add_kernel<bf16>), dim3((unsigned)blocks), dim3(256), (size_t)0, ctx->stream, p);
  };

  // ---- attention half: LayerNorm + modulate (text / video), spatial + temporal LIEM on the video tokens (:517-533)
  Buf y(ctx, (size_t)S * row_b), maps(ctx, (size_t)NV * 2 * 4);
  if (!y.p || !maps.p) return ctx->fail("dit: out of device memory");
  char* yv = (char*)y.p + (size_t)text_len * row_b;
  ln(xin, y.p, text_len, w.ln1, t_scale_msa, t_shift_msa, DLN_PLAIN, nullptr);
  ln(xin + (size_t)text_len * row_b, nullptr, NV, w.ln1, scale_msa, shift_msa, DLN_STATS, maps.as<float>());
  ln(xin + (size_t)text_len * row_b, yv, NV, w.ln1, scale_msa, shift_msa, DLN_GATE, maps.as<float>());
  maps.reset();
  // fused q|k|v projection, per-head QK LayerNorm + rotary (video tokens), full self-attention over text + video tokens
  Buf qkv(ctx, (size_t)S * 3 * row_b);
  if (!qkv.p) return ctx->fail("dit: out of device memory (qkv)");
  gemm(y.p, D, w.qkv, qkv.p, 3 * D, 0);
  {
    ProfScope ps(ctx, PK_MISC, 0.0, 4.0 * S * (double)D * 2.0);
    QkNormRopeParams p{qkv.p, 3 * D, D, m.heads, S, text_len, (const float*)w.qn.g.p, (const float*)w.qn.b.p, (const float*)w.kn.g.p,
                       (const float*)w.kn.b.p, (const float*)m.cosb, (const float*)m.sinb, 1e-6f};
    const long long items = (long long)S * m.heads;
    if (ctx->dtype == DT_F16) STAR_LAUNCH((qk_norm_rope_kernel<f16>), dim3((unsigned)((items + 3) / 4)), dim3(256), (size_t)0, ctx->stream, p);
    else STAR_LAUNCH((qk_norm_rope_kernel<bf16>), dim3((unsigned)((items + 3) / 4)), dim3(256), (size_t)0, ctx->stream, p);
  }
  {
    AttnArgs a;
    a.Q = qkv.p; a.K = (char*)qkv.p + row_b; a.V = (char*)qkv.p + 2 * row_b; a.O = y.p;
    a.ldq = a.ldk = a.ldv = 3 * D; a.ldo = D;
    a.bsq = a.bsk = a.bsv = a.bso = 0;
    a.Nq = a.Nk = S; a.heads = m.heads; a.batch = 1; a.scale = 0.125f;
    ok(op_flash_attn(ctx, a));
  }
  qkv.reset();
  Buf d(ctx, (size_t)S * row_b), h1(ctx, (size_t)S * row_b);
  if (!d.p || !h1.p) return ctx->fail("dit: out of device memory");
  gemm(y.p, D, w.dense, d.p, D, 0);
  gated_add(xin, d.p, h1.p, t_gate_msa, text_len);                                                       // :542
  gated_add(xin + (size_t)text_len * row_b, (char*)d.p + (size_t)text_len * row_b, (char*)h1.p + (size_t)text_len * row_b, gate_msa, NV);   // :541

  // ---- MLP half (:544-562): LayerNorm + modulate, dense_h_to_4h + tanh-GELU, dense_4h_to_h, gated residual
  ln(h1.p, y.p, text_len, w.ln2, t_scale_mlp, t_shift_mlp, DLN_PLAIN, nullptr);
  ln((char*)h1.p + (size_t)text_len * row_b, yv, NV, w.ln2, scale_mlp, shift_mlp, DLN_PLAIN, nullptr);
  Buf u(ctx, (size_t)S * w.fc1.N * es);
  if (!u.p) return ctx->fail("dit: out of device memory (mlp)");
  gemm(y.p, D, w.fc1, u.p, w.fc1.N, EPI_GELU_TANH);
  gemm(u.p, w.fc1.N, w.fc2, d.p, D, 0);
  u.reset();
  gated_add(h1.p, d.p, x_out, t_gate_mlp, text_len);
  gated_add((char*)h1.p + (size_t)text_len * row_b, (char*)d.p + (size_t)text_len * row_b, (char*)x_out + (size_t)text_len * row_b, gate_mlp, NV);
  return rc;
}

}  // namespace star
