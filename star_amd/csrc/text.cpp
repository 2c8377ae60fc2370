// text.cpp -- the OpenCLIP ViT-H/14 text tower of FrozenOpenCLIPEmbedder (reference video_to_video/modules/embedder.py:49-72;
// SURVEY.md section 8(f) rank 3) on the same runtime as the denoiser: every matmul is star's gemm_kernel, every LayerNorm its
// ln_kernel, the 77-token causal self-attention the product flash kernel with its causal mask (attn5.h).
//   x = token_embedding(tokens) + positional_embedding                     (host: an index gather)
//   for the first `run_layers` pre-LN blocks (23 of 24 for layer = 'penultimate', embedder.py:62-70):
//       x = x + out_proj(attn(ln_1(x)))         in_proj: fused q | k | v rows, 16 heads x 64, key j visible to query i for j <= i
//       x = x + c_proj(gelu(c_fc(ln_2(x))))     exact (erf) GELU
//   return ln_final(x)
// Token rows are [batch * tokens, width] in the context's 16-bit type with fp32 accumulation everywhere (the reference keeps the
// tower in fp32: its output feeds 16-bit cross-attention K / V projections).  Weight names are open_clip's (a CLIP state dict
// loads unchanged).  PARITY UNPINNED: open_clip is not installed in this image; the oracle is the torch restatement in
// star_amd/modules/embedder.py + oracle/text_oracle.py.
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include "graph.h"
#include "gemm.h"

namespace star {

struct TextLayerW { NormW ln1, ln2; LinW qkv, out, fc, proj; };
struct TextModel {
  int W = 0, heads = 0, n_layers = 0;
  std::vector<TextLayerW> layers;
  NormW ln_final;
  std::vector<void*> owned;
  ~TextModel() { for (void* p : owned) rt::dev_free(p); }
};

static std::mutex& text_mutex() { static std::mutex m; return m; }
static std::unordered_map<Ctx*, std::shared_ptr<TextModel>>& text_models() {
  static std::unordered_map<Ctx*, std::shared_ptr<TextModel>> m;
  return m;
}
static std::shared_ptr<TextModel> text_of(Ctx* ctx) {
  std::lock_guard<std::mutex> g(text_mutex());
  auto it = text_models().find(ctx);
  return it == text_models().end() ? nullptr : it->second;
}
void text_release(Ctx* ctx) {
  std::lock_guard<std::mutex> g(text_mutex());
  text_models().erase(ctx);
}

// nn.MultiheadAttention keeps its input projection as bare tensors (`in_proj_weight` [3W, W], `in_proj_bias`), not as a Linear
static LinW packed_in_proj(Builder& b, const std::string& p) {
  LinW l;
  const HostTensor* w = b.get(p + "in_proj_weight");
  const HostTensor* bias = b.get(p + "in_proj_bias");
  if (!w || !bias) return l;
  l.N = (int)w->shape[0]; l.K = (int)(w->data.size() / (size_t)l.N);
  l.w = b.upload_T(w->data);
  l.b = b.upload_f32(bias->data);
  return l;
}

int text_build(Ctx* ctx, int W, int heads, int n_layers) {
  if (W % 64 || heads * 64 != W) return ctx->fail("text_build: width must be heads x 64");
  if (n_layers < 1) return ctx->fail("text_build: no layers");
  auto m = std::make_shared<TextModel>();
  m->W = W; m->heads = heads; m->n_layers = n_layers;
  Builder b{ctx, &m->owned, ""};
  for (int i = 0; i < n_layers; ++i) {
    const std::string L = "transformer.resblocks." + std::to_string(i) + ".";
    TextLayerW w;
    w.ln1 = b.norm(L + "ln_1");
    w.ln2 = b.norm(L + "ln_2");
    w.qkv = packed_in_proj(b, L + "attn.");
    w.out = b.linear(L + "attn.out_proj");
    w.fc = b.linear(L + "mlp.c_fc");
    w.proj = b.linear(L + "mlp.c_proj");
    if (!b.err.empty()) return ctx->fail("text_build: " + b.err);
    if (w.qkv.N != 3 * W || w.qkv.K != W || w.out.N != W || w.out.K != W || w.fc.K != W || w.proj.N != W || w.proj.K != w.fc.N || w.ln1.C != W ||
        w.ln2.C != W)
      return ctx->fail("text_build: tensor shapes do not match the configuration");
    m->layers.push_back(w);
  }
  m->ln_final = b.norm("ln_final");
  if (!b.err.empty()) return ctx->fail("text_build: " + b.err);
  {
    std::lock_guard<std::mutex> g(text_mutex());
    text_models()[ctx] = m;
  }
  ctx->host_tensors.clear();
  return 0;
}

// exact GELU in place on 16-bit rows (the erfc form of gemm.h: one transcendental, |abs err| <= 6e-7)
struct GeluParams { void* x; long long n8; };
template <class T>
STAR_GLOBAL void gelu_rows_kernel(const GeluParams p) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n8; i += (long long)gridDim.x * blockDim.x) {
    vec<T, 8> v = reinterpret_cast<vec<T, 8>*>(p.x)[i];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = from_f32<T>(gelu_erf(to_f32<T>(v[e])));
    reinterpret_cast<vec<T, 8>*>(p.x)[i] = v;
  }
}

int text_forward(Ctx* ctx, const void* x_in, int batch, int tokens, int run_layers, void* out) {
  auto mp = text_of(ctx);
  if (!mp) return ctx->fail("text_forward: no model built (star_text_build)");
  TextModel& m = *mp;
  if (batch < 1 || tokens < 1) return ctx->fail("text_forward: empty input");
  if (run_layers < 0 || run_layers > m.n_layers) return ctx->fail("text_forward: bad layer count");
  const int W = m.W, M = batch * tokens;
  const size_t es = ctx->esize(), row_b = (size_t)W * es;
  int rc = 0;
  auto ok = [&](int r) { if (r && !rc) rc = r; };
  Buf x(ctx, (size_t)M * row_b), x2(ctx, (size_t)M * row_b), y(ctx, (size_t)M * row_b), qkv(ctx, (size_t)M * 3 * row_b), u(ctx, (size_t)M * 4 * row_b);
  if (!x.p || !x2.p || !y.p || !qkv.p || !u.p) return ctx->fail("text_forward: out of device memory");
  rt::memcpy_d2d(x.p, x_in, (size_t)M * row_b, ctx->stream);
  auto gemm = [&](const void* A, int lda, const LinW& lw, void* C, int ldc, const void* res) {
    GemmArgs g;
    g.A = A; g.W = lw.w.p; g.C = C; g.M = M; g.N = lw.N; g.K = lw.K; g.lda = lda; g.ldc = ldc;
    g.bias = (const float*)lw.b.p; g.res = res; g.ldr = W;
    g.epi = (g.bias ? EPI_BIAS : 0) | (res ? EPI_RES : 0);
    ok(op_gemm(ctx, g));
  };
  auto ln = [&](const void* a, void* b, const NormW& n) {
    ok(op_layer_norm(ctx, a, W, b, W, (const float*)n.g.p, (const float*)n.b.p, M, W, 1e-5f, LN_PLAIN, nullptr, nullptr, 0, 0));
  };
  for (int i = 0; i < run_layers && !rc; ++i) {
    const TextLayerW& w = m.layers[i];
    if (w.fc.N != 4 * W) return ctx->fail("text_forward: MLP width must be 4 x width");
    ln(x.p, y.p, w.ln1);
    gemm(y.p, W, w.qkv, qkv.p, 3 * W, nullptr);
    {
      AttnArgs a;
      a.Q = qkv.p; a.K = (char*)qkv.p + row_b; a.V = (char*)qkv.p + 2 * row_b; a.O = y.p;
      a.ldq = a.ldk = a.ldv = 3 * W; a.ldo = W;
      a.bsq = a.bsk = a.bsv = (long long)tokens * 3 * W; a.bso = (long long)tokens * W;
      a.Nq = a.Nk = tokens; a.heads = m.heads; a.batch = batch; a.scale = 0.125f; a.causal = 1;
      ok(op_flash_attn(ctx, a));
    }
    gemm(y.p, W, w.out, x2.p, W, x.p);                      // x2 = x + out_proj(attn)
    ln(x2.p, y.p, w.ln2);
    gemm(y.p, W, w.fc, u.p, 4 * W, nullptr);
    {
      ProfScope ps(ctx, PK_MISC, 0.0, 2.0 * M * 4.0 * W * 2.0);
      GeluParams gp{u.p, (long long)M * 4 * W / 8};
      long long blocks = (gp.n8 + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      if (ctx->dtype == DT_F16) STAR_LAUNCH((gelu_rows_kernel<f16>), dim3((unsigned)blocks), dim3(256), (size_t)0, ctx->stream, gp);
      else STAR_LAUNCH((gelu_rows_kernel<bf16>), dim3((unsigned)blocks), dim3(256), (size_t)0, ctx->stream, gp);
    }
    gemm(u.p, 4 * W, w.proj, x.p, W, x2.p);                 // x = x2 + c_proj(gelu(c_fc))
  }
  ln(x.p, out, m.ln_final);
  return rc;
}

}  // namespace star
