// graph.h -- shared pieces of the C++ graph executors (UNet, VAE): device weight handles, the weight
// repacking Builder (reference / diffusers tensor layouts -> kernel layouts) and the Runner with the
// activation bookkeeping and thin op wrappers.
#pragma once
#include "ops.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace star {

struct DevW {            // a device-resident parameter tensor
  void* p = nullptr;
  int64_t n = 0;
};
struct LinW { DevW w; DevW b; int N = 0, K = 0; DevW colsum; };   // w: T [N][K]; b: fp32 [N] (may be empty); colsum: fp32 [N], folded-LayerNorm layers only
struct NormW { DevW g, b; int C = 0; };               // fp32

struct Builder {
  Ctx* ctx;
  std::vector<void*>* owned;
  std::string err;

  const HostTensor* get(const std::string& name) {
    auto it = ctx->host_tensors.find(name);
    if (it == ctx->host_tensors.end()) { if (err.empty()) err = "missing tensor: " + name; return nullptr; }
    return &it->second;
  }
  static uint16_t cvt16(float v, int dtype) {
    if (dtype == DT_F16) { f16 h = (f16)v; uint16_t u; memcpy(&u, &h, 2); return u; }
    uint32_t u; memcpy(&u, &v, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
  }
  DevW upload_T(const std::vector<float>& v) {
    std::vector<uint16_t> h(v.size());
    for (size_t i = 0; i < v.size(); ++i) h[i] = cvt16(v[i], ctx->dtype);
    DevW d; d.n = (int64_t)v.size();
    if (rt::dev_malloc(&d.p, h.size() * 2 + 256)) { err = "device OOM uploading weights"; return d; }
    owned->push_back(d.p);
    rt::memcpy_h2d(d.p, h.data(), h.size() * 2, ctx->stream);
    rt::stream_sync(ctx->stream);
    return d;
  }
  DevW upload_f32(const std::vector<float>& v) {
    DevW d; d.n = (int64_t)v.size();
    if (rt::dev_malloc(&d.p, v.size() * 4 + 256)) { err = "device OOM uploading weights"; return d; }
    owned->push_back(d.p);
    rt::memcpy_h2d(d.p, v.data(), v.size() * 4, ctx->stream);
    rt::stream_sync(ctx->stream);
    return d;
  }
  NormW norm(const std::string& p) {
    NormW n;
    const HostTensor* g = get(p + ".weight"); const HostTensor* b = get(p + ".bias");
    if (!g || !b) return n;
    n.C = (int)g->data.size(); n.g = upload_f32(g->data); n.b = upload_f32(b->data);
    return n;
  }
  DevW bias(const std::string& name) {
    const HostTensor* b = get(name);
    return b ? upload_f32(b->data) : DevW{};
  }
  // nn.Linear / 1x1 conv / Conv1d(k=1): weight [N, K, (1,1)] -> [N][K]
  LinW linear(const std::string& p, bool has_bias = true) {
    LinW l;
    const HostTensor* w = get(p + ".weight");
    if (!w) return l;
    l.N = (int)w->shape[0]; l.K = (int)(w->data.size() / (size_t)l.N);
    l.w = upload_T(w->data);
    if (has_bias) l.b = bias(p + ".bias");
    return l;
  }
  // Conv2d 3x3 [N, C, 3, 3] -> [N][C / 64][tap][64] (K = 9C: the nine taps of a 64-channel block adjacent, conv_k below), tap = ky*3 + kx
  LinW conv3x3(const std::string& p, int pad_n_to = 1) {
    LinW l;
    const HostTensor* w = get(p + ".weight");
    if (!w) return l;
    const int N0 = (int)w->shape[0], C = (int)w->shape[1];
    if (C % 64) { if (err.empty()) err = "conv3x3: input channels must be a multiple of 64 (" + p + ")"; return l; }
    const int N = (N0 + pad_n_to - 1) / pad_n_to * pad_n_to;   // zero rows up to a multiple (fp32-out GEMMs need N % 4 == 0)
    if (N != N0) {
      std::vector<float> r((size_t)N * 9 * C, 0.f), rb(N, 0.f);
      const HostTensor* bb = get(p + ".bias");
      if (!bb) return l;
      for (int n = 0; n < N0; ++n) { rb[n] = bb->data[n]; for (int c = 0; c < C; ++c) for (int t = 0; t < 9; ++t) r[(size_t)n * 9 * C + conv_k(c, t)] = w->data[((size_t)n * C + c) * 9 + t]; }
      l.N = N; l.K = 9 * C; l.w = upload_T(r); l.b = upload_f32(rb);
      return l;
    }
    std::vector<float> r((size_t)N * 9 * C);
    for (int n = 0; n < N; ++n) for (int c = 0; c < C; ++c) for (int t = 0; t < 9; ++t)
      r[(size_t)n * 9 * C + conv_k(c, t)] = w->data[((size_t)n * C + c) * 9 + t];
    l.N = N; l.K = 9 * C; l.w = upload_T(r); l.b = bias(p + ".bias");
    return l;
  }
  // K index of (input channel c, tap t) of a 3x3 conv: 64-channel blocks outermost, the nine taps inside a block (gemm.h: the K loop
  // walks the taps of one block back to back, so that their shifted re-reads of the same input lines hit the L2); Cin % 64 == 0
  static size_t conv_k(int c, int t) { return ((size_t)(c >> 6) * 9 + t) * 64 + (c & 63); }
  // stem-like Conv2d 3x3 with tiny C (4): [N, C, 3, 3] -> [N][64] columns tap*C + c, zero padded (pairs with stem_im2col)
  LinW conv3x3_im2col64(const std::string& p) {
    LinW l;
    const HostTensor* w = get(p + ".weight");
    if (!w) return l;
    const int N = (int)w->shape[0], C = (int)w->shape[1];
    std::vector<float> r((size_t)N * 64, 0.f);
    for (int n = 0; n < N; ++n) for (int c = 0; c < C; ++c) for (int t = 0; t < 9; ++t)
      r[(size_t)n * 64 + t * C + c] = w->data[((size_t)n * C + c) * 9 + t];
    l.N = N; l.K = 64; l.w = upload_T(r); l.b = bias(p + ".bias");
    return l;
  }
  // Conv3d (3,1,1) [N, C, 3, 1, 1] -> [N][tap][C]
  LinW tconv(const std::string& p) {
    LinW l;
    const HostTensor* w = get(p + ".weight");
    if (!w) return l;
    const int N = (int)w->shape[0], C = (int)w->shape[1];
    std::vector<float> r((size_t)N * 3 * C);
    for (int n = 0; n < N; ++n) for (int c = 0; c < C; ++c) for (int t = 0; t < 3; ++t)
      r[((size_t)n * 3 + t) * C + c] = w->data[((size_t)n * C + c) * 3 + t];
    l.N = N; l.K = 3 * C; l.w = upload_T(r); l.b = bias(p + ".bias");
    return l;
  }
  // row-concatenate several [Ni, K] matrices (fused q|k|v)
  LinW fused(const std::vector<std::string>& names) {
    LinW l;
    std::vector<float> r;
    for (auto& nm : names) {
      const HostTensor* w = get(nm + ".weight");
      if (!w) return l;
      l.K = (int)(w->data.size() / (size_t)w->shape[0]);
      l.N += (int)w->shape[0];
      r.insert(r.end(), w->data.begin(), w->data.end());
    }
    l.w = upload_T(r);
    return l;
  }
  // GEGLU projection [2H, K] (value rows, gate rows) -> alternating 32-row (value, gate) blocks
  LinW geglu(const std::string& p) {
    LinW l;
    const HostTensor* w = get(p + ".weight"); const HostTensor* b = get(p + ".bias");
    if (!w || !b) return l;
    const int N2 = (int)w->shape[0], K = (int)w->shape[1], Hh = N2 / 2;
    std::vector<float> r((size_t)N2 * K), rb(N2);
    for (int blk = 0; blk < Hh / 32; ++blk)
      for (int half = 0; half < 2; ++half)
        for (int i = 0; i < 32; ++i) {
          const int src = half * Hh + blk * 32 + i, dst = blk * 64 + half * 32 + i;
          memcpy(&r[(size_t)dst * K], &w->data[(size_t)src * K], (size_t)K * 4);
          rb[dst] = b->data[src];
        }
    l.N = N2; l.K = K; l.w = upload_T(r); l.b = upload_f32(rb);
    return l;
  }
  static float round16(float v, int dtype) {
    const uint16_t u = cvt16(v, dtype);
    if (dtype == DT_F16) { f16 h; memcpy(&h, &u, 2); return (float)h; }
    const uint32_t w = (uint32_t)u << 16; float f; memcpy(&f, &w, 4); return f;
  }
  // A LayerNorm folded into the Linear(s) it feeds (the only consumers of its output): y = LN(x) W^T + b with
  // LN(x) = (a_m x + b_m) gamma + beta per row  =>  y[m][n] = a_m (x W'^T)[m][n] + b_m s_n + c_n,  W' = gamma o W (stored in T),
  // s_n = sum_k W'[n][k] (of the ROUNDED weights, so the mean term cancels exactly what the MFMA accumulated),
  // c_n = sum_k beta_k W[n][k] + b_n.  `rows` = the weight matrices stacked along N (fused q|k|v), already in kernel row order.
  LinW fold_ln(const std::vector<float>& w, const std::vector<float>& bias, int N, int K, const std::string& norm) {
    LinW l;
    const HostTensor* g = get(norm + ".weight"); const HostTensor* be = get(norm + ".bias");
    if (!g || !be) return l;
    if ((int)g->data.size() != K) { if (err.empty()) err = "fold_ln: LayerNorm width does not match " + norm; return l; }
    std::vector<float> wf((size_t)N * K), cs(N), cb(N);
    for (int n = 0; n < N; ++n) {
      double s = 0.0, c = 0.0;
      for (int k = 0; k < K; ++k) {
        const float v = w[(size_t)n * K + k] * g->data[k];
        wf[(size_t)n * K + k] = v;
        s += (double)round16(v, ctx->dtype);
        c += (double)be->data[k] * (double)w[(size_t)n * K + k];
      }
      cs[n] = (float)s;
      cb[n] = (float)c + (bias.empty() ? 0.f : bias[n]);
    }
    l.N = N; l.K = K; l.w = upload_T(wf); l.b = upload_f32(cb); l.colsum = upload_f32(cs);
    return l;
  }
  LinW linear_ln(const std::string& p, const std::string& norm, bool has_bias = true) {
    const HostTensor* w = get(p + ".weight");
    if (!w) return LinW{};
    const int N = (int)w->shape[0], K = (int)(w->data.size() / (size_t)N);
    std::vector<float> b;
    if (has_bias) { const HostTensor* bb = get(p + ".bias"); if (!bb) return LinW{}; b = bb->data; }
    return fold_ln(w->data, b, N, K, norm);
  }
  LinW fused_ln(const std::vector<std::string>& names, const std::string& norm) {
    std::vector<float> r; int N = 0, K = 0;
    for (auto& nm : names) {
      const HostTensor* w = get(nm + ".weight");
      if (!w) return LinW{};
      K = (int)(w->data.size() / (size_t)w->shape[0]); N += (int)w->shape[0];
      r.insert(r.end(), w->data.begin(), w->data.end());
    }
    return fold_ln(r, {}, N, K, norm);
  }
  // fused q | k | v with the LayerNorm folded in, rows re-ordered into 64-row tiles (q_h, k_h, v_h) per head (gemm_tq.h)
  LinW fused_ln_heads(const std::vector<std::string>& names, const std::string& norm, int heads) {
    std::vector<float> r; int K = 0, C = 0;
    for (auto& nm : names) {
      const HostTensor* w = get(nm + ".weight");
      if (!w) return LinW{};
      K = (int)(w->data.size() / (size_t)w->shape[0]); C = (int)w->shape[0];
      r.insert(r.end(), w->data.begin(), w->data.end());
    }
    if (names.size() != 3 || C != heads * 64) { if (err.empty()) err = "fused_ln_heads: q | k | v of heads x 64 rows expected"; return LinW{}; }
    std::vector<float> o(r.size());
    for (int h = 0; h < heads; ++h)
      for (int part = 0; part < 3; ++part)
        memcpy(&o[((size_t)(3 * h + part) * 64) * K], &r[((size_t)part * C + (size_t)h * 64) * K], (size_t)64 * K * 4);
    return fold_ln(o, {}, 3 * C, K, norm);
  }
  LinW geglu_ln(const std::string& p, const std::string& norm) {
    const HostTensor* w = get(p + ".weight"); const HostTensor* b = get(p + ".bias");
    if (!w || !b) return LinW{};
    const int N2 = (int)w->shape[0], K = (int)w->shape[1], Hh = N2 / 2;
    std::vector<float> r((size_t)N2 * K), rb(N2);
    for (int blk = 0; blk < Hh / 32; ++blk)
      for (int half = 0; half < 2; ++half)
        for (int i = 0; i < 32; ++i) {
          const int src = half * Hh + blk * 32 + i, dst = blk * 64 + half * 32 + i;
          memcpy(&r[(size_t)dst * K], &w->data[(size_t)src * K], (size_t)K * 4);
          rb[dst] = b->data[src];
        }
    return fold_ln(r, rb, N2, K, norm);
  }
  // FeedForward's second Linear and the transformer's proj_out are both affine maps of the SAME rows with only a residual between them:
  //   out = proj_out(ff2(g) + h2) + x_in = [W_po | W_po W_ff2] [h2 ; g] + (b_po + W_po b_ff2) + x_in            (unet_v2v.py:316,489-490,526-529)
  // so one GEMM over the concatenated operand [h2 | g] (K = 5 x inner) replaces two, and h3 = ff2(g) + h2 (one write + one read of the
  // activation) never exists.  The composite weight is computed ON THE DEVICE at build time from the 16-bit weights the two GEMMs would
  // have multiplied by, in ONE launch: W_po [I ; W_ff2^T]^T (fp32 accumulation, rounded once; the identity block reproduces W_po
  // exactly); the bias in fp32 on the host.  Round 6.
  LinW compose_ff2_proj_out(const std::string& ff2, const std::string& po) {
    LinW l;
    const HostTensor* w2 = get(ff2 + ".weight"); const HostTensor* b2 = get(ff2 + ".bias");
    const HostTensor* wp = get(po + ".weight"); const HostTensor* bp = get(po + ".bias");
    if (!w2 || !b2 || !wp || !bp) return l;
    const int inner = (int)w2->shape[0], hid = (int)(w2->data.size() / (size_t)inner);
    const int Co = (int)wp->shape[0];
    if ((int)(wp->data.size() / (size_t)Co) != inner) { if (err.empty()) err = "compose_ff2_proj_out: proj_out's input width must be ff's output width (" + po + ")"; return l; }
    const int Kc = inner + hid;
    std::vector<float> e((size_t)Kc * inner, 0.f), rb(Co);
    for (int i = 0; i < inner; ++i) e[(size_t)i * inner + i] = 1.0f;
    for (int j = 0; j < inner; ++j)
      for (int k = 0; k < hid; ++k) e[(size_t)(inner + k) * inner + j] = w2->data[(size_t)j * hid + k];
    for (int n = 0; n < Co; ++n) {
      double acc = bp->data[n];
      for (int j = 0; j < inner; ++j) acc += (double)wp->data[(size_t)n * inner + j] * b2->data[j];
      rb[n] = (float)acc;
    }
    // temporaries of the build (freed right after the launch has completed)
    std::vector<void*> tmp;
    std::vector<void*>* keep = owned;
    owned = &tmp;
    const DevW ed = upload_T(e), pd = upload_T(wp->data);
    owned = keep;
    l.N = Co; l.K = Kc;
    l.w.n = (int64_t)Co * Kc;
    int rc = 1;
    if (ed.p && pd.p && !rt::dev_malloc(&l.w.p, (size_t)Co * Kc * 2 + 256)) {
      owned->push_back(l.w.p);
      GemmArgs g;
      g.A = pd.p; g.W = ed.p; g.C = l.w.p; g.M = Co; g.N = Kc; g.K = inner; g.lda = inner; g.ldc = Kc;
      rc = op_gemm(ctx, g);
      rt::stream_sync(ctx->stream);
    }
    for (void* q : tmp) rt::dev_free(q);
    if (rc) { if (err.empty()) err = "compose_ff2_proj_out: composing " + po + " failed"; return LinW{}; }
    l.b = upload_f32(rb);
    return l;
  }
  DevW raw_f32(const std::string& name) {
    const HostTensor* t = get(name);
    return t ? upload_f32(t->data) : DevW{};
  }

};

struct Act {          // an activation tensor: rows = F*H*W tokens; the buffer returns to the pool with its last owner
  std::shared_ptr<Buf> buf; int C = 0, H = 0, W = 0;
  int ld = 0;          // row stride in elements when the rows live inside a wider buffer (Runner::make_wide); 0 = dense (C)
  int ldv() const { return ld ? ld : C; }
  // GroupNorm partial statistics of THIS tensor, written by the kernel that produced it (GemmArgs::gn_partial): a GroupNorm that
  // reads the tensor finalizes from them instead of running its own statistics pass; null = none
  std::shared_ptr<Buf> gnp;
  // ... and its per-row (LayerNorm) statistics (GemmArgs::ln_partial, ln_parts parts per row)
  std::shared_ptr<Buf> lnp; int ln_parts = 0;
  void* p() const { return buf ? buf->p : nullptr; }
  void drop() { buf.reset(); gnp.reset(); lnp.reset(); ln_parts = 0; }
};


struct Runner {
  Ctx* ctx;
  int F = 0;
  size_t es = 2;
  float* emb = nullptr;        // fp32 [E] (device) time embedding of the running net
  const void* context = nullptr;  // T [77][ctx_dim]
  int ctx_dim = 0, embed_dim = 0;
  int rc = 0;

  Act make(int C, int H, int W) {
    Act a; a.C = C; a.H = H; a.W = W;
    a.buf = std::make_shared<Buf>(ctx, (size_t)F * H * W * C * es);
    if (!a.buf->p) { rc = ctx->fail("out of device memory (activations)"); }
    return a;
  }
  // C columns at the head of rows of `ld` elements: the tail (ld - C columns per row) belongs to the caller (Fwd::ff_and_out keeps the
  // GEGLU output there, so that [h2 | g] is ONE operand of the composed FF-out / proj_out GEMM)
  Act make_wide(int C, int H, int W, int ld) {
    Act a; a.C = C; a.H = H; a.W = W; a.ld = ld;
    a.buf = std::make_shared<Buf>(ctx, (size_t)F * H * W * ld * es);
    if (!a.buf->p) { rc = ctx->fail("out of device memory (activations)"); }
    return a;
  }
  int rows(const Act& a) const { return F * a.H * a.W; }
  void ok(int r) { if (r && !rc) rc = r; }
  // GroupNorm statistics in the producers' epilogues (STAR_NO_GNEPI=1: every GroupNorm runs its own statistics pass; A/B switch, read once)
  static bool gn_epi_enabled() { static const bool v = std::getenv("STAR_NO_GNEPI") == nullptr; return v; }
  // op_gemm producing y; `stats`: also ask for y's GroupNorm partials (kept on y only if the launched tile wrote them)
  // rowstats: ask for y's per-row statistics instead (the next reader is a LayerNorm; levels of width <= 640: wider layers are
  // tail-split, and the remainder tile would cut their rows into a different number of parts)
  void gemm_to(GemmArgs& g, Act* y, bool stats, bool rowstats = false) {
    bool done = false, ldone = false;
    int lparts = 0;
    std::shared_ptr<Buf> part, lpart;
    if (stats && y && gn_epi_enabled() && !(g.N & 63)) {
      part = std::make_shared<Buf>(ctx, (size_t)((g.M + 31) / 32) * g.N * sizeof(float));
      if (part->p) { g.gn_partial = part->as<float>(); g.gn_done = &done; }
    } else if (rowstats && y && ln_epi_enabled() && g.mode == A_PLAIN && g.N <= 640 && !(g.N & 7)) {
      // two parts per column tile of the tile the launcher will choose (gemm_impl.h: 128 x 128 for small problems, else 256 x 320 for the
      // 320-multiple widths; any other choice drops the request, and the launcher re-checks the capacity): at level 0 that is 2 parts per
      // row (32 bytes), not the 128-column worst case's 6 (81 MB allocated per producer for 27 MB used)
      const bool small_tile = g.M <= 4096 && g.N <= 1024;
      const int cap = (!small_tile && g.N % 320 == 0) ? 2 * (g.N / 320) : 2 * ((g.N + 127) / 128);
      lpart = std::make_shared<Buf>(ctx, (size_t)g.M * cap * 4 * sizeof(float));
      if (lpart->p) { g.ln_partial = lpart->as<float>(); g.ln_parts_cap = cap; g.ln_parts = &lparts; g.ln_done = &ldone; }
    }
    ok(op_gemm(ctx, g));
    if (y) {
      y->gnp = done ? part : nullptr;
      y->lnp = ldone ? lpart : nullptr;
      y->ln_parts = ldone ? lparts : 0;
    }
  }
  // LayerNorm row statistics in the producers' epilogues (STAR_NO_LNEPI=1: every LayerNorm reads its rows; A/B switch, read once)
  static bool ln_epi_enabled() { static const bool v = std::getenv("STAR_NO_LNEPI") == nullptr; return v; }

  // y = x W^T (+b) (+res)
  void gemm(const void* A, int lda, int M, const LinW& w, void* C, int ldc, const void* res = nullptr, int ldr = 0, int extra_epi = 0,
            const float* bias_override = nullptr, Act* stats_of = nullptr, Act* rowstats_of = nullptr) {   // stats_of / rowstats_of: the Act that C is (dense, ldc == N): ask for its GroupNorm partials / LayerNorm row statistics
    GemmArgs g;
    g.A = A; g.W = w.w.p; g.C = C; g.M = M; g.N = w.N; g.K = w.K; g.lda = lda; g.ldc = ldc;
    g.bias = bias_override ? bias_override : (const float*)w.b.p;
    g.res = res; g.ldr = ldr;
    g.epi = (g.bias ? EPI_BIAS : 0) | (res ? EPI_RES : 0) | extra_epi;
    gemm_to(g, stats_of ? stats_of : rowstats_of, stats_of != nullptr, rowstats_of != nullptr);
  }
  void conv3x3(const Act& x, const LinW& w, Act& y, int mode, int stride, int pad_t, int pad_l, const void* res, const float* bias_override = nullptr,
               int extra_epi = 0, void* out_override = nullptr, int ldc_override = 0, int up_crop = 1, bool stats = false) {
    GemmArgs g;
    g.up_crop = up_crop;
    g.A = x.p(); g.W = w.w.p; g.C = out_override ? out_override : y.p();
    g.M = F * y.H * y.W; g.N = w.N; g.K = w.K; g.lda = x.C; g.ldc = ldc_override ? ldc_override : y.C;
    g.mode = mode; g.H = x.H; g.Wd = x.W; g.Cin = x.C; g.Ho = y.H; g.Wo = y.W; g.stride = stride; g.pad_t = pad_t; g.pad_l = pad_l;
    g.bias = bias_override ? bias_override : (const float*)w.b.p;
    g.res = res; g.ldr = y.C;
    g.epi = (g.bias ? EPI_BIAS : 0) | (res ? EPI_RES : 0) | extra_epi;
    gemm_to(g, (out_override || ldc_override) ? nullptr : &y, stats && !out_override && !ldc_override);
  }
  // y may be x itself (in place): its producer's partial statistics describe the un-normalised values and are dropped afterwards
  void gn(const Act& x, const NormW& n, Act& y, bool whole_chunk, float eps, bool silu) {
    const int rps = whole_chunk ? F * x.H * x.W : x.H * x.W;
    struct Drop { Act& y; bool on; ~Drop() { if (on) y.gnp.reset(); } } drop_stale{y, x.p() == y.p()};
    if (x.gnp && x.gnp->p)   // the producer of x left its partial statistics: finalize from them, no statistics pass
      ok(op_group_norm_fused(ctx, x.p(), x.C, y.p(), y.C, (const float*)n.g.p, (const float*)n.b.p, rows(x), x.C, rps, eps, silu, x.gnp->as<float>()));
    else
      ok(op_group_norm(ctx, x.p(), x.C, y.p(), y.C, (const float*)n.g.p, (const float*)n.b.p, rows(x), x.C, rps, eps, silu));
  }
  void ln(const void* x, void* y, int rws, int C, const NormW& n, int mode = LN_PLAIN, const float* gw = nullptr, float* maps = nullptr, int H = 0, int W = 0) {
    ok(op_layer_norm(ctx, x, C, y, C, (const float*)n.g.p, (const float*)n.b.p, rws, C, 1e-5f, mode, gw, maps, H, W));
  }
  // the row statistics of a LayerNorm that is folded into the next GEMM: rowab[m] = (a_m, b_m); x is read once, nothing is written back
  void ln_rows(const void* x, float* rowab, int rws, int C, int mode = LN_PLAIN, const float* gw = nullptr, float* maps = nullptr, int H = 0, int W = 0) {
    ok(op_layer_norm(ctx, x, C, nullptr, C, nullptr, nullptr, rws, C, 1e-5f, mode, gw, maps, H, W, rowab));
  }
  // the same for an activation whose producer left its row statistics (gemm(..., rowstats_of)): 16 x parts bytes per row instead of the row
  void ln_rows(const Act& x, float* rowab, int rws, int C, int mode = LN_PLAIN, const float* gw = nullptr, float* maps = nullptr, int H = 0, int W = 0) {
    if (x.lnp && x.lnp->p && x.ln_parts > 0)
      ok(op_layer_norm_from_partials(ctx, x.lnp->as<float>(), x.ln_parts, rws, C, 1e-5f, mode, gw, maps, H, W, rowab));
    else
      ok(op_layer_norm(ctx, x.p(), x.ldv(), nullptr, C, nullptr, nullptr, rws, C, 1e-5f, mode, gw, maps, H, W, rowab));
  }
  // y = LN(x) W^T + b through the folded form (w built by Builder::*_ln)
  void gemm_ln(const void* A, int lda, int M, const LinW& w, const float* rowab, void* C, int ldc, int extra_epi = 0) {
    GemmArgs g;
    g.A = A; g.W = w.w.p; g.C = C; g.M = M; g.N = w.N; g.K = w.K; g.lda = lda; g.ldc = ldc;
    g.bias = (const float*)w.b.p; g.rowab = rowab; g.colsum = (const float*)w.colsum.p;
    g.epi = EPI_BIAS | EPI_ROWAFF | extra_epi;
    ok(op_gemm(ctx, g));
  }

};

}  // namespace star
