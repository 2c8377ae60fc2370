// unet.cpp -- the ControlledV2VUNet graph executor: weight repacking and the static schedule of
// HIP kernels for one denoiser forward (reference: ControlledV2VUNet.forward unet_v2v.py:1717-1809,
// VideoControlNet.forward :2134-2206 and the block internals cited in ops below).
//
// Activations are channels-last token matrices [F*H*W, C] for the whole forward; the reference's
// (b f) c h w <-> b c f h w <-> (b h w) f c rearranges never materialise: spatial kernels index
// tokens by (frame, y, x), temporal kernels stride over frames by H*W rows.
#include "unet.h"
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace star {

UNetModel::~UNetModel() {
  for (auto& kv : graphs) {
    UNetGraph& g = kv.second;
    rt::graph_destroy(g.exec);
    for (float* p : {g.xt, g.hint, g.y[0], g.y[1], g.out[0], g.out[1], g.tsin}) if (p) rt::dev_free(p);
  }
  rt::stream_destroy(gstream);
  for (void* p : owned) rt::dev_free(p);
}

// ------------------------------------------------------------------ weight staging / repacking
namespace {

struct UBuilder : Builder {
  ResW res(const std::string& p) {
    ResW r;
    r.gn1 = norm(p + ".in_layers.0");
    r.conv1 = conv3x3(p + ".in_layers.2");
    r.emb = linear(p + ".emb_layers.1");
    r.gn2 = norm(p + ".out_layers.0");
    r.conv2 = conv3x3(p + ".out_layers.3");
    r.cin = r.gn1.C; r.cout = r.gn2.C;
    r.has_skip = ctx->host_tensors.count(p + ".skip_connection.weight") > 0;
    if (r.has_skip) r.skip = linear(p + ".skip_connection");
    const int ci[4] = {2, 3, 3, 3};
    for (int k = 0; k < 4; ++k) {
      const std::string q = p + ".temopral_conv.conv" + std::to_string(k + 1);
      r.tgn[k] = norm(q + ".0");
      r.tconv[k] = tconv(q + "." + std::to_string(ci[k]));
    }
    return r;
  }
  TBlockW tblock(const std::string& p, bool spatial) {
    TBlockW t;
    // norm1 / norm2 / norm3 feed only the projections behind them: each LayerNorm is folded into its GEMM (graph.h: fold_ln);
    // what is left of it at run time is one read of the row for (a_m, b_m)
    t.qkv1 = fused_ln({p + ".attn1.to_q", p + ".attn1.to_k", p + ".attn1.to_v"}, p + ".norm1");
    t.out1 = linear(p + ".attn1.to_out.0");
    if (spatial) {
      t.q2 = linear_ln(p + ".attn2.to_q", p + ".norm2", false);
      t.kv2 = fused({p + ".attn2.to_k", p + ".attn2.to_v"});
    } else {
      t.qkv2 = fused_ln({p + ".attn2.to_q", p + ".attn2.to_k", p + ".attn2.to_v"}, p + ".norm2");
      t.local2 = raw_f32(p + ".local2.conv1.weight");
      if (t.qkv1.K == 320 && t.qkv1.N == 960) {   // level-0 width: head-ordered copies for the fused projection + attention kernel
        t.qkv1_tq = fused_ln_heads({p + ".attn1.to_q", p + ".attn1.to_k", p + ".attn1.to_v"}, p + ".norm1", 5);
        t.qkv2_tq = fused_ln_heads({p + ".attn2.to_q", p + ".attn2.to_k", p + ".attn2.to_v"}, p + ".norm2", 5);
      }
    }
    t.out2 = linear(p + ".attn2.to_out.0");
    t.ff1 = geglu_ln(p + ".ff.net.0.proj", p + ".norm3");
    t.ff2 = linear(p + ".ff.net.2");
    t.local1 = raw_f32(p + ".local1.conv1.weight");
    return t;
  }
  STW st(const std::string& p, int heads) {
    STW s;
    s.norm = norm(p + ".norm"); s.C = s.norm.C; s.heads = heads;
    s.proj_in = linear(p + ".proj_in"); s.proj_out = linear(p + ".proj_out");
    s.tb = tblock(p + ".transformer_blocks.0", true);
    s.tb.ffpo = compose_ff2_proj_out(p + ".transformer_blocks.0.ff.net.2", p + ".proj_out");
    return s;
  }
  TTW tt(const std::string& p, int heads) {
    TTW s;
    s.norm = norm(p + ".norm"); s.C = s.norm.C; s.heads = heads;
    s.proj_in = linear(p + ".proj_in"); s.proj_out = linear(p + ".proj_out");
    s.inner = s.proj_in.N;
    s.tb = tblock(p + ".transformer_blocks.0", false);
    s.tb.ffpo = compose_ff2_proj_out(p + ".transformer_blocks.0.ff.net.2", p + ".proj_out");
    return s;
  }
};

// ------------------------------------------------------------------ forward helpers
struct Fwd : Runner {
  // ResBlock._forward + TemporalConvBlock_v2 (unet_v2v.py:666-692, 1266-1277)
  Act res_block(const ResW& r, Act x) {
    const int R = rows(x);
    Act n1 = make(r.cin, x.H, x.W);
    gn(x, r.gn1, n1, false, 1e-5f, true);
    // conv bias + Linear(SiLU(emb)) folded into one per-channel vector (batch = 1: same for every frame)
    Buf bias1(ctx, (size_t)r.cout * 4);
    ok(op_gemv(ctx, emb, r.emb.w.p, (const float*)r.emb.b.p, bias1.as<float>(), r.cout, r.emb.K, true, false));
    ok(op_vec_add_f32(ctx, bias1.as<float>(), (const float*)r.conv1.b.p, r.cout));
    Act h1 = make(r.cout, x.H, x.W);
    conv3x3(n1, r.conv1, h1, A_CONV3X3, 1, 1, 1, nullptr, bias1.as<float>(), 0, nullptr, 0, 1, true);   // + h1's GroupNorm partials (gn2)
    n1.drop();
    // out_layers.0 in place: nobody else reads h1 (STAR_NO_GN_INPLACE=1: a separate output tensor, the round-5 form; A/B switch, read once)
    static const bool gn_inplace = std::getenv("STAR_NO_GN_INPLACE") == nullptr;
    Act n2;
    if (gn_inplace) { gn(h1, r.gn2, h1, false, 1e-5f, true); n2 = std::move(h1); }
    else { n2 = make(r.cout, x.H, x.W); gn(h1, r.gn2, n2, false, 1e-5f, true); }
    h1.drop();
    Act skip;
    const void* sp = x.p();
    if (r.has_skip) {
      skip = make(r.cout, x.H, x.W);
      gemm(x.p(), x.C, R, r.skip, skip.p(), r.cout);
      sp = skip.p();
    }
    Act h2 = make(r.cout, x.H, x.W);
    conv3x3(n2, r.conv2, h2, A_CONV3X3, 1, 1, 1, sp, nullptr, 0, nullptr, 0, 1, true);   // + h2's partials (first temporal GroupNorm)
    n2.drop(); skip.drop(); x.drop();
    // temporal conv block: 4 x [GN(whole chunk) + SiLU + Conv3d(3,1,1)] + identity
    Act cur;  // null => h2
    for (int k = 0; k < 4; ++k) {
      // the first norm reads h2, which the block's identity adds back at the end; the other three normalise their input in place
      Act nn;
      if (k == 0 || !gn_inplace) { nn = make(r.cout, h2.H, h2.W); gn(k == 0 ? h2 : cur, r.tgn[k], nn, true, 1e-5f, true); }
      else { gn(cur, r.tgn[k], cur, true, 1e-5f, true); nn = std::move(cur); }
      Act nxt = make(r.cout, h2.H, h2.W);
      GemmArgs g;
      g.A = nn.p(); g.W = r.tconv[k].w.p; g.C = nxt.p(); g.M = R; g.N = r.cout; g.K = 3 * r.cout; g.lda = r.cout; g.ldc = r.cout;
      g.mode = A_TCONV3; g.Cin = r.cout; g.HW = h2.H * h2.W; g.F = F;
      g.bias = (const float*)r.tconv[k].b.p; g.epi = EPI_BIAS;
      if (k == 3) { g.res = h2.p(); g.ldr = r.cout; g.epi |= EPI_RES; }
      gemm_to(g, &nxt, k < 3 || want_out_stats);   // partials for the next temporal GroupNorm / for the module behind this block
      cur = std::move(nxt);
    }
    return cur;
  }

  // shared tail of both transformer kinds: LN3 -> GEGLU FF -> +res ; then proj_out (+ x_in).
  // Round 6: ff2 (+ h2) and proj_out are ONE GEMM over the operand [h2 | g] with the composed weight (graph.h: compose_ff2_proj_out):
  // h2 is allocated by make_h2() as the head of rows of 5 x inner elements, the GEGLU projection writes g behind it.  h3 = ff2(g) + h2
  // is never written or read (level 0: 1.1 GB per block).  STAR_NO_FFPO=1: the two GEMMs of rounds 1-5 (A/B switch, read once).
  static bool ffpo_enabled() { static const bool v = std::getenv("STAR_NO_FFPO") == nullptr; return v; }
  Act make_h2(const TBlockW& tb, int inner, int H, int W) {
    if (ffpo_enabled() && tb.ffpo.w.p && tb.ffpo.K == 5 * inner) return make_wide(inner, H, W, 5 * inner);
    return make(inner, H, W);
  }
  void ff_and_out(const TBlockW& tb, Act& h2, int R, int inner, const LinW& proj_out, const Act& x_in, Act& out) {
    Buf ab(ctx, (size_t)R * 2 * 4);
    ln_rows(h2, ab.as<float>(), R, inner);                             // norm3, folded into the GEGLU projection (row statistics from h2's producer where it left them)
    if (h2.ld == 5 * inner) {
      if (!ab.p) { rc = ctx->fail("out of device memory (ff)"); return; }
      char* gp = (char*)h2.p() + (size_t)inner * es;                   // g = columns [inner, 5 inner) of h2's rows
      gemm_ln(h2.p(), h2.ld, R, tb.ff1, ab.as<float>(), gp, h2.ld, EPI_GEGLU);
      ab.reset();
      gemm(h2.p(), h2.ld, R, tb.ffpo, out.p(), out.C, x_in.p(), x_in.C, 0, nullptr, want_out_stats ? &out : nullptr);
      h2.drop();
      return;
    }
    Buf g(ctx, (size_t)R * inner * 4 * es);
    if (!g.p || !ab.p) { rc = ctx->fail("out of device memory (ff)"); return; }
    gemm_ln(h2.p(), inner, R, tb.ff1, ab.as<float>(), g.p, inner * 4, EPI_GEGLU);
    ab.reset();
    Act h3 = make(inner, x_in.H, x_in.W);
    gemm(g.p, inner * 4, R, tb.ff2, h3.p(), inner, h2.p(), inner);
    g.reset(); h2.drop();
    gemm(h3.p(), inner, R, proj_out, out.p(), out.C, x_in.p(), x_in.C, 0, nullptr, want_out_stats ? &out : nullptr);
  }

  // SpatialTransformer.forward + BasicTransformerBlock space branch (unet_v2v.py:297-317, 466-477), in two halves:
  // st_self  = GroupNorm, proj_in, LIEM gate, LN1, QKV, self-attention, to_out (+res): does NOT depend on the text context;
  // st_cross = LN2, cross-attention to the context, LN3, GEGLU FF, proj_out (+ x_in): one per guidance branch.
  struct STMid { Act x_in, h1; };
  STMid st_self(const STW& s, const Act& x) {
    const int R = rows(x), C = s.C, HW = x.H * x.W;
    const TBlockW& tb = s.tb;
    STMid mid; mid.x_in = x;
    Act n = make(C, x.H, x.W);
    gn(x, s.norm, n, false, 1e-6f, false);
    Act h = make(C, x.H, x.W);
    gemm(n.p(), C, R, s.proj_in, h.p(), C, nullptr, 0, 0, nullptr, nullptr, &h);   // + h's row statistics
    // LIEM spatial gate + LN1 (folded into the QKV projection: rows of h are read for their statistics only -- or not at all)
    Buf maps(ctx, (size_t)R * 2 * 4), ab(ctx, (size_t)R * 2 * 4);
    ln_rows(h, nullptr, R, C, LN_STATS_ONLY, nullptr, maps.as<float>());
    ln_rows(h, ab.as<float>(), R, C, LN_GATE_MAP, (const float*)tb.local1.p, maps.as<float>(), x.H, x.W);
    maps.reset();
    // self attention over the H*W tokens of each frame
    Buf qkv(ctx, (size_t)R * 3 * C * es);
    if (!qkv.p || !ab.p) { rc = ctx->fail("out of device memory (qkv)"); return mid; }
    gemm_ln(h.p(), C, R, tb.qkv1, ab.as<float>(), qkv.p, 3 * C);
    ab.reset();
    {
      AttnArgs a;
      a.Q = qkv.p; a.K = (char*)qkv.p + (size_t)C * es; a.V = (char*)qkv.p + (size_t)2 * C * es; a.O = n.p();
      a.ldq = a.ldk = a.ldv = 3 * C; a.ldo = C;
      a.bsq = a.bsk = a.bsv = (long long)HW * 3 * C; a.bso = (long long)HW * C;
      a.Nq = a.Nk = HW; a.heads = s.heads; a.batch = F; a.scale = 0.125f;
      ok(op_flash_attn(ctx, a));
    }
    qkv.reset();
    mid.h1 = make(C, x.H, x.W);
    gemm(n.p(), C, R, tb.out1, mid.h1.p(), C, h.p(), C, 0, nullptr, nullptr, &mid.h1);   // + h1's row statistics (norm2)
    return mid;
  }
  Act st_cross(const STW& s, const STMid& mid, const void* context_T) {
    const Act& x = mid.x_in;
    const int R = rows(x), C = s.C, HW = x.H * x.W;
    const TBlockW& tb = s.tb;
    Act n = make(C, x.H, x.W);
    Buf ab(ctx, (size_t)R * 2 * 4);
    ln_rows(mid.h1, ab.as<float>(), R, C);                              // norm2, folded into to_q
    Act q2 = make(C, x.H, x.W);
    if (!ab.p) { rc = ctx->fail("out of device memory"); return n; }
    gemm_ln(mid.h1.p(), C, R, tb.q2, ab.as<float>(), q2.p(), C);
    ab.reset();
    Buf kv(ctx, (size_t)77 * 2 * C * es);
    gemm(context_T, ctx_dim, 77, tb.kv2, kv.p, 2 * C);
    {
      AttnArgs a;
      a.Q = q2.p(); a.K = kv.p; a.V = (char*)kv.p + (size_t)C * es; a.O = n.p();
      a.ldq = C; a.ldk = a.ldv = 2 * C; a.ldo = C;
      a.bsq = a.bso = (long long)HW * C; a.bsk = a.bsv = 0;
      a.Nq = HW; a.Nk = 77; a.heads = s.heads; a.batch = F; a.scale = 0.125f;
      ok(op_flash_attn(ctx, a));
    }
    q2.drop(); kv.reset();
    Act h2 = make_h2(tb, C, x.H, x.W);
    gemm(n.p(), C, R, tb.out2, h2.p(), h2.ldv(), mid.h1.p(), C, 0, nullptr, nullptr, &h2);   // + h2's row statistics (norm3)
    n.drop();
    Act out = make(C, x.H, x.W);
    ff_and_out(tb, h2, R, C, s.proj_out, x, out);
    return out;
  }
  Act spatial_transformer(const STW& s, Act x) {
    STMid mid = st_self(s, x);
    return st_cross(s, mid, context);
  }

  // TemporalTransformer.forward + BasicTransformerBlock temp branch (unet_v2v.py:1034-1092, 479-490)
  Act temporal_transformer(const TTW& s, Act x) {
    static const bool use_tq = std::getenv("STAR_NO_TQ") == nullptr;   // A/B switch, read once
    const int R = rows(x), C = s.C, I = s.inner, HW = x.H * x.W;
    const TBlockW& tb = s.tb;
    // GroupNorm over the whole chunk (no activation) feeds only proj_in: folded into its weights (norm.h: gn_fold_weights_kernel) --
    // the statistics pass stays, the apply pass and the normalised tensor do not exist; proj_in reads x itself.  The group means are
    // subtracted with the ROUNDED weights (bias'), so the fold is as accurate as norm + Linear for any mean / spread ratio.
    // STAR_NO_GNFOLD=1: the unfolded pair (A/B switch, read once).
    static const bool fold_gn = std::getenv("STAR_NO_GNFOLD") == nullptr;
    Act h = make(I, x.H, x.W);
    if (fold_gn) {
      Buf ab(ctx, (size_t)C * 2 * 4), mu(ctx, (size_t)C * 4), w2(ctx, (size_t)I * C * es), b2(ctx, (size_t)I * 4);
      if (!ab.p || !mu.p || !w2.p || !b2.p) { rc = ctx->fail("out of device memory (folded GroupNorm)"); return x; }
      if (x.gnp && x.gnp->p)   // x's producer (the spatial transformer's proj_out) left the partial statistics
        ok(op_group_norm_fused(ctx, x.p(), x.C, nullptr, 0, (const float*)s.norm.g.p, (const float*)s.norm.b.p, R, C, R, 1e-6f, false, x.gnp->as<float>(),
                               ab.as<float>(), mu.as<float>()));
      else
      ok(op_group_norm_stats(ctx, x.p(), x.C, (const float*)s.norm.g.p, (const float*)s.norm.b.p, R, C, R, 1e-6f, ab.as<float>(), mu.as<float>()));
      ok(op_gn_fold_weights(ctx, s.proj_in.w.p, (const float*)s.proj_in.b.p, ab.as<float>(), w2.p, b2.as<float>(), I, C, mu.as<float>()));
      LinW folded; folded.N = I; folded.K = C; folded.w.p = w2.p;
      gemm(x.p(), C, R, folded, h.p(), I, nullptr, 0, 0, b2.as<float>(), nullptr, &h);   // + h's row statistics (temporal gate + norm1)
    } else {
      Act xn = make(C, x.H, x.W);
      gn(x, s.norm, xn, true, 1e-6f, false);
      gemm(xn.p(), C, R, s.proj_in, h.p(), I, nullptr, 0, 0, nullptr, nullptr, &h);
    }
    Act l = make(I, x.H, x.W);
    Buf qkv(ctx, (size_t)R * 3 * I * es), ab(ctx, (size_t)R * 2 * 4);
    if (!qkv.p || !ab.p) { rc = ctx->fail("out of device memory (qkv)"); return x; }
    Act cur = std::move(h);
    for (int pass = 0; pass < 2; ++pass) {
      // temporal LIEM gate + LayerNorm, folded into the QKV projection
      ln_rows(cur, ab.as<float>(), R, I, LN_GATE_LINEAR, (const float*)(pass == 0 ? tb.local1.p : tb.local2.p));
      const LinW& tq = pass == 0 ? tb.qkv1_tq : tb.qkv2_tq;
      if (use_tq && tq.w.p && temporal_qkv_attn_covers(I, s.heads, F)) {
        // projection + attention in one kernel (gemm_tq.h): q | k | v never reach HBM; bit-identical to the two kernels below
        TqArgs a;
        a.A = cur.p(); a.lda = I; a.W = tq.w.p; a.bias = (const float*)tq.b.p; a.colsum = (const float*)tq.colsum.p; a.rowab = ab.as<float>();
        a.O = l.p(); a.ldo = I; a.HW = HW; a.F = F; a.C = I; a.heads = s.heads; a.scale = 0.125f;
        ok(op_temporal_qkv_attn(ctx, a));
      } else {
      gemm_ln(cur.p(), I, R, pass == 0 ? tb.qkv1 : tb.qkv2, ab.as<float>(), qkv.p, 3 * I);
      TAttnArgs a;
      a.Q = qkv.p; a.K = (char*)qkv.p + (size_t)I * es; a.V = (char*)qkv.p + (size_t)2 * I * es; a.O = l.p();
      a.ldq = a.ldk = a.ldv = 3 * I; a.ldo = I; a.F = F; a.HW = HW; a.heads = s.heads; a.scale = 0.125f;
      ok(op_temporal_attn(ctx, a));
      }
      Act nx = pass == 0 ? make(I, x.H, x.W) : make_h2(tb, I, x.H, x.W);     // the second pass's output is the FeedForward's h2
      gemm(l.p(), I, R, pass == 0 ? tb.out1 : tb.out2, nx.p(), nx.ldv(), cur.p(), I, 0, nullptr, nullptr, &nx);   // + nx's row statistics (next pass / norm3)
      cur = std::move(nx);
    }
    qkv.reset(); l.drop(); ab.reset();
    Act out = make(C, x.H, x.W);
    ff_and_out(tb, cur, R, I, s.proj_out, x, out);
    return out;
  }

  Act down(const ConvW& c, Act x) {   // Downsample: conv 3x3 stride 2 padding (2,1) (unet_v2v.py:709-722)
    Act y = make(c.C, x.H / 2 + 1, x.W / 2);
    if ((x.H + 4 - 3) / 2 + 1 != y.H || (x.W + 2 - 3) / 2 + 1 != y.W) { rc = ctx->fail("downsample: illegal latent size"); return y; }
    conv3x3(x, c.conv, y, A_CONV3X3, 2, 2, 1, nullptr, nullptr, 0, nullptr, 0, 1, want_out_stats);
    return y;
  }
  Act up(const ConvW& c, Act x) {     // Upsample: nearest x2, rows [1:-1], conv 3x3 (unet_v2v.py:556-567)
    Act y = make(c.C, 2 * x.H - 2, 2 * x.W);
    conv3x3(x, c.conv, y, A_CONV3X3_UP, 1, 1, 1, nullptr);
    return y;
  }
  // ---- classifier-free-guidance pair: up to two guidance branches (cond / uncond text context) share every activation
  // that does not depend on the context, i.e. everything up to and including the self-attention of the FIRST spatial
  // transformer of each net (stem conv, stem temporal transformer, first ResBlock, 28 ms of L0 self-attention ...).
  struct PA { Act a[2]; bool same = true; };
  int nb = 1;
  // set by the block walker before each module: the module BEHIND this one starts with a GroupNorm of this module's output
  // (ResBlock in_layers.0, SpatialTransformer.norm, TemporalTransformer.norm, the head's out.0), so the module's last kernel
  // also writes the output's partial statistics
  bool want_out_stats = false;
  // modules of one block in sequence; next_kind: what consumes the block's output (-1: not a GroupNorm, e.g. the decoder's concat)
  PA run_seq(const Net& net, const std::vector<Mod>& mods, PA x, int next_kind) {
    for (size_t i = 0; i < mods.size(); ++i) {
      const int nk = i + 1 < mods.size() ? mods[i + 1].kind : next_kind;
      want_out_stats = nk == M_RES || nk == M_ST || nk == M_TT;
      x = run(net, mods[i], std::move(x));
    }
    want_out_stats = false;
    return x;
  }
  const void* contexts[2] = {nullptr, nullptr};
  PA run(const Net& net, const Mod& m, PA x) {
    PA y;
    if (m.kind == M_ST) {
      const STW& st = net.st[m.idx];
      if (x.same) {
        STMid mid = st_self(st, x.a[0]);
        for (int b = 0; b < nb; ++b) y.a[b] = st_cross(st, mid, contexts[b]);
      } else {
        for (int b = 0; b < nb; ++b) { STMid mid = st_self(st, x.a[b]); y.a[b] = st_cross(st, mid, contexts[b]); }
      }
      y.same = (nb == 1);
      if (nb == 1) y.a[1] = y.a[0];
      return y;
    }
    if (x.same) {
      y.a[0] = run(net, m, x.a[0]);
      y.a[1] = y.a[0];
      y.same = true;
    } else {
      for (int b = 0; b < nb; ++b) y.a[b] = run(net, m, x.a[b]);
      y.same = false;
    }
    return y;
  }
  Act run(const Net& net, const Mod& m, Act x) {
    switch (m.kind) {
      case M_RES: return res_block(net.res[m.idx], std::move(x));
      case M_ST: return spatial_transformer(net.st[m.idx], std::move(x));
      case M_TT: return temporal_transformer(net.tt[m.idx], std::move(x));
      case M_DOWN: return down(net.convs[m.idx], std::move(x));
      case M_UP: return up(net.convs[m.idx], std::move(x));
    }
    rc = ctx->fail("bad module kind");
    return x;
  }
};

static void host_sinusoidal(long long t, int dim, std::vector<float>& out) {   // unet_v2v.py:96-108
  const int half = dim / 2;
  out.assign(dim, 0.f);
  for (int i = 0; i < half; ++i) {
    const float freq = powf(10000.f, -(float)i / (float)half);
    const float s = (float)t * freq;
    out[i] = cosf(s);
    out[half + i] = sinf(s);
  }
}

static int build_net(UBuilder& b, const UNetCfg& cfg, bool control, Net& net) {
  const std::string P = control ? "VideoControlNet." : "";
  net.time0 = b.linear(P + "time_embed.0");
  net.time2 = b.linear(P + "time_embed.2");
  net.stem = b.conv3x3_im2col64(P + "input_blocks.0.0");
  const int hd = cfg.head_dim;
  std::vector<int> enc; enc.push_back(cfg.dim);
  for (int i = 0; i < cfg.n_levels; ++i) enc.push_back(cfg.dim * cfg.dim_mult[i]);
  std::vector<int> shortcut;
  std::vector<int> zero_dims;
  // block 0: stem conv (separate) + stem TemporalTransformer
  {
    net.tt.push_back(b.tt(P + "input_blocks.0.1", cfg.num_heads));
    net.input_blocks.push_back({Mod{M_TT, (int)net.tt.size() - 1}});
    shortcut.push_back(cfg.dim); zero_dims.push_back(cfg.dim);
  }
  int idx = 1, level = 0;
  for (int i = 0; i < cfg.n_levels; ++i) {
    for (int j = 0; j < cfg.num_res_blocks; ++j) {
      const int cout = enc[i + 1];
      const std::string bp = P + "input_blocks." + std::to_string(idx);
      std::vector<Mod> mods;
      net.res.push_back(b.res(bp + ".0")); mods.push_back(Mod{M_RES, (int)net.res.size() - 1});
      if (level < cfg.attn_levels) {
        net.st.push_back(b.st(bp + ".1", cout / hd)); mods.push_back(Mod{M_ST, (int)net.st.size() - 1});
        net.tt.push_back(b.tt(bp + ".2", cout / hd)); mods.push_back(Mod{M_TT, (int)net.tt.size() - 1});
      }
      net.input_blocks.push_back(mods);
      shortcut.push_back(cout); zero_dims.push_back(cout);
      ++idx;
      if (i != cfg.n_levels - 1 && j == cfg.num_res_blocks - 1) {
        ConvW c; c.C = cout; c.conv = b.conv3x3(P + "input_blocks." + std::to_string(idx) + ".op");
        net.convs.push_back(c);
        net.input_blocks.push_back({Mod{M_DOWN, (int)net.convs.size() - 1}});
        shortcut.push_back(cout); zero_dims.push_back(cout);
        ++level; ++idx;
      }
    }
  }
  const int cm = enc.back();
  net.res.push_back(b.res(P + "middle_block.0")); net.middle.push_back(Mod{M_RES, (int)net.res.size() - 1});
  net.st.push_back(b.st(P + "middle_block.1", cm / hd)); net.middle.push_back(Mod{M_ST, (int)net.st.size() - 1});
  net.tt.push_back(b.tt(P + "middle_block.2", cm / hd)); net.middle.push_back(Mod{M_TT, (int)net.tt.size() - 1});
  net.res.push_back(b.res(P + "middle_block.3")); net.middle.push_back(Mod{M_RES, (int)net.res.size() - 1});
  if (control) {
    for (size_t i = 0; i < zero_dims.size(); ++i) net.zero_convs.push_back(b.linear(P + "zero_convs." + std::to_string(i) + ".0"));
    net.middle_out = b.linear(P + "middle_block_out.0");
    net.hint = b.conv3x3_im2col64(P + "input_hint_block");
  } else {
    int oidx = 0;
    for (int i = 0; i < cfg.n_levels; ++i) {
      const int cout = cfg.dim * cfg.dim_mult[cfg.n_levels - 1 - i];
      for (int j = 0; j < cfg.num_res_blocks + 1; ++j) {
        const std::string bp = "output_blocks." + std::to_string(oidx);
        std::vector<Mod> mods;
        net.res.push_back(b.res(bp + ".0")); mods.push_back(Mod{M_RES, (int)net.res.size() - 1});
        int k = 1;
        if (level < cfg.attn_levels) {
          net.st.push_back(b.st(bp + ".1", cout / hd)); mods.push_back(Mod{M_ST, (int)net.st.size() - 1});
          net.tt.push_back(b.tt(bp + ".2", cout / hd)); mods.push_back(Mod{M_TT, (int)net.tt.size() - 1});
          k = 3;
        }
        if (i != cfg.n_levels - 1 && j == cfg.num_res_blocks) {
          ConvW c; c.C = cout; c.conv = b.conv3x3(bp + "." + std::to_string(k) + ".conv");
          net.convs.push_back(c); mods.push_back(Mod{M_UP, (int)net.convs.size() - 1});
          --level;
        }
        net.output_blocks.push_back(mods);
        ++oidx;
      }
    }
    net.out_norm = b.norm("out.0");
    net.out_conv = b.conv3x3("out.2", 4);
  }
  return b.err.empty() ? 0 : 1;
}

}  // namespace

int unet_build(Ctx* ctx, const UNetCfg& cfg) {
  auto model = std::make_shared<UNetModel>();
  model->cfg = cfg;
  UBuilder b{{ctx, &model->owned, ""}};
  if (build_net(b, cfg, false, model->main) || build_net(b, cfg, true, model->control)) return ctx->fail("unet_build: " + b.err);
  if (!b.err.empty()) return ctx->fail("unet_build: " + b.err);
  ctx->unet = model;
  ctx->host_tensors.clear();
  return 0;
}

// the timestep MLP on the sinusoidal row `tsin` (device, fp32 [dim]; written by the caller: the only host-computed input of a forward)
static int time_embedding(Ctx* ctx, const Net& net, const float* tsin, int dim, int E, Buf& tmp_mid, float* out) {
  if (op_gemv(ctx, tsin, net.time0.w.p, (const float*)net.time0.b.p, tmp_mid.as<float>(), E, dim, false, true)) return 1;
  return op_gemv(ctx, tmp_mid.as<float>(), net.time2.w.p, (const float*)net.time2.b.p, out, E, E, false, false);
}

static int unet_forward_impl(Ctx* ctx, const float* xt, const float* tsin, const float* const* ys, const float* hint, float* const* outs, int nb,
                             int F, int H, int W, void* const* control_tap, int n_tap);

// One forward = a fixed sequence of a few thousand kernel launches (about 3400 per CFG pair at cfg2 size) that depends on
// (branches, frames, latent size) only.  With star_unet_graph(ctx, 1) the second forward of a shape is captured into a hipGraph and later ones replay it: the caller's tensors
// are copied into the staging buffers the graph was captured with, the sinusoidal row is refreshed, one hipGraphLaunch replaces
// the launches.  The first forward of a shape always runs eagerly: it sizes the activation pool and sets the kernels' LDS
// attributes, neither of which may happen under capture.
static int unet_forward_graph(Ctx* ctx, UNetModel& M, const float* xt, long long t, const float* const* ys, const float* hint,
                              float* const* outs, int nb, int F, int H, int W, bool* ran) {
  *ran = false;
  const UNetCfg& cfg = M.cfg;
  UNetGraph& g = M.graphs[{nb, F, H, W}];
  if (g.captured && g.pool_gen != ctx->pool.generation()) { rt::graph_destroy(g.exec); g.captured = false; g.n_x = 0; }   // trimmed pool: the eager path re-warms it
  if (g.n_x == 0) {   // first sight of this shape (or a stale graph): staging buffers now, eager forward by the caller
    g.n_x = (size_t)cfg.in_dim * F * H * W; g.n_hint = (size_t)4 * F * H * W; g.n_y = (size_t)77 * cfg.context_dim; g.n_out = (size_t)cfg.out_dim * F * H * W;
    auto need = [&](float*& p, size_t n) { return p ? 0 : rt::dev_malloc((void**)&p, n * 4); };
    int rc = need(g.xt, g.n_x) | need(g.hint, g.n_hint) | need(g.tsin, (size_t)cfg.dim);
    for (int b = 0; b < nb; ++b) rc |= need(g.y[b], g.n_y) | need(g.out[b], g.n_out);
    if (rc) { g.n_x = 0; return ctx->fail("unet_forward: out of device memory for the graph's staging buffers"); }
    return 0;
  }
  if (ctx->pool.in_use() != 0) return 0;   // a pool block is live across this call (ctx.h: Pool invariant): a replay could overwrite it -- eager forward
  if (!M.gstream && rt::stream_create(&M.gstream)) return ctx->fail("unet_forward: cannot create the graph stream");
  hipStream_t user = ctx->stream, gs = M.gstream;
  if (rt::stream_wait_stream(gs, user)) return ctx->fail("unet_forward: stream ordering failed");
  std::vector<float> se;
  host_sinusoidal(t, cfg.dim, se);
  rt::memcpy_h2d(g.tsin, se.data(), (size_t)cfg.dim * 4, gs);
  rt::stream_sync(gs);   // `se` is a stack temporary (the eager path has the same synchronisation point)
  rt::memcpy_d2d(g.xt, xt, g.n_x * 4, gs);
  rt::memcpy_d2d(g.hint, hint, g.n_hint * 4, gs);
  for (int b = 0; b < nb; ++b) rt::memcpy_d2d(g.y[b], ys[b], g.n_y * 4, gs);
  if (!g.captured) {
    if (rt::capture_begin(gs)) return ctx->fail(std::string("unet_forward: hipStreamBeginCapture failed: ") + rt::last_error_string());
    const uint64_t gen0 = ctx->pool.generation();
    ctx->stream = gs;
    const float* gys[2] = {g.y[0], g.y[1]};
    float* gouts[2] = {g.out[0], g.out[1]};
    const int rc = unet_forward_impl(ctx, g.xt, g.tsin, gys, g.hint, gouts, nb, F, H, W, nullptr, 0);
    ctx->stream = user;
    rt::GraphExec ex;
    const int rc2 = rt::capture_end(gs, &ex);
    if (rc) { if (!rc2) rt::graph_destroy(ex); return rc; }
    if (rc2) return ctx->fail(std::string("unet_forward: graph capture / instantiation failed: ") + rt::last_error_string());
    if (ctx->pool.generation() != gen0) {   // the pool dropped its cache while the forward was being recorded (an allocation did not fit):
      rt::graph_destroy(ex);                // blocks recorded earlier in this capture may be gone -- nothing was launched, start over eagerly
      g.n_x = 0;
      return 0;
    }
    g.exec = ex; g.captured = true; g.pool_gen = gen0;
  }
  if (rt::graph_launch(g.exec, gs)) return ctx->fail(std::string("unet_forward: hipGraphLaunch failed: ") + rt::last_error_string());
  for (int b = 0; b < nb; ++b) rt::memcpy_d2d(outs[b], g.out[b], g.n_out * 4, gs);
  if (rt::stream_wait_stream(user, gs)) return ctx->fail("unet_forward: stream ordering failed");
  *ran = true;
  return 0;
}

int unet_forward_n(Ctx* ctx, const float* xt, long long t, const float* const* ys, const float* hint, float* const* outs, int nb,
                   int F, int H, int W, void* const* control_tap, int n_tap) {
  if (!ctx->unet) return ctx->fail("unet_forward: no model built (star_unet_build)");
  if (nb < 1 || nb > 2) return ctx->fail("unet_forward: 1 or 2 guidance branches");
  UNetModel& M = *ctx->unet;
  const UNetCfg& cfg = M.cfg;
  if (F < 1 || F > 128) return ctx->fail("unet_forward: 1..128 frames per chunk");
  {  // legal latent sizes: every Downsample/Upsample pair must round-trip (H = 2 mod 8, W = 0 mod 8 for 3 levels)
    int h = H, w = W;
    for (int i = 0; i < cfg.n_levels - 1; ++i) { if (w % 2) return ctx->fail("unet_forward: illegal latent width"); h = h / 2 + 1; w /= 2; }
    for (int i = 0; i < cfg.n_levels - 1; ++i) { h = 2 * h - 2; w *= 2; }
    if (h != H || w != W) return ctx->fail("unet_forward: illegal latent size (need H = 2 mod 8, W = 0 mod 8)");
  }
  if (ctx->unet_graph && rt::graphs_available && !ctx->profiling && !control_tap) {
    bool ran = false;
    if (unet_forward_graph(ctx, M, xt, t, ys, hint, outs, nb, F, H, W, &ran)) return 1;
    if (ran) return 0;
  }
  Buf t_in(ctx, (size_t)cfg.dim * 4);
  if (!t_in.p) return ctx->fail("out of device memory");
  {
    std::vector<float> se;
    host_sinusoidal(t, cfg.dim, se);
    rt::memcpy_h2d(t_in.p, se.data(), (size_t)cfg.dim * 4, ctx->stream);
    rt::stream_sync(ctx->stream);  // `se` is a stack temporary
  }
  return unet_forward_impl(ctx, xt, t_in.as<float>(), ys, hint, outs, nb, F, H, W, control_tap, n_tap);
}

static int unet_forward_impl(Ctx* ctx, const float* xt, const float* tsin, const float* const* ys, const float* hint, float* const* outs, int nb,
                             int F, int H, int W, void* const* control_tap, int n_tap) {
  const UNetModel& M = *ctx->unet;
  const UNetCfg& cfg = M.cfg;
  Fwd f; f.ctx = ctx; f.F = F; f.es = ctx->esize(); f.ctx_dim = cfg.context_dim; f.embed_dim = cfg.embed_dim();
  f.nb = nb;
  const int E = cfg.embed_dim();
  const long long tok = (long long)F * H * W;
  Buf ctxT[2];
  for (int b = 0; b < nb; ++b) {
    ctxT[b] = Buf(ctx, (size_t)77 * cfg.context_dim * f.es);
    if (op_cast(ctx, ys[b], ctxT[b].p, (long long)77 * cfg.context_dim)) return 1;
    f.contexts[b] = ctxT[b].p;
  }
  f.context = f.contexts[0];
  Buf emb_main(ctx, (size_t)E * 4), emb_ctrl(ctx, (size_t)E * 4), t_mid(ctx, (size_t)E * 4);
  if (time_embedding(ctx, M.main, tsin, cfg.dim, E, t_mid, emb_main.as<float>())) return 1;
  if (time_embedding(ctx, M.control, tsin, cfg.dim, E, t_mid, emb_ctrl.as<float>())) return 1;

  // im2col rows of x (shared by both nets' stem convs) and of the hint
  Buf xcols(ctx, (size_t)tok * 64 * f.es), hcols(ctx, (size_t)tok * 64 * f.es);
  if (!xcols.p || !hcols.p) return ctx->fail("out of device memory");
  if (op_stem_im2col(ctx, xt, xcols.p, cfg.in_dim, F, H, W)) return 1;
  if (op_stem_im2col(ctx, hint, hcols.p, 4, F, H, W)) return 1;
  using PA = Fwd::PA;
  auto both = [&](const Act& a) { PA p; p.a[0] = a; p.a[1] = a; p.same = true; return p; };
  // y = zero_conv(x) per branch (shared while the branches still coincide)
  auto lin = [&](const PA& x, const LinW& w) {
    PA z;
    for (int b = 0; b < (x.same ? 1 : nb); ++b) {
      z.a[b] = f.make(x.a[b].C, x.a[b].H, x.a[b].W);
      f.gemm(x.a[b].p(), x.a[b].C, f.rows(x.a[b]), w, z.a[b].p(), x.a[b].C);
    }
    if (x.same) z.a[1] = z.a[0];
    z.same = x.same;
    return z;
  };

  // ---------------- VideoControlNet (unet_v2v.py:2134-2206)
  std::vector<PA> control;
  {
    const Net& net = M.control;
    f.emb = emb_ctrl.as<float>();
    Act hc = f.make(cfg.dim, H, W);
    f.gemm(hcols.p, 64, (int)tok, net.hint, hc.p(), cfg.dim);
    hcols.reset();
    Act x0 = f.make(cfg.dim, H, W);
    f.gemm(xcols.p, 64, (int)tok, net.stem, x0.p(), cfg.dim, hc.p(), cfg.dim);   // stem conv + hint (added before the stem TT, :2190-2194)
    hc.drop();
    PA x = both(x0);
    x0.drop();
    for (size_t bi = 0; bi < net.input_blocks.size(); ++bi) {
      const int next_kind = bi + 1 < net.input_blocks.size() ? net.input_blocks[bi + 1][0].kind : net.middle[0].kind;
      x = f.run_seq(net, net.input_blocks[bi], std::move(x), next_kind);
      if (f.rc) return f.rc;
      control.push_back(lin(x, net.zero_convs[bi]));
    }
    x = f.run_seq(net, net.middle, std::move(x), -1);
    if (f.rc) return f.rc;
    control.push_back(lin(x, net.middle_out));
  }
  if (control_tap) {   // star_controlnet_forward: hand the residuals out (channels-last rows, storage dtype) and stop
    if (n_tap != (int)control.size()) return ctx->fail("controlnet_forward: expected " + std::to_string(control.size()) + " residual buffers");
    for (int i = 0; i < n_tap; ++i) {
      const Act& a = control[i].a[0];
      rt::memcpy_d2d(control_tap[i], a.p(), (size_t)f.rows(a) * a.C * f.es, ctx->stream);
    }
    return f.rc;
  }
  // ---------------- main UNet (unet_v2v.py:1765-1808)
  const Net& net = M.main;
  f.emb = emb_main.as<float>();
  std::vector<PA> xs;
  PA x;
  {
    Act x0 = f.make(cfg.dim, H, W);
    f.gemm(xcols.p, 64, (int)tok, net.stem, x0.p(), cfg.dim);
    xcols.reset();
    x = both(x0);
  }
  for (size_t bi = 0; bi < net.input_blocks.size(); ++bi) {
    const int next_kind = bi + 1 < net.input_blocks.size() ? net.input_blocks[bi + 1][0].kind : net.middle[0].kind;
    x = f.run_seq(net, net.input_blocks[bi], std::move(x), next_kind);
    if (f.rc) return f.rc;
    xs.push_back(x);   // skip connection shares the buffer(s)
  }
  x = f.run_seq(net, net.middle, std::move(x), -1);   // (the control residual is added to the middle block's output: its statistics would be stale)
  if (f.rc) return f.rc;
  {
    PA& ctl = control.back();
    PA s2;
    s2.same = x.same && ctl.same;
    for (int b = 0; b < (s2.same ? 1 : nb); ++b) {
      s2.a[b] = f.make(x.a[b].C, x.a[b].H, x.a[b].W);
      f.ok(op_add(ctx, ctl.a[b].p(), x.a[b].p(), s2.a[b].p(), (long long)f.rows(x.a[b]) * x.a[b].C));
    }
    if (s2.same) s2.a[1] = s2.a[0];
    control.pop_back();
    x = std::move(s2);
  }
  for (size_t bi = 0; bi < net.output_blocks.size(); ++bi) {
    PA& skip = xs.back(); PA& ctl = control.back();
    PA cat;
    cat.same = x.same && skip.same && ctl.same;
    for (int b = 0; b < (cat.same ? 1 : nb); ++b) {
      if (skip.a[b].H != x.a[b].H || skip.a[b].W != x.a[b].W) return ctx->fail("unet_forward: skip shape mismatch");
      cat.a[b] = f.make(x.a[b].C + skip.a[b].C, x.a[b].H, x.a[b].W);
      // the block behind it starts with a GroupNorm of the concatenation (in_layers.0): the concat writes its partial statistics
      std::shared_ptr<Buf> part;
      if (Fwd::gn_epi_enabled() && !((x.a[b].C + skip.a[b].C) & 63)) {
        part = std::make_shared<Buf>(ctx, (size_t)((f.rows(x.a[b]) + 31) / 32) * (x.a[b].C + skip.a[b].C) * sizeof(float));
        if (!part->p) part.reset();
      }
      f.ok(op_concat_add(ctx, x.a[b].p(), skip.a[b].p(), ctl.a[b].p(), cat.a[b].p(), f.rows(x.a[b]), x.a[b].C, skip.a[b].C, part ? part->as<float>() : nullptr));
      cat.a[b].gnp = part;
    }
    if (cat.same) cat.a[1] = cat.a[0];
    xs.pop_back(); control.pop_back();
    x = std::move(cat);
    // the last block's output feeds the head's GroupNorm; the others go into the next block's concat
    x = f.run_seq(net, net.output_blocks[bi], std::move(x), bi + 1 == net.output_blocks.size() ? (int)M_RES : -1);
    if (f.rc) return f.rc;
  }
  // head: GN + SiLU + conv 3x3 -> out_dim, fp32 rows, then back to [1, C, F, H, W]
  for (int b = 0; b < nb; ++b) {
    const Act& xb = x.a[x.same ? 0 : b];
    Act n = f.make(xb.C, xb.H, xb.W);
    f.gn(xb, net.out_norm, n, false, 1e-5f, true);
    Buf rowsf(ctx, (size_t)tok * 8 * 4);
    Act dummy; dummy.C = 8; dummy.H = H; dummy.W = W;
    f.conv3x3(n, net.out_conv, dummy, A_CONV3X3, 1, 1, 1, nullptr, nullptr, EPI_OUT_F32, rowsf.p, 8);
    f.ok(op_rows_to_latent(ctx, rowsf.as<float>(), outs[b], cfg.out_dim, 8, tok));
  }
  return f.rc;
}

int unet_forward(Ctx* ctx, const float* xt, long long t, const float* y, const float* hint, float* out, int F, int H, int W) {
  const float* ys[1] = {y};
  float* outs[1] = {out};
  return unet_forward_n(ctx, xt, t, ys, hint, outs, 1, F, H, W);
}

int controlnet_forward(Ctx* ctx, const float* xt, long long t, const float* y, const float* hint, void* const* residuals, int n,
                       int F, int H, int W) {
  const float* ys[1] = {y};
  float* outs[1] = {nullptr};
  return unet_forward_n(ctx, xt, t, ys, hint, outs, 1, F, H, W, residuals, n);
}

int module_run(Ctx* ctx, int kind, const char* prefix, int cin, int cout, int heads, int embed_dim, int context_dim,
               const void* x, const float* emb, const float* context, void* out, int F, int H, int W) {
  UNetModel tmp;
  UBuilder b{{ctx, &tmp.owned, ""}};
  Fwd f; f.ctx = ctx; f.F = F; f.es = ctx->esize(); f.ctx_dim = context_dim; f.embed_dim = embed_dim;
  f.emb = const_cast<float*>(emb);
  Buf ctxT;
  if (context) {
    ctxT = Buf(ctx, (size_t)77 * context_dim * f.es);
    if (op_cast(ctx, context, ctxT.p, (long long)77 * context_dim)) return 1;
    f.context = ctxT.p;
  }
  Act xin = f.make(cin, H, W);
  rt::memcpy_d2d(xin.p(), x, (size_t)F * H * W * cin * f.es, ctx->stream);
  Act y;
  const std::string p(prefix);
  Net net;
  if (kind == M_RES) { net.res.push_back(b.res(p)); if (!b.err.empty()) return ctx->fail(b.err); y = f.res_block(net.res[0], std::move(xin)); }
  else if (kind == M_ST) { net.st.push_back(b.st(p, heads)); if (!b.err.empty()) return ctx->fail(b.err); y = f.spatial_transformer(net.st[0], std::move(xin)); }
  else if (kind == M_TT) { net.tt.push_back(b.tt(p, heads)); if (!b.err.empty()) return ctx->fail(b.err); y = f.temporal_transformer(net.tt[0], std::move(xin)); }
  else if (kind == M_DOWN) { ConvW c; c.C = cout; c.conv = b.conv3x3(p + ".op"); if (!b.err.empty()) return ctx->fail(b.err); y = f.down(c, std::move(xin)); }
  else if (kind == M_UP) { ConvW c; c.C = cout; c.conv = b.conv3x3(p + ".conv"); if (!b.err.empty()) return ctx->fail(b.err); y = f.up(c, std::move(xin)); }
  else return ctx->fail("module_run: bad kind");
  if (f.rc) return f.rc;
  rt::memcpy_d2d(out, y.p(), (size_t)F * y.H * y.W * y.C * f.es, ctx->stream);
  rt::stream_sync(ctx->stream);   // tmp weights are freed on return
  return 0;
}

}  // namespace star
