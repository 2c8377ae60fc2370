// vae.cpp -- the SVD temporal VAE on the HIP kernels.
//
// The reference calls diffusers==0.30.0 `AutoencoderKLTemporalDecoder` (video_to_video_model.py:16,57-63,141-161);
// diffusers is NOT vendored in /root/reference and is not installed here, so this file follows the published
// architecture of diffusers' models/autoencoders/{autoencoder_kl_temporal_decoder.py, vae.py},
// models/unets/unet_3d_blocks.py (MidBlockTemporalDecoder, UpBlockTemporalDecoder), models/resnet.py
// (ResnetBlock2D, TemporalResnetBlock, SpatioTemporalResBlock, AlphaBlender, Downsample2D, Upsample2D) and
// models/attention_processor.py (Attention) as restated in oracle/vae_oracle.py.  PARITY UNPINNED: no reference
// test or fixture pins this arithmetic; parity is HIP-vs-oracle on synthetic weights.
//
// Layout: channels-last token rows [n*H*W, C] as in the UNet.  The mid-block attention (1 head x 512 channels over
// all H*W tokens) is run unfused -- fp32 logits [N, N] materialised in HBM (1.4-2.8 GB at 122x216: sized for
// 288 GB), row softmax, then P V -- on the same MFMA GEMM kernel.
#include "vae.h"
#include <cstdlib>

namespace star {

VaeModel::~VaeModel() { for (void* p : owned) rt::dev_free(p); }

namespace {

struct VBuilder : Builder {
  Res2DW res2d(const std::string& p) {
    Res2DW r;
    r.n1 = norm(p + ".norm1"); r.c1 = conv3x3(p + ".conv1");
    r.n2 = norm(p + ".norm2"); r.c2 = conv3x3(p + ".conv2");
    r.cin = r.n1.C; r.cout = r.n2.C;
    r.has_sc = ctx->host_tensors.count(p + ".conv_shortcut.weight") > 0;
    if (r.has_sc) r.sc = linear(p + ".conv_shortcut");
    return r;
  }
  AttnVW attn(const std::string& p) {
    AttnVW a;
    a.gn = norm(p + ".group_norm"); a.C = a.gn.C;
    a.q = linear(p + ".to_q"); a.k = linear(p + ".to_k");
    a.v_as_a = linear(p + ".to_v", false);   // used as the A operand: V^T = Wv X^T
    a.bv = bias(p + ".to_v.bias");           // added after P V (softmax rows sum to 1)
    a.out = linear(p + ".to_out.0");
    return a;
  }
  STResW stres(const std::string& p) {
    STResW s;
    s.sp = res2d(p + ".spatial_res_block");
    s.tn1 = norm(p + ".temporal_res_block.norm1"); s.tc1 = tconv(p + ".temporal_res_block.conv1");
    s.tn2 = norm(p + ".temporal_res_block.norm2");
    // AlphaBlender (learned, switch_spatial_to_temporal_mix): out = a*xs + (1-a)*xt with a = 1 - sigmoid(mix) and
    // xt = xs + h  =>  out = xs + sigmoid(mix) * h : fold sigmoid(mix) into conv2's weights and bias
    const HostTensor* mix = get(p + ".time_mixer.mix_factor");
    const HostTensor* w = get(p + ".temporal_res_block.conv2.weight");
    const HostTensor* b = get(p + ".temporal_res_block.conv2.bias");
    if (!mix || !w || !b) return s;
    const float sg = 1.0f / (1.0f + expf(-mix->data[0]));
    const int N = (int)w->shape[0], C = (int)w->shape[1];
    std::vector<float> r((size_t)N * 3 * C), rb(N);
    for (int n = 0; n < N; ++n) {
      rb[n] = b->data[n] * sg;
      for (int c = 0; c < C; ++c) for (int t = 0; t < 3; ++t) r[((size_t)n * 3 + t) * C + c] = w->data[((size_t)n * C + c) * 3 + t] * sg;
    }
    s.tc2.N = N; s.tc2.K = 3 * C; s.tc2.w = upload_T(r); s.tc2.b = upload_f32(rb);
    return s;
  }
};

struct VRun : Runner {
  // GroupNorm statistics leave with the data (round 6, as in the UNet since round 5: gemm.h EPIF bit 4): every conv / temporal conv /
  // projection whose output is read by a GroupNorm next also writes that tensor's partial statistics, and the norm finalizes from them
  // instead of reading the tensor twice.  out_stats: the module BEHIND the running one starts with a GroupNorm of its output (set by the
  // block walkers below; false in front of a Downsample / Upsample conv).
  bool out_stats = true;
  // ResnetBlock2D without time embedding (eps 1e-6): x + conv2(silu(gn(conv1(silu(gn(x))))))
  Act res2d(const Res2DW& r, Act x, bool stats_out) {
    Act n1 = make(r.cin, x.H, x.W);
    gn(x, r.n1, n1, false, 1e-6f, true);
    Act h1 = make(r.cout, x.H, x.W);
    conv3x3(n1, r.c1, h1, A_CONV3X3, 1, 1, 1, nullptr, nullptr, 0, nullptr, 0, 1, true);   // + h1's partials (norm2)
    n1.drop();
    gn(h1, r.n2, h1, false, 1e-6f, true);              // in place: nobody else reads h1 (round 6)
    Act n2 = std::move(h1);
    Act sc;
    const void* sp = x.p();
    if (r.has_sc) { sc = make(r.cout, x.H, x.W); gemm(x.p(), x.C, rows(x), r.sc, sc.p(), r.cout); sp = sc.p(); }
    Act y = make(r.cout, x.H, x.W);
    conv3x3(n2, r.c2, y, A_CONV3X3, 1, 1, 1, sp, nullptr, 0, nullptr, 0, 1, stats_out);
    return y;
  }
  Act res2d(const Res2DW& r, Act x) { return res2d(r, std::move(x), out_stats); }
  // SpatioTemporalResBlock: spatial resnet, TemporalResnetBlock over the frame axis (GN stats over the whole group,
  // eps 1e-5), AlphaBlender folded into tc2
  Act stres(const STResW& s, Act x) {
    Act xs = res2d(s.sp, std::move(x), true);          // + xs's partials (the temporal block's norm1, over the whole group)
    const int C = xs.C;
    Act n1 = make(C, xs.H, xs.W);
    gn(xs, s.tn1, n1, true, 1e-5f, true);
    Act h1 = make(C, xs.H, xs.W);
    tconv(n1, s.tc1, h1, nullptr, true);               // + h1's partials (norm2)
    n1.drop();
    gn(h1, s.tn2, h1, true, 1e-5f, true);              // in place
    Act y = make(C, xs.H, xs.W);
    tconv(h1, s.tc2, y, xs.p(), out_stats);
    return y;
  }
  void tconv(const Act& x, const LinW& w, Act& y, const void* res, bool stats) {
    GemmArgs g;
    g.A = x.p(); g.W = w.w.p; g.C = y.p(); g.M = rows(x); g.N = w.N; g.K = w.K; g.lda = x.C; g.ldc = y.C;
    g.mode = A_TCONV3; g.Cin = x.C; g.HW = x.H * x.W; g.F = F;
    g.bias = (const float*)w.b.p; g.res = res; g.ldr = y.C;
    g.epi = EPI_BIAS | (res ? EPI_RES : 0);
    gemm_to(g, &y, stats);
  }
  // Attention(heads = 1, dim_head = C = 512, residual_connection, GroupNorm eps 1e-6), per frame.  One head of width 512 does not
  // fit the d = 64 flash kernel (the output tile alone would be 512 accumulators per lane), so the logits go through HBM -- but
  // only a bounded block of query rows at a time (<= 6 GiB of fp32 logits + 16-bit probabilities, every buffer below 4 GiB), so that
  // the 133 712-token frames of the 2160p configuration need 6 GiB of scratch instead of 107 GB.  Round 6: the bound was 2 GiB, which
  // cut a cfg2 frame (26 352 rows) into two blocks whose P V GEMM is 106 tiles of 256 x 256 on 256 CUs, and a 2160p frame into 2560-row
  // blocks = 20 tiles; one block per cfg2 frame is 206 tiles, a 2160p block 7936 rows = 62.  If the pool cannot give 6 GiB the bound is
  // halved down to the old one.  It is 0.5 % of a clip's time.
  Act attn(const AttnVW& a, Act x) {
    const int C = a.C, HW = x.H * x.W;
    const int Np = (HW + 63) & ~63;
    Act n = make(C, x.H, x.W);
    gn(x, a.gn, n, false, 1e-6f, false);
    Act q = make(C, x.H, x.W), k = make(C, x.H, x.W), o = make(C, x.H, x.W);
    gemm(n.p(), C, rows(x), a.q, q.p(), C);
    gemm(n.p(), C, rows(x), a.k, k.p(), C);
    Buf vt(ctx, (size_t)C * Np * es), S, P;
    long long qc = 0;
    for (long long budget = (long long)6 << 30; budget >= ((long long)1 << 31); budget >>= 1) {
      qc = budget / ((long long)Np * (4 + (long long)es));                  // query rows per block
      if (const char* e = getenv("STAR_VAE_ATTN_ROWS")) qc = atoll(e);      // tests: force several query blocks on a small frame
      qc = (qc / 256) * 256;
      if (qc < 256) qc = 256;
      if (qc > HW) qc = HW;
      S = Buf(ctx, (size_t)qc * Np * 4);
      P = Buf(ctx, (size_t)qc * Np * es);
      if (S.p && P.p) break;
      S.reset(); P.reset();
    }
    if (!vt.p || !S.p || !P.p) { rc = ctx->fail("out of device memory (VAE attention)"); return x; }
    const float scale = 1.0f / sqrtf((float)C);
    for (int f = 0; f < F; ++f) {
      const char* nf = (const char*)n.p() + (size_t)f * HW * C * es;
      rt::memset_async(vt.p, 0, (size_t)C * Np * es, ctx->stream);
      {  // V^T[d][key] = sum_c Wv[d][c] * n[key][c]
        GemmArgs g; g.A = a.v_as_a.w.p; g.W = nf; g.C = vt.p; g.M = C; g.N = HW; g.K = C; g.lda = C; g.ldc = Np;
        ok(op_gemm(ctx, g));
      }
      for (long long q0 = 0; q0 < HW; q0 += qc) {
        const int nq = (int)((HW - q0 < qc) ? HW - q0 : qc);
        {  // S = Q[q0 : q0 + nq] K^T (fp32)
          GemmArgs g; g.A = (const char*)q.p() + ((size_t)f * HW + q0) * C * es; g.W = (const char*)k.p() + (size_t)f * HW * C * es;
          g.C = S.p; g.M = nq; g.N = HW; g.K = C; g.lda = C; g.ldc = Np; g.epi = EPI_OUT_F32;
          ok(op_gemm(ctx, g));
        }
        ok(op_softmax_rows(ctx, S.as<float>(), Np, P.p, Np, nq, HW, scale));
        {  // O[q0 : q0 + nq] = P V + bv
          GemmArgs g; g.A = P.p; g.W = vt.p; g.C = (char*)o.p() + ((size_t)f * HW + q0) * C * es; g.M = nq; g.N = C; g.K = Np; g.lda = Np; g.ldc = C;
          g.bias = (const float*)a.bv.p; g.epi = EPI_BIAS;
          ok(op_gemm(ctx, g));
        }
      }
    }
    Act y = make(C, x.H, x.W);
    gemm(o.p(), C, rows(x), a.out, y.p(), C, x.p(), C, 0, nullptr, out_stats ? &y : nullptr);
    return y;
  }
};

}  // namespace

int vae_build(Ctx* ctx, const VaeCfg& cfg) {
  auto M = std::make_shared<VaeModel>();
  M->cfg = cfg;
  VBuilder b{{ctx, &M->owned, ""}};
  const int nb = cfg.n_blocks;
  // ---- encoder
  M->e_conv_in = b.conv3x3_im2col64("encoder.conv_in");
  for (int i = 0; i < nb; ++i) {
    std::vector<Res2DW> rs;
    for (int j = 0; j < cfg.layers_per_block; ++j) rs.push_back(b.res2d("encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j)));
    M->e_res.push_back(rs);
    if (i != nb - 1) M->e_down.push_back(b.conv3x3("encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv"));
  }
  M->e_mid0 = b.res2d("encoder.mid_block.resnets.0");
  M->e_attn = b.attn("encoder.mid_block.attentions.0");
  M->e_mid1 = b.res2d("encoder.mid_block.resnets.1");
  M->e_norm_out = b.norm("encoder.conv_norm_out");
  {  // conv_out (C -> 2L, 3x3) followed by quant_conv (2L -> 2L, 1x1): compose into one 3x3 conv
    const HostTensor* w = b.get("encoder.conv_out.weight"); const HostTensor* bb = b.get("encoder.conv_out.bias");
    const HostTensor* qw = b.get("quant_conv.weight"); const HostTensor* qb = b.get("quant_conv.bias");
    if (w && bb && qw && qb) {
      const int L2 = (int)w->shape[0], C = (int)w->shape[1];
      std::vector<float> r((size_t)L2 * 9 * C, 0.f), rb(L2, 0.f);
      for (int o = 0; o < L2; ++o) {
        rb[o] = qb->data[o];
        for (int m = 0; m < L2; ++m) {
          const float qv = qw->data[(size_t)o * L2 + m];
          rb[o] += qv * bb->data[m];
          for (int c = 0; c < C; ++c) for (int t = 0; t < 9; ++t) r[(size_t)o * 9 * C + Builder::conv_k(c, t)] += qv * w->data[((size_t)m * C + c) * 9 + t];   // K order of the 3x3 modes (graph.h: conv_k)
        }
      }
      M->e_conv_out.N = L2; M->e_conv_out.K = 9 * C; M->e_conv_out.w = b.upload_T(r); M->e_conv_out.b = b.upload_f32(rb);
    }
  }
  // ---- decoder
  M->d_conv_in = b.conv3x3_im2col64("decoder.conv_in");
  M->d_mid0 = b.stres("decoder.mid_block.resnets.0");
  M->d_attn = b.attn("decoder.mid_block.attentions.0");
  M->d_mid1 = b.stres("decoder.mid_block.resnets.1");
  for (int i = 0; i < nb; ++i) {
    std::vector<STResW> rs;
    for (int j = 0; j < cfg.layers_per_block + 1; ++j) rs.push_back(b.stres("decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j)));
    M->d_res.push_back(rs);
    if (i != nb - 1) M->d_up.push_back(b.conv3x3("decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv"));
  }
  M->d_norm_out = b.norm("decoder.conv_norm_out");
  M->d_conv_out = b.conv3x3("decoder.conv_out", 4);
  M->d_time_w = b.raw_f32("decoder.time_conv_out.weight");
  M->d_time_b = b.raw_f32("decoder.time_conv_out.bias");
  if (!b.err.empty()) return ctx->fail("vae_build: " + b.err);
  ctx->vae = M;
  ctx->host_tensors.clear();
  return 0;
}

int vae_encode(Ctx* ctx, const float* x, float* moments, int n, int H, int W) {
  if (!ctx->vae) return ctx->fail("vae_encode: no model built (star_vae_build)");
  const VaeModel& M = *ctx->vae;
  const VaeCfg& cfg = M.cfg;
  const int fdown = 1 << (cfg.n_blocks - 1);
  if (H % fdown || W % fdown) return ctx->fail("vae_encode: H and W must be multiples of the downsampling factor");
  if (((H / fdown) * (W / fdown)) % 8) return ctx->fail("vae_encode: latent H*W must be a multiple of 8 (legal STAR sizes have W % 64 == 0)");
  const int L2 = 2 * cfg.latent;
  for (int i = 0; i < n; ++i) {   // one frame per pass, as the reference does (video_to_video_model.py:153-161)
    VRun r; r.ctx = ctx; r.F = 1; r.es = ctx->esize();
    Buf cols(ctx, (size_t)H * W * 64 * r.es);
    if (!cols.p) return ctx->fail("out of device memory");
    if (op_stem_im2col(ctx, x + (size_t)i * cfg.in_ch * H * W, cols.p, cfg.in_ch, 1, H, W, true)) return 1;
    Act a = r.make(cfg.block_out[0], H, W);
    r.gemm(cols.p, 64, H * W, M.e_conv_in, a.p(), a.C, nullptr, 0, 0, nullptr, &a);   // + partials for the first resnet's norm1
    cols.reset();
    for (int b = 0; b < cfg.n_blocks; ++b) {
      for (size_t j = 0; j < M.e_res[b].size(); ++j)   // the last resnet of a level feeds the Downsample conv, not a GroupNorm
        a = r.res2d(M.e_res[b][j], std::move(a), j + 1 < M.e_res[b].size() || b == cfg.n_blocks - 1);
      if (b != cfg.n_blocks - 1) {   // Downsample2D(padding=0): pad (0,1,0,1) then conv stride 2
        Act d = r.make(a.C, a.H / 2, a.W / 2);
        r.conv3x3(a, M.e_down[b], d, A_CONV3X3, 2, 0, 0, nullptr, nullptr, 0, nullptr, 0, 1, true);
        a = std::move(d);
      }
      if (r.rc) return r.rc;
    }
    a = r.res2d(M.e_mid0, std::move(a));
    a = r.attn(M.e_attn, std::move(a));
    a = r.res2d(M.e_mid1, std::move(a));
    Act nrm = r.make(a.C, a.H, a.W);
    r.gn(a, M.e_norm_out, nrm, false, 1e-6f, true);
    Act dummy; dummy.C = L2; dummy.H = a.H; dummy.W = a.W;
    r.conv3x3(nrm, M.e_conv_out, dummy, A_CONV3X3, 1, 1, 1, nullptr, nullptr, EPI_OUT_F32, moments + (size_t)i * a.H * a.W * L2, L2);
    if (r.rc) return r.rc;
  }
  return 0;
}

int vae_decode(Ctx* ctx, const float* z, float* out, int n, int h, int w) {
  if (!ctx->vae) return ctx->fail("vae_decode: no model built (star_vae_build)");
  const VaeModel& M = *ctx->vae;
  const VaeCfg& cfg = M.cfg;
  if ((h * w) % 8) return ctx->fail("vae_decode: latent h*w must be a multiple of 8 (legal STAR sizes have w % 8 == 0)");
  VRun r; r.ctx = ctx; r.F = n; r.es = ctx->esize();
  Buf cols(ctx, (size_t)n * h * w * 64 * r.es);
  if (!cols.p) return ctx->fail("out of device memory");
  if (op_stem_im2col(ctx, z, cols.p, cfg.latent, n, h, w, true)) return 1;
  Act a = r.make(cfg.block_out[cfg.n_blocks - 1], h, w);
  r.gemm(cols.p, 64, n * h * w, M.d_conv_in, a.p(), a.C, nullptr, 0, 0, nullptr, &a);
  cols.reset();
  a = r.stres(M.d_mid0, std::move(a));
  a = r.attn(M.d_attn, std::move(a));
  a = r.stres(M.d_mid1, std::move(a));
  for (int b = 0; b < cfg.n_blocks; ++b) {
    for (size_t j = 0; j < M.d_res[b].size(); ++j) {   // the last block of a level feeds the Upsample conv, not a GroupNorm
      r.out_stats = j + 1 < M.d_res[b].size() || b == cfg.n_blocks - 1;
      a = r.stres(M.d_res[b][j], std::move(a));
    }
    r.out_stats = true;
    if (b != cfg.n_blocks - 1) {     // Upsample2D: nearest x2 + conv 3x3
      Act u = r.make(M.d_up[b].N, 2 * a.H, 2 * a.W);
      r.conv3x3(a, M.d_up[b], u, A_CONV3X3_UP, 1, 1, 1, nullptr, nullptr, 0, nullptr, 0, /*up_crop=*/0);
      a = std::move(u);
    }
    if (r.rc) return r.rc;
  }
  Act nrm = r.make(a.C, a.H, a.W);
  r.gn(a, M.d_norm_out, nrm, false, 1e-6f, true);
  Buf rowsf(ctx, (size_t)n * a.H * a.W * 4 * 4);
  Act dummy; dummy.C = 4; dummy.H = a.H; dummy.W = a.W;
  r.conv3x3(nrm, M.d_conv_out, dummy, A_CONV3X3, 1, 1, 1, nullptr, nullptr, EPI_OUT_F32, rowsf.p, 4);
  r.ok(op_time_conv_out(ctx, rowsf.as<float>(), 4, out, (const float*)M.d_time_w.p, (const float*)M.d_time_b.p, n, a.H * a.W, cfg.out_ch));
  return r.rc;
}

}  // namespace star
