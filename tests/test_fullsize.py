"""BASELINE config[1] sizes (32 frames, latent 122 x 216, the full 2.04 B-parameter architecture) on hardware: properties that
do not need a CPU oracle (the fp32 CPU path takes hours at this size: the one reference CFG pair that was afforded is the fixture of
tests/test_parity_cfg2.py)."""
import math

import pytest
import torch


@pytest.mark.gpu
def test_spatial_attention_full_length_vs_fp32():
    """one (frame, head) of the L0 spatial self-attention at its real length N = 26352 against fp32 softmax(QK^T/8)V."""
    from util import make_ctx
    ctx = make_ctx("hip", torch.float16, None)
    dev = ctx.torch_device
    g = torch.Generator().manual_seed(3)
    N = 122 * 216
    qkv = (torch.randn(2, N, 192, generator=g) * 1.5).to(torch.float16).to(dev)      # 2 frames, 1 head, fused QKV rows
    out = ctx.attention(qkv[..., :64], qkv[..., 64:128], qkv[..., 128:], 1).float()
    q, k, v = (qkv[..., i * 64:(i + 1) * 64].float() for i in range(3))
    ref = torch.empty_like(out)
    for b in range(2):
        for s in range(0, N, 4096):                                                  # 4096 x 26352 fp32 logits at a time
            p = torch.softmax(q[b, s:s + 4096] @ k[b].T / 8.0, dim=-1)
            ref[b, s:s + 4096] = p @ v[b]
    err = float((out - ref).abs().max())
    assert err <= 4e-3 * max(1.0, float(ref.abs().max())), err
    ctx.sync()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_spatial_attention_full_length_peaked_logits(dtype):
    """N = 26352 with the logit statistics of a TRAINED attention layer instead of N(0, 1) inputs: queries / keys with a strong
    shared component and spatially smooth structure, logits spread over ~+-40 (log2 units ~+-60), a handful of keys carrying most
    of each row's mass, rows whose maximum arrives late in the key range, and a block of near-duplicate keys (flat rows).  This is
    the regime the lazy-maxima probe (row sum of P <= 2^10), its recompute path and the packed 16-bit row sums (f16) were
    built for but random-init weights never produce."""
    from util import make_ctx
    ctx = make_ctx("hip", dtype, None)
    dev = ctx.torch_device
    g = torch.Generator().manual_seed(7)
    H, W = 122, 216
    N = H * W
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    pos = torch.stack([torch.sin(yy / 9.0), torch.cos(yy / 9.0), torch.sin(xx / 13.0), torch.cos(xx / 13.0)], dim=-1).reshape(N, 4)
    basis = torch.randn(4, 64, generator=g)
    q = pos @ basis * 2.5 + torch.randn(N, 64, generator=g) * 0.7          # locality: a query looks at keys near its own position
    k = pos @ basis * 2.5 + torch.randn(N, 64, generator=g) * 0.7
    k[N - 3000:] *= 1.6                                                      # the strongest keys sit at the END of the key range
    k[5000:5512] = k[5000:5001]                                              # 512 duplicates of one key
    q[100:164] = k[N - 10] * 0.9                                             # 64 rows whose maximum is a very late key
    v = torch.randn(N, 64, generator=g)
    qkv = torch.cat([q, k, v], dim=-1)[None].to(dtype).to(dev)
    out = ctx.attention(qkv[..., :64], qkv[..., 64:128], qkv[..., 128:], 1).float()[0]
    qf, kf, vf = (qkv[0, :, i * 64:(i + 1) * 64].float() for i in range(3))
    ref = torch.empty_like(out)
    lmax = 0.0
    for s0 in range(0, N, 4096):
        logit = qf[s0:s0 + 4096] @ kf.T / 8.0
        lmax = max(lmax, float(logit.abs().max()))
        ref[s0:s0 + 4096] = torch.softmax(logit, dim=-1) @ vf
    assert lmax > 25.0, lmax                                                 # the test really is in the peaked regime
    eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    err = float((out - ref).abs().max())
    assert torch.isfinite(out).all() and err <= 4.0 * eps * max(1.0, float(ref.abs().max())), (err, lmax)
    ctx.sync()
    ctx.close()


@pytest.mark.gpu
def test_spatial_attention_cfg4_length_vs_fp32():
    """BASELINE config[3] (32 f 540x960 -> x4, latent 274 x 488): one (frame, head) of the L0 spatial self-attention at its real
    length N = 133712 (2090 key tiles, ragged tail of 16) against fp32 softmax(QK^T/8)V taken 2048 query rows at a time."""
    from util import make_ctx
    ctx = make_ctx("hip", torch.float16, None)
    dev = ctx.torch_device
    g = torch.Generator().manual_seed(4)
    N = 274 * 488
    qkv = (torch.randn(1, N, 192, generator=g) * 1.5).to(torch.float16).to(dev)
    qkv[0, N - 500, 64:128] = qkv[0, 77, :64] * 2.5                                   # a late key that moves one row's maximum
    out = ctx.attention(qkv[..., :64], qkv[..., 64:128], qkv[..., 128:], 1).float()[0]
    q, k, v = (qkv[0, :, i * 64:(i + 1) * 64].float() for i in range(3))
    ref = torch.empty_like(out)
    for s0 in range(0, N, 2048):
        ref[s0:s0 + 2048] = torch.softmax(q[s0:s0 + 2048] @ k.T / 8.0, dim=-1) @ v
    err = float((out - ref).abs().max())
    rel = float((out - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    print(f"spatial attention at N = {N}: max abs err {err:.2e}, relative rms {rel:.2e}")
    assert torch.isfinite(out).all() and err <= 4e-3 * max(1.0, float(ref.abs().max())) and rel <= 2e-3, (err, rel)
    ctx.sync()
    ctx.close()


@pytest.mark.gpu
def test_vae_mid_attention_at_its_real_block_size(monkeypatch):
    """the single-head d = 512 mid-block attention of the VAE sends its fp32 logits through HBM in blocks of query rows bounded
    by 6 GiB of scratch (vae.cpp; 2 GiB until round 6).  Full-width encoder on one cfg2 frame (976 x 1728 -> 26352 tokens): the
    natural single block (one-read register softmax) against two 13568-row blocks -- bit-identical; on one cfg4 frame (2192 x 3904
    -> 133712 tokens: 17 natural blocks of 7936 rows, a 4.25 GB logits buffer, rows too long for registers -> the online two-read
    softmax) against 5120-row blocks -- bit-identical and finite."""
    from star_amd.vae import AutoencoderKLTemporalDecoder
    from star_amd.vae_topology import VaeConfig, random_vae_state_dict
    cfg = VaeConfig()
    vae = AutoencoderKLTemporalDecoder(cfg, dtype=torch.float16).load_state_dict(random_vae_state_dict(cfg, seed=2))
    g = torch.Generator().manual_seed(5)
    for (H, W, other_rows) in ((976, 1728, 13568), (2192, 3904, 5120)):
        x = (torch.randn(1, 3, H // 8, W // 8, generator=g) * 0.5).clamp(-1, 1)
        x = torch.nn.functional.interpolate(x, size=(H, W), mode="bilinear").cuda()
        monkeypatch.delenv("STAR_VAE_ATTN_ROWS", raising=False)
        natural = vae.encode(x).latent_dist.parameters.float().cpu()
        monkeypatch.setenv("STAR_VAE_ATTN_ROWS", str(other_rows))
        other = vae.encode(x).latent_dist.parameters.float().cpu()
        assert natural.shape == (1, 8, H // 8, W // 8) and torch.isfinite(natural).all()
        assert torch.equal(natural, other), (H, W, float((natural - other).abs().max()))
    monkeypatch.delenv("STAR_VAE_ATTN_ROWS", raising=False)
    vae.ctx.sync()


# (the whole denoiser at cfg2 size -- finite, shared-prefix CFG pair == two plain forwards -- is asserted AGAINST THE REFERENCE in
# tests/test_parity_cfg2.py since round 4; rounds 1-3 compared that shape only with itself here)
