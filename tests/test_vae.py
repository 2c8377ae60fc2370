"""VAE parity: HIP path vs the CPU oracle restatement of diffusers' AutoencoderKLTemporalDecoder
(oracle/vae_oracle.py -- PARITY UNPINNED against real diffusers, see its header) on synthetic weights."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import vae_oracle as VO  # noqa: E402
from star_amd.vae import AutoencoderKLTemporalDecoder  # noqa: E402
from star_amd.vae_topology import SMALL_VAE_CONFIG, VaeConfig, random_vae_state_dict, vae_param_shapes  # noqa: E402
from util import BACKENDS  # noqa: E402

torch.set_grad_enabled(False)
REL = {torch.float16: 1e-2, torch.bfloat16: 6e-2}


def rel_rms(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-12))


@pytest.fixture(params=BACKENDS)
def backend(request):
    return request.param


def test_full_vae_parameter_count():
    n = sum(int(torch.Size(s).numel()) for s in vae_param_shapes(VaeConfig()).values())
    assert abs(n / 1e6 - 97.7) < 0.5   # SVD VAE: 34.2 M encoder + 63.6 M temporal decoder


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_small_vae_encode_decode(backend, dtype, request):
    emu = request.getfixturevalue("emu_lib") if backend == "emu" else None
    cfg = SMALL_VAE_CONFIG
    sd = random_vae_state_dict(cfg, seed=0)
    vae = AutoencoderKLTemporalDecoder(cfg, dtype=dtype, library=emu).load_state_dict(sd)
    dev = vae.ctx.torch_device
    g = torch.Generator().manual_seed(1)
    H, W = (16, 32) if backend == "emu" else (40, 64)   # latent H*W must be a multiple of 8 (always true for legal UNet sizes)
    x = torch.randn(2, 3, H, W, generator=g).clamp(-1, 1)
    n0 = vae.ctx.lib.gn_fused_count(vae.ctx.h)
    mom = vae.encode(x.to(dev)).latent_dist.parameters
    ref = VO.encode_moments(sd, cfg, x)
    assert rel_rms(mom, ref) < REL[dtype], rel_rms(mom, ref)
    # round 6: every GroupNorm of the encoder finalizes from the statistics its producer's epilogue wrote (vae.cpp; 3 levels: 8 resnets x 2
    # norms + the attention's norm + conv_norm_out = 18 per frame, 2 frames)
    n1 = vae.ctx.lib.gn_fused_count(vae.ctx.h)
    assert n1 - n0 == 2 * 18, n1 - n0
    z = torch.randn(3, 4, H // cfg.downsample, W // cfg.downsample, generator=g)
    out = vae.decode(z.to(dev), num_frames=3).sample
    # ... and of the decoder, except the two norms behind an Upsample conv (the nearest-x2 gather has no statistics flavour): 11 blocks x 4 + 2 - 2
    assert vae.ctx.lib.gn_fused_count(vae.ctx.h) - n1 == 11 * 4 + 2 - 2, vae.ctx.lib.gn_fused_count(vae.ctx.h) - n1
    refd = VO.decode(sd, cfg, z, 3)
    assert out.shape == refd.shape
    assert rel_rms(out, refd) < REL[dtype], rel_rms(out, refd)
    # a 2-frame tail group, as the reference's last decode call has (video_to_video_model.py:144-151)
    out2 = vae.decode(z[:2].to(dev), num_frames=2).sample
    assert rel_rms(out2, VO.decode(sd, cfg, z[:2], 2)) < REL[dtype]


@pytest.mark.gpu
def test_full_width_vae_small_image():
    """all real channel widths (128/256/512) of the SVD VAE on a 64x96 frame, fp16."""
    cfg = VaeConfig()
    sd = random_vae_state_dict(cfg, seed=2)
    vae = AutoencoderKLTemporalDecoder(cfg, dtype=torch.float16).load_state_dict(sd)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 3, 64, 96, generator=g).clamp(-1, 1)
    mom = vae.encode(x.cuda()).latent_dist.parameters
    assert rel_rms(mom, VO.encode_moments(sd, cfg, x)) < REL[torch.float16]
    z = torch.randn(3, 4, 8, 12, generator=g)
    out = vae.decode(z.cuda(), num_frames=3).sample
    assert rel_rms(out, VO.decode(sd, cfg, z, 3)) < REL[torch.float16]


def test_mid_block_attention_in_query_blocks(backend, request, monkeypatch):
    """the mid-block attention materialises its logits for a bounded block of query rows at a time (2 GiB at real sizes, so
    that 133 712-token frames fit); forced to 256-row blocks here, the result must not change."""
    emu = request.getfixturevalue("emu_lib") if backend == "emu" else None
    cfg = SMALL_VAE_CONFIG
    sd = random_vae_state_dict(cfg, seed=0)
    g = torch.Generator().manual_seed(4)
    H, W = (128, 192) if backend == "emu" else (192, 256)      # latent 16 x 24 = 384 / 24 x 32 = 768 tokens: 2 / 3 blocks of 256
    x = torch.randn(1, 3, H, W, generator=g).clamp(-1, 1)
    vae = AutoencoderKLTemporalDecoder(cfg, dtype=torch.float16, library=emu).load_state_dict(sd)
    whole = vae.encode(x.to(vae.ctx.torch_device)).latent_dist.parameters.cpu()
    monkeypatch.setenv("STAR_VAE_ATTN_ROWS", "256")
    blocked = vae.encode(x.to(vae.ctx.torch_device)).latent_dist.parameters.cpu()
    assert torch.equal(whole, blocked)
    assert rel_rms(whole, VO.encode_moments(sd, cfg, x)) < REL[torch.float16]
