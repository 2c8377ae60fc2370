"""BASELINE.json's parity bar -- PSNR >= 50 dB against the reference CPU path on identical latents / noise -- at FULL WIDTH on
BASELINE config[0]'s geometry (8 frames 128x128 -> x4, padded to 720x1280, latent 90x160, 5 solver steps).

tests/golden/cfg1_full.pt was produced by oracle/make_golden_cfg1.py in the build container: the REFERENCE's own
`ControlledV2VUNet` (2.04 B parameters) and `GaussianDiffusion.sample_sr` executed in fp32 on the CPU (40 minutes), the VAE by
oracle/vae_oracle.py (parity unpinned: diffusers is absent).  Inputs and weights are re-derived here from the same seeds.
"""
import math
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import fmt_metrics, frames_u8, parity_metrics  # noqa: E402
GOLD = os.path.join(ROOT, "tests", "golden", "cfg1_full.pt")
torch.set_grad_enabled(False)
# PSNR convention (stated next to every number, round-2 review): the asserted 50 dB bar uses peak = max - min of the reference
# tensor -- latents have no nominal range, and random-init weights decode to frames spanning [-3, 3.4] instead of [-1, 1].  The
# figure at the nominal peak 2.0, the range-free relative rms and (for frames) the PSNR of the clamped 8-bit frames are printed
# and bounded as well.  For scale: the REFERENCE's own fp16 arithmetic (half + autocast) differs from its fp32 arithmetic by a
# relative rms of 1.5e-2 after 6 forwards (tests/golden/fp16ref_small.pt, asserted in tests/test_pipeline.py).


def psnr(a, b, data_range):
    mse = float((a.double() - b.double()).pow(2).mean())
    return 10 * math.log10(data_range ** 2 / max(mse, 1e-30))


def test_golden_fixture_is_consistent():
    """CPU: the fixture matches the generator script's configuration and its tensors have the cfg1 shapes."""
    from make_golden_cfg1 import CFG1, cfg1_inputs
    g = torch.load(GOLD)
    assert g["cfg"] == CFG1
    assert tuple(g["z"].shape) == tuple(g["x0_final"].shape) == tuple(g["x0_first"].shape) == (1, 4, 8, 90, 160)
    assert tuple(g["video_out_f16"].shape) == (1, 3, 8, 512, 512) and g["video_out_f16"].dtype == torch.float16
    video, y, neg = cfg1_inputs()
    assert video.shape == (8, 3, 128, 128) and float(video.abs().max()) <= 1.0
    assert all(torch.isfinite(g[k].float()).all() for k in ("z", "noised", "x0_first", "x0_final", "video_out_f16"))


@pytest.fixture(scope="module")
def pipeline():
    from make_golden_cfg1 import CFG1, cfg1_inputs
    from star_amd.topology import UNetConfig, random_state_dict
    from star_amd.vae_topology import VaeConfig, random_vae_state_dict
    from star_amd.video_to_video_model import VideoToVideo_sr
    video, y, neg = cfg1_inputs()
    model = VideoToVideo_sr(dict(state_dict=random_state_dict(UNetConfig(), seed=CFG1["wseed"]),
                                 vae_state_dict=random_vae_state_dict(VaeConfig(), seed=CFG1["wseed"]),
                                 dtype=torch.float16, negative_y=neg, rng=torch.Generator().manual_seed(CFG1["rng_seed"])))
    return model, video, y, neg, torch.load(GOLD), CFG1


@pytest.mark.gpu
def test_denoiser_on_identical_latents_matches_the_reference(pipeline):
    """the chunk's whole sampling loop (10 full-width denoiser forwards, CFG + rescale, DPM++(2M) SDE) started from the golden
    VAE latent and noised latent, with the same injected solver noise: latent-space PSNR and the first evaluation's x0."""
    model, video, y, neg, gold, CFG1 = pipeline
    dev = model._tensor_device
    gen = torch.Generator().manual_seed(CFG1["rng_seed"])
    for _ in range(CFG1["frames"]):                       # the generator's consumption order: posterior noise per frame,
        torch.randn(1, 4, 90, 160, generator=gen)
    torch.randn(gold["z"].shape, generator=gen)           # diffuse noise, then one tensor per solver step
    z, noised = gold["z"].to(dev), gold["noised"].to(dev)

    class Sampler:
        def __init__(self, x, a, b, seed=None):
            self.shape, self.device = x.shape, x.device

        def __call__(self, s, sn):
            return torch.randn(self.shape, generator=gen).to(self.device)

    kw = [{"y": y.to(dev)}, {"y": neg.to(dev)}, {"hint": z}]
    t = torch.LongTensor([CFG1["total_noise_levels"] - 1]).to(dev)
    first = model.diffusion.denoise_x0(noised, t, model.generator, kw, CFG1["guide_scale"], 0.2).cpu()
    m1 = parity_metrics(first, gold["x0_first"])
    p1 = m1["psnr_range"]
    x0 = model.diffusion.sample_sr(noise=noised, model=model.generator, model_kwargs=kw, guide_scale=CFG1["guide_scale"], guide_rescale=0.2,
                                   solver="dpmpp_2m_sde", solver_mode=CFG1["solver_mode"], steps=CFG1["steps"], t_max=CFG1["total_noise_levels"] - 1,
                                   t_min=0, discretization="trailing", chunk_inds=None, noise_sampler_cls=Sampler).cpu()
    m = parity_metrics(x0, gold["x0_final"])
    p = m["psnr_range"]
    print(f"cfg1 full width, fp16: first-evaluation x0 {fmt_metrics(m1)}; final latent after 10 forwards {fmt_metrics(m)}")
    assert torch.isfinite(x0).all()
    assert p1 >= 50.0 and p >= 50.0, (p1, p)
    assert m1["rel_rms"] <= 1.2e-2 and m["rel_rms"] <= 2e-2, (m1, m)   # measured on MI355X: 8.0e-3 / 1.28e-2


@pytest.mark.gpu
def test_vae_encode_16_bit_storage_vs_the_fp32_encode_of_the_reference_path(pipeline):
    """the reference encodes in fp32 (vae.encode runs outside autocast, video_to_video_model.py:93); the HIP encoder keeps fp32
    accumulation but 16-bit activations.  Measured here on the eight 720x1280 frames of cfg1 against the fp32 CPU encode with
    the same posterior noise: the latent must agree far inside the 50 dB bar, otherwise an fp32-storage encoder would be needed."""
    model, video, y, neg, gold, CFG1 = pipeline
    from star_amd.geometry import pad_to_fit
    dev = model._tensor_device
    th, tw = CFG1["target"]
    padded = model.generator.ctx.resize_pad(video.to(dev, torch.float32), (th, tw), pad_to_fit(th, tw), 1.0).unsqueeze(0)
    model.rng.manual_seed(CFG1["rng_seed"])
    z = model.vae_encode(padded).cpu()
    m = parity_metrics(z, gold["z"])
    print(f"cfg1 VAE encode, 16-bit activations vs fp32 CPU encode: latent {fmt_metrics(m)}")
    assert m["psnr_range"] >= 60.0 and m["rel_rms"] <= 1e-3, m


@pytest.mark.gpu
def test_whole_test_call_psnr_at_least_50_db(pipeline):
    """`VideoToVideo_sr.test()` end to end (resize + pad, VAE encode, sampling, VAE decode, crop) against the decoded reference
    output: the stated bar, asserted."""
    model, video, y, neg, gold, CFG1 = pipeline
    model.rng.manual_seed(CFG1["rng_seed"])
    out = model.test({"video_data": video.to(model._tensor_device), "y": y, "target_res": CFG1["target"]},
                     total_noise_levels=CFG1["total_noise_levels"], steps=CFG1["steps"], solver_mode=CFG1["solver_mode"],
                     guide_scale=CFG1["guide_scale"], max_chunk_len=32)
    ref = gold["video_out_f16"].float()
    assert out.shape == ref.shape and out.dtype == torch.float32 and out.device.type == "cpu"
    m = parity_metrics(out, ref, nominal_peak=2.0)
    u8o, u8r = frames_u8(out), frames_u8(ref)
    p8 = 10 * math.log10(255.0 ** 2 / max(float((u8o - u8r).pow(2).mean()), 1e-30))
    print(f"cfg1 full width, fp16: decoded frames vs the reference CPU path {fmt_metrics(m)}; clamped 8-bit frames (what save_video "
          f"writes) {p8:.1f} dB at peak 255, {float((u8o != u8r).float().mean()) * 100:.1f} % of the bytes differ (by at most {int((u8o - u8r).abs().max())})")
    assert torch.isfinite(out).all() and m["psnr_range"] >= 50.0, m
    # the same error at the nominal [-1, 1] peak and on the 8-bit frames: bounded, not at 50 dB -- nor is the reference's own
    # fp16 path (tests/test_pipeline.py::test_hip_fp16_is_as_close_to_fp32_as_the_references_own_fp16)
    assert m["psnr_nominal"] >= 46.0 and p8 >= 46.0 and m["rel_rms"] <= 1.5e-2, (m, p8)
