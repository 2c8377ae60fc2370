"""BASELINE configs[2]'s per-step CHUNK LOOP at its own geometry against the reference: 72 frames, latent 122 x 216, `max_chunk_len=16`
-> the 8 overlapping chunks (0,16), (8,24) ... (56,72), each denoised with its own `hint_chunk` slice, overlaps trimmed (first [:12],
inner [4:12], last [4:]) and concatenated.

tests/golden/cfg3_chunkloop.pt was produced in the build container by oracle/make_golden_cfg3.py: ONE solver step (steps = 1) of the
REFERENCE's own `GaussianDiffusion.sample_sr` / `model_chunk_fn` (video_to_video/diffusion/diffusion_sdedit.py:330-353) and
`sample_dpmpp_2m_sde` (solvers_sdedit.py:144-204) around the REFERENCE's own full-width `ControlledV2VUNet`
(video_to_video/modules/unet_v2v.py:1717-1809), fp32 on the CPU (8 chunks x 2 guidance forwards of 16 frames, 4.5 PFLOP).  The loop
had been compared with the reference at 11 frames / latent 90x160 only (tests/test_pipeline.py).

PSNR convention as in tests/test_parity_cfg2.py (peak = range of the reference latent; the nominal-peak figure is printed).
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import fmt_metrics, parity_metrics  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "cfg3_chunkloop.pt")
torch.set_grad_enabled(False)
needs_fixture = pytest.mark.skipif(not os.path.isfile(GOLD), reason="tests/golden/cfg3_chunkloop.pt not generated (oracle/make_golden_cfg3.py, ~4 CPU-hours)")


@needs_fixture
def test_golden_fixture_is_consistent():
    """CPU: the fixture's chunk list is the product's make_chunks(72, 0, 16) (the generator asserts it is the reference's), its x0 has
    the cfg3 shape, and the noised latent re-derived here is the one the reference saw (checksum)."""
    from make_golden_cfg3 import CFG3, cfg3_inputs
    from star_amd.diffusion import GaussianDiffusion, noise_schedule
    from star_amd.geometry import make_chunks
    g = torch.load(GOLD)
    assert g["cfg"] == CFG3 and CFG3["frames"] == 72 and tuple(CFG3["latent"]) == (122, 216)
    chunks = make_chunks(72, 0, 16)
    assert [tuple(c) for c in g["chunks"]] == [tuple(c) for c in chunks] == [(8 * i, 8 * i + 16) for i in range(8)]
    assert tuple(g["x0_f16"].shape) == (1, 4, 72, 122, 216) and torch.isfinite(g["x0_f16"].float()).all()
    z, eps, y, neg = cfg3_inputs()
    gd = GaussianDiffusion(noise_schedule(schedule="logsnr_cosine_interp", n=1000, zero_terminal_snr=True, scale_min=2.0, scale_max=4.0))
    noised = gd.diffuse(z, torch.LongTensor([CFG3["t"]]), noise=eps)
    assert abs(float(noised.double().sum()) - g["noised_sum"]) <= 1e-6 * max(1.0, abs(g["noised_sum"])) + 1e-3


@needs_fixture
@pytest.mark.gpu
def test_hip_chunk_loop_matches_the_reference_at_cfg3_geometry():
    """star_amd's sample_sr (host Python, bit-exact solver arithmetic) driving the HIP fp16 denoiser over the 8 chunks: the stitched x0
    against the reference's, PSNR(range) >= 50 dB and relative rms <= 1.2e-2 (the cfg2 bar); every chunk's core is checked on its own,
    so an overlap-trim error cannot hide in the average."""
    from make_golden_cfg3 import CFG3, cfg3_inputs
    from star_amd.diffusion import GaussianDiffusion, noise_schedule
    from star_amd.geometry import make_chunks
    from star_amd.modules.unet_v2v import ControlledV2VUNet
    from star_amd.topology import UNetConfig, random_state_dict
    gold = torch.load(GOLD)
    ref = gold["x0_f16"].float()
    cfg = UNetConfig()
    net = ControlledV2VUNet(cfg, dtype=torch.float16, device=0)
    net.load_state_dict(random_state_dict(cfg, seed=CFG3["wseed"]))
    net.release_host_weights()
    z, eps, y, neg = cfg3_inputs()
    gd = GaussianDiffusion(noise_schedule(schedule="logsnr_cosine_interp", n=1000, zero_terminal_snr=True, scale_min=2.0, scale_max=4.0))
    noised = gd.diffuse(z, torch.LongTensor([CFG3["t"]]), noise=eps)
    dev = torch.device("cuda", 0)
    chunks = make_chunks(CFG3["frames"], 0, CFG3["max_chunk_len"])

    class NoNoise:
        def __init__(self, x, a, b, seed=None):
            pass

        def __call__(self, s, sn):
            raise AssertionError("a one-step trajectory draws no noise")

    x0 = gd.sample_sr(noise=noised.to(dev), model=net, model_kwargs=[{"y": y.to(dev)}, {"y": neg.to(dev)}, {"hint": z.to(dev)}],
                      guide_scale=CFG3["guide_scale"], guide_rescale=CFG3["guide_rescale"], solver="dpmpp_2m_sde", solver_mode="normal",
                      steps=CFG3["steps"], t_max=CFG3["t"], t_min=0, discretization="trailing", chunk_inds=chunks, noise_sampler_cls=NoNoise).cpu()
    m = parity_metrics(x0, ref, nominal_peak=2.0)
    print(f"cfg3 chunk loop (72 f, 8 chunks of 16, 122x216, full width), HIP fp16 vs the reference's fp32: x0 {fmt_metrics(m)}")
    assert x0.shape == ref.shape and torch.isfinite(x0).all()
    assert m["psnr_range"] >= 50.0 and m["rel_rms"] <= 1.2e-2, m
    # per chunk core: frames [0,12), [12,20), ... [52,60), [60,72)
    cores = [(0, 12)] + [(12 + 8 * i, 20 + 8 * i) for i in range(6)] + [(60, 72)]
    for a, b in cores:
        mc = parity_metrics(x0[:, :, a:b], ref[:, :, a:b])
        assert mc["rel_rms"] <= 1.5e-2, ((a, b), mc)
