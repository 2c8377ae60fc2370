"""End-to-end `VideoToVideo_sr.test()` on the GPU against the CPU oracle pipeline (oracle/pipeline_oracle.py) with
identical weights, inputs and injected noise.  BASELINE.json's parity statement: PSNR of the decoded output vs the
CPU reference path.  Reduced-width UNet/VAE so that the CPU oracle finishes in about a minute; the padded frame is the
real 720x1280 (latent 90x160), i.e. BASELINE config[0]'s geometry."""
import math
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import fmt_metrics, frames_u8, parity_metrics  # noqa: E402
torch.set_grad_enabled(False)


def psnr(a, b, data_range=2.0):
    mse = float((a.float() - b.float()).pow(2).mean())
    return 10 * math.log10(data_range ** 2 / max(mse, 1e-30))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,min_psnr", [(torch.float16, 50.0), (torch.bfloat16, 30.0)], ids=["f16", "bf16"])   # f16 = the parity dtype (bar: 50 dB); bf16 is an opt-in speed mode that does not meet it
def test_pipeline_psnr_vs_cpu_oracle(dtype, min_psnr):
    import pipeline_oracle as PO
    from star_amd.topology import SMALL_TEST_CONFIG, random_state_dict
    from star_amd.vae_topology import VaeConfig, random_vae_state_dict
    from star_amd.video_to_video_model import VideoToVideo_sr
    ucfg = SMALL_TEST_CONFIG
    vcfg = VaeConfig(block_out_channels=(64, 64, 128, 128))
    sd, vsd = random_state_dict(ucfg, seed=0), random_vae_state_dict(vcfg, seed=0)
    g = torch.Generator().manual_seed(666)
    frames = 4
    video = (torch.randn(frames, 3, 32, 32, generator=g) * 0.5).clamp(-1, 1)
    y = torch.randn(1, 77, ucfg.context_dim, generator=g)
    neg = torch.randn(1, 77, ucfg.context_dim, generator=g)
    kw = dict(total_noise_levels=900, steps=3, solver_mode="normal", guide_scale=7.5, max_chunk_len=32)
    ref, aux = PO.run_pipeline(sd, vsd, ucfg, vcfg, video, y, neg, (128, 128), torch.Generator().manual_seed(1), **kw)
    model = VideoToVideo_sr(dict(state_dict=sd, vae_state_dict=vsd, unet_config=ucfg, vae_config=vcfg, dtype=dtype, negative_y=neg,
                                 rng=torch.Generator().manual_seed(1)))
    out = model.test({"video_data": video.cuda(), "y": y, "target_res": (128, 128)}, **kw)
    assert out.shape == ref.shape == (1, 3, frames, 128, 128) and out.dtype == torch.float32 and out.device.type == "cpu"
    m = parity_metrics(out, ref, nominal_peak=2.0)   # conventions: tests/test_parity_cfg1.py header
    p8 = psnr(frames_u8(out), frames_u8(ref), data_range=255.0)
    print(f"pipeline vs CPU oracle ({dtype}): {fmt_metrics(m)}; clamped 8-bit frames {p8:.1f} dB at peak 255")
    assert torch.isfinite(out).all()
    assert m["psnr_range"] >= min_psnr, m
    if dtype == torch.float16:   # measured on MI355X: 57.0 dB (range) / 48.3 dB (nominal) / 48.0 dB (8-bit frames), relative rms 1.6e-2
        assert m["psnr_nominal"] >= 46.0 and p8 >= 46.0 and m["rel_rms"] <= 2.5e-2, (m, p8)


@pytest.mark.gpu
@pytest.mark.parametrize("width", ["reduced", "full"])
def test_chunked_solver_loop_with_the_hip_denoiser(width):
    """the per-step chunk loop (diffusion_sdedit.py:330-353: overlapping chunks, hint_chunk slices, overlap trim, concat) driven
    with the HIP denoiser on the GPU against the same loop with the CPU oracle denoiser.  reduced width: 41 frames -> chunks
    (0,32), (16,41), 3 steps; FULL width (2.04 B parameters, round-2 review): 11 frames at max_chunk_len = 8 -> chunks (0,8), (4,11)
    on the smallest legal latent, 2 steps = 8 full-width forwards per side."""
    import unet_oracle as O
    from star_amd.diffusion import GaussianDiffusion, noise_schedule
    from star_amd.geometry import make_chunks
    from star_amd.modules.unet_v2v import ControlledV2VUNet
    from star_amd.topology import SMALL_TEST_CONFIG, UNetConfig, random_state_dict
    if width == "full":
        cfg, (F_, h, w), max_len, steps, want = UNetConfig(), (11, 10, 8), 8, 2, [(0, 8), (4, 11)]
        torch.set_num_threads(min(64, os.cpu_count() or 8))
    else:
        cfg, (F_, h, w), max_len, steps, want = SMALL_TEST_CONFIG, (41, 18, 16), 32, 3, [(0, 32), (16, 41)]
    sd = random_state_dict(cfg, seed=0)
    net = ControlledV2VUNet(cfg, dtype=torch.float16, device=0)
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(12)
    noise = torch.randn(1, 4, F_, h, w, generator=g)
    hint = torch.randn(1, 4, F_, h, w, generator=g) * 0.5
    y, neg = torch.randn(1, 77, cfg.context_dim, generator=g), torch.randn(1, 77, cfg.context_dim, generator=g)
    chunks = make_chunks(F_, 0, max_len)
    assert chunks == want
    gd = GaussianDiffusion(noise_schedule(schedule="logsnr_cosine_interp", n=1000, zero_terminal_snr=True, scale_min=2.0, scale_max=4.0))

    def run(model, dev):
        gen = torch.Generator().manual_seed(3)

        class Sampler:
            def __init__(self, x, a, b, seed=None):
                self.shape = x.shape

            def __call__(self, s, sn):
                return torch.randn(self.shape, generator=gen).to(dev)

        return gd.sample_sr(noise=noise.to(dev), model=model, model_kwargs=[{"y": y.to(dev)}, {"y": neg.to(dev)}, {"hint": hint.to(dev)}],
                            guide_scale=7.5, guide_rescale=0.2, solver="dpmpp_2m_sde", solver_mode="normal", steps=steps, t_max=899, t_min=0,
                            discretization="trailing", chunk_inds=chunks, noise_sampler_cls=Sampler).cpu()

    def oracle(x, t=None, y=None, hint=None, hint_chunk=None, variant_info=None):
        return O.unet_forward(sd, cfg, x, t, y, hint_chunk if hint_chunk is not None else hint)

    out = run(net, torch.device("cuda", 0))
    ref = run(oracle, torch.device("cpu"))
    m = parity_metrics(out, ref)
    print(f"chunked loop ({width} width), HIP denoiser vs CPU oracle: latent {fmt_metrics(m)}")
    assert out.shape == ref.shape == (1, 4, F_, h, w) and torch.isfinite(out).all()
    assert m["psnr_range"] >= 50.0 and m["rel_rms"] <= 1.5e-2, m


def test_no_bf16_arithmetic_meets_the_bar_the_references_own_does_not():
    """CPU: BASELINE configs[1] says "bf16" while its parity bar says "stated fp16 tolerance (PSNR >= 50 dB vs reference output)".
    The fixture holds the REFERENCE's own modules run in fp32, in fp16 (half + autocast, its GPU arithmetic) and in bf16 (bfloat16
    weights + autocast(bfloat16)) on identical inputs (oracle/make_golden_fp16ref.py): its fp16 arithmetic clears 50 dB, its bf16
    arithmetic misses it by more than 10 dB -- 8 mantissa bits against 11, three bits = 18 dB.  So f16 is the dtype that meets the
    bar (bench.py's default) and bf16 is an opt-in speed mode on any implementation, not a gap of this one."""
    gold = torch.load(os.path.join(ROOT, "tests", "golden", "fp16ref_small.pt"))
    f16, b16 = parity_metrics(gold["x0_ref_fp16"], gold["x0_fp32"]), parity_metrics(gold["x0_ref_bf16"], gold["x0_fp32"])
    print(f"the reference's own arithmetic vs its fp32 (6 forwards):  fp16 {fmt_metrics(f16)}  |  bf16 {fmt_metrics(b16)}")
    assert f16["psnr_range"] >= 50.0 and b16["psnr_range"] <= 40.0
    assert 6.0 <= b16["rel_rms"] / f16["rel_rms"] <= 10.0        # ~2^3: the three mantissa bits


@pytest.mark.gpu
def test_hip_bf16_is_as_close_to_fp32_as_the_references_own_bf16():
    """--dtype bf16 of the HIP path against the same yardstick as the fp16 test below: at least as close to the reference's fp32
    result as the reference's OWN bf16 arithmetic is (10 % slack).  (Neither reaches 50 dB; see the CPU test above.)"""
    out, gold = _run_fp16ref_case(torch.bfloat16)
    ours, theirs = parity_metrics(out, gold["x0_fp32"]), parity_metrics(gold["x0_ref_bf16"], gold["x0_fp32"])
    print(f"final latent after 6 forwards vs the reference in fp32:  HIP bf16 path {fmt_metrics(ours)}  |  the reference's own bf16 (bfloat16 + "
          f"autocast) {fmt_metrics(theirs)}")
    assert torch.isfinite(out).all() and ours["rel_rms"] <= 1.1 * theirs["rel_rms"], (ours, theirs)


def _run_fp16ref_case(dtype):
    from make_golden_fp16ref import CFG16, fp16ref_inputs
    from star_amd.diffusion import GaussianDiffusion, noise_schedule
    from star_amd.modules.unet_v2v import ControlledV2VUNet
    from star_amd.topology import SMALL_TEST_CONFIG, random_state_dict
    gold = torch.load(os.path.join(ROOT, "tests", "golden", "fp16ref_small.pt"))
    assert gold["cfg"] == CFG16
    cfg = SMALL_TEST_CONFIG
    net = ControlledV2VUNet(cfg, dtype=dtype, device=0)
    net.load_state_dict(random_state_dict(cfg, seed=CFG16["wseed"]))
    noise, hint, y, neg = fp16ref_inputs()
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(CFG16["noise_seed"])

    class Sampler:
        def __init__(self, x, a, b, seed=None):
            self.shape = x.shape

        def __call__(self, s, sn):
            return torch.randn(self.shape, generator=gen).to(dev)

    gd = GaussianDiffusion(noise_schedule(schedule="logsnr_cosine_interp", n=1000, zero_terminal_snr=True, scale_min=2.0, scale_max=4.0))
    out = gd.sample_sr(noise=noise.to(dev), model=net, model_kwargs=[{"y": y.to(dev)}, {"y": neg.to(dev)}, {"hint": hint.to(dev)}], guide_scale=7.5,
                       guide_rescale=0.2, solver="dpmpp_2m_sde", solver_mode="normal", steps=CFG16["steps"], t_max=899, t_min=0,
                       discretization="trailing", chunk_inds=None, noise_sampler_cls=Sampler).cpu()
    return out, gold


@pytest.mark.gpu
def test_hip_fp16_is_as_close_to_fp32_as_the_references_own_fp16():
    """"within a stated fp16 tolerance" (BASELINE.json): tests/golden/fp16ref_small.pt holds the final latent of the REFERENCE's own
    UNet + sampler run twice on the CPU by oracle/make_golden_fp16ref.py -- in fp32, and with half weights under autocast(float16),
    the arithmetic of its GPU path (video_to_video_model.py:42,98).  The HIP fp16 path on the same inputs, weights and injected
    noise must be at least as close to the fp32 result as the reference's own fp16 result is (10 % slack), and meet the 50 dB bar."""
    out, gold = _run_fp16ref_case(torch.float16)
    ours, theirs = parity_metrics(out, gold["x0_fp32"]), parity_metrics(gold["x0_ref_fp16"], gold["x0_fp32"])
    print(f"final latent after 6 forwards vs the reference in fp32:  HIP fp16 path {fmt_metrics(ours)}  |  the reference's own fp16 (half + "
          f"autocast) {fmt_metrics(theirs)}")
    assert torch.isfinite(out).all() and ours["psnr_range"] >= 50.0, ours
    assert ours["rel_rms"] <= 1.1 * theirs["rel_rms"], (ours, theirs)
