"""Shared helpers for the parity tests."""
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BACKENDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]
DTYPES = [pytest.param(torch.float16, id="f16"), pytest.param(torch.bfloat16, id="bf16")]

# unit-kernel tolerances relative to the fp32 torch reference of the same op, in units of the
# output dtype's epsilon times the output magnitude (fp32 accumulation everywhere)
EPS = {torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}


def make_ctx(backend, dtype, emu_lib=None):
    from star_amd import lib as L
    if backend == "emu":
        return L.Context(0, dtype, emu_lib)
    return L.Context(0, dtype)


def assert_close(out, ref, dtype, scale=4.0, what=""):
    out = out.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert out.shape == ref.shape, f"{what}: shape {tuple(out.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(out).all(), f"{what}: non-finite output"
    tol = scale * EPS[dtype] * max(1.0, float(ref.abs().max()))
    err = float((out - ref).abs().max())
    assert err <= tol, f"{what}: max abs err {err:.4g} > tol {tol:.4g}"
    return err


def parity_metrics(out, ref, nominal_peak=None):
    """The figures every PSNR-type parity test prints and asserts on (round-2 review: the peak of a PSNR is a free parameter, so
    state it).  `range`: peak = max - min of the reference tensor (what the tests of rounds 1-2 used; the only meaningful peak for
    latents and for random-init weights, whose decoded frames span [-3, 3.4] instead of [-1, 1]).  `nominal`: peak = nominal_peak
    (2.0 for frames that nominally live in [-1, 1]).  `rel_rms`: range-free, ||out - ref|| / ||ref||."""
    import math
    o, r = out.detach().double().cpu(), ref.detach().double().cpu()
    mse = float((o - r).pow(2).mean())
    rng = float(r.max() - r.min())
    m = {"rel_rms": math.sqrt(mse) / max(float(r.pow(2).mean().sqrt()), 1e-30), "range": rng,
         "psnr_range": 10 * math.log10(rng ** 2 / max(mse, 1e-30))}
    if nominal_peak is not None:
        m["psnr_nominal"] = 10 * math.log10(nominal_peak ** 2 / max(mse, 1e-30))
    return m


def fmt_metrics(m):
    s = f"PSNR {m['psnr_range']:.1f} dB at peak = range {m['range']:.2f}"
    if "psnr_nominal" in m:
        s += f", {m['psnr_nominal']:.1f} dB at the nominal peak 2.0"
    return s + f", relative rms {m['rel_rms']:.2e}"


def frames_u8(x):
    """what save_video writes: tensor2vid's [-1, 1] -> [0, 255] map, clamped and rounded (inference_utils.py:16-23)"""
    return ((x.detach().float().cpu() + 1.0) * 127.5).clamp(0, 255).round()
