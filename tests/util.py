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
