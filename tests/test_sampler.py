"""Host-side sampler logic against fixtures produced by the reference's own diffusion code
(tests/golden/sampler.pt <- oracle/make_golden.py: schedules_sdedit.py, diffusion_sdedit.py, solvers_sdedit.py,
and the geometry helpers of video_to_video_model.py executed unmodified)."""
import os

import pytest
import torch

from star_amd.diffusion import GaussianDiffusion, noise_schedule
from star_amd.geometry import make_chunks, pad_to_fit

GOLD = torch.load(os.path.join(os.path.dirname(__file__), "golden", "sampler.pt"))


@pytest.fixture(scope="module")
def gd():
    sig = noise_schedule(schedule="logsnr_cosine_interp", n=1000, zero_terminal_snr=True, scale_min=2.0, scale_max=4.0)
    return GaussianDiffusion(sigmas=sig)


def test_schedule_bit_exact(gd):
    assert torch.equal(gd.sigmas, GOLD["sigmas"]) and torch.equal(gd.alphas, GOLD["alphas"])
    # SURVEY.md appendix B.1 spot values
    assert abs(float(gd.sigmas[899]) - 0.999094) < 1e-6 and abs(float(gd.alphas[899]) - 0.042566) < 1e-6


@pytest.mark.parametrize("mode,steps", [("fast", 15), ("normal", 50), ("normal", 5), ("normal", 7)])
def test_sigma_ladders(gd, mode, steps):
    want = GOLD[f"ladder_{mode}_{steps}"]
    sig = gd.sr_sigmas(steps, mode, 899, 0)
    assert torch.equal(sig, want["sigmas"])
    ts = torch.stack([gd._sigma_to_t(s).round().long()[0] for s in sig[:-1]])
    assert torch.equal(ts, want["t"])
    assert len(sig) - 1 == (14 if mode == "fast" else steps)


def test_geometry_tables():
    for (h, w), pad in GOLD["pad_to_fit"].items():
        assert pad_to_fit(h, w) == pad
        ph, pw = h + pad[2] + pad[3], w + pad[0] + pad[1]
        assert (ph // 8) % 8 == 2 and (pw // 8) % 8 == 0
    for (f, mx), chunks in GOLD["make_chunks"].items():
        assert make_chunks(f, 0, mx) == chunks


def _toy_model(A):
    def model(x, t, y=None, hint=None, hint_chunk=None, variant_info=None):
        h_ = hint_chunk if hint_chunk is not None else hint
        return torch.einsum("oc,bcfhw->bofhw", A, x) * (1.0 + 0.1 * float(y.mean())) + 0.05 * h_ + 0.001 * float(t[0])
    return model


def test_denoise_cfg_rescale_x0(gd):
    d = GOLD["denoise"]
    x0 = gd.denoise_x0(d["xt"], d["t"], _toy_model(d["A"]), [{"y": d["y1"]}, {"y": d["y2"]}, {"hint": d["hint"]}], 7.5, 0.2)
    assert torch.allclose(x0, d["x0"], rtol=0, atol=1e-6)


class _InjectedNoise:   # the same seeded source make_golden.py injected into the reference solver
    def __init__(self, x, smin, smax, seed=None):
        self.g = torch.Generator().manual_seed(1234)
        self.shape = x.shape

    def __call__(self, s, s_next):
        return torch.randn(self.shape, generator=self.g)


@pytest.mark.parametrize("name", ["nochunk", "chunked", "chunked3"])
def test_sample_sr_trajectory(gd, name):
    """full solver loop incl. the per-step chunk loop / overlap trim, with identical injected noise."""
    d, s = GOLD["denoise"], GOLD[f"sample_{name}"]
    x0 = gd.sample_sr(noise=s["noise"], model=_toy_model(d["A"]), model_kwargs=[{"y": d["y1"]}, {"y": d["y2"]}, {"hint": s["hint"]}],
                      guide_scale=7.5, guide_rescale=0.2, solver="dpmpp_2m_sde", solver_mode=s["mode"], steps=s["steps"],
                      t_max=899, t_min=0, discretization="trailing", chunk_inds=s["chunks"], noise_sampler_cls=_InjectedNoise)
    assert x0.shape == s["x0"].shape
    assert torch.allclose(x0, s["x0"], rtol=1e-5, atol=1e-5), float((x0 - s["x0"]).abs().max())


def test_single_chunk_list_raises_like_reference(gd):
    d = GOLD["denoise"]
    with pytest.raises(IndexError):
        gd.sample_sr(noise=torch.zeros(1, 4, 33, 10, 8), model=_toy_model(d["A"]),
                     model_kwargs=[{"y": d["y1"]}, {"y": d["y2"]}, {"hint": torch.zeros(1, 4, 33, 10, 8)}], guide_scale=7.5,
                     guide_rescale=0.2, steps=3, solver_mode="normal", t_max=899, t_min=0, discretization="trailing",
                     chunk_inds=[(0, 33)], noise_sampler_cls=_InjectedNoise)
