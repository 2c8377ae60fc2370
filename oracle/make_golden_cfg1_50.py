"""Long-horizon full-width parity fixture: 50 stochastic solver steps = 100 denoiser forwards (TEST INFRASTRUCTURE ONLY;
needs /root/reference).

    python oracle/make_golden_cfg1_50.py         # ~1.5-2 h on 8 cores

The reference runs `solver_mode='normal', steps=50` as its quality setting (video_super_resolution/scripts/inference_sr.py:43,
diffusion_sdedit.py:356-411); tests/golden/cfg1_full.pt stops at 5 steps.  This fixture measures the drift of the 16-bit HIP
path over a whole 50-step trajectory against the REFERENCE's own code:
  * the REFERENCE's `ControlledV2VUNet` (unet_v2v.py), full 2.04 B-parameter width, `random_state_dict(UNetConfig(), seed=0)`,
  * the REFERENCE's `GaussianDiffusion.sample_sr` / `sample_dpmpp_2m_sde` with the Brownian tree replaced by one seeded
    N(0,1) tensor per solver step (torchsde absent),
  * no VAE at all (it is parity-unpinned): the start latent is the first 4 frames of cfg1_full.pt's `z`,
on 4 frames, latent 90x160 (cfg1's level-0 geometry, 23.6 TFLOP per forward).

Stored (fp32): the noised start, the x0 prediction of every 10th evaluation and the final latent.
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_loader  # noqa: E402
from star_amd.topology import UNetConfig, random_state_dict  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

CFG50 = dict(frames=4, latent=(90, 160), steps=50, solver_mode="normal", total_noise_levels=900, guide_scale=7.5,
             ctx_seed=666, rng_seed=3, wseed=0, keep_every=10)


def cfg50_inputs():
    """text contexts (N(0,1)) and the start latent (first frames of the cfg1 fixture's VAE latent)."""
    g = torch.Generator().manual_seed(CFG50["ctx_seed"])
    y = torch.randn(1, 77, 1024, generator=g)
    neg = torch.randn(1, 77, 1024, generator=g)
    z = torch.load(os.path.join(GOLD, "cfg1_full.pt"))["z"][:, :, :CFG50["frames"]].clone()
    return z, y, neg


def main():
    torch.set_grad_enabled(False)
    torch.set_num_threads(os.cpu_count())
    assert ref_loader.reference_available()
    t0 = time.time()
    m = ref_loader.load_unet_module()
    dif, sol, sch = ref_loader.load_diffusion_modules()
    net = m.ControlledV2VUNet().eval()
    net.load_state_dict(random_state_dict(UNetConfig(), seed=CFG50["wseed"]), strict=True)
    print("model built", time.time() - t0, flush=True)
    z, y, neg = cfg50_inputs()
    gen = torch.Generator().manual_seed(CFG50["rng_seed"])
    sig = sch.noise_schedule(schedule="logsnr_cosine_interp", n=1000, zero_terminal_snr=True, scale_min=2.0, scale_max=4.0)
    gd = dif.GaussianDiffusion(sigmas=sig)
    t = torch.LongTensor([CFG50["total_noise_levels"] - 1])
    noised = gd.diffuse(z, t, noise=torch.randn(z.shape, generator=gen))

    class InjectedNoise:   # stands in for BrownianTreeNoiseSampler (solvers_sdedit.py:110-140)
        def __init__(self, x, smin, smax, seed=None, transform=None):
            self.shape = x.shape

        def __call__(self, s, s_next):
            return torch.randn(self.shape, generator=gen)

    sol.BrownianTreeNoiseSampler = InjectedNoise
    kept = {}
    calls = [0]
    orig_denoise = gd.denoise

    def logged_denoise(*a, **k):
        out = orig_denoise(*a, **k)
        calls[0] += 1
        if calls[0] % CFG50["keep_every"] == 0 or calls[0] == 1:
            kept[calls[0]] = out[-2].clone() if isinstance(out, (tuple, list)) else out.clone()
        print("denoise", calls[0], time.time() - t0, flush=True)
        return out

    gd.denoise = logged_denoise

    def model(x, t=None, y=None, hint=None, hint_chunk=None, variant_info=None):
        return net(x, t=t, y=y, hint=hint, hint_chunk=hint_chunk, variant_info=variant_info)

    x0 = gd.sample_sr(noise=noised, model=model, model_kwargs=[{"y": y}, {"y": neg}, {"hint": z}], guide_scale=CFG50["guide_scale"],
                      guide_rescale=0.2, solver="dpmpp_2m_sde", solver_mode=CFG50["solver_mode"], steps=CFG50["steps"],
                      t_max=CFG50["total_noise_levels"] - 1, t_min=0, discretization="trailing", chunk_inds=None)
    torch.save({"cfg": CFG50, "noised": noised.clone(), "x0_final": x0.clone(), "x0_at": kept},
               os.path.join(GOLD, "cfg1_50step.pt"))
    print("wrote cfg1_50step.pt", tuple(x0.shape), time.time() - t0)


if __name__ == "__main__":
    main()
