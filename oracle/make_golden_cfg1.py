"""Full-width parity fixture at BASELINE config[0] geometry (TEST INFRASTRUCTURE ONLY; needs /root/reference).

    python oracle/make_golden_cfg1.py            # ~40 min on 8 cores, 20 GB of RAM

Runs `VideoToVideo_sr.test()`'s arithmetic (video_to_video/video_to_video_model.py:75-139) in fp32 on the CPU with
  * the REFERENCE's own `ControlledV2VUNet` (video_to_video/modules/unet_v2v.py, imported by oracle/ref_loader.py), the full
    2.04 B-parameter architecture with `random_state_dict(UNetConfig(), seed=0)` weights,
  * the REFERENCE's own `GaussianDiffusion.sample_sr` / `sample_dpmpp_2m_sde` (diffusion/*.py) with the Brownian tree
    replaced by one seeded N(0,1) tensor per solver step (torchsde is absent),
  * oracle/vae_oracle.py for the VAE (full width; parity unpinned, diffusers is absent),
on cfg1: 8 frames 128x128 -> x4 = 512x512, padded to 720x1280 (latent 90x160), solver_mode='normal', steps=5.

All inputs are re-derived from seeds by tests/test_parity_cfg1.py; stored are the VAE latent `z` (so that denoiser
parity can be measured on identical latents, BASELINE.json north_star), the final latent, the first step's x0 and the
decoded, cropped frames (fp16 storage: quantisation error 72 dB below the signal range, far under the 50 dB bar).
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_loader  # noqa: E402
import vae_oracle as VO  # noqa: E402
from star_amd.geometry import pad_to_fit  # noqa: E402
from star_amd.topology import UNetConfig, random_state_dict  # noqa: E402
from star_amd.vae_topology import VaeConfig, random_vae_state_dict  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

CFG1 = dict(frames=8, lr=(128, 128), target=(512, 512), steps=5, solver_mode="normal", total_noise_levels=900,
            guide_scale=7.5, video_seed=666, rng_seed=1, wseed=0)


def cfg1_inputs():
    """LR clip and the two text contexts (clamp(randn*0.5), N(0,1) embeddings)."""
    g = torch.Generator().manual_seed(CFG1["video_seed"])
    video = (torch.randn(CFG1["frames"], 3, *CFG1["lr"], generator=g) * 0.5).clamp(-1, 1)
    y = torch.randn(1, 77, 1024, generator=g)
    neg = torch.randn(1, 77, 1024, generator=g)
    return video, y, neg


def main():
    torch.set_grad_enabled(False)
    torch.set_num_threads(os.cpu_count())
    assert ref_loader.reference_available()
    t0 = time.time()
    ucfg, vcfg = UNetConfig(), VaeConfig()
    m = ref_loader.load_unet_module()
    dif, sol, sch = ref_loader.load_diffusion_modules()
    net = m.ControlledV2VUNet().eval()
    net.load_state_dict(random_state_dict(ucfg, seed=CFG1["wseed"]), strict=True)
    vsd = random_vae_state_dict(vcfg, seed=CFG1["wseed"])
    print("models built", time.time() - t0, flush=True)

    video, y, neg = cfg1_inputs()
    gen = torch.Generator().manual_seed(CFG1["rng_seed"])
    th, tw = CFG1["target"]
    video_up = F.interpolate(video.float(), [th, tw], mode="bilinear")
    padding = pad_to_fit(th, tw)
    video_up = F.pad(video_up, padding, "constant", 1)
    frames = video.shape[0]
    zs = []
    for i in range(frames):
        mom = VO.encode_moments(vsd, vcfg, video_up[i:i + 1])
        zs.append(VO.sample_posterior(mom, torch.randn(mom[:, :vcfg.latent_channels].shape, generator=gen)))
        print("encoded frame", i, time.time() - t0, flush=True)
    z = torch.cat(zs).unsqueeze(0).permute(0, 2, 1, 3, 4) * vcfg.scaling_factor

    sig = sch.noise_schedule(schedule="logsnr_cosine_interp", n=1000, zero_terminal_snr=True, scale_min=2.0, scale_max=4.0)
    gd = dif.GaussianDiffusion(sigmas=sig)
    t = torch.LongTensor([CFG1["total_noise_levels"] - 1])
    noised = gd.diffuse(z, t, noise=torch.randn(z.shape, generator=gen))

    class InjectedNoise:   # stands in for BrownianTreeNoiseSampler (solvers_sdedit.py:110-140)
        def __init__(self, x, smin, smax, seed=None, transform=None):
            self.shape = x.shape

        def __call__(self, s, s_next):
            return torch.randn(self.shape, generator=gen)

    sol.BrownianTreeNoiseSampler = InjectedNoise
    x0_steps = []
    calls = [0]
    orig_denoise = gd.denoise

    def logged_denoise(*a, **k):
        out = orig_denoise(*a, **k)
        calls[0] += 1
        print("denoise", calls[0], time.time() - t0, flush=True)
        return out

    gd.denoise = logged_denoise

    def model(x, t=None, y=None, hint=None, hint_chunk=None, variant_info=None):
        return net(x, t=t, y=y, hint=hint, hint_chunk=hint_chunk, variant_info=variant_info)

    x0 = gd.sample_sr(noise=noised, model=model, model_kwargs=[{"y": y}, {"y": neg}, {"hint": z}], guide_scale=CFG1["guide_scale"],
                      guide_rescale=0.2, solver="dpmpp_2m_sde", solver_mode=CFG1["solver_mode"], steps=CFG1["steps"],
                      t_max=CFG1["total_noise_levels"] - 1, t_min=0, discretization="trailing", chunk_inds=None)
    # first model evaluation alone (x0 prediction at t = 899 from the same noised latent): single-forward drift reference
    first = orig_denoise(noised, t, None, model, [{"y": y}, {"y": neg}, {"hint": z}], CFG1["guide_scale"], 0.2)[-2]
    zf = x0.permute(0, 2, 1, 3, 4).reshape(-1, x0.shape[1], x0.shape[3], x0.shape[4])
    outs = []
    for i in range(0, frames, 3):
        n = min(3, frames - i)
        outs.append(VO.decode(vsd, vcfg, zf[i:i + n] / vcfg.scaling_factor, n))
        print("decoded group", i, time.time() - t0, flush=True)
    vid = torch.cat(outs)
    w1, w2, h1, h2 = padding
    vid = vid[:, :, h1:th + h1, w1:tw + w1]
    out = vid.reshape(1, frames, *vid.shape[1:]).permute(0, 2, 1, 3, 4).float()
    torch.save({"cfg": CFG1, "z": z.clone(), "noised": noised.clone(), "x0_first": first.clone(), "x0_final": x0.clone(),
                "video_out_f16": out.to(torch.float16), "out_range": (float(out.min()), float(out.max()))},
               os.path.join(GOLD, "cfg1_full.pt"))
    print("wrote cfg1_full.pt", tuple(out.shape), time.time() - t0)


if __name__ == "__main__":
    main()
