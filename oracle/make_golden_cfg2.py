"""Full-width parity fixture on BASELINE config[1]'s OWN geometry (TEST INFRASTRUCTURE ONLY; needs /root/reference).

    nice python oracle/make_golden_cfg2.py       # 1.14 PFLOP: 4.4 h on 8 shared cores (2.4 h per forward; the CPU flash-attention of the
                                                 # 7 level-0 layers dominates), ~35 GB of RAM

cfg2 = 32 frames, latent 122x216 (240x426 -> x4 = 960x1704, padded to 976x1728): level sizes 122 -> 62 -> 32 -> 17 rows, the
5-D GroupNorms and temporal attention over all 32 frames x 26 352 pixels, four temporal blocks per level.  Until round 4 that
shape had only been compared with itself on the GPU; every reference comparison was at f <= 11 and latent 90x160.

What runs here, in fp32 on the CPU:
  * the REFERENCE's own `ControlledV2VUNet` + `VideoControlNet` (video_to_video/modules/unet_v2v.py:1717-1809, imported by
    oracle/ref_loader.py) at full width, weights `random_state_dict(UNetConfig(), seed=0)`;
  * ONE classifier-free-guidance pair through the REFERENCE's own `GaussianDiffusion.denoise`
    (video_to_video/diffusion/diffusion_sdedit.py:44-115) at t = 899: y_out, u_out, CFG 7.5 with the 0.2 std-rescale, v -> x0.
The VAE is not involved (it is unpinned anyway): the hint latent z is drawn from a seeded generator with the statistics of the
cfg1 VAE latents (std 0.21, half of the energy in a smooth low-frequency field), xt = alpha_899 z + sigma_899 eps as `diffuse`.

Stored: x0 (fp32, 13.5 MB) and the two raw denoiser outputs y_out / u_out (fp16: quantisation 72 dB below their range).
All inputs are re-derived from the seeds by tests/test_parity_cfg2.py.
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

GOLD = os.path.join(ROOT, "tests", "golden")

CFG2 = dict(frames=32, latent=(122, 216), t=899, guide_scale=7.5, guide_rescale=0.2, seed=2666, wseed=0)


def cfg2_inputs(cfg=CFG2):
    """hint latent z [1,4,F,h,w], SDEdit noise eps, the two text contexts -- all from one seeded CPU generator."""
    g = torch.Generator().manual_seed(cfg["seed"])
    f, (h, w) = cfg["frames"], cfg["latent"]
    coarse = torch.randn(1, 4, f // 4 + 1, h // 8 + 1, w // 8 + 1, generator=g)
    smooth = F.interpolate(coarse, size=(f, h, w), mode="trilinear", align_corners=True)
    smooth = smooth / smooth.std()
    z = 0.21 * (smooth + torch.randn(1, 4, f, h, w, generator=g)) / 2 ** 0.5
    eps = torch.randn(1, 4, f, h, w, generator=g)
    y = torch.randn(1, 77, 1024, generator=g)
    neg = torch.randn(1, 77, 1024, generator=g)
    return z, eps, y, neg


def main(t_eval=None, out_name="cfg2_pair.pt"):
    """t_eval: timestep of the pair (default CFG2['t'] = 899, the first evaluation; `python oracle/make_golden_cfg2.py 449` writes the
    mid-trajectory fixture cfg2_pair_t449.pt from the same z / eps / contexts)"""
    import ref_loader
    from star_amd.topology import UNetConfig, random_state_dict
    torch.set_grad_enabled(False)
    torch.set_num_threads(int(os.environ.get("STAR_GOLDEN_THREADS", os.cpu_count())))
    assert ref_loader.reference_available()
    t0 = time.time()
    m = ref_loader.load_unet_module()
    dif, sol, sch = ref_loader.load_diffusion_modules()
    net = m.ControlledV2VUNet().eval()
    net.load_state_dict(random_state_dict(UNetConfig(), seed=CFG2["wseed"]), strict=True)
    print("model built", time.time() - t0, flush=True)

    z, eps, y, neg = cfg2_inputs()
    sig = sch.noise_schedule(schedule="logsnr_cosine_interp", n=1000, zero_terminal_snr=True, scale_min=2.0, scale_max=4.0)
    gd = dif.GaussianDiffusion(sigmas=sig)
    cfg = dict(CFG2, t=int(t_eval)) if t_eval is not None else dict(CFG2)
    t = torch.LongTensor([cfg["t"]])
    xt = gd.diffuse(z, t, noise=eps)
    raw = []

    def model(x, t=None, y=None, hint=None, hint_chunk=None, variant_info=None):
        out = net(x, t=t, y=y, hint=hint, hint_chunk=hint_chunk, variant_info=variant_info)
        raw.append(out.clone())
        print("forward", len(raw), time.time() - t0, flush=True)
        return out

    x0 = gd.denoise(xt, t, None, model, [{"y": y}, {"y": neg}, {"hint": z}], CFG2["guide_scale"], CFG2["guide_rescale"])[-2]
    assert len(raw) == 2 and torch.isfinite(x0).all()
    torch.save({"cfg": cfg, "x0": x0.clone(), "y_out_f16": raw[0].to(torch.float16), "u_out_f16": raw[1].to(torch.float16),
                "xt_sum": float(xt.double().sum()), "x0_range": (float(x0.min()), float(x0.max()))},
               os.path.join(GOLD, out_name))
    print("wrote", out_name, tuple(x0.shape), time.time() - t0)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        main(int(sys.argv[1]), f"cfg2_pair_t{int(sys.argv[1])}.pt")
    else:
        main()
