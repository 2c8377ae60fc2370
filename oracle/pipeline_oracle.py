"""CPU oracle of the whole `VideoToVideo_sr.test()` pipeline (TEST INFRASTRUCTURE ONLY).

Restates video_to_video/video_to_video_model.py:75-161 with the CPU oracles of the denoiser (unet_oracle.py, pinned
against the reference's own code) and of the VAE (vae_oracle.py, parity unpinned), in fp32.  The sampler is
star_amd.diffusion, which tests/test_sampler.py pins bit-exactly against the reference's diffusion modules.  All
randomness comes from one CPU generator in the reference's consumption order: VAE posterior noise per frame ->
diffuse noise -> one N(0,1) tensor per solver step.
"""
import torch
import torch.nn.functional as F

import unet_oracle as O
import vae_oracle as VO
from star_amd.diffusion import GaussianDiffusion, noise_schedule
from star_amd.geometry import make_chunks, pad_to_fit


def run_pipeline(sd, vsd, ucfg, vcfg, video, y, neg_y, target_res, gen, total_noise_levels=900, steps=50, solver_mode="fast",
                 guide_scale=7.5, max_chunk_len=32):
    video = F.interpolate(video.float(), list(target_res), mode="bilinear")
    frames, _, h, w = video.shape
    padding = pad_to_fit(h, w)
    video = F.pad(video, padding, "constant", 1)
    zs = []
    for i in range(frames):   # vae_encode: one frame per call, .sample(), x scaling_factor (:153-161)
        mom = VO.encode_moments(vsd, vcfg, video[i:i + 1])
        zs.append(VO.sample_posterior(mom, torch.randn(mom[:, :vcfg.latent_channels].shape, generator=gen)))
    z = torch.cat(zs).unsqueeze(0).permute(0, 2, 1, 3, 4) * vcfg.scaling_factor
    sig = noise_schedule(schedule="logsnr_cosine_interp", n=1000, zero_terminal_snr=True, scale_min=2.0, scale_max=4.0)
    gd = GaussianDiffusion(sig)
    t = torch.LongTensor([total_noise_levels - 1])
    noised = gd.diffuse(z, t, noise=torch.randn(z.shape, generator=gen))

    def model(x, t=None, y=None, hint=None, hint_chunk=None, variant_info=None):
        return O.unet_forward(sd, ucfg, x, t, y, hint_chunk if hint_chunk is not None else hint)

    class Sampler:
        def __init__(self, x, a, b, seed=None):
            self.shape = x.shape

        def __call__(self, s, sn):
            return torch.randn(self.shape, generator=gen)

    chunks = make_chunks(frames, 0, max_chunk_len) if frames > max_chunk_len else None
    x0 = gd.sample_sr(noise=noised, model=model, model_kwargs=[{"y": y}, {"y": neg_y}, {"hint": z}], guide_scale=guide_scale,
                      guide_rescale=0.2, solver="dpmpp_2m_sde", solver_mode=solver_mode, steps=steps, t_max=total_noise_levels - 1,
                      t_min=0, discretization="trailing", chunk_inds=chunks, noise_sampler_cls=Sampler)
    zf = x0.permute(0, 2, 1, 3, 4).reshape(-1, x0.shape[1], x0.shape[3], x0.shape[4])
    outs = []
    for i in range(0, frames, 3):   # vae_decode_chunk(chunk_size=3) (:144-151)
        n = min(3, frames - i)
        outs.append(VO.decode(vsd, vcfg, zf[i:i + n] / vcfg.scaling_factor, n))
    vid = torch.cat(outs)
    w1, w2, h1, h2 = padding
    vid = vid[:, :, h1:h + h1, w1:w + w1]
    return vid.reshape(1, frames, *vid.shape[1:]).permute(0, 2, 1, 3, 4).float(), {"latent_x0": x0, "z": z}
