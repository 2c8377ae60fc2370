"""GaussianDiffusion: v-prediction -> x0, classifier-free guidance with rescale, the per-step chunk loop and the
timestep/sigma ladders (host Python; the denoiser it calls is the HIP UNet).

Restates video_to_video/diffusion/diffusion_sdedit.py: `diffuse` :26-30, `denoise` :44-115 (only the x0 it returns is
used by the SR sampler), `sample_sr` :265-411 with `model_fn` :315-327 / `model_chunk_fn` :330-353,
`_sigma_to_t` :415-433, `_t_to_sigma` :435-442.  Chunk execution is pluggable so that the chunks of one solver step
can be sharded over GPUs (star_amd/parallel.py); everything else is bit-for-bit the reference's arithmetic in fp32.
"""
import random

import torch

from .solvers_sdedit import sample_dpmpp_2m_sde


def _at(tensor, t, x):
    return tensor[t.to(tensor.device)].view((x.size(0),) + (1,) * (x.ndim - 1)).to(x.device)


class GaussianDiffusion:
    def __init__(self, sigmas):
        self.sigmas = sigmas
        self.alphas = torch.sqrt(1 - sigmas ** 2)
        self.num_timesteps = len(sigmas)

    def diffuse(self, x0, t, noise=None):
        noise = torch.randn_like(x0) if noise is None else noise
        return _at(self.alphas, t, x0) * x0 + _at(self.sigmas, t, x0) * noise

    def get_velocity(self, x0, xt, t):
        return (_at(self.alphas, t, xt) * xt - x0) / _at(self.sigmas, t, xt)

    def get_x0(self, v, xt, t):
        return _at(self.alphas, t, xt) * xt - _at(self.sigmas, t, xt) * v

    def denoise_x0(self, xt, t, model, model_kwargs, guide_scale=None, guide_rescale=None):
        """x0 of `denoise` (diffusion_sdedit.py:76-99): two sequential denoiser calls (cond, uncond), CFG,
        std-rescale, v -> x0."""
        sig, alp = _at(self.sigmas, t, xt), _at(self.alphas, t, xt)
        if guide_scale is None:
            out = model(xt, t=t, **model_kwargs)
        else:
            cond = {**model_kwargs[0], **model_kwargs[2]}
            unc = {**model_kwargs[1], **model_kwargs[2]}
            pair = getattr(model, "forward_cfg_pair", None) if guide_scale != 1.0 else None
            if pair is not None:
                # same two evaluations as the reference's sequential calls (bit-identical), sharing the part of the
                # denoiser that does not see the text context (star_unet_forward_cfg)
                y_out, u_out = pair(xt, t, cond.pop("y"), unc.pop("y"), **cond)
            else:
                y_out = model(xt, t=t, **cond)
            if guide_scale == 1.0:
                out = y_out
            else:
                if pair is None:
                    u_out = model(xt, t=t, **unc)
                out = u_out + guide_scale * (y_out - u_out)
                if guide_rescale is not None:
                    ratio = (y_out.flatten(1).std(dim=1) / (out.flatten(1).std(dim=1) + 1e-12)).view((-1,) + (1,) * (y_out.ndim - 1))
                    out = out * (guide_rescale * ratio + (1 - guide_rescale))
        return alp * xt - sig * out

    # ------------------------------------------------------------------ ladders
    def _log_sigmas(self, like):
        return torch.sqrt(self.sigmas ** 2 / (1 - self.sigmas ** 2)).log().to(like)

    def _sigma_to_t(self, sigma):
        if sigma == float("inf"):
            t = torch.full_like(sigma, len(self.sigmas) - 1)
        else:
            ls = self._log_sigmas(sigma)
            log_sigma = sigma.log()
            dists = log_sigma - ls[:, None]
            low = dists.ge(0).cumsum(dim=0).argmax(dim=0).clamp(max=ls.shape[0] - 2)
            high = low + 1
            w = ((ls[low] - log_sigma) / (ls[low] - ls[high])).clamp(0, 1)
            t = ((1 - w) * low + w * high).view(sigma.shape)
        return t.unsqueeze(0) if t.ndim == 0 else t

    def _t_to_sigma(self, t):
        t = t.float()
        lo, hi, w = t.floor().long(), t.ceil().long(), t.frac()
        ls = self._log_sigmas(t)
        log_sigma = (1 - w) * ls[lo] + w * ls[hi]
        log_sigma[torch.isnan(log_sigma) | torch.isinf(log_sigma)] = float("inf")
        return log_sigma.exp()

    def sr_sigmas(self, steps, solver_mode, t_max, t_min, device="cpu"):
        """The k-diffusion sigma ladder handed to the solver ('trailing' discretisation, penultimate step dropped;
        diffusion_sdedit.py:356-406).  `fast` = 4 steps over [t_max, 500] + 11 steps over [500, t_min]."""
        n = steps + 1
        ts = torch.arange(t_max, t_min - 1, -((t_max - t_min + 1) / n))
        if solver_mode == "fast":
            t_mid = 500
            ts = torch.concat([torch.arange(t_max, t_mid - 1, -((t_max - t_mid + 1) / 4)),
                               torch.arange(t_mid, t_min - 1, -((t_mid - t_min + 1) / 11))])
        ts = torch.as_tensor(ts.clamp_(t_min, t_max), dtype=torch.float32, device=device)
        sig = self._t_to_sigma(ts)
        sig = torch.cat([sig, sig.new_zeros([1])])
        return torch.cat([sig[:-2], sig[-1:]])

    # ------------------------------------------------------------------ sampling
    @staticmethod
    def chunk_core(i, n_chunks, cur_f, o_len):
        """frames of chunk i's x0 that survive the overlap trim (diffusion_sdedit.py:345-350)."""
        cut = o_len // 2
        if i == 0:
            return 0, cur_f + cut - o_len
        if i == n_chunks - 1:
            return cut, cur_f
        return cut, cur_f + cut - o_len

    @torch.no_grad()
    def sample_sr(self, noise, model, model_kwargs={}, condition_fn=None, guide_scale=None, guide_rescale=None,
                  clamp=None, percentile=None, solver="dpmpp_2m_sde", solver_mode="fast", steps=20, t_max=None,
                  t_min=None, discretization=None, discard_penultimate_step=None, return_intermediate=None,
                  show_progress=False, seed=-1, chunk_inds=None, variant_info=None, chunk_executor=None, **kwargs):
        assert solver == "dpmpp_2m_sde", "the SR pipeline only uses DPM-Solver++(2M) SDE (video_to_video_model.py:109)"
        assert discretization in (None, "trailing"), "video_to_video_model.py:121 passes 'trailing'"
        assert isinstance(steps, int) and clamp is None and percentile is None and return_intermediate is None
        _ = seed if seed >= 0 else random.randint(0, 2 ** 31)   # reference consumes python's RNG here (:297)
        t_max = self.num_timesteps - 1 if t_max is None else t_max
        t_min = 0 if t_min is None else t_min
        sigmas = self.sr_sigmas(steps, solver_mode, t_max, t_min, device=noise.device)

        def model_fn(xt, sigma):
            t = self._sigma_to_t(sigma).repeat(len(xt)).round().long()
            return self.denoise_x0(xt, t, model, model_kwargs, guide_scale, guide_rescale)

        def model_chunk_fn(xt, sigma):
            t = self._sigma_to_t(sigma).repeat(len(xt)).round().long()
            if len(chunk_inds) < 2:
                raise IndexError("chunk_inds has a single chunk: the reference indexes chunk_inds[1] "
                                 "(diffusion_sdedit.py:333); use max_chunk_len >= frames instead")
            o_len = chunk_inds[0][-1] - chunk_inds[1][0]
            hint_full = model_kwargs[2]["hint"]

            def run_chunk(i):
                s, e = chunk_inds[i]
                kw = [model_kwargs[0], model_kwargs[1], {**model_kwargs[2], "hint_chunk": hint_full[:, :, s:e].clone()}]
                x0c = self.denoise_x0(xt[:, :, s:e].clone(), t, model, kw, guide_scale, guide_rescale)
                a, b = self.chunk_core(i, len(chunk_inds), e - s, o_len)
                return x0c[:, :, a:b]

            if chunk_executor is not None:
                cores = chunk_executor(run_chunk, len(chunk_inds), key=(tuple(xt.shape), tuple(map(tuple, chunk_inds))))
            else:
                cores = [run_chunk(i) for i in range(len(chunk_inds))]
            return torch.concat(cores, dim=2)

        fn = model_chunk_fn if chunk_inds is not None else model_fn
        return sample_dpmpp_2m_sde(noise, fn, sigmas, **kwargs)
