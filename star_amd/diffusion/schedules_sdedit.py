"""Noise schedule of the reference sampler (host Python; defines WHAT is computed per step).

Restates video_to_video/diffusion/schedules_sdedit.py:27-84: a cosine log-SNR schedule interpolated between two
resolution shifts, mapped to sigma = sqrt(sigmoid(-logsnr)), optionally rescaled to zero terminal SNR.
"""
import math

import torch


def _logsnr_cosine(n, logsnr_min=-15.0, logsnr_max=15.0):
    t_min = math.atan(math.exp(-0.5 * logsnr_min))
    t_max = math.atan(math.exp(-0.5 * logsnr_max))
    t = torch.linspace(1, 0, n)
    return -2 * torch.log(torch.tan(t_min + t * (t_max - t_min)))


def logsnr_cosine_interp_schedule(n, logsnr_min=-15.0, logsnr_max=15.0, scale_min=2.0, scale_max=4.0):
    t = torch.linspace(1, 0, n)
    base = _logsnr_cosine(n, logsnr_min, logsnr_max)
    lo = base + 2 * math.log(1 / scale_min)
    hi = base + 2 * math.log(1 / scale_max)
    logsnr = t * lo + (1 - t) * hi
    return torch.sqrt(torch.sigmoid(-logsnr))


def noise_schedule(schedule="logsnr_cosine_interp", n=1000, zero_terminal_snr=False, **kwargs):
    if schedule != "logsnr_cosine_interp":
        raise KeyError(schedule)
    sigmas = logsnr_cosine_interp_schedule(n, **kwargs)
    if zero_terminal_snr and sigmas.max() != 1.0:
        lo = sigmas.min()
        sigmas = lo + (1.0 - lo) / (sigmas.max() - lo) * (sigmas - lo)
    return sigmas
