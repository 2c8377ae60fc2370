"""DPM-Solver++(2M) SDE on k-diffusion sigmas (host Python over device tensors).

Restates video_to_video/diffusion/solvers_sdedit.py:144-204 (`sample_dpmpp_2m_sde`, eta = 1, s_noise = 1,
midpoint correction).  The reference draws its noise from a torchsde Brownian tree seeded from the global RNG
(:83-140); the solver only ever asks for increments over consecutive disjoint sigma intervals, which after the
1/sqrt(|dt|) normalisation (:137-140) are i.i.d. N(0, 1).  `BrownianIntervalNoise` therefore consumes the global
RNG exactly like the reference (one `torch.randint` for the seed) and then draws one N(0,1) tensor per step from
its own generator; the real tree is used instead whenever torchsde is importable.
"""
import torch


class BrownianIntervalNoise:
    def __init__(self, x, sigma_min, sigma_max, seed=None):
        if seed is None:
            seed = int(torch.randint(0, 2 ** 63 - 1, []).item())   # same global-RNG consumption as solvers_sdedit.py:79-80
        self.shape, self.device, self.dtype = x.shape, x.device, x.dtype
        self._tree = None
        try:
            import torchsde  # noqa: F401
            t0, t1 = torch.as_tensor(sigma_min), torch.as_tensor(sigma_max)
            t0, t1 = (t0, t1) if t0 < t1 else (t1, t0)
            self._tree = torchsde.BrownianTree(t0, torch.zeros_like(x), t1, entropy=seed)
        except ImportError:
            self.gen = torch.Generator(device="cpu").manual_seed(seed % (2 ** 63 - 1))

    def __call__(self, sigma, sigma_next):
        if self._tree is not None:
            t0, t1 = torch.as_tensor(sigma), torch.as_tensor(sigma_next)
            lo, hi, sign = (t0, t1, 1) if t0 < t1 else (t1, t0, -1)
            return self._tree(lo, hi) * sign / (t1 - t0).abs().sqrt()
        return torch.randn(self.shape, generator=self.gen, dtype=torch.float32).to(device=self.device, dtype=self.dtype)


def sample_dpmpp_2m_sde(noise, model, sigmas, eta=1.0, s_noise=1.0, noise_sampler_cls=BrownianIntervalNoise,
                        step_callback=None, **unused):
    """x0 = solver(noise * sigma_0): `model(x_in, sigma)` returns the denoised estimate of the FULL-length latent."""
    x = noise * sigmas[0]
    pos = sigmas[sigmas > 0]
    sampler = noise_sampler_cls(x, pos.min(), sigmas[sigmas < float("inf")].max())
    old_denoised, h_last = None, None
    for i in range(len(sigmas) - 1):
        s, s_next = sigmas[i], sigmas[i + 1]
        c_in = 1.0 / (s ** 2 + 1.0) ** 0.5                 # get_scalings, solvers_sdedit.py:21-24
        denoised = model(x * c_in, s)
        if s_next == 0:
            x = denoised
            h = None
        else:
            t, t_next = -s.log(), -s_next.log()
            h = t_next - t
            eta_h = eta * h
            x = s_next / s * (-eta_h).exp() * x + (-h - eta_h).expm1().neg() * denoised
            if old_denoised is not None:
                r = h_last / h
                x = x + 0.5 * (-h - eta_h).expm1().neg() * (1 / r) * (denoised - old_denoised)   # midpoint
            x = x + sampler(s, s_next) * s_next * (-2 * eta_h).expm1().neg().sqrt() * s_noise
        old_denoised, h_last = denoised, h
        if step_callback is not None:
            step_callback(i, x)
    return x
