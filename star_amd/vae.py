"""Host-side mirror of the VAE object the pipeline uses (B3 in SURVEY.md section 8b):
`vae.encode(x).latent_dist.sample()`, `vae.decode(z, num_frames=n).sample`, `vae.config.scaling_factor`
(video_to_video_model.py:141-161).  Arithmetic runs in libstar_hip.so (star_vae_encode / star_vae_decode)."""
import ctypes
from types import SimpleNamespace

import torch

from . import lib as L
from .modules.unet_v2v import stage_tensor
from .vae_topology import VaeConfig, vae_param_shapes


class VaeConfigC(ctypes.Structure):
    _fields_ = [("in_ch", ctypes.c_int32), ("out_ch", ctypes.c_int32), ("latent", ctypes.c_int32), ("n_blocks", ctypes.c_int32),
                ("block_out", ctypes.c_int32 * 8), ("layers_per_block", ctypes.c_int32)]


def _bind(lib):
    if getattr(lib, "_vae_bound", False):
        return
    c = lib.cdll
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    lib.vae_build = L._sig(c, "star_vae_build", i32, vp, ctypes.POINTER(VaeConfigC))
    lib.vae_encode = L._sig(c, "star_vae_encode", i32, vp, vp, vp, i32, i32, i32)
    lib.vae_decode = L._sig(c, "star_vae_decode", i32, vp, vp, vp, i32, i32, i32)
    lib._vae_bound = True


class DiagonalGaussianDistribution:
    """mean / logvar (clamped to [-30, 20]) with `.sample()` = mean + std * randn on the parameters' device."""

    def __init__(self, moments):
        self.parameters = moments
        self.mean, logvar = torch.chunk(moments, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, device=self.parameters.device, dtype=self.parameters.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class AutoencoderKLTemporalDecoder:
    def __init__(self, cfg: VaeConfig = VaeConfig(), dtype=torch.float16, device=0, library=None):
        self.cfg = cfg
        self.config = SimpleNamespace(scaling_factor=cfg.scaling_factor, force_upcast=True, latent_channels=cfg.latent_channels)
        self.dtype = dtype
        self._device, self._library = device, library
        self.ctx = None

    def eval(self):
        return self

    def requires_grad_(self, flag=False):
        return self

    def to(self, device):
        return self

    def load_state_dict(self, sd, strict=True):
        want = vae_param_shapes(self.cfg)
        missing = [k for k in want if k not in sd]
        if missing:
            raise L.StarError(f"VAE load_state_dict: {len(missing)} tensors missing (first: {missing[:3]})")
        if self.ctx is not None:
            self.ctx.close()
        self.ctx = L.Context(self._device, self.dtype, self._library)
        _bind(self.ctx.lib)
        for k, shp in want.items():
            if tuple(sd[k].shape) != tuple(shp):
                raise L.StarError(f"VAE load_state_dict: {k} has shape {tuple(sd[k].shape)}, expected {tuple(shp)}")
            stage_tensor(self.ctx, k, sd[k])
        c = VaeConfigC()
        c.in_ch, c.out_ch, c.latent = self.cfg.in_channels, self.cfg.out_channels, self.cfg.latent_channels
        c.n_blocks = len(self.cfg.block_out_channels)
        for i, v in enumerate(self.cfg.block_out_channels):
            c.block_out[i] = v
        c.layers_per_block = self.cfg.layers_per_block
        self.ctx._check(self.ctx.lib.vae_build(self.ctx.h, ctypes.byref(c)), "vae_build")
        return self

    def encode(self, x):
        """x: [n, 3, H, W] -> object with .latent_dist (moments [n, 2L, H/f, W/f] fp32)."""
        ctx = self.ctx
        ctx.use_current_stream()
        n, _, H, W = x.shape
        f = self.cfg.downsample
        xf = x.to(device=ctx.torch_device, dtype=torch.float32).contiguous()
        L2 = 2 * self.cfg.latent_channels
        rows = torch.empty(n * (H // f) * (W // f), L2, dtype=torch.float32, device=ctx.torch_device)
        ctx._check(ctx.lib.vae_encode(ctx.h, L._ptr(xf), L._ptr(rows), n, H, W), "vae_encode")
        moments = rows.reshape(n, H // f, W // f, L2).permute(0, 3, 1, 2).contiguous()
        return SimpleNamespace(latent_dist=DiagonalGaussianDistribution(moments))

    def decode(self, z, num_frames=1):
        """z: [n, 4, h, w] with n == num_frames (the reference decodes one <=3-frame group per call) -> .sample [n, 3, 8h, 8w]."""
        ctx = self.ctx
        ctx.use_current_stream()
        n, _, h, w = z.shape
        f = self.cfg.downsample
        outs = []
        for g0 in range(0, n, num_frames):    # batch > 1: independent temporal groups
            zz = z[g0:g0 + num_frames].to(device=ctx.torch_device, dtype=torch.float32).contiguous()
            out = torch.empty(zz.shape[0], self.cfg.out_channels, h * f, w * f, dtype=torch.float32, device=ctx.torch_device)
            ctx._check(ctx.lib.vae_decode(ctx.h, L._ptr(zz), L._ptr(out), zz.shape[0], h, w), "vae_decode")
            outs.append(out)
        out = torch.cat(outs) if len(outs) > 1 else outs[0]
        return SimpleNamespace(sample=out if z.dtype == torch.float32 else out.to(z.dtype))
