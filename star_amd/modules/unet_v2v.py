"""Host-side mirror of the reference denoiser object (B2 in SURVEY.md section 8b).

`ControlledV2VUNet` keeps the reference's construction / loading / call conventions
(video_to_video/modules/unet_v2v.py:1712-1809; used by video_to_video_model.py:32-42 and
diffusion_sdedit.py:81,88) but owns no torch parameters: `load_state_dict` hands the reference-keyed
tensors to the C ABI (star_load_tensor + star_unet_build), and `__call__` is one
star_unet_forward on the caller's stream.
"""
import ctypes
import os

import torch

from .. import lib as L
from ..topology import UNetConfig, param_shapes


class UNetConfigC(ctypes.Structure):
    _fields_ = [("in_dim", ctypes.c_int32), ("dim", ctypes.c_int32), ("context_dim", ctypes.c_int32),
                ("out_dim", ctypes.c_int32), ("n_levels", ctypes.c_int32), ("dim_mult", ctypes.c_int32 * 8),
                ("num_heads", ctypes.c_int32), ("head_dim", ctypes.c_int32), ("num_res_blocks", ctypes.c_int32),
                ("attn_levels", ctypes.c_int32)]


def _bind(lib):
    if getattr(lib, "_unet_bound", False):
        return
    c = lib.cdll
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    lib.load_tensor = L._sig(c, "star_load_tensor", i32, vp, ctypes.c_char_p, vp, ctypes.POINTER(i64), i32, i32)
    lib.unet_build = L._sig(c, "star_unet_build", i32, vp, ctypes.POINTER(UNetConfigC))
    lib.unet_forward = L._sig(c, "star_unet_forward", i32, vp, vp, i64, vp, vp, vp, i32, i32, i32)
    lib.controlnet_forward = L._sig(c, "star_controlnet_forward", i32, vp, vp, i64, vp, vp, ctypes.POINTER(vp), i32, i32, i32, i32)
    lib.unet_forward_cfg = L._sig(c, "star_unet_forward_cfg", i32, vp, vp, i64, vp, vp, vp, vp, vp, i32, i32, i32)
    lib.module_run = L._sig(c, "star_module_run", i32, vp, i32, ctypes.c_char_p, i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, i32, i32)
    lib.clear_staged = L._sig(c, "star_clear_staged", i32, vp)
    lib.unet_graph = L._sig(c, "star_unet_graph", i32, vp, i32)
    lib._unet_bound = True


def stage_tensor(ctx, name, t):
    """star_load_tensor: copy one host tensor (reference key name) into the context's staging area."""
    _bind(ctx.lib)
    t = t.detach().cpu().contiguous()
    if t.dtype not in L._TORCH2STAR:
        t = t.float()
    shape = (ctypes.c_int64 * max(1, t.dim()))(*t.shape)
    ctx._check(ctx.lib.load_tensor(ctx.h, name.encode(), ctypes.c_void_p(t.data_ptr()), shape, t.dim(), L._TORCH2STAR[t.dtype]),
               f"load_tensor({name})")


class _LoadResult:
    def __init__(self, missing, unexpected):
        self.missing_keys, self.unexpected_keys = missing, unexpected

    def __repr__(self):
        return "<All keys matched successfully>" if not (self.missing_keys or self.unexpected_keys) else \
            f"_IncompatibleKeys(missing_keys={self.missing_keys}, unexpected_keys={self.unexpected_keys})"


class ControlledV2VUNet:
    """Drop-in for the reference class: `ControlledV2VUNet()`, `.to(device)`, `.eval()`, `.load_state_dict(sd)`,
    `.half()` / `.bfloat16()`, then `model(xt, t=t, y=y, hint=z[, hint_chunk=...])`."""

    def __init__(self, cfg: UNetConfig = UNetConfig(), dtype=torch.float16, device=None, library=None):
        self.cfg = cfg
        self.dtype = dtype
        self._library = library
        self._device = device if device is not None else 0
        self.ctx = None
        self.batch = 1
        self.training = False
        self._pending_sd = None
        self._graph = os.environ.get("STAR_UNET_GRAPH", "0") not in ("", "0")

    # -- reference-style fluent no-ops
    def to(self, device):
        if isinstance(device, (torch.device, str)):
            self._device = L.device_index(torch.device(device))
        return self

    def eval(self):
        return self

    def requires_grad_(self, flag=False):
        return self

    def half(self):
        return self._set_dtype(torch.float16)

    def bfloat16(self):
        return self._set_dtype(torch.bfloat16)

    def _set_dtype(self, dt):
        if self.ctx is not None and dt != self.dtype:
            sd = self._pending_sd
            if sd is None:
                raise L.StarError("change the dtype before load_state_dict (weights are repacked for one dtype)")
        self.dtype = dt
        if self._pending_sd is not None:
            self._build(self._pending_sd)
        return self

    def state_dict_shapes(self):
        return param_shapes(self.cfg)

    def load_state_dict(self, sd, strict=False):
        """B4: same keys as the reference (2247 tensors at full width)."""
        want = param_shapes(self.cfg)
        missing = [k for k in want if k not in sd]
        unexpected = [k for k in sd if k not in want]
        if missing:
            raise L.StarError(f"load_state_dict: {len(missing)} tensors missing (first: {missing[:3]}); "
                              "the HIP path has no random-init fallback")
        for k, shp in want.items():
            if tuple(sd[k].shape) != tuple(shp):
                raise L.StarError(f"load_state_dict: {k} has shape {tuple(sd[k].shape)}, expected {tuple(shp)}")
        self._pending_sd = sd
        self._build(sd)
        return _LoadResult(missing, unexpected)

    def _build(self, sd):
        if self.ctx is not None:
            self.ctx.close()
        self.ctx = L.Context(self._device, self.dtype, self._library)
        _bind(self.ctx.lib)
        for k in param_shapes(self.cfg):
            stage_tensor(self.ctx, k, sd[k])
        c = UNetConfigC()
        cfg = self.cfg
        c.in_dim, c.dim, c.context_dim, c.out_dim = cfg.in_dim, cfg.dim, cfg.context_dim, cfg.out_dim
        c.n_levels = len(cfg.dim_mult)
        for i, m in enumerate(cfg.dim_mult):
            c.dim_mult[i] = m
        c.num_heads, c.head_dim, c.num_res_blocks = cfg.num_heads, cfg.head_dim, cfg.num_res_blocks
        levels = 0
        scale = 1.0
        for _ in cfg.dim_mult:
            if scale in cfg.attn_scales:
                levels += 1
            scale /= 2.0
        c.attn_levels = levels
        self.ctx._check(self.ctx.lib.unet_build(self.ctx.h, ctypes.byref(c)), "unet_build")
        if self._graph:
            self.ctx._check(self.ctx.lib.unet_graph(self.ctx.h, 1), "unet_graph")

    def use_graph(self, enable=True):
        """star_unet_graph: replay the forward of each (branches, frames, latent size) from a captured hipGraph (bit-identical results;
        default: the STAR_UNET_GRAPH environment variable, else off)."""
        self._graph = bool(enable)
        if self.ctx is not None:
            self.ctx._check(self.ctx.lib.unet_graph(self.ctx.h, int(self._graph)), "unet_graph")
        return self

    def release_host_weights(self):
        self._pending_sd = None

    def __call__(self, x, t=None, y=None, hint=None, variant_info=None, hint_chunk=None, **unused):
        if self.ctx is None:
            raise L.StarError("ControlledV2VUNet: load_state_dict first")
        if hint_chunk is not None:          # unet_v2v.py:1743-1744
            hint = hint_chunk
        b, c, f, h, w = x.shape
        if b != 1:
            raise L.StarError("the reference always calls the denoiser with batch 1 (diffusion_sdedit.py:81,88)")
        ctx = self.ctx
        ctx.use_current_stream()
        dev = ctx.torch_device
        xf = x.to(device=dev, dtype=torch.float32).contiguous()
        hf = hint.to(device=dev, dtype=torch.float32).contiguous()
        yf = y.to(device=dev, dtype=torch.float32).reshape(77, self.cfg.context_dim).contiguous()
        tt = int(t.reshape(-1)[0]) if torch.is_tensor(t) else int(t)
        out = torch.empty(1, self.cfg.out_dim, f, h, w, dtype=torch.float32, device=dev)
        ctx._check(ctx.lib.unet_forward(ctx.h, L._ptr(xf), tt, L._ptr(yf), L._ptr(hf), L._ptr(out), f, h, w), "unet_forward")
        self.batch = b
        return out if x.dtype == torch.float32 else out.to(x.dtype)


def _control_residuals(self, x, t, y, hint):
    """`VideoControlNet.forward` alone (unet_v2v.py:2134-2206) -> list of 13 float32 tensors [(f), C_l, H_l, W_l] (the reference's
    return value): 12 zero-conv'd encoder outputs + the middle block's."""
    cfg, ctx = self.cfg, self.ctx
    ctx.use_current_stream()
    dev = ctx.torch_device
    b, c, f, h, w = x.shape
    shapes = []
    ch, hh, ww = cfg.dim, h, w
    shapes.append((ch, hh, ww))                                   # input_blocks.0 (stem)
    for li, mult in enumerate(cfg.dim_mult):
        ch = cfg.dim * mult
        shapes += [(ch, hh, ww)] * cfg.num_res_blocks
        if li != len(cfg.dim_mult) - 1:
            hh, ww = hh // 2 + 1, ww // 2
            shapes.append((ch, hh, ww))                           # Downsample
    shapes.append((ch, hh, ww))                                   # middle_block_out
    bufs = [torch.empty(f * sh * sw, sc, dtype=ctx.dtype, device=dev) for (sc, sh, sw) in shapes]
    ptrs = (ctypes.c_void_p * len(bufs))(*[bb.data_ptr() for bb in bufs])
    xf = x.to(device=dev, dtype=torch.float32).contiguous()
    hf = hint.to(device=dev, dtype=torch.float32).contiguous()
    yf = y.to(device=dev, dtype=torch.float32).reshape(77, cfg.context_dim).contiguous()
    tt = int(t.reshape(-1)[0]) if torch.is_tensor(t) else int(t)
    ctx._check(ctx.lib.controlnet_forward(ctx.h, L._ptr(xf), tt, L._ptr(yf), L._ptr(hf), ptrs, len(bufs), f, h, w), "controlnet_forward")
    return [bb.float().reshape(f, sh, sw, sc).permute(0, 3, 1, 2).contiguous() for bb, (sc, sh, sw) in zip(bufs, shapes)]


ControlledV2VUNet.control_residuals = _control_residuals


def _cfg_pair(self, x, t, y_cond, y_uncond, hint=None, hint_chunk=None, **unused):
    """Both classifier-free-guidance forwards of one denoise step (diffusion_sdedit.py:81,88) in one call:
    -> (y_out, u_out), bit-identical to two __call__s; the context-independent prefix of both nets runs once."""
    if self.ctx is None:
        raise L.StarError("ControlledV2VUNet: load_state_dict first")
    if hint_chunk is not None:
        hint = hint_chunk
    b, c, f, h, w = x.shape
    if b != 1:
        raise L.StarError("the reference always calls the denoiser with batch 1 (diffusion_sdedit.py:81,88)")
    ctx = self.ctx
    ctx.use_current_stream()
    dev = ctx.torch_device
    xf = x.to(device=dev, dtype=torch.float32).contiguous()
    hf = hint.to(device=dev, dtype=torch.float32).contiguous()
    yc = y_cond.to(device=dev, dtype=torch.float32).reshape(77, self.cfg.context_dim).contiguous()
    yu = y_uncond.to(device=dev, dtype=torch.float32).reshape(77, self.cfg.context_dim).contiguous()
    tt = int(t.reshape(-1)[0]) if torch.is_tensor(t) else int(t)
    oc = torch.empty(1, self.cfg.out_dim, f, h, w, dtype=torch.float32, device=dev)
    ou = torch.empty_like(oc)
    ctx._check(ctx.lib.unet_forward_cfg(ctx.h, L._ptr(xf), tt, L._ptr(yc), L._ptr(yu), L._ptr(hf), L._ptr(oc), L._ptr(ou), f, h, w),
               "unet_forward_cfg")
    if x.dtype != torch.float32:
        oc, ou = oc.to(x.dtype), ou.to(x.dtype)
    return oc, ou


ControlledV2VUNet.forward_cfg_pair = _cfg_pair


def run_module(ctx, kind, sd, prefix, x_nchw, emb=None, context=None, heads=1, cout=None):
    """Unit-parity helper: run ONE reference module (weights `sd`, keys prefixed `prefix.`) through the HIP graph code.
    x_nchw: [F, C, H, W] float; returns [F, C', H', W'] float32."""
    _bind(ctx.lib)
    kinds = {"res": 0, "st": 1, "tt": 2, "down": 3, "up": 4}
    ctx.lib.clear_staged(ctx.h)
    for k, v in sd.items():
        stage_tensor(ctx, f"{prefix}.{k}", v)
    F_, C, H, W = x_nchw.shape
    cout = cout or C
    Ho, Wo = (H // 2 + 1, W // 2) if kind == "down" else ((2 * H - 2, 2 * W) if kind == "up" else (H, W))
    rows = x_nchw.permute(0, 2, 3, 1).reshape(-1, C).contiguous().to(device=ctx.torch_device, dtype=ctx.dtype)
    out = torch.empty(F_ * Ho * Wo, cout, dtype=ctx.dtype, device=ctx.torch_device)
    embd = emb.float().contiguous().to(ctx.torch_device) if emb is not None else None
    ctxd = context.float().contiguous().to(ctx.torch_device) if context is not None else None
    edim = embd.numel() if embd is not None else 0
    cdim = ctxd.shape[-1] if ctxd is not None else 0
    ctx._check(ctx.lib.module_run(ctx.h, kinds[kind], prefix.encode(), C, cout, heads, edim, cdim, L._ptr(rows), L._ptr(embd),
                                  L._ptr(ctxd), L._ptr(out), F_, H, W), f"module_run({kind})")
    ctx.lib.clear_staged(ctx.h)
    return out.float().reshape(F_, Ho, Wo, cout).permute(0, 3, 1, 2)
