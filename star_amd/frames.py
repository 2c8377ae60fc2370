"""Host side of the full-resolution frame kernels either side of the diffusion path (star_amd/csrc/frames.h):
resize + pad in front of the VAE encoder (video_to_video_model.py:81-87), tensor2vid + AdaIN colour fix behind the
decoder (inference_utils.py:16-23, color_fix.py:15-29).  Every function runs the HIP kernels through the C ABI on the
context of the device that holds (or is to hold) the frames; there is no torch / CPU fallback."""
import torch

from . import lib as L

_contexts = {}


def register_context(ctx):
    """Make `ctx` the context the module-level functions use for its device (VideoToVideo_sr registers its own)."""
    _contexts[ctx.device_index] = ctx
    return ctx


def context_for(device=None):
    """The registered context of `device` (index, torch.device, or None / an index-less 'cuda' = this process's current device);
    created on first use."""
    device = L.device_index(torch.device("cuda") if device is None else device)
    if device not in _contexts:
        _contexts[device] = L.Context(device, torch.float16)
    return _contexts[device]


def _on(ctx, t):
    return t.to(ctx.torch_device, torch.float32).contiguous()


def resize_pad(video, target_hw, padding, value=1.0, ctx=None):
    """[F, C, h, w] -> bilinear resize to target_hw, then constant pad (left, right, top, bottom); result on the device."""
    ctx = ctx or context_for(video.device if video.is_cuda else None)
    return ctx.resize_pad(_on(ctx, video), target_hw, padding, value)


def tensor2vid_color_fix(video, source, ctx=None, as_uint8=False):
    """tensor2vid followed by adain_color_fix in one pass over the frames: video [1, C, F, H, W] in ~[-1, 1] (as
    VideoToVideo_sr.test returns it), source [F, C, h, w] in [-1, 1]  ->  fp32 [F, H, W, C] in [0, 255] on video's device;
    as_uint8: the uint8 frames save_video would make of them (truncation, inference_utils.py:92) -- what the CLI moves off the GPU."""
    ctx = ctx or context_for(video.device if video.is_cuda else None)
    out = ctx.color_fix(_on(ctx, video), _on(ctx, source), as_uint8=as_uint8)
    return out if video.is_cuda else out.cpu()


def adain_color_fix(target, source, ctx=None):
    """color_fix.py:15-29 with the reference's signature: target [F, H, W, C] in [0, 255] (a tensor2vid result),
    source [F, C, h, w] in [-1, 1]  ->  fp32 [F, H, W, C] in [0, 255] on target's device."""
    ctx = ctx or context_for(target.device if target.is_cuda else None)
    out = ctx.adain_color_fix(_on(ctx, target), _on(ctx, source))
    return out if target.is_cuda else out.cpu()


def calc_mean_std(feat, eps=1e-5, ctx=None):
    """color_fix.py:62-74: [b, c, H, W] -> (mean, std) each [b, c, 1, 1]."""
    ctx = ctx or context_for(feat.device if feat.is_cuda else None)
    st = ctx.plane_stats(_on(ctx, feat), eps=eps)
    st = st if feat.is_cuda else st.cpu()
    b, c = feat.shape[:2]
    return st[..., 0].reshape(b, c, 1, 1), st[..., 1].reshape(b, c, 1, 1)
