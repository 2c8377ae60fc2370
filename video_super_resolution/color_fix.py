"""Colour transfer from the LR input to the SR frames (the reference's video_super_resolution/color_fix.py:15-74),
restated on torch tensors: AdaIN (match per-channel mean/std) and a wavelet variant (swap the low-frequency band).
Bandwidth-bound post-processing ("next" row 1 of SURVEY.md section 8f) -- plain torch ops on whatever device holds the frames."""
import torch
import torch.nn.functional as F


def _mean_std(feat, eps=1e-5):
    b, c = feat.shape[:2]
    var = feat.reshape(b, c, -1).var(dim=2) + eps
    return feat.reshape(b, c, -1).mean(dim=2).reshape(b, c, 1, 1), var.sqrt().reshape(b, c, 1, 1)


def adaptive_instance_normalization(content, style):
    sm, ss = _mean_std(style)
    cm, cs = _mean_std(content)
    return (content - cm) / cs * ss + sm


def _wavelet_blur(image, radius):
    k = torch.tensor([[0.0625, 0.125, 0.0625], [0.125, 0.25, 0.125], [0.0625, 0.125, 0.0625]], dtype=image.dtype, device=image.device)
    k = k[None, None].repeat(3, 1, 1, 1)
    image = F.pad(image, (radius, radius, radius, radius), mode="replicate")
    return F.conv2d(image, k, groups=3, dilation=radius)


def wavelet_decomposition(image, levels=5):
    high = torch.zeros_like(image)
    low = image
    for i in range(levels):
        low_next = _wavelet_blur(low, 2 ** i)
        high = high + (low - low_next)
        low = low_next
    return high, low


def wavelet_reconstruction(content, style):
    ch, _ = wavelet_decomposition(content)
    _, sl = wavelet_decomposition(style)
    return ch + sl


def _to_frames(x):  # [F, H, W, 3] uint8-like in [0, 255]  ->  [F, 3, H, W] in [-1, 1]
    return x.permute(0, 3, 1, 2).float() / 127.5 - 1.0


def adain_color_fix(target, source):
    """target: SR frames [F, H, W, 3] (0..255); source: LR frames [F, 3, h, w] in [-1, 1] -> uint8 [F, H, W, 3]."""
    t = _to_frames(target)
    s = F.interpolate(source.float().to(t.device), size=t.shape[-2:], mode="bilinear")
    out = adaptive_instance_normalization(t, s)
    return ((out.clamp(-1, 1) + 1) * 127.5).permute(0, 2, 3, 1).round().to(torch.uint8)


def wavelet_color_fix(target, source):
    t = _to_frames(target)
    s = F.interpolate(source.float().to(t.device), size=t.shape[-2:], mode="bilinear")
    out = wavelet_reconstruction(t, s)
    return ((out.clamp(-1, 1) + 1) * 127.5).permute(0, 2, 3, 1).round().to(torch.uint8)
