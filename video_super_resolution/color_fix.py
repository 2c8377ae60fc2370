"""Colour transfer from the LR input to the SR frames, with the reference's names and conventions
(video_super_resolution/color_fix.py:15-89).  The AdaIN path -- the one inference_sr.py:48 uses -- runs as HIP kernels
through the C ABI (star_plane_stats / star_adain_color_fix, star_amd/csrc/frames.h); it needs the gfx950 library and
a GPU like the rest of the package.  The reference's wavelet variant (:31-45, :91-130) is not called anywhere on its
inference path and is out of scope here."""
from star_amd import frames as _frames


def calc_mean_std(feat, eps=1e-5):
    """[b, c, H, W] -> (mean, sqrt(unbiased var + eps)), each [b, c, 1, 1] (color_fix.py:62-74)."""
    assert feat.dim() == 4, "The input feature should be 4D tensor."
    return _frames.calc_mean_std(feat, eps)


def adaptive_instance_normalization(content_feat, style_feat):
    """(content - mean_c) / std_c * std_s + mean_s per (b, c) plane (color_fix.py:76-89)."""
    style_mean, style_std = calc_mean_std(style_feat)
    content_mean, content_std = calc_mean_std(content_feat)
    return (content_feat - content_mean) / content_std * style_std + style_mean


def adain_color_fix(target, source):
    """target: tensor2vid frames [T, H, W, C] in 0..255; source: LR clip [T, C, h, w] in [-1, 1]
    -> float frames [T, H, W, C] in 0..255 (color_fix.py:15-29)."""
    return _frames.adain_color_fix(target, source)
