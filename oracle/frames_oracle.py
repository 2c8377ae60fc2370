"""CPU restatement of the frame pre/post-processing either side of the diffusion path (TEST INFRASTRUCTURE ONLY).

Nothing in the product package imports this file: it is the checker for star_resize_pad / star_plane_stats /
star_color_fix (star_amd/csrc/frames.h), used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline only.

Pinned: tests/golden/frames.pt holds outputs of the REFERENCE's own tensor2vid (inference_utils.py:16-23) and
adain_color_fix (color_fix.py:15-29) executed in the build container by oracle/make_golden.py (torchvision / cv2,
which those files import but never use on this path, replaced by empty stand-ins), and of the two torch calls the
reference makes inline for the resize + pad (video_to_video_model.py:81,87); tests/test_frames.py checks this
restatement against them bit for bit.
"""
import torch
import torch.nn.functional as F


def resize_pad(video, target_hw, padding, value=1.0):
    """video_to_video_model.py:81-87: F.interpolate(video_data, [target_h, target_w], mode='bilinear') then
    F.pad(video_data, padding, 'constant', 1) with padding = pad_to_fit(h, w) = (left, right, top, bottom)."""
    v = F.interpolate(video.float(), [int(target_hw[0]), int(target_hw[1])], mode="bilinear")
    return F.pad(v, tuple(int(p) for p in padding), "constant", value)


def tensor2vid(video, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    """inference_utils.py:16-23 (without the in-place mutation of the argument): [1, C, F, H, W] -> [F, H, W, C] in 0..255."""
    m = torch.tensor(mean).reshape(1, -1, 1, 1, 1)
    s = torch.tensor(std).reshape(1, -1, 1, 1, 1)
    v = (video.float() * s + m).clamp(0, 1) * 255.0
    return v.permute(0, 2, 3, 4, 1)[0]


def calc_mean_std(feat, eps=1e-5):
    """color_fix.py:62-74: per (b, c) mean and sqrt(unbiased var + eps) over H*W."""
    b, c = feat.shape[:2]
    var = feat.reshape(b, c, -1).var(dim=2) + eps
    return feat.reshape(b, c, -1).mean(dim=2).reshape(b, c, 1, 1), var.sqrt().reshape(b, c, 1, 1)


def adain_color_fix(target, source):
    """color_fix.py:15-29 + :76-89: target [T, H, W, C] in 0..255, source [T, C, h, w] in [-1, 1] -> [T, H, W, C] in 0..255."""
    t = target.permute(0, 3, 1, 2) / 255
    s = (source + 1) / 2
    out = []
    for i in range(t.shape[0]):
        c, st = t[i:i + 1], s[i:i + 1]
        sm, ss = calc_mean_std(st)
        cm, cs = calc_mean_std(c)
        out.append((c - cm) / cs * ss + sm)
    r = torch.cat(out, 0).clamp(0.0, 1.0)
    return r.permute(0, 2, 3, 1) * 255


def postprocess(video, source):
    """inference_sr.py:47-48: tensor2vid then adain_color_fix against the low-resolution clip."""
    return adain_color_fix(tensor2vid(video), source)


def frames_inputs(F_, h, w, up, seed):
    """Seeded synthetic clip shared by golden generation and the parity tests: a smooth field plus noise, in [-1, 1]."""
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(F_, 3, max(h // 4, 2), max(w // 4, 2), generator=g)
    lr = F.interpolate(base, [h, w], mode="bicubic", align_corners=False) * 0.5 + torch.randn(F_, 3, h, w, generator=g) * 0.1
    lr = lr.clamp(-1, 1)
    H, W = h * up, w * up
    sr = F.interpolate(lr, [H, W], mode="bicubic", align_corners=False) * 1.2 + 0.1 + torch.randn(F_, 3, H, W, generator=g) * 0.05
    video = sr.permute(1, 0, 2, 3).unsqueeze(0).contiguous()   # [1, C, F, H, W], partly outside [-1, 1] to exercise the clamps
    return lr.contiguous(), video
