"""Frame pre/post-processing either side of the diffusion path: resize + pad before the
VAE encoder, tensor2vid + AdaIN colour fix after the decoder.  Floating point: the kernels must agree with the reference
within 2e-5 of the value range (|x| <= 1 planes; 5e-3 on the 0..255 output), the tolerance written in each test."""
import os
import sys

import pytest
import torch

from util import BACKENDS, make_ctx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import frames_oracle as fo  # noqa: E402

GOLD = torch.load(os.path.join(ROOT, "tests", "golden", "frames.pt"))


@pytest.fixture(params=BACKENDS)
def ctx(request):
    emu = request.getfixturevalue("emu_lib") if request.param == "emu" else None
    c = make_ctx(request.param, torch.float16, emu)
    yield c
    c.sync()
    c.close()


@pytest.mark.parametrize("name", sorted(GOLD))
def test_oracle_matches_reference_golden(name):
    """oracle/frames_oracle.py restates inference_utils.py:16-23 / color_fix.py:15-29,62-89 / video_to_video_model.py:81-87;
    the fixture holds what the reference's own functions returned on the same seeded clip: bit-exact."""
    g = GOLD[name]
    F_, h, w, up, seed, target, padding = g["case"]
    lr, video = fo.frames_inputs(F_, h, w, up, seed)
    assert torch.equal(fo.postprocess(video, lr), g["color_fix"])
    assert torch.equal(fo.resize_pad(lr, target, padding), g["resize_pad"])
    cm, cs = fo.calc_mean_std(fo.tensor2vid(video).permute(0, 3, 1, 2) / 255)
    assert torch.equal(torch.stack([cm.flatten(1), cs.flatten(1)], -1), g["content_stats"])


@pytest.mark.parametrize("name", sorted(GOLD))
def test_resize_pad_vs_reference(ctx, name):
    g = GOLD[name]
    F_, h, w, up, seed, target, padding = g["case"]
    lr, _ = fo.frames_inputs(F_, h, w, up, seed)
    out = ctx.resize_pad(lr.to(ctx.torch_device), target, padding).cpu()
    assert out.shape == g["resize_pad"].shape
    assert float((out - g["resize_pad"]).abs().max()) <= 2e-5
    pl, pr, pt, pb = padding
    if pt:
        assert float((out[:, :, :pt] - 1).abs().max()) == 0   # the pad value is exact


@pytest.mark.parametrize("name", sorted(GOLD))
def test_color_fix_vs_reference(ctx, name):
    g = GOLD[name]
    F_, h, w, up, seed, target, padding = g["case"]
    lr, video = fo.frames_inputs(F_, h, w, up, seed)
    dev = ctx.torch_device
    stats = ctx.plane_stats((lr.to(dev)), scale=0.5, shift=0.5)
    assert float((stats.cpu() - g["style_stats"]).abs().max()) <= 2e-6
    out = ctx.color_fix(video.to(dev), lr.to(dev)).cpu()
    assert out.shape == g["color_fix"].shape and out.dtype == torch.float32
    assert float((out - g["color_fix"]).abs().max()) <= 5e-3      # 0..255 scale: 2e-5 of the range
    assert float(out.min()) >= 0.0 and float(out.max()) <= 255.0
    # the uint8 form the CLI moves off the GPU = save_video's `.astype('uint8')` (truncation) of the very same fp32 values
    u8 = ctx.color_fix(video.to(dev), lr.to(dev), as_uint8=True).cpu()
    assert u8.dtype == torch.uint8 and torch.equal(u8, out.to(torch.uint8))
    # the reference's two-call form: tensor2vid on the host side, adain_color_fix as its own entry point
    alone = ctx.adain_color_fix(fo.tensor2vid(video).contiguous().to(dev), lr.to(dev)).cpu()
    assert float((alone - g["color_fix"]).abs().max()) <= 5e-3


def test_frames_edge_cases(ctx):
    """identity resize, single-pixel planes, constant planes (variance 0 -> std = sqrt(eps)), wrong shapes."""
    dev = ctx.torch_device
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 3, 7, 5, generator=g)
    same = ctx.resize_pad(x.to(dev), (7, 5)).cpu()
    assert float((same - x).abs().max()) <= 1e-6
    one = torch.full((1, 3, 1, 1), 0.25)
    up = ctx.resize_pad(one.to(dev), (4, 6), (1, 0, 0, 2), pad_value=-1.0).cpu()
    assert torch.equal(up, fo.resize_pad(one, (4, 6), (1, 0, 0, 2), value=-1.0))
    const = torch.full((1, 3, 8, 8), 0.5)
    st = ctx.plane_stats(const.to(dev)).cpu()
    assert float((st[..., 0] - 0.5).abs().max()) <= 1e-7 and float((st[..., 1] - 1e-5 ** 0.5).abs().max()) <= 1e-7
    video = torch.zeros(1, 3, 2, 8, 8)
    with pytest.raises(AssertionError):
        ctx.color_fix(video.to(dev), torch.zeros(3, 3, 4, 4).to(dev))   # frame count mismatch


@pytest.mark.gpu
def test_color_fix_full_size_properties():
    """cfg2 size (32 f, 960x1704): after the fix every (frame, channel) plane of the output carries the style statistics
    (where no clamp bites), and the op is idempotent in its statistics -- size-independent properties, no CPU oracle."""
    ctx = make_ctx("hip", torch.float16, None)
    dev = ctx.torch_device
    g = torch.Generator(device="cpu").manual_seed(5)
    lr = (torch.rand(32, 3, 240, 426, generator=g) * 0.6 - 0.3).to(dev)
    video = (torch.rand(1, 3, 32, 960, 1704, generator=g) * 0.8 - 0.4).to(dev)
    out = ctx.color_fix(video, lr)
    assert out.shape == (32, 960, 1704, 3)
    planes = (out / 255).permute(0, 3, 1, 2).contiguous()
    got = ctx.plane_stats(planes)
    want = ctx.plane_stats(lr, scale=0.5, shift=0.5)
    assert float((got - want).abs().max()) <= 1e-4
    ctx.sync()
    ctx.close()
