"""Import the reference's own hot-path modules on CPU (TEST INFRASTRUCTURE ONLY).

Only usable where /root/reference exists (the build container) -- never on the
GPU box and never from the product package.  Used by oracle/make_golden.py to
generate the fixtures in tests/golden/ and by tests that pin the restatement in
oracle/unet_oracle.py against the real reference code.

Three third-party imports of video_to_video/modules/unet_v2v.py are absent in
this image and are replaced by numerically equivalent stand-ins:
  * xformers.ops.memory_efficient_attention(q,k,v) on [B*h, N, d] tensors
        -> torch.nn.functional.scaled_dot_product_attention (softmax(QK^T/sqrt d)V)
           (unet_v2v.py:179-185)
  * fairscale.nn.checkpoint.checkpoint_wrapper -> identity (no_grad inference; unet_v2v.py:13)
  * timm.models.vision_transformer.Mlp -> 2-layer MLP (only used by the
        never-instantiated CaptionEmbedder; unet_v2v.py:14,27)
and torchsde (solvers_sdedit.py:4) by an empty module: the Brownian tree is
replaced by an injected noise source in the oracle.
"""
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = os.environ.get("STAR_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REF_ROOT, "video_to_video/modules/unet_v2v.py"))


def _install_stubs():
    if "xformers" not in sys.modules:
        xf = types.ModuleType("xformers")
        xops = types.ModuleType("xformers.ops")

        def memory_efficient_attention(q, k, v, attn_bias=None, op=None):
            assert attn_bias is None
            if q.dim() == 3:   # [B*h, N, d]: the 4-D form takes torch's memory-efficient CPU kernel (the 3-D form materialises
                return F.scaled_dot_product_attention(q.unsqueeze(0), k.unsqueeze(0), v.unsqueeze(0)).squeeze(0)   # B*h x N x N logits)
            return F.scaled_dot_product_attention(q, k, v)

        xops.memory_efficient_attention = memory_efficient_attention
        xf.ops = xops
        sys.modules["xformers"] = xf
        sys.modules["xformers.ops"] = xops
    if "fairscale" not in sys.modules:
        fs = types.ModuleType("fairscale")
        fsnn = types.ModuleType("fairscale.nn")
        fsck = types.ModuleType("fairscale.nn.checkpoint")
        fsck.checkpoint_wrapper = lambda m, *a, **k: m
        fs.nn = fsnn
        fsnn.checkpoint = fsck
        sys.modules["fairscale"] = fs
        sys.modules["fairscale.nn"] = fsnn
        sys.modules["fairscale.nn.checkpoint"] = fsck
    if "timm" not in sys.modules:
        tm = types.ModuleType("timm")
        tmm = types.ModuleType("timm.models")
        tmv = types.ModuleType("timm.models.vision_transformer")

        class Mlp(nn.Module):
            def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
                super().__init__()
                self.fc1 = nn.Linear(in_features, hidden_features)
                self.act = act_layer() if isinstance(act_layer, type) else act_layer
                self.fc2 = nn.Linear(hidden_features, out_features)

            def forward(self, x):
                return self.fc2(self.act(self.fc1(x)))

        tmv.Mlp = Mlp
        tm.models = tmm
        tmm.vision_transformer = tmv
        sys.modules["timm"] = tm
        sys.modules["timm.models"] = tmm
        sys.modules["timm.models.vision_transformer"] = tmv
    if "torchsde" not in sys.modules:
        sys.modules["torchsde"] = types.ModuleType("torchsde")
    if "easydict" not in sys.modules:
        ed = types.ModuleType("easydict")

        class EasyDict(dict):
            __getattr__ = dict.get
            __setattr__ = dict.__setitem__

        ed.EasyDict = EasyDict
        sys.modules["easydict"] = ed


def _load_by_path(name, relpath):
    path = os.path.join(REF_ROOT, relpath)
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


_cache = {}


def load_unet_module():
    """-> the reference module object of video_to_video/modules/unet_v2v.py"""
    if "unet" not in _cache:
        _install_stubs()
        _cache["unet"] = _load_by_path("_star_ref_unet_v2v", "video_to_video/modules/unet_v2v.py")
    return _cache["unet"]


def load_diffusion_modules():
    """-> (diffusion_sdedit, solvers_sdedit, schedules_sdedit) reference modules."""
    if "diff" not in _cache:
        _install_stubs()
        # the reference files import `video_to_video.utils.logger`; provide a minimal package tree
        for pkg in ("video_to_video", "video_to_video.utils", "video_to_video.diffusion"):
            if pkg not in sys.modules:
                m = types.ModuleType(pkg)
                m.__path__ = []
                sys.modules[pkg] = m
        if "video_to_video.utils.logger" not in sys.modules:
            lg = types.ModuleType("video_to_video.utils.logger")
            import logging

            lg.get_logger = lambda *a, **k: logging.getLogger("star_ref")
            sys.modules["video_to_video.utils.logger"] = lg
        sch = _load_by_path("video_to_video.diffusion.schedules_sdedit", "video_to_video/diffusion/schedules_sdedit.py")
        sol = _load_by_path("video_to_video.diffusion.solvers_sdedit", "video_to_video/diffusion/solvers_sdedit.py")
        dif = _load_by_path("video_to_video.diffusion.diffusion_sdedit", "video_to_video/diffusion/diffusion_sdedit.py")
        _cache["diff"] = (dif, sol, sch)
    return _cache["diff"]


def load_frame_modules():
    """-> (inference_utils, color_fix) reference modules.  Both import torchvision / cv2 at module level but the
    functions on this path (tensor2vid; adain_color_fix, calc_mean_std, adaptive_instance_normalization) use torch
    and einops only, so empty stand-ins are enough."""
    if "frames" not in _cache:
        load_diffusion_modules()   # installs the video_to_video.utils.logger stand-in
        for name in ("cv2", "torchvision", "torchvision.transforms", "torchvision.transforms.functional"):
            if name not in sys.modules:
                m = types.ModuleType(name)
                m.__path__ = []
                sys.modules[name] = m
        tvt = sys.modules["torchvision.transforms"]
        for attr in ("ToTensor", "ToPILImage"):
            if not hasattr(tvt, attr):
                setattr(tvt, attr, object)
        sys.modules["torchvision"].transforms = tvt
        tvt.functional = sys.modules["torchvision.transforms.functional"]
        iu = _load_by_path("_star_ref_inference_utils", "inference_utils.py")
        cf = _load_by_path("_star_ref_color_fix", "video_super_resolution/color_fix.py")
        _cache["frames"] = (iu, cf)
    return _cache["frames"]


def randomize_zero_init(model, seed=0, std=0.02):
    """Re-draw every all-zero parameter N(0, std^2) so parity is non-vacuous
    (the reference zero-initialises proj_out / zero_convs / out_layers[-1] /
    temopral_conv.conv4 / input_hint_block / out[-1].weight: SURVEY.md section 7)."""
    g = torch.Generator().manual_seed(seed)
    n = 0
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.numel() > 0 and float(p.abs().max()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * std)
                n += 1
    return n
