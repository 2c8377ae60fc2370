"""Parameter names/shapes of diffusers' AutoencoderKLTemporalDecoder (the SVD temporal VAE the reference loads from
`stabilityai/stable-video-diffusion-img2vid`, subfolder `vae`; video_to_video_model.py:57-59).

diffusers==0.30.0 is not vendored in the reference and not installed here: names follow the published module tree of
models/autoencoders/autoencoder_kl_temporal_decoder.py (restated from memory -- PARITY UNPINNED).
"""
import zlib
from collections import OrderedDict
from dataclasses import dataclass
from typing import Tuple

import torch


@dataclass(frozen=True)
class VaeConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    scaling_factor: float = 0.18215

    @property
    def downsample(self):
        return 2 ** (len(self.block_out_channels) - 1)


SMALL_VAE_CONFIG = VaeConfig(block_out_channels=(64, 128, 128))


def _resnet2d(p, name, cin, cout):
    p[f"{name}.norm1.weight"] = (cin,); p[f"{name}.norm1.bias"] = (cin,)
    p[f"{name}.conv1.weight"] = (cout, cin, 3, 3); p[f"{name}.conv1.bias"] = (cout,)
    p[f"{name}.norm2.weight"] = (cout,); p[f"{name}.norm2.bias"] = (cout,)
    p[f"{name}.conv2.weight"] = (cout, cout, 3, 3); p[f"{name}.conv2.bias"] = (cout,)
    if cin != cout:
        p[f"{name}.conv_shortcut.weight"] = (cout, cin, 1, 1); p[f"{name}.conv_shortcut.bias"] = (cout,)


def _attention(p, name, c):
    p[f"{name}.group_norm.weight"] = (c,); p[f"{name}.group_norm.bias"] = (c,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        p[f"{name}.{n}.weight"] = (c, c); p[f"{name}.{n}.bias"] = (c,)


def _st_resblock(p, name, cin, cout):
    _resnet2d(p, f"{name}.spatial_res_block", cin, cout)
    t = f"{name}.temporal_res_block"
    p[f"{t}.norm1.weight"] = (cout,); p[f"{t}.norm1.bias"] = (cout,)
    p[f"{t}.conv1.weight"] = (cout, cout, 3, 1, 1); p[f"{t}.conv1.bias"] = (cout,)
    p[f"{t}.norm2.weight"] = (cout,); p[f"{t}.norm2.bias"] = (cout,)
    p[f"{t}.conv2.weight"] = (cout, cout, 3, 1, 1); p[f"{t}.conv2.bias"] = (cout,)
    p[f"{name}.time_mixer.mix_factor"] = (1,)


def vae_param_shapes(cfg: VaeConfig = VaeConfig()):
    p = OrderedDict()
    boc = cfg.block_out_channels
    p["encoder.conv_in.weight"] = (boc[0], cfg.in_channels, 3, 3); p["encoder.conv_in.bias"] = (boc[0],)
    cin = boc[0]
    for i, c in enumerate(boc):
        for j in range(cfg.layers_per_block):
            _resnet2d(p, f"encoder.down_blocks.{i}.resnets.{j}", cin, c)
            cin = c
        if i != len(boc) - 1:
            p[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (c, c, 3, 3)
            p[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (c,)
    c = boc[-1]
    _resnet2d(p, "encoder.mid_block.resnets.0", c, c)
    _attention(p, "encoder.mid_block.attentions.0", c)
    _resnet2d(p, "encoder.mid_block.resnets.1", c, c)
    p["encoder.conv_norm_out.weight"] = (c,); p["encoder.conv_norm_out.bias"] = (c,)
    L2 = 2 * cfg.latent_channels
    p["encoder.conv_out.weight"] = (L2, c, 3, 3); p["encoder.conv_out.bias"] = (L2,)
    p["quant_conv.weight"] = (L2, L2, 1, 1); p["quant_conv.bias"] = (L2,)
    # decoder
    p["decoder.conv_in.weight"] = (c, cfg.latent_channels, 3, 3); p["decoder.conv_in.bias"] = (c,)
    _st_resblock(p, "decoder.mid_block.resnets.0", c, c)
    _attention(p, "decoder.mid_block.attentions.0", c)
    _st_resblock(p, "decoder.mid_block.resnets.1", c, c)
    rev = list(reversed(boc))
    cout = rev[0]
    for i in range(len(boc)):
        cprev, cout = cout, rev[i]
        for j in range(cfg.layers_per_block + 1):
            _st_resblock(p, f"decoder.up_blocks.{i}.resnets.{j}", cprev if j == 0 else cout, cout)
        if i != len(boc) - 1:
            p[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            p[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (cout,)
    p["decoder.conv_norm_out.weight"] = (boc[0],); p["decoder.conv_norm_out.bias"] = (boc[0],)
    p["decoder.conv_out.weight"] = (cfg.out_channels, boc[0], 3, 3); p["decoder.conv_out.bias"] = (cfg.out_channels,)
    p["decoder.time_conv_out.weight"] = (cfg.out_channels, cfg.out_channels, 3, 1, 1)
    p["decoder.time_conv_out.bias"] = (cfg.out_channels,)
    return p


def random_vae_state_dict(cfg: VaeConfig = VaeConfig(), seed=0):
    """Seeded synthetic weights (no checkpoint offline), per-tensor generators keyed by name."""
    sd = OrderedDict()
    for k, shape in vae_param_shapes(cfg).items():
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(k.encode())) % (2 ** 62))
        if k.endswith("mix_factor"):
            t = torch.randn(shape, generator=g) * 0.5
        elif len(shape) == 1:
            t = torch.randn(shape, generator=g) * (0.1 if k.endswith("weight") else 0.02)
            if k.endswith("weight"):
                t += 1.0
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g) / (fan_in ** 0.5)
        sd[k] = t
    return sd
