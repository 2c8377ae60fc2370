"""Architecture description of ControlledV2VUNet (+ VideoControlNet) as data.

Everything here is derived from the constructor logic of the reference
(video_to_video/modules/unet_v2v.py:1283-1555 Vid2VidSDUNet, :1896-2128 VideoControlNet,
:1712-1715 ControlledV2VUNet); it names every parameter with the reference's state-dict key
(B4 in SURVEY.md section 8b, including the `temopral_conv` typo) so that a reference `.pt`
loads unchanged.  tests/test_topology.py pins the default config against the 2247 key/shape
pairs dumped from the reference (tests/golden/unet_state_dict_shapes.json).
"""
import zlib
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import List, Tuple

import torch


@dataclass(frozen=True)
class UNetConfig:
    in_dim: int = 4
    dim: int = 320
    context_dim: int = 1024
    out_dim: int = 4
    dim_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_heads: int = 8            # heads of the stem TemporalTransformer (inner dim = num_heads * head_dim)
    head_dim: int = 64
    num_res_blocks: int = 2
    attn_scales: Tuple[float, ...] = (1.0, 0.5, 0.25)

    @property
    def embed_dim(self):
        return self.dim * 4


# a reduced-width config used by fast parity tests (same topology rules, 1/5 of the width)
SMALL_TEST_CONFIG = UNetConfig(dim=64, num_heads=2)


@dataclass
class BlockSpec:
    """One entry of input_blocks / middle_block / output_blocks: a list of (kind, name, dims...) modules."""
    name: str
    modules: list = field(default_factory=list)


def _temporal_transformer(prefix, c, heads, head_dim):
    return ("tt", prefix, c, heads * head_dim, heads)


def build_blocks(cfg: UNetConfig, control: bool):
    """-> dict with keys input_blocks, middle_block, output_blocks (empty for the control net): lists of module
    tuples: ("conv_in", name, cin, cout) | ("res", name, cin, cout) | ("st", name, c, heads) |
    ("tt", name, c, inner, heads) | ("down", name, c) | ("up", name, c)."""
    dim, hd = cfg.dim, cfg.head_dim
    enc_dims = [dim * u for u in (1,) + tuple(cfg.dim_mult)]
    dec_dims = [dim * u for u in (cfg.dim_mult[-1],) + tuple(cfg.dim_mult[::-1])]
    shortcut = []
    scale = 1.0
    inp = []
    blk0 = [("conv_in", "input_blocks.0.0", cfg.in_dim, dim), _temporal_transformer("input_blocks.0.1", dim, cfg.num_heads, hd)]
    inp.append(blk0)
    shortcut.append(dim)
    idx = 1
    for i, (cin, cout) in enumerate(zip(enc_dims[:-1], enc_dims[1:])):
        for j in range(cfg.num_res_blocks):
            mods = [("res", f"input_blocks.{idx}.0", cin, cout)]
            if scale in cfg.attn_scales:
                mods.append(("st", f"input_blocks.{idx}.1", cout, cout // hd))
                mods.append(_temporal_transformer(f"input_blocks.{idx}.2", cout, cout // hd, hd))
            cin = cout
            inp.append(mods)
            shortcut.append(cout)
            idx += 1
            if i != len(cfg.dim_mult) - 1 and j == cfg.num_res_blocks - 1:
                inp.append([("down", f"input_blocks.{idx}", cout)])
                shortcut.append(cout)
                scale /= 2.0
                idx += 1
    c = enc_dims[-1]
    mid = [("res", "middle_block.0", c, c), ("st", "middle_block.1", c, c // hd),
           _temporal_transformer("middle_block.2", c, c // hd, hd), ("res", "middle_block.3", c, c)]
    out = []
    if not control:
        idx = 0
        for i, (cin, cout) in enumerate(zip(dec_dims[:-1], dec_dims[1:])):
            for j in range(cfg.num_res_blocks + 1):
                mods = [("res", f"output_blocks.{idx}.0", cin + shortcut.pop(), cout)]
                k = 1
                if scale in cfg.attn_scales:
                    mods.append(("st", f"output_blocks.{idx}.1", cout, cout // hd))
                    mods.append(_temporal_transformer(f"output_blocks.{idx}.2", cout, cout // hd, hd))
                    k = 3
                cin = cout
                if i != len(cfg.dim_mult) - 1 and j == cfg.num_res_blocks:
                    mods.append(("up", f"output_blocks.{idx}.{k}", cout))
                    scale *= 2.0
                out.append(mods)
                idx += 1
    return {"input_blocks": inp, "middle_block": mid, "output_blocks": out,
            "skip_dims": [m[-1][3] if m[-1][0] in ("conv_in",) else None for m in []]}


def _res_params(p, name, cin, cout, embed):
    s = OrderedDict()
    s[f"{name}.in_layers.0.weight"] = (cin,)
    s[f"{name}.in_layers.0.bias"] = (cin,)
    s[f"{name}.in_layers.2.weight"] = (cout, cin, 3, 3)
    s[f"{name}.in_layers.2.bias"] = (cout,)
    s[f"{name}.emb_layers.1.weight"] = (cout, embed)
    s[f"{name}.emb_layers.1.bias"] = (cout,)
    s[f"{name}.out_layers.0.weight"] = (cout,)
    s[f"{name}.out_layers.0.bias"] = (cout,)
    s[f"{name}.out_layers.3.weight"] = (cout, cout, 3, 3)
    s[f"{name}.out_layers.3.bias"] = (cout,)
    if cin != cout:
        s[f"{name}.skip_connection.weight"] = (cout, cin, 1, 1)
        s[f"{name}.skip_connection.bias"] = (cout,)
    for k, conv_idx in ((1, 2), (2, 3), (3, 3), (4, 3)):
        s[f"{name}.temopral_conv.conv{k}.0.weight"] = (cout,)
        s[f"{name}.temopral_conv.conv{k}.0.bias"] = (cout,)
        s[f"{name}.temopral_conv.conv{k}.{conv_idx}.weight"] = (cout, cout, 3, 1, 1)
        s[f"{name}.temopral_conv.conv{k}.{conv_idx}.bias"] = (cout,)
    p.update(s)


def _tblock_params(p, name, inner, ctx_dim_attn2, spatial):
    tb = f"{name}.transformer_blocks.0"
    for a, cdim in (("attn1", inner), ("attn2", ctx_dim_attn2)):
        p[f"{tb}.{a}.to_q.weight"] = (inner, inner)
        p[f"{tb}.{a}.to_k.weight"] = (inner, cdim)
        p[f"{tb}.{a}.to_v.weight"] = (inner, cdim)
        p[f"{tb}.{a}.to_out.0.weight"] = (inner, inner)
        p[f"{tb}.{a}.to_out.0.bias"] = (inner,)
        if a == "attn1":
            p[f"{tb}.ff.net.0.proj.weight"] = (inner * 8, inner)
            p[f"{tb}.ff.net.0.proj.bias"] = (inner * 8,)
            p[f"{tb}.ff.net.2.weight"] = (inner, inner * 4)
            p[f"{tb}.ff.net.2.bias"] = (inner,)
    for n in ("norm1", "norm2", "norm3"):
        p[f"{tb}.{n}.weight"] = (inner,)
        p[f"{tb}.{n}.bias"] = (inner,)
    if spatial:
        p[f"{tb}.local1.conv1.weight"] = (1, 2, 7, 7)
    else:
        p[f"{tb}.local1.conv1.weight"] = (1, 2)
        p[f"{tb}.local2.conv1.weight"] = (1, 2)


def _module_params(p, m, cfg):
    kind, name = m[0], m[1]
    if kind == "conv_in":
        p[f"{name}.weight"] = (m[3], m[2], 3, 3)
        p[f"{name}.bias"] = (m[3],)
    elif kind == "res":
        _res_params(p, name, m[2], m[3], cfg.embed_dim)
    elif kind == "st":
        c = m[2]
        p[f"{name}.norm.weight"] = (c,)
        p[f"{name}.norm.bias"] = (c,)
        p[f"{name}.proj_in.weight"] = (c, c)
        p[f"{name}.proj_in.bias"] = (c,)
        _tblock_params(p, name, c, cfg.context_dim, spatial=True)
        p[f"{name}.proj_out.weight"] = (c, c)
        p[f"{name}.proj_out.bias"] = (c,)
    elif kind == "tt":
        c, inner = m[2], m[3]
        p[f"{name}.norm.weight"] = (c,)
        p[f"{name}.norm.bias"] = (c,)
        p[f"{name}.proj_in.weight"] = (inner, c, 1)
        p[f"{name}.proj_in.bias"] = (inner,)
        _tblock_params(p, name, inner, inner, spatial=False)
        p[f"{name}.proj_out.weight"] = (c, inner, 1)
        p[f"{name}.proj_out.bias"] = (c,)
    elif kind == "down":
        p[f"{name}.op.weight"] = (m[2], m[2], 3, 3)
        p[f"{name}.op.bias"] = (m[2],)
    elif kind == "up":
        p[f"{name}.conv.weight"] = (m[2], m[2], 3, 3)
        p[f"{name}.conv.bias"] = (m[2],)
    else:
        raise ValueError(kind)


def _net_params(cfg, control):
    p = OrderedDict()
    p["time_embed.0.weight"] = (cfg.embed_dim, cfg.dim)
    p["time_embed.0.bias"] = (cfg.embed_dim,)
    p["time_embed.2.weight"] = (cfg.embed_dim, cfg.embed_dim)
    p["time_embed.2.bias"] = (cfg.embed_dim,)
    blocks = build_blocks(cfg, control)
    zero_dims = []
    for mods in blocks["input_blocks"]:
        for m in mods:
            _module_params(p, m, cfg)
        last = mods[-1]
        zero_dims.append(last[3] if last[0] in ("conv_in", "res") else last[2])
    for m in blocks["middle_block"]:
        _module_params(p, m, cfg)
    if control:
        for i, c in enumerate(zero_dims):
            p[f"zero_convs.{i}.0.weight"] = (c, c, 1, 1)
            p[f"zero_convs.{i}.0.bias"] = (c,)
        cm = blocks["middle_block"][-1][3]
        p["middle_block_out.0.weight"] = (cm, cm, 1, 1)
        p["middle_block_out.0.bias"] = (cm,)
        p["input_hint_block.weight"] = (cfg.dim, 4, 3, 3)
        p["input_hint_block.bias"] = (cfg.dim,)
    else:
        for mods in blocks["output_blocks"]:
            for m in mods:
                _module_params(p, m, cfg)
        p["out.0.weight"] = (cfg.dim,)
        p["out.0.bias"] = (cfg.dim,)
        p["out.2.weight"] = (cfg.out_dim, cfg.dim, 3, 3)
        p["out.2.bias"] = (cfg.out_dim,)
    return p


def param_shapes(cfg: UNetConfig = UNetConfig()):
    """OrderedDict: reference state-dict key -> shape tuple for ControlledV2VUNet(cfg)."""
    p = _net_params(cfg, control=False)
    for k, v in _net_params(cfg, control=True).items():
        p["VideoControlNet." + k] = v
    return p


def random_state_dict(cfg: UNetConfig = UNetConfig(), seed=0, dtype=torch.float32):
    """Seeded random-init weights of the architecture (no checkpoint is available offline).  Every tensor is
    drawn from its own generator keyed by (seed, crc32(key)), so the values do not depend on construction order:
    weights ~ N(0, 1/fan_in) (x0.5 for the reference's zero-initialised layers so residual branches stay
    non-vacuous but small), biases ~ N(0, 0.02^2), norm scales 1 + N(0, 0.1^2)."""
    sd = OrderedDict()
    for i, (k, shape) in enumerate(param_shapes(cfg).items()):
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(k.encode())) % (2 ** 62))
        if len(shape) == 1:
            is_norm_scale = k.endswith("weight")
            t = torch.randn(shape, generator=g) * (0.1 if is_norm_scale else 0.02)
            if is_norm_scale:
                t += 1.0
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g) / (fan_in ** 0.5)
            if "local" in k:
                t = torch.randn(shape, generator=g) * 0.3
        sd[k] = t.to(dtype)
    return sd


def geglu_interleave(n_half):
    """Row permutation of a GEGLU projection [2*n_half, K] (value rows then gate rows, unet_v2v.py:500-504)
    into alternating 32-row (value, gate) blocks, the layout the GEMM's GEGLU epilogue pairs in registers."""
    assert n_half % 32 == 0
    idx = []
    for blk in range(n_half // 32):
        idx += list(range(blk * 32, blk * 32 + 32))
        idx += list(range(n_half + blk * 32, n_half + blk * 32 + 32))
    return torch.tensor(idx, dtype=torch.long)
