"""CPU oracle: a plain-PyTorch fp32 restatement of the reference's denoiser forward.

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product path (star_amd/) never imports it.

It restates ControlledV2VUNet.forward (video_to_video/modules/unet_v2v.py:1717-1809) and everything it
calls as pure functions over a reference-keyed state dict, so it also works for reduced-width configs.
Pinned against the reference's own code: oracle/make_golden.py runs the real unet_v2v.py (this container
only) and stores outputs in tests/golden/; tests/test_oracle.py checks this file against them and, where
/root/reference exists, against the live reference modules.
"""
import math

import torch
import torch.nn.functional as F


def sinusoidal_embedding(t, dim):
    """unet_v2v.py:96-108"""
    half = dim // 2
    t = t.float()
    freqs = torch.pow(10000, -torch.arange(half).to(t).div(half))
    s = torch.outer(t, freqs)
    return torch.cat([torch.cos(s), torch.sin(s)], dim=1)


def _lin(sd, name, x, bias=True):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias") if bias else None)


def attention(sd, p, x, context, heads):
    """MemoryEfficientCrossAttention.forward, unet_v2v.py:158-195 (xformers call = softmax(QK^T/sqrt(d))V)."""
    ctx = x if context is None else context
    q = F.linear(x, sd[p + ".to_q.weight"])
    k = F.linear(ctx, sd[p + ".to_k.weight"])
    v = F.linear(ctx, sd[p + ".to_v.weight"])
    b = q.shape[0]
    sp = lambda t: t.reshape(b, t.shape[1], heads, -1).transpose(1, 2)
    o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v))
    o = o.transpose(1, 2).reshape(b, q.shape[1], -1)
    return _lin(sd, p + ".to_out.0", o)


def feed_forward(sd, p, x):
    """FeedForward with GEGLU, unet_v2v.py:496-529 (exact erf GELU)."""
    h = _lin(sd, p + ".net.0.proj", x)
    a, gate = h.chunk(2, dim=-1)
    return _lin(sd, p + ".net.2", a * F.gelu(gate))


def liem_spatial(sd, p, x, h, w):
    """SpatialAttention, unet_v2v.py:380-394, applied to tokens [b, hw, c]."""
    xi = x.transpose(1, 2).reshape(x.shape[0], x.shape[2], h, w)
    m = torch.cat([xi.max(dim=1, keepdim=True)[0], xi.mean(dim=1, keepdim=True)], dim=1)
    g = torch.sigmoid(F.conv2d(m, sd[p + ".conv1.weight"], padding=3))
    return (g * xi).reshape(x.shape[0], x.shape[2], h * w).transpose(1, 2)


def liem_temporal(sd, p, x):
    """TemporalLocalAttention, unet_v2v.py:396-411."""
    m = torch.cat([x.max(dim=-1, keepdim=True)[0], x.mean(dim=-1, keepdim=True)], dim=-1)
    return torch.sigmoid(F.linear(m, sd[p + ".conv1.weight"])) * x


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def transformer_block_space(sd, p, x, context, heads, h, w):
    """BasicTransformerBlock.forward, space branch, unet_v2v.py:466-477 (residual adds the un-gated x)."""
    xl = liem_spatial(sd, p + ".local1", x, h, w)
    x = attention(sd, p + ".attn1", _ln(sd, p + ".norm1", xl), None, heads) + x
    x = attention(sd, p + ".attn2", _ln(sd, p + ".norm2", x), context, heads) + x
    x = feed_forward(sd, p + ".ff", _ln(sd, p + ".norm3", x)) + x
    return x


def transformer_block_temp(sd, p, x, heads):
    """BasicTransformerBlock.forward, temp branch, unet_v2v.py:479-490 (attn2 is self-attention too)."""
    xl = liem_temporal(sd, p + ".local1", x)
    x = attention(sd, p + ".attn1", _ln(sd, p + ".norm1", xl), None, heads) + x
    xl = liem_temporal(sd, p + ".local2", x)
    x = attention(sd, p + ".attn2", _ln(sd, p + ".norm2", xl), None, heads) + x
    x = feed_forward(sd, p + ".ff", _ln(sd, p + ".norm3", x)) + x
    return x


def spatial_transformer(sd, p, x, context, heads):
    """SpatialTransformer.forward (use_linear=True), unet_v2v.py:297-317.  x: [(b f), c, h, w]."""
    n, c, h, w = x.shape
    y = F.group_norm(x, 32, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)
    y = y.reshape(n, c, h * w).transpose(1, 2)
    y = _lin(sd, p + ".proj_in", y)
    y = transformer_block_space(sd, p + ".transformer_blocks.0", y, context, heads, h, w)
    y = _lin(sd, p + ".proj_out", y)
    return y.transpose(1, 2).reshape(n, c, h, w) + x


def temporal_transformer(sd, p, x, heads):
    """TemporalTransformer.forward (use_linear=False, only_self_att=True), unet_v2v.py:1034-1092.  x: [b, c, f, h, w]."""
    b, c, f, h, w = x.shape
    y = F.group_norm(x, 32, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)
    y = y.permute(0, 3, 4, 1, 2).reshape(b * h * w, c, f)
    y = F.conv1d(y, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
    y = y.transpose(1, 2)                                   # (b h w) f inner
    y = transformer_block_temp(sd, p + ".transformer_blocks.0", y, heads)
    y = y.transpose(1, 2)                                   # (b h w) inner f
    y = F.conv1d(y, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    y = y.reshape(b, h, w, c, f).permute(0, 3, 4, 1, 2)
    return y + x


def temporal_conv_block(sd, p, x):
    """TemporalConvBlock_v2.forward, variant_info=None branch, unet_v2v.py:1266-1277.  x: [b, c, f, h, w]."""
    y = x
    for k, ci in ((1, 2), (2, 3), (3, 3), (4, 3)):
        q = f"{p}.conv{k}"
        y = F.silu(F.group_norm(y, 32, sd[q + ".0.weight"], sd[q + ".0.bias"], 1e-5))
        y = F.conv3d(y, sd[f"{q}.{ci}.weight"], sd[f"{q}.{ci}.bias"], padding=(1, 0, 0))
    return x + y


def res_block(sd, p, x, emb, batch):
    """ResBlock._forward (no up/down, no scale-shift), unet_v2v.py:666-692.  x: [(b f), c, h, w], emb: [(b f), E]."""
    h = F.silu(F.group_norm(x, 32, sd[p + ".in_layers.0.weight"], sd[p + ".in_layers.0.bias"], 1e-5))
    h = F.conv2d(h, sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    e = _lin(sd, p + ".emb_layers.1", F.silu(emb))
    h = h + e[:, :, None, None]
    h = F.silu(F.group_norm(h, 32, sd[p + ".out_layers.0.weight"], sd[p + ".out_layers.0.bias"], 1e-5))
    h = F.conv2d(h, sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if p + ".skip_connection.weight" in sd:
        x = F.conv2d(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    h = x + h
    n, c, hh, ww = h.shape
    h5 = h.reshape(batch, n // batch, c, hh, ww).permute(0, 2, 1, 3, 4)
    h5 = temporal_conv_block(sd, p + ".temopral_conv", h5)
    return h5.permute(0, 2, 1, 3, 4).reshape(n, c, hh, ww)


def downsample(sd, p, x):
    """Downsample.forward, conv stride 2 padding (2, 1), unet_v2v.py:709-729."""
    return F.conv2d(x, sd[p + ".op.weight"], sd[p + ".op.bias"], stride=2, padding=(2, 1))


def upsample(sd, p, x):
    """Upsample.forward (dims=2.0 -> the 2-D branch), nearest x2, drop first/last row, conv; unet_v2v.py:556-567."""
    x = F.interpolate(x, scale_factor=2, mode="nearest")[..., 1:-1, :]
    return F.conv2d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=1)


def _run_module(sd, m, x, e, context, batch, prefix):
    kind, name = m[0], prefix + m[1]
    if kind == "conv_in":
        return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], padding=1)
    if kind == "res":
        return res_block(sd, name, x, e, batch)
    if kind == "st":
        return spatial_transformer(sd, name, x, context, m[3])
    if kind == "tt":
        n, c, h, w = x.shape
        x5 = x.reshape(batch, n // batch, c, h, w).permute(0, 2, 1, 3, 4)
        x5 = temporal_transformer(sd, name, x5, m[4])
        return x5.permute(0, 2, 1, 3, 4).reshape(n, c, h, w)
    if kind == "down":
        return downsample(sd, name, x)
    if kind == "up":
        return upsample(sd, name, x)
    raise ValueError(kind)


def _time_embed(sd, prefix, t, dim):
    e = sinusoidal_embedding(t, dim)
    e = _lin(sd, prefix + "time_embed.0", e)
    return _lin(sd, prefix + "time_embed.2", F.silu(e))


def control_net_forward(sd, cfg, blocks, x, t, y, hint):
    """VideoControlNet.forward, unet_v2v.py:2134-2206 -> list of 13 residual tensors."""
    P = "VideoControlNet."
    b, _, f, h, w = x.shape
    hint2 = hint.permute(0, 2, 1, 3, 4).reshape(b * f, hint.shape[1], h, w)
    hint2 = F.conv2d(hint2, sd[P + "input_hint_block.weight"], sd[P + "input_hint_block.bias"], padding=1)
    e = _time_embed(sd, P, t, cfg.dim).repeat_interleave(f, dim=0)
    context = y.repeat_interleave(f, dim=0)
    x = x.permute(0, 2, 1, 3, 4).reshape(b * f, x.shape[1], h, w)
    outs = []
    for i, mods in enumerate(blocks["input_blocks"]):
        for m in mods:
            x = _run_module(sd, m, x, e, context, b, P)
            if hint2 is not None and m[0] != "tt":   # added once, after the first non-temporal module (:2190-2194)
                x = x + hint2
                hint2 = None
        outs.append(F.conv2d(x, sd[f"{P}zero_convs.{i}.0.weight"], sd[f"{P}zero_convs.{i}.0.bias"]))
    for m in blocks["middle_block"]:
        x = _run_module(sd, m, x, e, context, b, P)
    outs.append(F.conv2d(x, sd[P + "middle_block_out.0.weight"], sd[P + "middle_block_out.0.bias"]))
    return outs


def unet_forward(sd, cfg, x, t, y, hint):
    """ControlledV2VUNet.forward, unet_v2v.py:1717-1809.
    x, hint: [b, 4, f, h, w]; t: LongTensor [b]; y: [b, 77, context_dim] -> [b, 4, f, h, w]."""
    from star_amd.topology import build_blocks
    main = build_blocks(cfg, control=False)
    ctrl = build_blocks(cfg, control=True)
    control = control_net_forward(sd, cfg, ctrl, x, t, y, hint)
    b, _, f, h, w = x.shape
    e = _time_embed(sd, "", t, cfg.dim).repeat_interleave(f, dim=0)
    context = y.repeat_interleave(f, dim=0)
    x = x.permute(0, 2, 1, 3, 4).reshape(b * f, x.shape[1], h, w)
    xs = []
    for mods in main["input_blocks"]:
        for m in mods:
            x = _run_module(sd, m, x, e, context, b, "")
        xs.append(x)
    for m in main["middle_block"]:
        x = _run_module(sd, m, x, e, context, b, "")
    x = control.pop() + x
    for mods in main["output_blocks"]:
        x = torch.cat([x, xs.pop() + control.pop()], dim=1)
        for m in mods:
            x = _run_module(sd, m, x, e, context, b, "")
    x = F.silu(F.group_norm(x, 32, sd["out.0.weight"], sd["out.0.bias"], 1e-5))
    x = F.conv2d(x, sd["out.2.weight"], sd["out.2.bias"], padding=1)
    n, c, hh, ww = x.shape
    return x.reshape(b, f, c, hh, ww).permute(0, 2, 1, 3, 4)
