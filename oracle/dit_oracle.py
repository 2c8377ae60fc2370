"""TEST INFRASTRUCTURE (never imported by the product path): fp32 PyTorch restatement of one CogVideoX-5B DiT block as STAR's
CogVideoX variant runs it (BASELINE config #5).

Pinned part: the block logic the reference itself holds -- AdaLNMixin.layer_forward (cogvideox-based/sat/dit_video_concat.py:
482-563: modulation chunk order, LayerNorm + modulate, spatial then temporal LIEM on the video tokens, gated residuals), the
3-D rotary embedding (:254-346), the QK LayerNorm (:571-598), modulate (:349-350) and the LIEM gates
(cogvideox-based/transformer.py:316-348) -- is checked against tests/golden/dit_block.pt, which oracle/make_golden_dit.py
produced by EXECUTING those reference functions.

PARITY UNPINNED part: what the mixin calls into -- fused QKV dense, attention core, output dense, LayerNorm, MLP with the
tanh-form GELU -- lives in SwissArmyTransformer==0.4.12 (cogvideox-based/sat/requirements.txt:1), which is neither vendored
nor installed here; it is restated from the published package (sat/transformer_defaults.py: attention_forward_default,
standard_attention, mlp_forward_default; sat/mpu/utils.py: gelu_impl), in the golden generator as well as here.
"""
import math

import torch
import torch.nn.functional as F


class DitConfig:
    def __init__(self, hidden=3072, heads=48, time_embed_dim=512, n_layers=42, ln_eps=1e-5):
        self.hidden, self.heads, self.time_embed_dim, self.n_layers, self.ln_eps = hidden, heads, time_embed_dim, n_layers, ln_eps


SMALL_DIT_CONFIG = DitConfig(hidden=128, heads=2, time_embed_dim=64, n_layers=2)


def random_dit_state_dict(cfg, seed=0):
    """SAT-checkpoint key names (below `model.diffusion_model.`), N(0, s^2) values; nothing is left at the reference's zero
    initialisation (adaLN_modulations are zero-initialised there, dit_video_concat.py:565-568), so every path is exercised."""
    g = torch.Generator().manual_seed(seed)
    D, E = cfg.hidden, cfg.time_embed_dim
    rn = lambda *s, std=1.0: torch.randn(*s, generator=g) * std
    sd = {}
    for i in range(cfg.n_layers):
        L, A = f"transformer.layers.{i}.", "mixins.adaln_layer."
        for n in ("input_layernorm", "post_attention_layernorm"):
            sd[L + n + ".weight"] = 1.0 + rn(D, std=0.1)
            sd[L + n + ".bias"] = rn(D, std=0.1)
        sd[L + "attention.query_key_value.weight"] = rn(3 * D, D, std=D ** -0.5)
        sd[L + "attention.query_key_value.bias"] = rn(3 * D, std=0.1)
        sd[L + "attention.dense.weight"] = rn(D, D, std=D ** -0.5)
        sd[L + "attention.dense.bias"] = rn(D, std=0.1)
        sd[L + "mlp.dense_h_to_4h.weight"] = rn(4 * D, D, std=D ** -0.5)
        sd[L + "mlp.dense_h_to_4h.bias"] = rn(4 * D, std=0.1)
        sd[L + "mlp.dense_4h_to_h.weight"] = rn(D, 4 * D, std=(4 * D) ** -0.5)
        sd[L + "mlp.dense_4h_to_h.bias"] = rn(D, std=0.1)
        sd[L + "spa_local.conv1.weight"] = rn(1, 2, 7, 7, std=0.15)
        sd[L + "temp_local.conv1.weight"] = rn(1, 2, std=0.5)
        sd[A + f"adaLN_modulations.{i}.1.weight"] = rn(12 * D, E, std=0.3 * E ** -0.5)
        sd[A + f"adaLN_modulations.{i}.1.bias"] = rn(12 * D, std=0.2)
        for n in ("query_layernorm_list", "key_layernorm_list"):
            sd[A + f"{n}.{i}.weight"] = 1.0 + rn(64, std=0.1)
            sd[A + f"{n}.{i}.bias"] = rn(64, std=0.1)
    return sd


def dit_inputs(cfg, text_len, T, H, W, seed=1):
    g = torch.Generator().manual_seed(seed)
    S = text_len + T * H * W
    x = torch.randn(1, S, cfg.hidden, generator=g)
    emb = torch.randn(1, cfg.time_embed_dim, generator=g)
    return x, emb


def rotary_tables(T, H, W, theta=10000.0):
    """Rotary3DPositionEmbeddingMixin.__init__ (dit_video_concat.py:267-295) for head dim 64: 16 | 24 | 24 channels."""
    def fr(dim, n):
        f = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
        return torch.outer(torch.arange(n, dtype=torch.float32), f).repeat_interleave(2, dim=-1)
    ft, fh, fw = fr(16, T), fr(24, H), fr(24, W)
    freqs = torch.cat([ft[:, None, None, :].expand(T, H, W, 16), fh[None, :, None, :].expand(T, H, W, 24),
                       fw[None, None, :, :].expand(T, H, W, 24)], dim=-1).reshape(T * H * W, 64)
    return freqs.cos(), freqs.sin()


def rotate_half(x):   # :247-251
    x1, x2 = x[..., 0::2], x[..., 1::2]
    return torch.stack((-x2, x1), dim=-1).flatten(-2)


def gelu_tanh(x):     # sat.mpu.utils.gelu_impl
    return 0.5 * x * (1.0 + torch.tanh(0.7978845608028654 * x * (1.0 + 0.044715 * x * x)))


def dit_block_forward(sd, cfg, layer, x, emb, text_len, T, H, W):
    """x: [1, S, D] fp32 (text tokens first, video tokens in (t h w) order); emb: [1, E].  Returns [1, S, D]."""
    D, heads = cfg.hidden, cfg.heads
    L, A = f"transformer.layers.{layer}.", "mixins.adaln_layer."
    x = x.float()
    mod = F.linear(F.silu(emb.float()), sd[A + f"adaLN_modulations.{layer}.1.weight"], sd[A + f"adaLN_modulations.{layer}.1.bias"])
    (shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp,
     t_shift_msa, t_scale_msa, t_gate_msa, t_shift_mlp, t_scale_mlp, t_gate_mlp) = mod.chunk(12, dim=1)          # :497-510
    ln = lambda v, n: F.layer_norm(v, (D,), sd[L + n + ".weight"], sd[L + n + ".bias"], cfg.ln_eps)
    modulate = lambda v, sh, sc: v * (1 + sc.unsqueeze(1)) + sh.unsqueeze(1)                                    # :349-350
    txt, img = x[:, :text_len], x[:, text_len:]
    img_in = modulate(ln(img, "input_layernorm"), shift_msa, scale_msa)
    txt_in = modulate(ln(txt, "input_layernorm"), t_shift_msa, t_scale_msa)
    # spatial LIEM (transformer.py:316-331) per frame, then temporal LIEM (:333-348) per pixel; both gate by per-token statistics
    f = img_in.reshape(T, H, W, D).permute(0, 3, 1, 2)                                                          # (b t) c h w
    w = torch.cat([f.max(dim=1, keepdim=True)[0], f.mean(dim=1, keepdim=True)], dim=1)
    f = torch.sigmoid(F.conv2d(w, sd[L + "spa_local.conv1.weight"], padding=3)) * f
    tf = f.permute(2, 3, 0, 1).reshape(H * W, T, D)                                                             # (b h w) t c
    w2 = torch.cat([tf.max(dim=-1, keepdim=True)[0], tf.mean(dim=-1, keepdim=True)], dim=-1)
    tf = torch.sigmoid(F.linear(w2, sd[L + "temp_local.conv1.weight"])) * tf
    img_in = tf.reshape(H, W, T, D).permute(2, 0, 1, 3).reshape(1, T * H * W, D)
    a_in = torch.cat([txt_in, img_in], dim=1)
    # sat attention_forward_default: fused dense, split into q | k | v, heads of 64
    qkv = F.linear(a_in, sd[L + "attention.query_key_value.weight"], sd[L + "attention.query_key_value.bias"])
    q, k, v = [t.reshape(1, -1, heads, 64).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=-1)]
    q = F.layer_norm(q, (64,), sd[A + f"query_layernorm_list.{layer}.weight"], sd[A + f"query_layernorm_list.{layer}.bias"], 1e-6)   # :585-589
    k = F.layer_norm(k, (64,), sd[A + f"key_layernorm_list.{layer}.weight"], sd[A + f"key_layernorm_list.{layer}.bias"], 1e-6)
    cos, sin = rotary_tables(T, H, W)
    rot = lambda t: t * cos + rotate_half(t) * sin                                                              # :306-311
    q = torch.cat([q[:, :, :text_len], rot(q[:, :, text_len:])], dim=2)                                         # :332-333
    k = torch.cat([k[:, :, :text_len], rot(k[:, :, text_len:])], dim=2)
    o = F.scaled_dot_product_attention(q, k, v)                                                                 # standard_attention, scale 1/8
    o = o.permute(0, 2, 1, 3).reshape(1, -1, D)
    o = F.linear(o, sd[L + "attention.dense.weight"], sd[L + "attention.dense.bias"])
    img = img + gate_msa.unsqueeze(1) * o[:, text_len:]                                                         # :541
    txt = txt + t_gate_msa.unsqueeze(1) * o[:, :text_len]                                                       # :542
    m_in = torch.cat([modulate(ln(txt, "post_attention_layernorm"), t_shift_mlp, t_scale_mlp),
                      modulate(ln(img, "post_attention_layernorm"), shift_mlp, scale_mlp)], dim=1)              # :545-549
    u = gelu_tanh(F.linear(m_in, sd[L + "mlp.dense_h_to_4h.weight"], sd[L + "mlp.dense_h_to_4h.bias"]))
    e = F.linear(u, sd[L + "mlp.dense_4h_to_h.weight"], sd[L + "mlp.dense_4h_to_h.bias"])
    img = img + gate_mlp.unsqueeze(1) * e[:, text_len:]                                                         # :561
    txt = txt + t_gate_mlp.unsqueeze(1) * e[:, :text_len]                                                       # :562
    return torch.cat([txt, img], dim=1)


# ---------------------------------------------------------------------------------------------------------------------------
# The whole DiffusionTransformer.forward (dit_video_concat.py:791-817) around the blocks: timestep embedding + time_embed MLP
# (:688-693, sgm timestep_embedding), patch embedding + text projection (:51-76), layers, sat's final_layernorm (BaseTransformer:
# applied to the layer stack's output before the final_forward hook -- sat internals, PARITY UNPINNED), FinalLayerMixin (:395-410).
def random_dit_model_state_dict(cfg, in_channels=8, out_channels=8, patch=2, text_hidden=64, seed=0):
    sd = random_dit_state_dict(cfg, seed)
    g = torch.Generator().manual_seed(seed + 1000)
    rn = lambda *s, std=1.0: torch.randn(*s, generator=g) * std
    D, E = cfg.hidden, cfg.time_embed_dim
    sd["time_embed.0.weight"], sd["time_embed.0.bias"] = rn(E, D, std=D ** -0.5), rn(E, std=0.1)
    sd["time_embed.2.weight"], sd["time_embed.2.bias"] = rn(E, E, std=E ** -0.5), rn(E, std=0.1)
    kin = 2 * in_channels * patch * patch
    sd["mixins.patch_embed.proj_sr.weight"], sd["mixins.patch_embed.proj_sr.bias"] = rn(D, 2 * in_channels, patch, patch, std=kin ** -0.5), rn(D, std=0.1)
    sd["mixins.patch_embed.text_proj.weight"], sd["mixins.patch_embed.text_proj.bias"] = rn(D, text_hidden, std=text_hidden ** -0.5), rn(D, std=0.1)
    for k in ("transformer.final_layernorm", "mixins.final_layer.norm_final"):
        sd[k + ".weight"], sd[k + ".bias"] = 1.0 + rn(D, std=0.1), rn(D, std=0.1)
    sd["mixins.final_layer.linear.weight"], sd["mixins.final_layer.linear.bias"] = rn(patch * patch * out_channels, D, std=D ** -0.5), rn(patch * patch * out_channels, std=0.1)
    sd["mixins.final_layer.adaLN_modulation.1.weight"], sd["mixins.final_layer.adaLN_modulation.1.bias"] = rn(2 * D, E, std=0.3 * E ** -0.5), rn(2 * D, std=0.2)
    return sd


def dit_forward(sd, cfg, x, timesteps, context, out_channels, patch=2):
    """x [1, T, 2C, H, W], timesteps [1], context [1, n_text, text_hidden] -> [1, T, C_out, H, W] (fp32)."""
    D = cfg.hidden
    half = D // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    emb = F.linear(F.silu(F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    _, T, C2, H, W = x.shape
    hid = patch_embed(sd, x, context, patch)
    h, w = H // patch, W // patch
    n_text = context.shape[1]
    for i in range(cfg.n_layers):
        hid = dit_block_forward(sd, cfg, i, hid, emb, n_text, T, h, w)
    hid = F.layer_norm(hid, (D,), sd["transformer.final_layernorm.weight"], sd["transformer.final_layernorm.bias"], cfg.ln_eps)
    return final_layer(sd, cfg, hid, emb, n_text, T, h, w, out_channels, patch)


def patch_embed(sd, x, context, patch=2):
    """ImagePatchEmbeddingMixin.word_embedding_forward (dit_video_concat.py:51-76): Conv2d(k = s = patch) per frame, tokens in
    (t h w) order behind the projected text tokens.  Pinned to the reference's own mixin by tests/golden/dit_block.pt."""
    _, T, C2, H, W = x.shape
    e = F.conv2d(x[0].float(), sd["mixins.patch_embed.proj_sr.weight"], sd["mixins.patch_embed.proj_sr.bias"], stride=patch)   # (t, d, h/p, w/p)
    e = e.flatten(2).transpose(1, 2).reshape(1, -1, e.shape[1])
    txt = F.linear(context.float(), sd["mixins.patch_embed.text_proj.weight"], sd["mixins.patch_embed.text_proj.bias"])
    return torch.cat([txt, e], dim=1)


def final_layer(sd, cfg, hid, emb, n_text, T, h, w, out_channels, patch=2):
    """FinalLayerMixin.final_forward (:395-410) + unpatchify (:353-369) on the video tokens.  Pinned like patch_embed."""
    D = cfg.hidden
    v = hid[:, n_text:]
    shift, scale = F.linear(F.silu(emb), sd["mixins.final_layer.adaLN_modulation.1.weight"], sd["mixins.final_layer.adaLN_modulation.1.bias"]).chunk(2, dim=1)
    v = F.layer_norm(v, (D,), sd["mixins.final_layer.norm_final.weight"], sd["mixins.final_layer.norm_final.bias"], 1e-6)
    v = v * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)
    v = F.linear(v, sd["mixins.final_layer.linear.weight"], sd["mixins.final_layer.linear.bias"])
    c, p = out_channels, patch
    return v.reshape(1, T, h, w, c, p, p).permute(0, 1, 4, 2, 5, 3, 6).reshape(1, T, c, h * p, w * p)
