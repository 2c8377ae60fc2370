"""Independent statement of one OpenCLIP text-tower block (TEST INFRASTRUCTURE ONLY): open_clip.transformer.ResidualAttentionBlock
(pre-LN, causal nn.MultiheadAttention, exact GELU) written on torch.nn.functional.multi_head_attention_forward, against which
tests/test_embedder.py checks the nn.Module restatement in star_amd/modules/embedder.py.  PARITY UNPINNED: open_clip itself is not
installed in this image (reference call sites: video_to_video/modules/embedder.py:25,49-72)."""
import math

import torch
import torch.nn.functional as F


def reference_block(x, p, heads, mask):
    """independent statement of one block on F.multi_head_attention_forward (tests only; x: [L, B, W], p: dict of tensors)."""
    h = F.layer_norm(x, x.shape[-1:], p["ln_1.weight"], p["ln_1.bias"])
    a, _ = F.multi_head_attention_forward(h, h, h, x.shape[-1], heads, p["attn.in_proj_weight"], p["attn.in_proj_bias"], None, None,
                                          False, 0.0, p["attn.out_proj.weight"], p["attn.out_proj.bias"], training=False,
                                          need_weights=False, attn_mask=mask)
    x = x + a
    h = F.layer_norm(x, x.shape[-1:], p["ln_2.weight"], p["ln_2.bias"])
    h = F.linear(h, p["mlp.c_fc.weight"], p["mlp.c_fc.bias"])
    h = 0.5 * h * (1.0 + torch.erf(h / math.sqrt(2.0)))
    return x + F.linear(h, p["mlp.c_proj.weight"], p["mlp.c_proj.bias"])
