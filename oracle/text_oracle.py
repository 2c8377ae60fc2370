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


# ---------------------------------------------------------------------------------------------------------------------------
# Cross-pin with an INDEPENDENT PUBLISHED implementation that IS installed: HuggingFace `transformers.CLIPTextModel` implements the
# same pre-LN causal text transformer (the class Stable Diffusion 2 runs the OpenCLIP ViT-H/14 text weights through, taking the
# penultimate hidden state and applying the final LayerNorm -- exactly FrozenOpenCLIPEmbedder(layer='penultimate'),
# video_to_video/modules/embedder.py:49-72).  open_clip itself stays absent: "cross-pinned (HF), open_clip absent".
def hf_clip_text_model(width=1024, heads=16, layers=24, vocab_size=49408, context_length=77, seed=0):
    """random-init transformers.CLIPTextModel in open_clip ViT-H/14's text configuration (exact-erf GELU, eps 1e-5)"""
    from transformers import CLIPTextConfig, CLIPTextModel
    cfg = CLIPTextConfig(vocab_size=vocab_size, hidden_size=width, intermediate_size=4 * width, num_hidden_layers=layers,
                         num_attention_heads=heads, max_position_embeddings=context_length, hidden_act="gelu", layer_norm_eps=1e-5,
                         projection_dim=width, bos_token_id=0, eos_token_id=vocab_size - 1, pad_token_id=1)
    torch.manual_seed(seed)
    m = CLIPTextModel(cfg).eval()
    with torch.no_grad():   # HF initialises biases to zero and LayerNorms to (1, 0): perturb them so that every tensor matters
        g = torch.Generator().manual_seed(seed + 1)
        for n, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
    return m


def hf_to_open_clip_state_dict(hf_model):
    """transformers CLIPTextModel state dict -> open_clip CLIP text names (token_embedding.weight, positional_embedding,
    transformer.resblocks.{i}.{ln_1, ln_2, attn.in_proj_*, attn.out_proj, mlp.c_fc, mlp.c_proj}, ln_final): q | k | v stacked into
    in_proj in open_clip's order."""
    sd = {k.split("text_model.")[-1]: v for k, v in hf_model.state_dict().items()}
    out = {"token_embedding.weight": sd["embeddings.token_embedding.weight"].clone(),
           "positional_embedding": sd["embeddings.position_embedding.weight"].clone(),
           "ln_final.weight": sd["final_layer_norm.weight"].clone(), "ln_final.bias": sd["final_layer_norm.bias"].clone()}
    n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layers."))
    for i in range(n_layers):
        s, d = f"encoder.layers.{i}.", f"transformer.resblocks.{i}."
        for wb in ("weight", "bias"):
            out[d + f"attn.in_proj_{wb}"] = torch.cat([sd[s + f"self_attn.{x}_proj.{wb}"] for x in ("q", "k", "v")], dim=0)
            out[d + f"attn.out_proj.{wb}"] = sd[s + f"self_attn.out_proj.{wb}"].clone()
            out[d + f"ln_1.{wb}"] = sd[s + f"layer_norm1.{wb}"].clone()
            out[d + f"ln_2.{wb}"] = sd[s + f"layer_norm2.{wb}"].clone()
            out[d + f"mlp.c_fc.{wb}"] = sd[s + f"mlp.fc1.{wb}"].clone()
            out[d + f"mlp.c_proj.{wb}"] = sd[s + f"mlp.fc2.{wb}"].clone()
    return out


def hf_penultimate_embedding(hf_model, tokens):
    """what FrozenOpenCLIPEmbedder(layer='penultimate') returns, computed by the HF implementation: hidden state behind the
    second-to-last block, then the final LayerNorm"""
    o = hf_model(input_ids=tokens, output_hidden_states=True)
    tm = getattr(hf_model, "text_model", hf_model)
    return tm.final_layer_norm(o.hidden_states[-2])


def oracle_tower(sd, tokens, heads, skip_last=1):
    """the text tower on reference_block (above): embeddings, all blocks but the last `skip_last`, ln_final -> [B, 77, W]"""
    x = (sd["token_embedding.weight"][tokens] + sd["positional_embedding"]).permute(1, 0, 2)
    L = x.shape[0]
    mask = torch.full((L, L), float("-inf")).triu_(1)
    n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("transformer.resblocks."))
    for i in range(n_layers - skip_last):
        pre = f"transformer.resblocks.{i}."
        x = reference_block(x, {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, heads, mask)
    return F.layer_norm(x.permute(1, 0, 2), x.shape[-1:], sd["ln_final.weight"], sd["ln_final.bias"])
