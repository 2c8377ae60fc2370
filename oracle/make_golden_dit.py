"""Generates tests/golden/dit_block.pt by EXECUTING the reference's own DiT-block code on the CPU (build container only:
/root/reference does not exist on the GPU box; the fixture is committed).

What runs from the reference, unmodified, loaded by path:
  cogvideox-based/sat/dit_video_concat.py : AdaLNMixin.layer_forward / .attention_fn, Rotary3DPositionEmbeddingMixin (tables,
      rotary, attention_fn), modulate, rotate_half
  cogvideox-based/transformer.py          : SpatialAttention, TemporalLocalAttention (the LIEM gates)
What is a stand-in (SwissArmyTransformer==0.4.12 is not vendored and not installed): the `sat` / `sgm` imports of those two
files -- BaseMixin / BaseModel shells, LayerNorm = torch.nn.LayerNorm, and sat's default attention / MLP forward restated from
the published package (fused dense -> q | k | v heads -> attention_fn hooks -> dense; dense_h_to_4h -> tanh GELU -> dense_4h_to_h).

    python oracle/make_golden_dit.py
"""
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
REF = "/root/reference/cogvideox-based"


def attention_fn_default(q, k, v, mask, attention_dropout=None, log_attention_weights=None, scaling_attention_score=True, **kw):
    return F.scaled_dot_product_attention(q, k, v)   # sat.transformer_defaults.standard_attention without mask / dropout


def _stub_modules():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class BaseMixin(nn.Module):
        pass

    class BaseModel(nn.Module):
        pass

    def non_conflict(f):
        return f

    ident = lambda *a, **k: None
    mod("sat", mpu=None)
    mod("sat.model")
    mod("sat.model.base_model", BaseModel=BaseModel, non_conflict=non_conflict)
    mod("sat.model.mixins", BaseMixin=BaseMixin)
    mod("sat.transformer_defaults", HOOKS_DEFAULT={"attention_fn": attention_fn_default}, attention_fn_default=attention_fn_default,
        standard_attention=attention_fn_default, split_tensor_along_last_dim=lambda t, n: t.chunk(n, dim=-1))
    mpu = mod("sat.mpu", get_model_parallel_world_size=lambda: 1, ColumnParallelLinear=nn.Linear, RowParallelLinear=nn.Linear,
              VocabParallelEmbedding=nn.Embedding, gather_from_model_parallel_region=lambda x: x,
              copy_to_model_parallel_region=lambda x: x, checkpoint=ident)
    sys.modules["sat"].mpu = mpu
    mod("sat.mpu.layers", ColumnParallelLinear=nn.Linear)
    mod("sat.mpu.utils", divide=lambda a, b: a // b, sqrt=lambda x: x ** 0.5, scaled_init_method=ident, unscaled_init_method=ident,
        gelu=None)
    mod("sat.ops")
    mod("sat.ops.layernorm", LayerNorm=nn.LayerNorm, RMSNorm=nn.LayerNorm)
    mod("sgm")
    mod("sgm.util", instantiate_from_config=ident)
    mod("sgm.modules")
    mod("sgm.modules.diffusionmodules")
    mod("sgm.modules.diffusionmodules.openaimodel", Timestep=nn.Identity)
    mod("sgm.modules.diffusionmodules.util", linear=nn.Linear, timestep_embedding=ident)


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    import dit_oracle as O
    torch.set_grad_enabled(False)
    _stub_modules()
    dit = _load(os.path.join(REF, "sat", "dit_video_concat.py"), "ref_dit_video_concat")
    tr = _load(os.path.join(REF, "transformer.py"), "ref_transformer")
    cfg = O.SMALL_DIT_CONFIG
    text_len, T, H, W = 5, 3, 4, 6
    D, heads = cfg.hidden, cfg.heads
    sd = O.random_dit_state_dict(cfg, seed=0)
    x, emb = O.dit_inputs(cfg, text_len, T, H, W, seed=1)
    layer_id = 1
    L, A = f"transformer.layers.{layer_id}.", "mixins.adaln_layer."

    # ---- the reference mixins, with this layer's weights
    ada = dit.AdaLNMixin(width=W, height=H, hidden_size=D, num_layers=cfg.n_layers, time_embed_dim=cfg.time_embed_dim,
                         compressed_num_frames=T, qk_ln=True, hidden_size_head=64, elementwise_affine=True)
    ada.load_state_dict({k[len(A):]: v for k, v in sd.items() if k.startswith(A)})
    rope = dit.Rotary3DPositionEmbeddingMixin(height=H, width=W, compressed_num_frames=T, hidden_size=D, hidden_size_head=64,
                                              text_length=text_len)

    # ---- a transformer layer shell: the reference's LIEM modules + sat's defaults restated
    class Layer(nn.Module):
        def __init__(self):
            super().__init__()
            self.input_layernorm = nn.LayerNorm(D, eps=cfg.ln_eps)
            self.post_attention_layernorm = nn.LayerNorm(D, eps=cfg.ln_eps)
            self.spa_local = tr.SpatialAttention()
            self.temp_local = tr.TemporalLocalAttention()
            self.query_key_value = nn.Linear(D, 3 * D)
            self.dense = nn.Linear(D, D)
            self.dense_h_to_4h = nn.Linear(D, 4 * D)
            self.dense_4h_to_h = nn.Linear(4 * D, D)

        def attention(self, hidden_states, mask, **kw):          # sat attention_forward_default
            qkv = self.query_key_value(hidden_states)
            q, k, v = [t.reshape(1, -1, heads, 64).permute(0, 2, 1, 3).contiguous() for t in qkv.chunk(3, dim=-1)]
            # hook chain of the model: AdaLNMixin.attention_fn (QK LayerNorm) wraps Rotary3D...attention_fn (rotary, then the default)
            ctx = ada.attention_fn(q, k, v, mask, old_impl=rope.attention_fn, **kw)
            return self.dense(ctx.permute(0, 2, 1, 3).reshape(1, -1, D))

        def mlp(self, hidden_states, **kw):                      # sat mlp_forward_default with gelu_impl
            return self.dense_4h_to_h(O.gelu_tanh(self.dense_h_to_4h(hidden_states)))

    layer = Layer()
    layer.load_state_dict({
        "input_layernorm.weight": sd[L + "input_layernorm.weight"], "input_layernorm.bias": sd[L + "input_layernorm.bias"],
        "post_attention_layernorm.weight": sd[L + "post_attention_layernorm.weight"],
        "post_attention_layernorm.bias": sd[L + "post_attention_layernorm.bias"],
        "spa_local.conv1.weight": sd[L + "spa_local.conv1.weight"], "temp_local.conv1.weight": sd[L + "temp_local.conv1.weight"],
        "query_key_value.weight": sd[L + "attention.query_key_value.weight"], "query_key_value.bias": sd[L + "attention.query_key_value.bias"],
        "dense.weight": sd[L + "attention.dense.weight"], "dense.bias": sd[L + "attention.dense.bias"],
        "dense_h_to_4h.weight": sd[L + "mlp.dense_h_to_4h.weight"], "dense_h_to_4h.bias": sd[L + "mlp.dense_h_to_4h.bias"],
        "dense_4h_to_h.weight": sd[L + "mlp.dense_4h_to_h.weight"], "dense_4h_to_h.bias": sd[L + "mlp.dense_4h_to_h.bias"],
    })
    shell = types.SimpleNamespace(layers={layer_id: layer}, layernorm_order="pre")
    object.__setattr__(ada, "transformer", shell)
    out = ada.layer_forward(x.clone(), None, text_length=text_len, layer_id=layer_id, emb=emb)
    # ---- the reference's patch embedding and final layer (ImagePatchEmbeddingMixin :23-83, FinalLayerMixin :372-415, unpatchify)
    C, p_, th = 8, 2, 64
    msd = O.random_dit_model_state_dict(cfg, in_channels=C, out_channels=C, patch=p_, text_hidden=th, seed=0)
    gm = torch.Generator().manual_seed(3)
    Hl, Wl, n_text = 6, 8, 5
    xm = torch.randn(1, T, 2 * C, Hl, Wl, generator=gm)
    cm = torch.randn(1, n_text, th, generator=gm)
    pe = dit.ImagePatchEmbeddingMixin(in_channels=C, hidden_size=D, patch_size=p_, text_hidden_size=th)
    pe.load_state_dict({k[len("mixins.patch_embed."):]: v for k, v in msd.items() if k.startswith("mixins.patch_embed.")})
    pe_out = pe.word_embedding_forward(None, images=xm, encoder_outputs=cm)
    fl = dit.FinalLayerMixin(hidden_size=D, time_embed_dim=cfg.time_embed_dim, patch_size=p_, out_channels=C, latent_width=Wl,
                             latent_height=Hl, elementwise_affine=True)
    fl.load_state_dict({k[len("mixins.final_layer."):]: v for k, v in msd.items() if k.startswith("mixins.final_layer.")})
    hid_m = torch.randn(1, n_text + T * (Hl // p_) * (Wl // p_), D, generator=gm)
    emb_m = torch.randn(1, cfg.time_embed_dim, generator=gm)
    fl_out = fl.final_forward(hid_m, text_length=n_text, emb=emb_m)
    d_pe = float((O.patch_embed(msd, xm, cm, p_) - pe_out).abs().max())
    d_fl = float((O.final_layer(msd, cfg, hid_m, emb_m, n_text, T, Hl // p_, Wl // p_, C, p_) - fl_out).abs().max())
    path = os.path.join(ROOT, "tests", "golden", "dit_block.pt")
    torch.save({"cfg": dict(hidden=D, heads=heads, time_embed_dim=cfg.time_embed_dim, n_layers=cfg.n_layers, ln_eps=cfg.ln_eps),
                "geometry": (text_len, T, H, W), "layer": layer_id, "sd_seed": 0, "in_seed": 1, "out": out.float().clone(),
                "rope_cos": rope.freqs_cos.clone(), "rope_sin": rope.freqs_sin.clone(),
                "model_parts": {"C": C, "patch": p_, "text_hidden": th, "geometry": (T, Hl, Wl, n_text), "seed": 3,
                                "patch_embed_out": pe_out.float().clone(), "final_layer_out": fl_out.float().clone()}}, path)
    print("patch embedding / final layer: restatement vs reference max abs diff %.3e / %.3e" % (d_pe, d_fl))
    ours = O.dit_block_forward(sd, cfg, layer_id, x, emb, text_len, T, H, W)
    print("saved", path, "out rms %.4f" % float(out.pow(2).mean().sqrt()), "| restatement vs reference: max abs diff %.3e" % float((ours - out).abs().max()))


if __name__ == "__main__":
    main()
