"""Host-side mirror of one CogVideoX-5B DiT block of STAR's CogVideoX variant: what
`AdaLNMixin.layer_forward(hidden_states, mask, text_length=..., layer_id=..., emb=...)` computes for a layer
(cogvideox-based/sat/dit_video_concat.py:482-563, with the rotary / QK-LayerNorm attention hooks :254-346, :571-598 and the LIEM
gates of cogvideox-based/transformer.py:316-348).  The arithmetic runs in libstar_hip.so (star_dit_build /
star_dit_block_forward); weights are taken from a SAT-checkpoint state dict (keys below `model.diffusion_model.`)."""
import ctypes

import torch

from .. import lib as L
from .unet_v2v import stage_tensor


class DitConfigC(ctypes.Structure):
    _fields_ = [("hidden", ctypes.c_int32), ("heads", ctypes.c_int32), ("time_embed_dim", ctypes.c_int32), ("n_layers", ctypes.c_int32),
                ("ln_eps", ctypes.c_float)]


def _bind(lib):
    if getattr(lib, "_dit_bound", False):
        return
    c = lib.cdll
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    lib.dit_build = L._sig(c, "star_dit_build", i32, vp, ctypes.POINTER(DitConfigC))
    lib.dit_block_forward = L._sig(c, "star_dit_block_forward", i32, vp, i32, vp, vp, vp, i32, i32, i32, i32)
    lib._dit_bound = True


def dit_keys(n_layers):
    keys = []
    for i in range(n_layers):
        Lp, A = f"transformer.layers.{i}.", "mixins.adaln_layer."
        for n in ("input_layernorm", "post_attention_layernorm", "attention.query_key_value", "attention.dense", "mlp.dense_h_to_4h",
                  "mlp.dense_4h_to_h"):
            keys += [Lp + n + ".weight", Lp + n + ".bias"]
        keys += [Lp + "spa_local.conv1.weight", Lp + "temp_local.conv1.weight"]
        for n in (f"adaLN_modulations.{i}.1", f"query_layernorm_list.{i}", f"key_layernorm_list.{i}"):
            keys += [A + n + ".weight", A + n + ".bias"]
    return keys


class DiTBlocks:
    """`blocks = DiTBlocks(hidden=3072, heads=48, time_embed_dim=512, n_layers=42).load_state_dict(sd)`;
    `hidden_states = blocks.layer_forward(hidden_states, layer_id, emb, text_length, (T, H, W))` with hidden_states
    `[1, text_length + T*H*W, hidden]` (device tensor, any float dtype) and emb `[1, time_embed_dim]`."""

    def __init__(self, hidden=3072, heads=48, time_embed_dim=512, n_layers=42, ln_eps=1e-5, dtype=torch.float16, device=0, library=None):
        self.cfg = dict(hidden=hidden, heads=heads, time_embed_dim=time_embed_dim, n_layers=n_layers, ln_eps=ln_eps)
        self.dtype, self._device, self._library = dtype, device, library
        self.ctx = None

    def load_state_dict(self, sd, prefix=""):
        self.ctx = L.Context(self._device, self.dtype, self._library)
        _bind(self.ctx.lib)
        missing = [k for k in dit_keys(self.cfg["n_layers"]) if prefix + k not in sd]
        if missing:
            raise L.StarError(f"DiT state dict: {len(missing)} missing keys, e.g. {missing[:3]}")
        for k in dit_keys(self.cfg["n_layers"]):
            stage_tensor(self.ctx, k, sd[prefix + k])
        c = DitConfigC(self.cfg["hidden"], self.cfg["heads"], self.cfg["time_embed_dim"], self.cfg["n_layers"], self.cfg["ln_eps"])
        self.ctx._check(self.ctx.lib.dit_build(self.ctx.h, ctypes.byref(c)), "dit_build")
        return self

    def layer_forward(self, hidden_states, layer_id, emb, text_length, thw):
        if self.ctx is None:
            raise L.StarError("DiTBlocks: load_state_dict first")
        T, H, W = thw
        ctx = self.ctx
        S, D = text_length + T * H * W, self.cfg["hidden"]
        if tuple(hidden_states.shape) != (1, S, D):
            raise L.StarError(f"hidden_states must be [1, {S}, {D}], got {tuple(hidden_states.shape)}")
        ctx.use_current_stream()
        x = hidden_states.to(device=ctx.torch_device, dtype=self.dtype).contiguous()
        e = emb.to(device=ctx.torch_device, dtype=torch.float32).contiguous().reshape(-1)
        out = torch.empty_like(x)
        ctx._check(ctx.lib.dit_block_forward(ctx.h, int(layer_id), L._ptr(x), L._ptr(e), L._ptr(out), int(text_length), T, H, W),
                   "dit_block_forward")
        return out.to(hidden_states.dtype)
