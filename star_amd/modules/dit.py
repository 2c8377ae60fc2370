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


class DiffusionTransformer:
    """The whole denoiser of STAR's CogVideoX variant, `DiffusionTransformer.forward(x, timesteps, context)`
    (cogvideox-based/sat/dit_video_concat.py:791-817) = sinusoidal timestep embedding + `time_embed` MLP (:688-693), patch
    embedding of the (noisy latent | LQ latent) pair + text projection (`ImagePatchEmbeddingMixin.word_embedding_forward`,
    :51-76), the transformer layers (`DiTBlocks`, one `star_dit_block_forward` each), sat's `final_layernorm`, and
    `FinalLayerMixin.final_forward` (:395-410: LayerNorm, adaLN modulate, Linear, unpatchify).  Host Python over device tensors as
    in the reference; every matmul / LayerNorm / attention runs in libstar_hip.so through the ABI entries the blocks use.
    LoRA adapters of a fine-tuned checkpoint (`sat.model.finetune.lora2`, configs/cogvideox_5b/*.yaml) are expected merged into
    the dense weights.  x: [1, T, 2*C, H, W] (latent | LQ latent, C = 16), timesteps: [1], context: [1, 226, 4096] -> [1, T, C, H, W]."""

    def __init__(self, hidden=3072, heads=48, time_embed_dim=512, n_layers=42, in_channels=16, out_channels=16, patch_size=2,
                 text_hidden=4096, ln_eps=1e-5, dtype=torch.float16, device=0, library=None):
        self.hidden, self.E, self.C_in, self.C_out, self.p, self.text_hidden = hidden, time_embed_dim, in_channels, out_channels, patch_size, text_hidden
        self.ln_eps = ln_eps
        self.blocks = DiTBlocks(hidden, heads, time_embed_dim, n_layers, ln_eps, dtype=dtype, device=device, library=library)
        self.n_layers, self.dtype = n_layers, dtype
        self.w = None

    HOST_KEYS = ("time_embed.0", "time_embed.2", "mixins.patch_embed.proj_sr", "mixins.patch_embed.text_proj", "transformer.final_layernorm",
                 "mixins.final_layer.norm_final", "mixins.final_layer.linear", "mixins.final_layer.adaLN_modulation.1")

    def load_state_dict(self, sd, prefix=""):
        missing = [k + s_ for k in self.HOST_KEYS for s_ in (".weight", ".bias") if prefix + k + s_ not in sd]
        if missing:
            raise L.StarError(f"DiT state dict: {len(missing)} missing keys, e.g. {missing[:3]}")
        self.blocks.load_state_dict(sd, prefix)
        ctx = self.blocks.ctx
        dev = ctx.torch_device
        g = lambda k: sd[prefix + k].detach().float()
        w = {}
        for k in ("time_embed.0", "time_embed.2", "mixins.final_layer.adaLN_modulation.1"):       # tiny fp32 MLPs on the [1, E] embedding
            w[k] = (g(k + ".weight").to(dev), g(k + ".bias").to(dev))
        for k in ("mixins.patch_embed.proj_sr", "mixins.patch_embed.text_proj", "mixins.final_layer.linear"):   # GEMM operands [N, K]
            wt = g(k + ".weight")
            w[k] = (wt.reshape(wt.shape[0], -1).to(dev, self.dtype).contiguous(), g(k + ".bias").to(dev))
        for k in ("transformer.final_layernorm", "mixins.final_layer.norm_final"):
            w[k] = (g(k + ".weight").to(dev), g(k + ".bias").to(dev))
        self.w = w
        return self

    @staticmethod
    def timestep_embedding(timesteps, dim, max_period=10000):       # sgm.modules.diffusionmodules.util.timestep_embedding
        half = dim // 2
        freqs = torch.exp(-torch.log(torch.tensor(float(max_period))) * torch.arange(half, dtype=torch.float32) / half).to(timesteps.device)
        args = timesteps[:, None].float() * freqs[None]
        return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)

    def forward(self, x, timesteps, context):
        if self.w is None:
            raise L.StarError("DiffusionTransformer: load_state_dict first")
        ctx, w, p, D = self.blocks.ctx, self.w, self.p, self.hidden
        dev = ctx.torch_device
        b, T, C2, H, W = x.shape
        if b != 1 or C2 != 2 * self.C_in or H % p or W % p:
            raise L.StarError(f"x must be [1, T, {2 * self.C_in}, H, W] with H, W multiples of {p}")
        ctx.use_current_stream()
        lin = torch.nn.functional.linear
        emb = self.timestep_embedding(timesteps.to(dev), D)
        emb = lin(torch.nn.functional.silu(lin(emb, *w["time_embed.0"])), *w["time_embed.2"])                    # [1, E]
        h, wd = H // p, W // p
        text_len = context.shape[1]
        # patch rows in Conv2d(k = s = p) weight order (c, py, px); tokens in (t h w) order, text tokens first
        patches = x[0].to(dev, self.dtype).reshape(T, C2, h, p, wd, p).permute(0, 2, 4, 1, 3, 5).reshape(T * h * wd, C2 * p * p).contiguous()
        hidden = torch.empty(1, text_len + T * h * wd, D, dtype=self.dtype, device=dev)
        ctx.gemm(context[0].to(dev, self.dtype).contiguous(), w["mixins.patch_embed.text_proj"][0], bias=w["mixins.patch_embed.text_proj"][1],
                 out=hidden[0, :text_len])
        ctx.gemm(patches, w["mixins.patch_embed.proj_sr"][0], bias=w["mixins.patch_embed.proj_sr"][1], out=hidden[0, text_len:])
        for i in range(self.n_layers):
            hidden = self.blocks.layer_forward(hidden, i, emb, text_len, (T, h, wd))
        v = hidden[0, text_len:].contiguous()
        v = ctx.layer_norm(v, *w["transformer.final_layernorm"], eps=self.ln_eps)                                  # sat BaseTransformer.final_layernorm
        v = ctx.layer_norm(v, *w["mixins.final_layer.norm_final"], eps=1e-6)
        shift, scale = lin(torch.nn.functional.silu(emb), *w["mixins.final_layer.adaLN_modulation.1"]).chunk(2, dim=1)
        v = (v.float() * (1 + scale) + shift).to(self.dtype)                                                       # modulate (:349-350)
        o = ctx.gemm(v, w["mixins.final_layer.linear"][0], bias=w["mixins.final_layer.linear"][1], out_f32=True)   # [T h w, c p p]
        c = self.C_out
        return o.reshape(T, h, wd, c, p, p).permute(0, 3, 1, 4, 2, 5).reshape(1, T, c, h * p, wd * p)              # unpatchify (:353-369)

    __call__ = forward
