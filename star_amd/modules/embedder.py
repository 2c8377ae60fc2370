"""Text encoder of the pipeline: `FrozenOpenCLIPEmbedder` (reference video_to_video/modules/embedder.py:12-72).

The reference runs OpenCLIP ViT-H/14's text tower (24 pre-LN blocks, width 1024, 16 heads, 77 tokens, causal mask), stops
one block early (`layer='penultimate'`) and applies `ln_final`: a `[1, 77, 1024]` tensor per prompt, twice per video
(positive and negative prompt).  It is host-side orchestration, not part of the per-chunk hot path (SURVEY.md section 8f
rank 3).  Two execution paths for the 23 transformer blocks + ln_final:

  * `runtime="hip"` (default whenever the embedder sits on a GPU): the tower runs on the SAME runtime as the denoiser --
    star_text_build / star_text_forward in libstar_hip.so (star_amd/csrc/text.cpp): star's GEMM, LayerNorm and flash-attention
    kernels, the latter with the causal mask added for this; only the token-embedding gather stays in torch;
  * `runtime="torch"`: the nn.Module below, used on the CPU and as the parity reference of the HIP path in the tests.

Where the model and the tokenizer come from:

  * with `open_clip` installed the wrapper builds the same model the reference builds (`create_model_and_transforms`,
    visual tower deleted) and tokenises with `open_clip.tokenize`;
  * without it (this image) the tower itself is still available: `OpenCLIPTextTransformer` restates the text transformer
    with open_clip's parameter names, so an open_clip text state dict loads unchanged (`text_state_dict=` / a path), and
    any callable `str | list[str] -> LongTensor[B, 77]` serves as the tokenizer.  The BPE vocabulary ships only inside the
    open_clip package, so a prompt STRING cannot be tokenised without it -- that case raises ImportError with this text.

PARITY UNPINNED for the restated tower: open_clip is not installed here; tests/test_embedder.py checks the nn.Module against an
independent statement of the block (on torch.nn.functional.multi_head_attention_forward, kept with the test infrastructure) and the HIP path
against the nn.Module.
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


class _ResidualAttentionBlock(nn.Module):
    """open_clip.transformer.ResidualAttentionBlock (pre-LN, exact GELU, no layer scale): parameter names kept."""

    def __init__(self, width, heads):
        super().__init__()
        self.ln_1 = nn.LayerNorm(width)
        self.attn = nn.MultiheadAttention(width, heads)
        self.ln_2 = nn.LayerNorm(width)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(width, width * 4)), ("gelu", nn.GELU()),
                                              ("c_proj", nn.Linear(width * 4, width))]))

    def forward(self, x, attn_mask=None):   # x: [L, B, W]
        h = self.ln_1(x)
        x = x + self.attn(h, h, h, need_weights=False, attn_mask=attn_mask)[0]
        return x + self.mlp(self.ln_2(x))


class _Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.resblocks = nn.ModuleList([_ResidualAttentionBlock(width, heads) for _ in range(layers)])
        self.grad_checkpointing = False


class OpenCLIPTextTransformer(nn.Module):
    """The text half of an open_clip CLIP model (ViT-H-14 defaults), state-dict compatible with it."""

    def __init__(self, vocab_size=49408, context_length=77, width=1024, heads=16, layers=24):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab_size, width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, width).normal_(std=0.01))
        self.transformer = _Transformer(width, layers, heads)
        self.ln_final = nn.LayerNorm(width)
        mask = torch.empty(context_length, context_length).fill_(float("-inf")).triu_(1)   # causal: open_clip build_attention_mask
        self.register_buffer("attn_mask", mask, persistent=False)

    def load_text_state_dict(self, sd):
        """accepts a full open_clip CLIP state dict: visual.*, text_projection and logit_scale are not used by the embedder."""
        own = self.state_dict()
        missing = [k for k in own if k not in sd]
        if missing:
            raise KeyError(f"text tower: {len(missing)} tensors missing (first: {missing[:3]})")
        self.load_state_dict({k: sd[k] for k in own}, strict=True)
        return self


class FrozenOpenCLIPEmbedder(nn.Module):
    """Same constructor / call surface as the reference class (embedder.py:12-72)."""
    LAYERS = ["last", "penultimate"]

    def __init__(self, pretrained="laion2b_s32b_b79k", arch="ViT-H-14", device="cuda", max_length=77, freeze=True,
                 layer="penultimate", model=None, tokenizer=None, text_state_dict=None, runtime=None, dtype=torch.float16, library=None):
        super().__init__()
        assert layer in self.LAYERS
        assert runtime in (None, "hip", "torch")
        self._runtime, self._hip_dtype, self._library, self._hip = runtime, dtype, library, None
        if model is None and text_state_dict is not None:
            if isinstance(text_state_dict, str):
                text_state_dict = torch.load(text_state_dict, map_location="cpu")
            model = OpenCLIPTextTransformer().load_text_state_dict(text_state_dict)
        if model is None or tokenizer is None:
            try:
                import open_clip
            except ImportError as e:
                raise ImportError("FrozenOpenCLIPEmbedder: open_clip is not installed.  Install it, or pass precomputed "
                                  "[1, 77, 1024] embeddings (input['y'], opt.negative_y), or pass text_state_dict= (an open_clip "
                                  "text state dict) together with tokenizer= (str -> LongTensor[B, 77])") from e
            if model is None:
                model, _, _ = open_clip.create_model_and_transforms(arch, device=torch.device("cpu"), pretrained=pretrained)
                del model.visual
            if tokenizer is None:
                tokenizer = open_clip.tokenize
        self.model = model
        self.tokenizer = tokenizer
        self.device = device
        self.max_length = max_length
        if freeze:
            self.freeze()
        self.layer = layer
        self.layer_idx = 0 if layer == "last" else 1
        self.model.to(device)

    def freeze(self):
        self.model = self.model.eval()
        for param in self.parameters():
            param.requires_grad = False

    @torch.no_grad()
    def forward(self, text):
        tokens = self.tokenizer(text)
        return self.encode_with_transformer(tokens.to(self.device))

    def _use_hip(self):
        if self._runtime is not None:
            return self._runtime == "hip"
        return torch.device(self.device).type == "cuda" or self._library is not None

    def _hip_tower(self):
        """stage the tower's weights once (open_clip names) and build it on the device context: star_text_build"""
        if self._hip is None:
            from .. import lib as L
            from .unet_v2v import stage_tensor
            dev = torch.device(self.device)
            ctx = L.Context(L.device_index(dev), self._hip_dtype, self._library)   # 'cuda' without an index = this rank's CURRENT device, not GPU 0
            sd = self.model.state_dict()
            blocks = self.model.transformer.resblocks
            keys = ["ln_final.weight", "ln_final.bias"]
            for i in range(len(blocks)):
                keys += [f"transformer.resblocks.{i}.{n}" for n in ("ln_1.weight", "ln_1.bias", "ln_2.weight", "ln_2.bias", "attn.in_proj_weight",
                                                                     "attn.in_proj_bias", "attn.out_proj.weight", "attn.out_proj.bias", "mlp.c_fc.weight",
                                                                     "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias")]
            for k in keys:
                stage_tensor(ctx, k, sd[k])
            width = self.model.ln_final.weight.shape[0]
            heads = blocks[0].attn.num_heads
            ctx._check(ctx.lib.text_build(ctx.h, int(width), int(heads), len(blocks)), "text_build")
            n_blocks = len(blocks)
            # the runtime holds its own 16-bit copy: the fp32 torch blocks leave the device (0.6 GB at full size); the embedding
            # tables stay where the tokens are looked up
            self.model.transformer.resblocks.to("cpu")
            self._hip = (ctx, L, width, n_blocks)
        return self._hip

    def encode_with_transformer(self, text):
        x = self.model.token_embedding(text)             # [B, 77, W]
        x = x + self.model.positional_embedding
        if self._use_hip():                              # blocks + ln_final on the HIP runtime (star_amd/csrc/text.cpp)
            ctx, L, width, n_blocks = self._hip_tower()
            ctx.use_current_stream()
            B, T, _ = x.shape
            xin = x.to(device=ctx.torch_device, dtype=self._hip_dtype).reshape(B * T, width).contiguous()
            out = torch.empty_like(xin)
            ctx._check(ctx.lib.text_forward(ctx.h, L._ptr(xin), B, T, n_blocks - self.layer_idx, L._ptr(out)), "text_forward")
            return out.reshape(B, T, width).float().to(x.device)   # back on the caller's device (the tower's context may sit elsewhere)
        blocks = self.model.transformer.resblocks        # the torch path after a HIP tower was built: its blocks were parked on the host
        if next(blocks.parameters()).device != x.device:
            blocks.to(x.device)
        x = x.permute(1, 0, 2)                           # NLD -> LND
        x = self.text_transformer_forward(x, attn_mask=self.model.attn_mask)
        x = x.permute(1, 0, 2)
        return self.model.ln_final(x)

    def text_transformer_forward(self, x, attn_mask=None):
        blocks = self.model.transformer.resblocks
        for i, r in enumerate(blocks):
            if i == len(blocks) - self.layer_idx:        # 'penultimate': skip the last block
                break
            x = r(x, attn_mask=attn_mask)
        return x

    def encode(self, text):
        return self(text)
