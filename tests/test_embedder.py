"""Text-encoder wrapper (star_amd/modules/embedder.py; reference video_to_video/modules/embedder.py:12-72): the prompt-string path
with a stub tokenizer, the penultimate-layer rule, the causal mask, and the restated OpenCLIP text block against an independent
statement on F.multi_head_attention_forward.  open_clip is not installed in this image; the tower is CROSS-PINNED to the HuggingFace
CLIPTextModel implementation of the same architecture (test_text_tower_is_cross_pinned_to_the_hf_clip_implementation)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from text_oracle import (hf_clip_text_model, hf_penultimate_embedding, hf_to_open_clip_state_dict, oracle_tower,  # noqa: E402
                         reference_block)
from star_amd.modules.embedder import FrozenOpenCLIPEmbedder, OpenCLIPTextTransformer  # noqa: E402
from util import BACKENDS  # noqa: E402

torch.set_grad_enabled(False)


def small_tower(layers=3, width=64, heads=4):
    torch.manual_seed(0)
    m = OpenCLIPTextTransformer(vocab_size=100, context_length=77, width=width, heads=heads, layers=layers)
    for p in m.parameters():
        if p.dim() == 1:
            p.add_(torch.randn_like(p) * 0.1)
    return m


def tok(text):
    texts = [text] if isinstance(text, str) else text
    out = torch.zeros(len(texts), 77, dtype=torch.long)
    for i, t in enumerate(texts):
        ids = [1] + [3 + (ord(c) % 90) for c in t][:75] + [2]
        out[i, :len(ids)] = torch.tensor(ids)
    return out


def test_prompt_string_goes_through_tokenizer_and_stops_one_block_early():
    m = small_tower(3)
    emb = FrozenOpenCLIPEmbedder(device="cpu", model=m, tokenizer=tok, runtime="torch")
    z = emb("a cat")
    assert z.shape == (1, 77, 64) and z.dtype == torch.float32
    sd = {k: v for k, v in m.state_dict().items()}
    x = (m.token_embedding(tok("a cat")) + m.positional_embedding).permute(1, 0, 2)
    for i in range(2):    # 3 blocks, 'penultimate' -> the first two
        p = {k[len(f"transformer.resblocks.{i}."):]: v for k, v in sd.items() if k.startswith(f"transformer.resblocks.{i}.")}
        x = reference_block(x, p, 4, m.attn_mask)
    ref = torch.nn.functional.layer_norm(x.permute(1, 0, 2), (64,), sd["ln_final.weight"], sd["ln_final.bias"])
    assert float((z - ref).abs().max()) < 2e-5
    last = FrozenOpenCLIPEmbedder(device="cpu", model=m, tokenizer=tok, layer="last", runtime="torch")("a cat")
    assert float((last - z).abs().max()) > 1e-3


def test_causal_mask_later_tokens_do_not_leak_backwards():
    m = small_tower(2)
    emb = FrozenOpenCLIPEmbedder(device="cpu", model=m, tokenizer=tok, layer="last", runtime="torch")
    a, b = emb("hello world"), emb("hello there")
    assert float((a[0, :6] - b[0, :6]).abs().max()) == 0.0      # <start> + "hello" are identical, what follows must not matter
    assert float((a[0, 8:] - b[0, 8:]).abs().max()) > 0


def test_state_dict_surface_matches_open_clip_names():
    keys = set(OpenCLIPTextTransformer(100, 77, 64, 4, 1).state_dict())
    assert {"token_embedding.weight", "positional_embedding", "ln_final.weight", "transformer.resblocks.0.attn.in_proj_weight",
            "transformer.resblocks.0.attn.out_proj.bias", "transformer.resblocks.0.mlp.c_fc.weight",
            "transformer.resblocks.0.mlp.c_proj.bias", "transformer.resblocks.0.ln_1.weight", "transformer.resblocks.0.ln_2.bias"} <= keys
    full = {"visual.proj": torch.zeros(1), "logit_scale": torch.zeros(()), **OpenCLIPTextTransformer(100, 77, 64, 4, 1).state_dict()}
    OpenCLIPTextTransformer(100, 77, 64, 4, 1).load_text_state_dict(full)


def test_missing_open_clip_is_reported_only_when_a_string_needs_it():
    try:
        import open_clip  # noqa: F401
        pytest.skip("open_clip is installed")
    except ImportError:
        pass
    with pytest.raises(ImportError):
        FrozenOpenCLIPEmbedder(device="cpu")


def _rel(a, b):
    return float((a.float().cpu() - b.float().cpu()).pow(2).mean().sqrt() / b.float().cpu().pow(2).mean().sqrt())


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_text_tower_on_the_hip_runtime_matches_the_torch_tower(backend, dtype, request):
    """the blocks + ln_final through star_text_forward (GEMM, LayerNorm, causal flash attention of
    libstar_hip.so) against the nn.Module restatement in fp32; penultimate and last layer; two prompts of different length (the
    causal mask is what keeps the padding behind the end token from leaking forward)."""
    emu = request.getfixturevalue("emu_lib") if backend == "emu" else None
    m = small_tower(3, width=128, heads=2)
    ref = FrozenOpenCLIPEmbedder(device="cpu", model=m, tokenizer=tok, runtime="torch")(["a cat", "a much longer prompt about a dog"])
    dev = "cpu" if backend == "emu" else "cuda:0"
    for layer in ("penultimate", "last"):
        want = ref if layer == "penultimate" else FrozenOpenCLIPEmbedder(device="cpu", model=m, tokenizer=tok, runtime="torch", layer="last")(
            ["a cat", "a much longer prompt about a dog"])
        hip = FrozenOpenCLIPEmbedder(device=dev, model=small_tower(3, width=128, heads=2), tokenizer=tok, runtime="hip", dtype=dtype, library=emu,
                                     layer=layer)
        z = hip(["a cat", "a much longer prompt about a dog"])
        assert z.shape == (2, 77, 128) and z.dtype == torch.float32 and torch.isfinite(z).all()
        assert _rel(z, want) < (4e-3 if dtype == torch.float16 else 3e-2), (layer, _rel(z, want))
    a = hip(["hello world"])
    b = hip(["hello there"])
    assert float((a[0, :6] - b[0, :6]).abs().max()) == 0.0 and float((a[0, 8:] - b[0, 8:]).abs().max()) > 0      # causal on the HIP path too


@pytest.mark.parametrize("size", ["small", "vit_h_14"])
def test_text_tower_is_cross_pinned_to_the_hf_clip_implementation(size):
    """CROSS-PIN (HF), open_clip absent: transformers.CLIPTextModel -- an independent published implementation of the same pre-LN
    causal text transformer, installed in this image -- random-init in ViT-H/14's text configuration, its state dict mapped to
    open_clip's names; (a) the oracle's block statement (oracle/text_oracle.py) and (b) the product's nn.Module restatement
    (star_amd/modules/embedder.py, what FrozenOpenCLIPEmbedder runs with runtime='torch') reproduce HF's penultimate hidden state +
    final LayerNorm to fp32 round-off.  Reference call site: video_to_video/modules/embedder.py:49-72."""
    width, heads, layers, vocab = (64, 4, 3, 100) if size == "small" else (1024, 16, 24, 49408)
    hf = hf_clip_text_model(width, heads, layers, vocab_size=vocab, seed=3)
    sd = hf_to_open_clip_state_dict(hf)
    g = torch.Generator().manual_seed(5)
    tokens = torch.randint(2, vocab - 1, (2, 77), generator=g)
    tokens[0, 20:] = 1                                        # a short prompt: padding behind the end token
    want = hf_penultimate_embedding(hf, tokens)
    got_oracle = oracle_tower(sd, tokens, heads)
    m = OpenCLIPTextTransformer(vocab_size=vocab, context_length=77, width=width, heads=heads, layers=layers).load_text_state_dict(sd)
    got_module = FrozenOpenCLIPEmbedder(device="cpu", model=m, tokenizer=lambda t: tokens, runtime="torch")("x")
    scale = float(want.abs().max())
    eo, em = float((got_oracle - want).abs().max()), float((got_module - want).abs().max())
    print(f"text tower {size}: |oracle - HF| = {eo:.2e}, |module - HF| = {em:.2e} (max |HF| = {scale:.2f})")
    assert want.shape == (2, 77, width) and eo <= 1e-5 * max(1.0, scale) and em <= 1e-5 * max(1.0, scale)
    # the 'last' layer too (one more block): HF's last_hidden_state is final_layer_norm(hidden_states[-1])
    last = FrozenOpenCLIPEmbedder(device="cpu", model=m, tokenizer=lambda t: tokens, runtime="torch", layer="last")("x")
    assert float((last - hf(input_ids=tokens).last_hidden_state).abs().max()) <= 1e-5 * max(1.0, scale)


@pytest.mark.gpu
def test_full_size_text_tower_on_the_gpu_against_hf_clip():
    """the pinned chain on hardware: HF CLIPTextModel (fp32, CPU) in ViT-H/14's text configuration is the reference, its weights go
    through the open_clip-named state dict into the HIP runtime (star_text_forward, fp16)."""
    hf = hf_clip_text_model(seed=4)
    sd = hf_to_open_clip_state_dict(hf)
    g = torch.Generator().manual_seed(6)
    tokens = torch.randint(2, 49000, (2, 77), generator=g)
    tokens[0, 20:] = 1
    want = hf_penultimate_embedding(hf, tokens)
    m = OpenCLIPTextTransformer().load_text_state_dict(sd)
    hip = FrozenOpenCLIPEmbedder(device="cuda:0", model=m, tokenizer=lambda t: tokens)
    assert hip._use_hip()
    z = hip("x")
    print(f"text tower, 23 blocks + ln_final, fp16 HIP vs HF CLIPTextModel fp32: relative rms {_rel(z, want):.2e}")
    assert z.shape == (2, 77, 1024) and torch.isfinite(z).all() and _rel(z, want) < 1e-2
    assert next(m.transformer.resblocks.parameters()).device.type == "cpu"      # the torch blocks left the device once staged


@pytest.mark.gpu
def test_full_size_text_tower_on_the_gpu():
    """ViT-H/14 text tower at its real size (24 blocks, width 1024, 16 heads), random weights, fp16 on the HIP runtime vs the fp32
    torch tower; `penultimate` as the pipeline uses it (embedder.py:24)."""
    torch.manual_seed(1)
    m = OpenCLIPTextTransformer()
    for p in m.parameters():
        if p.dim() == 1:
            p.add_(torch.randn_like(p) * 0.05)
    tokens = torch.randint(3, 49000, (2, 77))
    tokens[0, 20:] = 0
    want = FrozenOpenCLIPEmbedder(device="cpu", model=m, tokenizer=lambda t: tokens, runtime="torch")("x")
    hip = FrozenOpenCLIPEmbedder(device="cuda:0", model=m, tokenizer=lambda t: tokens)
    assert hip._use_hip()
    z = hip("x")
    print(f"text tower, 23 blocks + ln_final, fp16 HIP vs fp32 torch: relative rms {_rel(z, want):.2e}")
    assert z.shape == (2, 77, 1024) and torch.isfinite(z).all() and _rel(z, want) < 1e-2
