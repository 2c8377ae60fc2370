"""One CogVideoX-5B DiT block at full size (hidden 3072, 48 heads, 226 text + 7 x 30 x 45 video tokens = 9676) on the GPU:
time per block and the matmul / attention FLOP rate.  python tools/bench_dit.py [f16|bf16]"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import dit_oracle as O
from star_amd.modules.dit import DiTBlocks

dt = {"f16": torch.float16, "bf16": torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else "f16"]
torch.set_grad_enabled(False)
cfg = O.DitConfig(hidden=3072, heads=48, time_embed_dim=512, n_layers=1)
text_len, T, H, W = 226, 7, 30, 45
sd = O.random_dit_state_dict(cfg, seed=0)
blocks = DiTBlocks(cfg.hidden, cfg.heads, cfg.time_embed_dim, cfg.n_layers, cfg.ln_eps, dtype=dt).load_state_dict(sd)
x, emb = O.dit_inputs(cfg, text_len, T, H, W, seed=1)
x, emb = x.cuda().to(dt), emb.cuda()
for _ in range(2):
    y = blocks.layer_forward(x, 0, emb, text_len, (T, H, W))
torch.cuda.synchronize()
blocks.ctx.profile_begin()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 5
e0.record()
for _ in range(n):
    y = blocks.layer_forward(x, 0, emb, text_len, (T, H, W))
e1.record(); torch.cuda.synchronize()
prof = blocks.ctx.profile_end()
ms = e0.elapsed_time(e1) / n
S, D = text_len + T * H * W, cfg.hidden
flops = 24.0 * S * D * D + 4.0 * S * S * D
res = {"ms_per_block": ms, "TFLOP_per_block": flops / 1e12, "TFLOP/s": flops / ms / 1e9, "tokens": S, "finite": bool(torch.isfinite(y).all()),
       "families_ms": {k: round(v["ms"] / n, 3) for k, v in prof.items() if v["ms"] > 0},
       "attn_TFLOP/s": prof["attn_self"]["flops"] / prof["attn_self"]["ms"] / 1e9 if prof.get("attn_self", {}).get("ms") else None,
       "gemm_TFLOP/s": prof["gemm"]["flops"] / prof["gemm"]["ms"] / 1e9 if prof.get("gemm", {}).get("ms") else None}
print(json.dumps(res))
