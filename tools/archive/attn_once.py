"""Launch one spatial self-attention variant at a cfg2 level a few times (for rocprofv3 --pmc passes).
   python tools/attn_once.py <variant> [L0|L1|L2] [f16|bf16]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from star_amd import lib as L
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 9
B, heads, N = {"L0": (32, 5, 26352), "L1": (32, 10, 6696), "L2": (32, 20, 1728)}[sys.argv[2] if len(sys.argv) > 2 else "L0"]
dt = {"f16": torch.float16, "bf16": torch.bfloat16}[sys.argv[3] if len(sys.argv) > 3 else "f16"]
ctx = L.Context(0, dt)
C = heads * 64
qkv = torch.randn(B, N, 3 * C, device="cuda", dtype=dt)
out = torch.empty(B, N, C, device="cuda", dtype=dt)
for _ in range(3):
    ctx.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, out=out, variant=variant)
torch.cuda.synchronize()
