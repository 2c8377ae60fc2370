"""Launch the spatial self-attention kernel at the full cfg2 L0 shape (32 frames x 5 heads, N = 26352) a few times
(for rocprofv3 --pmc passes: FETCH_SIZE / WRITE_SIZE per launch)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from star_amd import lib as L
dt = torch.float16
ctx = L.Context(0, dt)
B, heads, N = 32, 5, 26352
qkv = torch.randn(B, N, 960, device="cuda", dtype=dt)
out = torch.empty(B, N, 320, device="cuda", dtype=dt)
for _ in range(3):
    ctx.attention(qkv[..., :320], qkv[..., 320:640], qkv[..., 640:], heads, out=out)
torch.cuda.synchronize()
