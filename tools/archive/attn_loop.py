"""Run one attention variant at the L0 shape in a loop for N seconds (for power / clock sampling with rocm-smi alongside)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from star_amd import lib as L
variant, secs = int(sys.argv[1]), float(sys.argv[2])
fill = sys.argv[3] if len(sys.argv) > 3 else "randn"
ctx = L.Context(0, torch.float16)
B, heads, N = 32, 5, 26352
C = heads * 64
qkv = torch.randn(B, N, 3 * C, device="cuda", dtype=torch.float16) if fill == "randn" else torch.zeros(B, N, 3 * C, device="cuda", dtype=torch.float16)
out = torch.empty(B, N, C, device="cuda", dtype=torch.float16)
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(10):
        ctx.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, out=out, variant=variant)
    torch.cuda.synchronize(); n += 10
dt = time.time() - t0
print(f"variant {variant} {fill}: {dt / n * 1e3:.3f} ms per launch, {4.0 * B * heads * N * N * 64 / (dt / n) / 1e12:.0f} TF/s", flush=True)
