"""Ablation probes of the spatial self-attention kernel (variant 6) at the cfg2 L0 shape: which part of the key-tile
loop the time belongs to.  Results are NOT valid attention outputs; bench only."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_amd import lib as L
ctx = L.Context(0, torch.float16, L.Library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench", "libstar_hip_bench.so")))   # bench build: make bench
dev = ctx.torch_device
B, heads, N = 8, 5, 26352
C = heads * 64
qkv = torch.randn(B, N, 3 * C, device=dev, dtype=torch.float16)
out = torch.empty(B, N, C, device=dev, dtype=torch.float16)
flops = 4.0 * B * heads * N * N * 64
names = {9: "full kernel (variant 9)", 16: "variant 9, no softmax VALU (constant P)", 17: "variant 9, no K/V staging after tile 1", 6: "full kernel (variant 6)", 11: "exp2 -> one v_mul", 12: "no PV MFMAs", 13: "no K/V staging after tile 1", 14: "no staging, no barriers"}
def t_ms(v, iters=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ctx.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, out=out, variant=v)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for v in names: t_ms(v, 1)
for rnd in range(2):
    for v, nm in names.items():
        ms = t_ms(v)
        print("%-34s %7.3f ms  %6.0f TF/s-equivalent" % (nm, ms, flops / ms / 1e9), flush=True)
