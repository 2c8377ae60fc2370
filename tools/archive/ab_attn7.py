"""A/B of the shipped spatial-attention kernel (variant 9, attn5.h) against the one-wave-per-SIMD query-block pipeline (variants
40 / 41 = 2 / 3 query blocks per wave, attn7.h) at the cfg2 shapes, product library, interleaved rounds in one process;
errors against fp32 softmax(QK^T/8)V on the smaller levels.   python tools/ab_attn7.py [f16|bf16] [9,40,41] [levels: L0,L1,L2]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from star_amd import lib as L
dt = {"f16": torch.float16, "bf16": torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else "f16"]
variants = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["9", "40", "41"])]
levels = (sys.argv[3].split(",") if len(sys.argv) > 3 else ["L0", "L1", "L2"])
ctx = L.Context(0, dt)
dev = ctx.torch_device
def t_ms(fn, iters=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (B, heads, N, tag) in [(32, 5, 26352, "L0"), (32, 10, 6696, "L1"), (32, 20, 1728, "L2")]:
    if tag not in levels: continue
    C = heads * 64
    qkv = torch.randn(B, N, 3 * C, device=dev, dtype=dt)
    out = torch.empty(B, N, C, device=dev, dtype=dt)
    flops = 4.0 * B * heads * N * N * 64
    res = {v: [] for v in variants}
    run = lambda v: ctx.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, out=out, variant=v)
    for v in variants: run(v)
    torch.cuda.synchronize()
    for rnd in range(3):
        for v in variants:
            res[v].append(t_ms(lambda: run(v), 2 if tag == "L0" else 4))
    print(tag, {v: "%.3f ms %.0f TF/s" % (min(r), flops / min(r) / 1e9) for v, r in res.items()}, flush=True)
    o = {}
    for v in variants:
        run(v); o[v] = out.float().clone()
    for v in variants[1:]:
        print("   max |v%d - v%d| = %.2e" % (variants[0], v, float((o[variants[0]] - o[v]).abs().max())))
    nb = 1 if tag == "L0" else 2
    hh = 1 if tag == "L0" else heads
    sp = lambda t: t[:nb, :, :hh * 64].float().reshape(nb, N, hh, 64).transpose(1, 2)
    q, k, v_ = sp(qkv[..., :C]), sp(qkv[..., C:2 * C]), sp(qkv[..., 2 * C:])
    ref = torch.empty(nb, hh, N, 64, device=dev)
    for s0 in range(0, N, 4096):
        ref[:, :, s0:s0 + 4096] = torch.softmax(q[:, :, s0:s0 + 4096] @ k.transpose(-1, -2) / 8.0, dim=-1) @ v_
    ref = ref.transpose(1, 2).reshape(nb, N, hh * 64)
    for v in variants:
        d = o[v][:nb, :, :hh * 64] - ref
        print("   v%d vs fp32: rel rms %.3e  max abs %.3e" % (v, float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()), float(d.abs().max())))
    del qkv, out, o
