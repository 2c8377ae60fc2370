"""A/B of the flash-attention kernel variants at the cfg2 shapes (interleaved rounds, same process)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_amd import lib as L
variants = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "1"])]
dt = {"f16": torch.float16, "bf16": torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else "f16"]
ctx = L.Context(0, dt, L.Library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench", "libstar_hip_bench.so")))   # bench build: make bench
dev = ctx.torch_device
def t_ms(fn, iters=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (B, heads, N, tag) in [(8, 5, 26352, "L0"), (16, 10, 6696, "L1"), (32, 20, 1728, "L2")]:
    C = heads * 64
    fill = sys.argv[3] if len(sys.argv) > 3 else "randn"
    qkv = torch.zeros(B, N, 3 * C, device=dev, dtype=dt) if fill == "zeros" else torch.randn(B, N, 3 * C, device=dev, dtype=dt) * (float(fill) if fill not in ("randn", "zeros") else 1.0)
    out = torch.empty(B, N, C, device=dev, dtype=dt)
    flops = 4.0 * B * heads * N * N * 64
    res = {v: [] for v in variants}
    for v in variants:
        ctx.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, out=out, variant=v)
    for rnd in range(3):
        for v in variants:
            res[v].append(t_ms(lambda: ctx.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, out=out, variant=v)))
    print(tag, fill, {v: "%.3f ms %.0f TF/s" % (min(r), flops / min(r) / 1e9) for v, r in res.items()}, flush=True)
    if len(variants) > 1:
        o = {}
        for v in variants:
            ctx.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, out=out, variant=v)
            o[v] = out.float().clone()
        for v in variants[1:]:
            print("   max |v%d - v%d| = %.2e" % (variants[0], v, float((o[variants[0]] - o[v]).abs().max())))
        if tag != "L0":   # error against an fp32 softmax(QK^T)V of the same 16-bit inputs (first frame)
            sp = lambda t: t[:1].float().reshape(1, N, heads, 64).transpose(1, 2)
            ref = torch.nn.functional.scaled_dot_product_attention(sp(qkv[..., :C]), sp(qkv[..., C:2 * C]), sp(qkv[..., 2 * C:])).transpose(1, 2).reshape(1, N, C)
            for v in variants:
                d = o[v][:1] - ref
                print("   v%d vs fp32: rel rms %.3e  max abs %.3e" % (v, float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()), float(d.abs().max())))
    del qkv, out
