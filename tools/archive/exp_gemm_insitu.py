"""Experiment: why are the K=320 GEMMs 3x slower inside the forward than in the micro-benchmark?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from star_amd import lib as L
dt = torch.float16
ctx = L.Context(0, dt)
dev = ctx.torch_device
M, N, K = 843264, 960, 320

def ev_time(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

A = torch.randn(M, K, device=dev, dtype=dt)
W = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
C = torch.empty(M, N, device=dev, dtype=dt)
print("1 baseline randn A:            %.3f ms" % ev_time(lambda: ctx.gemm(A, W, out=C)))
g = torch.randn(K, device=dev); b = torch.randn(K, device=dev)
A2 = ctx.layer_norm(A, g, b)
print("2 A = layer_norm output:       %.3f ms" % ev_time(lambda: ctx.gemm(A2, W, out=C)))
A3 = torch.zeros_like(A)
print("3 A = zeros:                   %.3f ms" % ev_time(lambda: ctx.gemm(A3, W, out=C)))
A4 = (torch.randn(M, K, device=dev) * 20).to(dt)
print("4 A = randn*20:                %.3f ms" % ev_time(lambda: ctx.gemm(A4, W, out=C)))
A5 = (torch.randn(M, K, device=dev) * 1e-6).to(dt)   # fp16 denormals
print("5 A = denormal-range:          %.3f ms" % ev_time(lambda: ctx.gemm(A5, W, out=C)))
# 6: producer->consumer: LN writes A each iteration, then gemm (cache / dirty-line effects)
def f6():
    ctx.layer_norm(A, g, b, out=A2)
    ctx.gemm(A2, W, out=C)
t_ln = ev_time(lambda: ctx.layer_norm(A, g, b, out=A2))
print("6 LN -> gemm chain (gemm part): %.3f ms  (LN alone %.3f)" % (ev_time(f6) - t_ln, t_ln))
# 7: heavy attention before the gemm
qkv = torch.randn(4, 26352, 960, device=dev, dtype=dt)
ao = torch.empty(4, 26352, 320, device=dev, dtype=dt)
def att(): ctx.attention(qkv[..., :320], qkv[..., 320:640], qkv[..., 640:], 5, out=ao)
t_att = ev_time(att, iters=3, warm=1)
def f7():
    att(); ctx.gemm(A, W, out=C)
print("7 attn -> gemm (gemm part):    %.3f ms  (attn alone %.3f)" % (ev_time(f7, iters=3, warm=1) - t_att, t_att))
# 8: per-launch event profiling as in the forward
ctx.profile_begin()
for _ in range(5): ctx.gemm(A, W, out=C)
p = ctx.profile_end()
print("8 per-launch events:           %.3f ms" % (p["gemm"]["ms"] / 5))
# 9: C written as a slice of a wider buffer / residual read
R = torch.randn(M, 320, device=dev, dtype=dt); W2 = (torch.randn(320, 320, device=dev) / 18).to(dt); C2 = torch.empty(M, 320, device=dev, dtype=dt); bias = torch.randn(320, device=dev)
print("9 320x320 +bias+res:           %.3f ms" % ev_time(lambda: ctx.gemm(A, W2, bias=bias, res=R, out=C2)))
print("9b 320x320 no res:             %.3f ms" % ev_time(lambda: ctx.gemm(A, W2, bias=bias, out=C2)))
# 10: many live allocations of odd sizes (fragmentation / TLB)
junk = [torch.empty(int(3e6 + 1e6 * i), device=dev, dtype=dt) for i in range(200)]
A6 = torch.randn(M, K, device=dev, dtype=dt); C6 = torch.empty(M, N, device=dev, dtype=dt)
print("10 after 200 odd allocations:  %.3f ms" % ev_time(lambda: ctx.gemm(A6, W, out=C6)))
# 11: after a full-size weights footprint (4 GB of other data touched)
big = torch.randn(2 * 1024 ** 3, device=dev, dtype=dt)
s = big.sum()
print("11 after touching 4 GB:        %.3f ms" % ev_time(lambda: ctx.gemm(A, W, out=C)))
