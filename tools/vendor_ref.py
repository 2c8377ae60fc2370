"""Reference points from the vendor libraries on the same box, same shapes (NOT used by the product: star links neither hipBLASLt nor a
library attention): torch.matmul (hipBLASLt / rocBLAS) against star's GEMM, torch SDPA (its flash backend) against star's attention.
   python tools/vendor_ref.py"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_amd import lib as L
dt = torch.float16
ctx = L.Context(0, dt)
def t_ms(fn, iters=5, rounds=3):
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best
print("GEMM  C[M,N] = A[M,K] W[N,K]^T, fp16, min of 3x5 launches")
for (M, N, K) in [(8192, 8192, 8192), (16384, 16384, 4096), (55296, 3840, 1280), (55296, 1280, 5120), (214272, 640, 2560), (214272, 1920, 640),
                  (843264, 960, 320), (843264, 320, 1280), (55296, 1280, 11520)]:
    A = torch.randn(M, K, device="cuda", dtype=dt); W = torch.randn(N, K, device="cuda", dtype=dt) * 0.05
    out = torch.empty(M, N, device="cuda", dtype=dt); out2 = torch.empty_like(out)
    ctx.gemm(A, W, out=out); torch.matmul(A, W.t(), out=out2)
    err = (out.float() - out2.float()).abs().max().item()
    a = t_ms(lambda: ctx.gemm(A, W, out=out)); b = t_ms(lambda: torch.matmul(A, W.t(), out=out2))
    fl = 2.0 * M * N * K / 1e9
    print(f"{M}x{N}x{K}: star {a:.3f} ms {fl / a:.0f} TF/s   torch.matmul {b:.3f} ms {fl / b:.0f} TF/s   star/vendor time {a / b:.2f}   max|diff| {err:.3g}", flush=True)
    del A, W, out, out2
print("self-attention, head dim 64, fp16 (batch x heads x tokens)")
for (B, H, N) in [(32, 5, 26352), (32, 10, 6696), (32, 20, 1728)]:
    q, k, v = (torch.randn(B, N, H * 64, device="cuda", dtype=dt) for _ in range(3))
    o = ctx.attention(q, k, v, H).view(B, N, H, 64)
    qt, kt, vt = (x.view(B, N, H, 64).transpose(1, 2) for x in (q, k, v))
    try:
        o2 = F.scaled_dot_product_attention(qt, kt, vt).transpose(1, 2)
        err = (o.float() - o2.float()).abs().max().item()
        b = t_ms(lambda: F.scaled_dot_product_attention(qt, kt, vt), iters=2)
    except Exception as e:
        print("  SDPA failed:", type(e).__name__, str(e)[:200]); b, err = float("nan"), float("nan")
    a = t_ms(lambda: ctx.attention(q, k, v, H), iters=2)
    fl = 4.0 * B * H * N * N * 64 / 1e9
    print(f"{B}x{H}x{N}: star {a:.3f} ms {fl / a:.0f} TF/s   torch SDPA {b:.3f} ms {fl / b:.0f} TF/s   star/vendor time {a / b:.2f}   max|diff| {err:.3g}", flush=True)
    del q, k, v
