"""A/B of the folded-LayerNorm projections of level 0 (K = 320): the tiled kernel of gemm.h (the tile its launcher picks) against the
A-stationary kernel of gemm_as.h (force_tile 30), product library, interleaved rounds."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from star_amd import lib as L
dt = {"f16": torch.float16, "bf16": torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else "f16"]
ctx = L.Context(0, dt, L.Library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench", "libstar_hip_bench.so")) if len(sys.argv) > 2 else None)   # ablations: bench build
def t_ms(fn, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (M, N, geglu) in [(843264, 960, 0), (843264, 2560, 1), (843264, 640, 0), (115200, 960, 0), (115200, 2560, 1)]:
    K = 320
    A = torch.randn(M, K, device="cuda", dtype=dt); W = torch.randn(N, K, device="cuda", dtype=dt) * 0.05
    b = torch.randn(N, device="cuda"); rowab = torch.rand(M, 2, device="cuda") + 0.5; colsum = W.float().sum(1).contiguous()
    out = torch.empty(M, N // 2 if geglu else N, device="cuda", dtype=dt)
    run = {}
    os.environ["STAR_NO_ASTAT"] = "1"
    run["tiled"] = lambda: ctx.gemm(A, W, bias=b, geglu=bool(geglu), rowab=rowab, colsum=colsum, out=out)
    run["astat"] = lambda: ctx.gemm(A, W, bias=b, geglu=bool(geglu), rowab=rowab, colsum=colsum, out=out, force_tile=30)
    if not geglu and len(sys.argv) > 2:
        for k, ft in (("no_epi", 31), ("no_dma", 32), ("no_lds", 33), ("kloop", 34), ("mfma", 35)):
            run[k] = lambda ft=ft: ctx.gemm(A, W, bias=b, rowab=rowab, colsum=colsum, out=out, force_tile=ft)
    o = {}
    for k, f in run.items():
        f(); o[k] = out.float().clone()
    d = (o["tiled"] - o["astat"]).abs().max()
    res = {k: [] for k in run}
    for rnd in range(3):
        for k, f in run.items(): res[k].append(t_ms(f))
    fl = 2.0 * M * N * K
    by = 2.0 * M * (K + (N // 2 if geglu else N))
    print(f"{M}x{N}x{K}{' geglu' if geglu else ''}: " + "  ".join(f"{k} {min(r):.3f} ms {fl / min(r) / 1e9:.0f} TF/s {by / min(r) / 1e9:.2f} TB/s" for k, r in res.items()) + f"   max|diff| {float(d):.2e}", flush=True)
    del A, W, out
