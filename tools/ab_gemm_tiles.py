"""A/B of GEMM tile ids (ops.h force_tile) on long-K shapes, interleaved rounds, bit-equality asserted; torch.matmul (vendor library) beside
them as a reference point.   python tools/ab_gemm_tiles.py 1,14 [bench]   (2nd argument: bench library, which holds the experimental tiles)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_amd import lib as L
tiles = [int(g) for g in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "14"])]
dt = torch.float16
ABL = {20, 21, 22, 23}   # timing ablations: results are garbage by construction
ctx = L.Context(0, dt, L.Library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench", "libstar_hip_bench.so")) if len(sys.argv) > 2 and sys.argv[2] == "bench" else None)
def t_ms(fn, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
only_plain = any(g in ABL for g in tiles)
shapes = [(8192, 8192, 8192, 0), (8192, 8192, 8256, 0), (8192, 8192, 8320, 0), (8192, 8192, 8448, 0), (55296, 1280, 5120, 0), (55296, 1280, 5184, 0)] if len(sys.argv) > 3 else [(8192, 8192, 8192, 0), (16384, 16384, 4096, 0), (55296, 10240, 1280, 1), (55296, 3840, 1280, 0), (55296, 1280, 5120, 0),
          (214272, 640, 2560, 0), (55296, 1280, 1280, 0), (214272, 5120, 640, 1), (55296, 1280, 11520, 0)]
for (M, N, K, geglu) in shapes:
    if geglu and only_plain: continue
    A = torch.randn(M, K, device="cuda", dtype=dt); W = torch.randn(N, K, device="cuda", dtype=dt) * 0.05
    out = torch.empty(M, N // 2 if geglu else N, device="cuda", dtype=dt)
    res = {g: [] for g in tiles}; res["vendor"] = []
    ref = None
    for g in tiles:
        ctx.gemm(A, W, out=out, geglu=bool(geglu), force_tile=g)
        o = out.clone()
        if ref is None: ref = o
        assert g in ABL or torch.equal(o, ref), (M, N, K, g)
    for rnd in range(3):
        for g in tiles:
            res[g].append(t_ms(lambda: ctx.gemm(A, W, out=out, geglu=bool(geglu), force_tile=g)))
        if not geglu:
            res["vendor"].append(t_ms(lambda: torch.matmul(A, W.t())))
    fl = 2.0 * M * N * K
    print(f"{M}x{N}x{K}{' geglu' if geglu else ''}: " + "  ".join(f"{'t' + str(g) if g != 'vendor' else g} {min(r):.3f} ms {fl / min(r) / 1e9:.0f} TF/s" for g, r in res.items() if r), flush=True)
    del A, W, out
