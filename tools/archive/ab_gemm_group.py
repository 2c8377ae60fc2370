"""A/B of the GEMM tile walk (row-major vs column-major inside groups of tile rows, gemm.h) on the layer shapes of cfg2, product
library, interleaved rounds.   python tools/ab_gemm_group.py [groups: 0,4,8,16]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from star_amd import lib as L
groups = [int(g) for g in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "4", "8", "16"])]
dt = torch.float16
ctx = L.Context(0, dt)
def t_ms(fn, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
shapes = [(8192, 8192, 8192, 0), (214272, 5120, 640, 1), (55296, 10240, 1280, 1), (214272, 1920, 640, 0), (55296, 3840, 1280, 0),
          (55296, 1280, 5120, 0), (214272, 640, 2560, 0), (843264, 4096, 512, 1), (843264, 2560, 320, 1), (843264, 960, 320, 0),
          (843264, 320, 1280, 0), (55296, 1280, 1280, 0), (214272, 640, 640, 0), (14688, 10240, 1280, 1)]
for (M, N, K, geglu) in shapes:
    A = torch.randn(M, K, device="cuda", dtype=dt); W = torch.randn(N, K, device="cuda", dtype=dt) * 0.05
    out = torch.empty(M, N // 2 if geglu else N, device="cuda", dtype=dt)
    res = {g: [] for g in groups}
    ref = None
    for g in groups:
        os.environ["STAR_GEMM_GROUP_M"] = str(g)
        ctx.gemm(A, W, out=out, geglu=bool(geglu))
        o = out.clone()
        if ref is None: ref = o
        assert torch.equal(o, ref), (M, N, K, g)      # the walk must not change a single bit
    for rnd in range(3):
        for g in groups:
            os.environ["STAR_GEMM_GROUP_M"] = str(g)
            res[g].append(t_ms(lambda: ctx.gemm(A, W, out=out, geglu=bool(geglu))))
    fl = 2.0 * M * N * K
    print(f"{M}x{N}x{K}{' geglu' if geglu else ''}: " + "  ".join(f"g{g} {min(r):.3f} ms {fl / min(r) / 1e9:.0f} TF/s" for g, r in res.items()), flush=True)
    del A, W, out
os.environ.pop("STAR_GEMM_GROUP_M", None)
