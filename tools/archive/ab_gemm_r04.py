"""Round-4 GEMM A/B on the UNet's own layer shapes (cfg2), product library, interleaved rounds, bit-equality asserted:
  * the persistent one-wave-per-SIMD tile 18 (gemm_p.h) against the 8-wave tiles 1 / 2 and tile 17, per epilogue flavour;
  * the tail split (a poorly filled last round of big tiles -> a second launch of 128 x 128 tiles): automatic choice (0) against
    the forced whole-launch tile.
   python tools/ab_gemm_r04.py [plain|conv]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from star_amd import lib as L
dt = torch.float16
ctx = L.Context(0, dt)
what = sys.argv[1] if len(sys.argv) > 1 else "plain"


def t_ms(fn, iters=6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def run(label, M, N, K, variants, flops):
    res = {k: [] for k in variants}
    ref = None
    for k, fn in variants.items():
        o = fn().clone()
        if ref is None: ref = o
        assert torch.equal(o, ref), (label, k)
    for rnd in range(3):
        for k, fn in variants.items():
            res[k].append(t_ms(fn))
    print(f"{label:34s} " + "  ".join(f"{k} {min(r):.3f} ms {flops / min(r) / 1e9:5.0f} TF/s" for k, r in res.items()), flush=True)


if what == "plain":
    # (M, N, K, flavour): the plain-A layers of levels 1-2 with K >= 640 (profiles/r03_forward_detail_f16_v1.txt)
    shapes = [(214272, 1920, 640, "rowaff"), (55296, 3840, 1280, "rowaff"), (214272, 640, 640, "res"), (55296, 1280, 1280, "res"),
              (214272, 640, 2560, "res"), (55296, 1280, 5120, "res"), (55296, 1280, 1280, "bias"), (214272, 640, 640, "bias"),
              (843264, 320, 1280, "res"), (8192, 8192, 8192, "none"), (9676, 9216, 3072, "bias"), (9676, 3072, 12288, "bias"),
              (214272, 5120, 640, "geglu"), (55296, 10240, 1280, "geglu"), (843264, 4096, 512, "geglu"), (843264, 1536, 512, "rowaff")]
    if len(sys.argv) > 2:
        shapes = [sh for sh in shapes if sh[3] in sys.argv[2].split(",")]
    for (M, N, K, fl) in shapes:
        A = torch.randn(M, K, device="cuda", dtype=dt); W = torch.randn(N, K, device="cuda", dtype=dt) * 0.05
        b = torch.randn(N, device="cuda"); R = torch.randn(M, N, device="cuda", dtype=dt) if fl == "res" else None
        rowab = torch.rand(M, 2, device="cuda") + 0.5; colsum = W.float().sum(1).contiguous()
        out = torch.empty(M, N // 2 if fl == "geglu" else N, device="cuda", dtype=dt)
        kw = {"none": {}, "bias": dict(bias=b), "res": dict(bias=b, res=R), "rowaff": dict(bias=b, rowab=rowab, colsum=colsum),
              "geglu": dict(bias=b, rowab=rowab, colsum=colsum, geglu=True)}[fl]
        v = {}
        for t in ([1, 18, 0] if fl == "geglu" else [1, 2, 17, 18, 0] if N % 320 == 0 else [1, 17, 18, 0]):
            if t == 17 and fl in ("rowaff", "geglu"): continue
            v[f"t{t}"] = (lambda t=t: ctx.gemm(A, W, out=out, force_tile=t, **kw))
        run(f"{M}x{N}x{K} {fl}", M, N, K, v, 2.0 * M * N * K)
        del A, W, R, out
else:
    # gathered layers of level 2 (N = 1280: 4.2 rounds of 256 x 256 tiles, 3.4 of 256 x 320): automatic choice (tail split) against whole launches
    F_, H, Wd = 32, 32, 54
    for (Cin, Cout, mode) in [(1280, 1280, "conv"), (2560, 1280, "conv"), (1280, 1280, "tconv")]:
        M = F_ * H * Wd
        x = torch.randn(M, Cin, device="cuda", dtype=dt)
        b = torch.randn(Cout, device="cuda")
        out = torch.empty(M, Cout, device="cuda", dtype=dt)
        if mode == "conv":
            W = torch.randn(Cout, 9 * Cin, device="cuda", dtype=dt) * 0.02
            v = {f"t{t}": (lambda t=t: ctx.gemm(x, W, bias=b, out=out, mode=L.A_CONV3X3, conv=(F_, H, Wd, Cin, H, Wd, 1, 1, 1), force_tile=t)) for t in (2, 17, 0)}
            run(f"conv3x3 {M}x{Cout}x{9 * Cin}", M, Cout, 9 * Cin, v, 2.0 * M * Cout * 9 * Cin)
        else:
            W = torch.randn(Cout, 3 * Cin, device="cuda", dtype=dt) * 0.02
            v = {f"t{t}": (lambda t=t: ctx.gemm(x, W, bias=b, res=x, out=out, mode=L.A_TCONV3, temporal=(F_, H * Wd, Cin), force_tile=t)) for t in (2, 17, 0)}
            run(f"tconv {M}x{Cout}x{3 * Cin}", M, Cout, 3 * Cin, v, 2.0 * M * Cout * 3 * Cin)
        del x, W, out
print("splits:", ctx.lib.gemm_split_count(ctx.h))
