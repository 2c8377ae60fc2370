"""A/B of GEMM tile ids on the gathered-A layers (3x3 conv, temporal conv) whose width is a multiple of 256, against the auto
choice (tile 0); interleaved rounds, bit-equality asserted.   python tools/ab_conv_tiles.py 0,17"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from star_amd import lib as L
tiles = [int(g) for g in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "17"])]
dt = torch.float16
ctx = L.Context(0, dt)
def t_ms(fn, iters=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
def run(name, flops, call):
    res = {g: [] for g in tiles}; ref = None
    for g in tiles:
        o = call(g).clone()
        if ref is None: ref = o
        assert torch.equal(o, ref), (name, g, (o.float() - ref.float()).abs().max().item())
    for rnd in range(3):
        for g in tiles: res[g].append(t_ms(lambda: call(g)))
    print(f"{name}: " + "  ".join(f"t{g} {min(r):.3f} ms {flops / min(r) / 1e9:.0f} TF/s" for g, r in res.items()), flush=True)
for (NB, Cin, Hh, Ww, Cout, tag, with_res) in [(32, 1280, 32, 54, 1280, "L2 1280->1280", 0), (32, 1280, 32, 54, 1280, "L2 1280->1280 +res", 1), (32, 2560, 32, 54, 1280, "L2 2560->1280", 0),
                                               (32, 1920, 32, 54, 1280, "L2 1920->1280", 0), (32, 2560, 17, 27, 1280, "L3 2560->1280", 0), (32, 1280, 17, 27, 1280, "L3 1280->1280", 0),
                                               (32, 1280, 62, 108, 1280, "L1up 1280->1280", 0)]:
    x = torch.randn(NB * Hh * Ww, Cin, device="cuda", dtype=dt); w = torch.randn(Cout, 9 * Cin, device="cuda", dtype=dt) * 0.02
    b = torch.randn(Cout, device="cuda"); out = torch.empty(NB * Hh * Ww, Cout, device="cuda", dtype=dt)
    r = torch.randn(NB * Hh * Ww, Cout, device="cuda", dtype=dt) if with_res else None
    run(f"conv3x3 {tag}", 2.0 * NB * Hh * Ww * Cout * 9 * Cin,
        lambda g: ctx.gemm(x, w, bias=b, res=r, out=out, mode=L.A_CONV3X3, conv=(NB, Hh, Ww, Cin, Hh, Ww, 1, 1, 1), force_tile=g))
    del x, w, out, r
for (F_, HW, C, tag) in [(32, 32 * 54, 1280, "L2"), (32, 17 * 27, 1280, "L3")]:
    x = torch.randn(F_ * HW, C, device="cuda", dtype=dt); w = torch.randn(C, 3 * C, device="cuda", dtype=dt) * 0.02
    out = torch.empty(F_ * HW, C, device="cuda", dtype=dt)
    run(f"tconv {tag} {C} +res", 2.0 * F_ * HW * C * 3 * C, lambda g: ctx.gemm(x, w, out=out, res=x, mode=L.A_TCONV3, temporal=(F_, HW, C), force_tile=g))
    del x, w, out
for (M, N, K, res) in [(55296, 1280, 5120, 1), (55296, 1280, 1280, 1), (14688, 1280, 5120, 1), (55296, 3840, 1280, 0)]:
    A = torch.randn(M, K, device="cuda", dtype=dt); W = torch.randn(N, K, device="cuda", dtype=dt) * 0.05; b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=dt); r = torch.randn(M, N, device="cuda", dtype=dt) if res else None
    run(f"gemm {M}x{N}x{K}{' +res' if res else ''}", 2.0 * M * N * K, lambda g: ctx.gemm(A, W, bias=b, res=r, out=out, force_tile=g))
    del A, W, out, r
