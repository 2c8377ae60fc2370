"""Interleaved A/B of two builds of the C-ABI library on the GEMM / implicit-conv layer shapes of cfg2 (same process, same
buffers, alternating rounds; min over rounds).  usage: python tools/ab_libs.py <libA.so> <libB.so> [f16|bf16] [gemm,conv,res]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_amd import lib as L

dt = {"f16": torch.float16, "bf16": torch.bfloat16}[sys.argv[3] if len(sys.argv) > 3 else "f16"]
what = set((sys.argv[4] if len(sys.argv) > 4 else "gemm,conv,res").split(","))
ctxs = [L.Context(0, dt, L.Library(os.path.abspath(p))) for p in sys.argv[1:3]]
PERSIST_B = int(os.environ.get("AB_PERSIST_B", "0"))   # build B walks its tiles with 256 * PERSIST_B persistent workgroups


def auto_tile(M, N, K, geglu=False, plain=True):   # gemm_impl.h: launch_gemm
    if M <= 4096 and N <= 1024: return 3
    if not geglu and N % 320 == 0: return 2
    if geglu and K <= 320 and plain: return 9
    if N <= 128: return 4
    return 1


def ftile(M, N, K, geglu=False, plain=True):
    return [0, 100 * PERSIST_B + auto_tile(M, N, K, geglu, plain) if PERSIST_B else 0]
dev = ctxs[0].torch_device


def t_ms(fn, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def ab(tag, flops, fns):
    for f in fns:
        f()
    res = [[], []]
    for rnd in range(4):
        for i, f in enumerate(fns):
            res[i].append(t_ms(f))
    a, b = min(res[0]), min(res[1])
    print("%-44s A %.3f ms %6.0f TF/s | B %.3f ms %6.0f TF/s | B/A speed %.3f" % (tag, a, flops / a / 1e9, b, flops / b / 1e9, a / b), flush=True)


F_, H, W = 32, 122, 216
tok = F_ * H * W
if "gemm" in what:
    for (M, N, K, tag, geglu) in [
        (tok, 960, 320, "L0 qkv", False), (tok, 320, 320, "L0 proj", False), (tok, 2560, 320, "L0 geglu", True),
        (tok, 320, 1280, "L0 ff-out", False), (tok // 4, 1920, 640, "L1 qkv", False), (tok // 4, 5120, 640, "L1 geglu", True),
        (tok // 16 + 1536, 3840, 1280, "L2 qkv", False), (tok // 16 + 1536, 10240, 1280, "L2 geglu", True),
        (8192, 8192, 8192, "square 8192", False), (tok // 16 + 1536, 1280, 5120, "L2 ff-out", False),
        (tok // 4, 640, 2560, "L1 ff-out", False), (tok, 1536, 512, "L0 tt qkv", False), (tok, 4096, 512, "L0 tt geglu", True),
    ]:
        A = torch.randn(M, K, device=dev, dtype=dt)
        Wt = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
        b = torch.randn(N, device=dev)
        outs = [torch.empty(M, N // 2 if geglu else N, device=dev, dtype=dt) for _ in ctxs]
        ab(f"gemm {tag} {M}x{N}x{K}", 2.0 * M * N * K,
           [lambda c=c, o=o, t=t: c.gemm(A, Wt, bias=b, out=o, geglu=geglu, force_tile=t) for c, o, t in zip(ctxs, outs, ftile(M, N, K, geglu))])
        if not torch.equal(outs[0], outs[1]): print("   results differ: max |A-B| = %.3e" % float((outs[0].float() - outs[1].float()).abs().max()))
        del A, Wt, outs
if "res" in what:
    for (M, N, K, tag) in [(tok, 320, 320, "L0 proj+res"), (tok, 320, 1280, "L0 ff-out+res"), (tok // 4, 640, 640, "L1 proj+res"),
                           (tok // 16 + 1536, 1280, 1280, "L2 proj+res")]:
        A = torch.randn(M, K, device=dev, dtype=dt)
        Wt = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
        b = torch.randn(N, device=dev)
        R = torch.randn(M, N, device=dev, dtype=dt)
        outs = [torch.empty(M, N, device=dev, dtype=dt) for _ in ctxs]
        ab(f"gemm {tag} {M}x{N}x{K}", 2.0 * M * N * K,
           [lambda c=c, o=o, t=t: c.gemm(A, Wt, bias=b, res=R, out=o, force_tile=t) for c, o, t in zip(ctxs, outs, ftile(M, N, K))])
        if not torch.equal(outs[0], outs[1]): print("   results differ: max |A-B| = %.3e" % float((outs[0].float() - outs[1].float()).abs().max()))
        del A, Wt, outs, R
if "conv" in what:
    for (NB, Cin, Hh, Ww, Cout, tag) in [(32, 320, 122, 216, 320, "L0 320->320"), (32, 640, 62, 108, 640, "L1 640->640"),
                                         (32, 1280, 32, 54, 1280, "L2 1280->1280"), (32, 2560, 17, 27, 1280, "L3 2560->1280")]:
        x = torch.randn(NB * Hh * Ww, Cin, device=dev, dtype=dt)
        w = (torch.randn(Cout, 9 * Cin, device=dev) / (9 * Cin) ** 0.5).to(dt)
        b = torch.randn(Cout, device=dev)
        outs = [torch.empty(NB * Hh * Ww, Cout, device=dev, dtype=dt) for _ in ctxs]
        ab(f"conv3x3 {tag}", 2.0 * NB * Hh * Ww * Cout * 9 * Cin,
           [lambda c=c, o=o, t=t: c.gemm(x, w, bias=b, out=o, mode=L.A_CONV3X3, conv=(NB, Hh, Ww, Cin, Hh, Ww, 1, 1, 1), force_tile=t) for c, o, t in zip(ctxs, outs, ftile(NB * Hh * Ww, Cout, 9 * Cin, plain=False))])
        if not torch.equal(outs[0], outs[1]): print("   results differ: max |A-B| = %.3e" % float((outs[0].float() - outs[1].float()).abs().max()))
        del x, w, outs
    for (C, hw, tag) in [(320, H * W, "L0 320"), (1280, (H * W) // 16, "L2 1280")]:
        rows = F_ * hw
        x = torch.randn(rows, C, device=dev, dtype=dt)
        w = (torch.randn(C, 3 * C, device=dev) / (3 * C) ** 0.5).to(dt)
        outs = [torch.empty(rows, C, device=dev, dtype=dt) for _ in ctxs]
        ab(f"tconv {tag}", 2.0 * rows * C * 3 * C,
           [lambda c=c, o=o, t=t: c.gemm(x, w, out=o, res=x, mode=L.A_TCONV3, temporal=(F_, hw, C), force_tile=t) for c, o, t in zip(ctxs, outs, ftile(rows, C, 3 * C, plain=False))])
        if not torch.equal(outs[0], outs[1]): print("   results differ: max |A-B| = %.3e" % float((outs[0].float() - outs[1].float()).abs().max()))
        del x, w, outs
