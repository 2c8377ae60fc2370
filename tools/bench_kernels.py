"""Micro-benchmarks of the individual HIP kernels at the cfg2 (32 f, latent 122x216) shapes.
Run on the GPU box:  python tools/bench_kernels.py [--dtype bf16] [--only attn,gemm,...]
Prints achieved TFLOP/s (MFMA-bound kernels) or GB/s (HBM-bound kernels)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_amd import lib as L


def timeit(fn, iters=5, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--tiles", default="")
    ap.add_argument("--lib", default="")
    args = ap.parse_args()
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
    only = set(args.only.split(",")) if args.only else None
    ctx = L.Context(0, dt, L.Library(args.lib) if args.lib else None)
    dev = ctx.torch_device
    res = []

    def rec(name, secs, flops=None, bytes_=None):
        r = {"kernel": name, "ms": secs * 1e3}
        if flops:
            r["TFLOP/s"] = flops / secs / 1e12
        if bytes_:
            r["GB/s"] = bytes_ / secs / 1e9
        res.append(r)
        print(json.dumps(r), flush=True)

    def want(k):
        return only is None or k in only

    F_, H, W = 32, 122, 216
    HW = H * W
    tok = F_ * HW
    if want("attn"):
        for (B, heads, N, tag) in [(4, 5, HW, "L0 N=26352"), (8, 10, 62 * 108, "L1 N=6696"), (32, 20, 32 * 54, "L2 N=1728")]:
            C = heads * 64
            qkv = torch.randn(B, N, 3 * C, device=dev, dtype=dt)
            out = torch.empty(B, N, C, device=dev, dtype=dt)
            fn = lambda: ctx.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, out=out)
            s = timeit(fn, iters=3, warmup=1)
            rec(f"flash_attn self {tag} B={B} h={heads}", s, flops=4.0 * B * heads * N * N * 64)
            del qkv, out
        # cross attention, L0
        B, heads, N = 32, 5, HW
        q = torch.randn(B, N, 320, device=dev, dtype=dt)
        kv = torch.randn(1, 77, 640, device=dev, dtype=dt)
        out = torch.empty(B, N, 320, device=dev, dtype=dt)
        s = timeit(lambda: ctx.attention(q, kv[..., :320], kv[..., 320:], heads, out=out))
        rec("flash_attn cross L0 (Nk=77)", s, flops=4.0 * B * heads * N * 77 * 64, bytes_=2 * q.numel() * 2)
        del q, out
    if want("tattn"):
        heads = 5
        qkv = torch.randn(tok, 960, device=dev, dtype=dt)
        out = torch.empty(tok, 320, device=dev, dtype=dt)
        s = timeit(lambda: ctx.temporal_attention(qkv[:, :320], qkv[:, 320:640], qkv[:, 640:], F_, HW, heads, out=out))
        rec("temporal_attn L0", s, flops=4.0 * HW * heads * 32 * 32 * 64, bytes_=(qkv.numel() + out.numel()) * 2)
        del qkv, out
    if want("gemm"):
        for (M, N, K, tag, geglu) in [
            (tok, 960, 320, "L0 qkv", False), (tok, 320, 320, "L0 proj", False), (tok, 2560, 320, "L0 geglu", True),
            (tok, 320, 1280, "L0 ff-out", False), (tok // 4, 1920, 640, "L1 qkv", False), (tok // 4, 5120, 640, "L1 geglu", True),
            (tok // 16 + 1536, 3840, 1280, "L2 qkv", False), (tok // 16 + 1536, 10240, 1280, "L2 geglu", True),
            (8192, 8192, 8192, "square 8192", False), (4096, 4096, 4096, "square 4096", False), (tok // 16 + 1536, 1280, 5120, "L2 ff-out", False),
            (tok // 4, 640, 2560, "L1 ff-out", False), (tok, 1536, 512, "L0 tt qkv", False),
        ]:
            A = torch.randn(M, K, device=dev, dtype=dt)
            Wt = torch.randn(N, K, device=dev, dtype=dt) * 0.05
            b = torch.randn(N, device=dev)
            out = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=dt)
            for tile in ([0] if not args.tiles else [int(t) for t in args.tiles.split(",")]):
                if geglu and tile in (2, 6):
                    continue
                s = timeit(lambda: ctx.gemm(A, Wt, bias=b, out=out, geglu=geglu, force_tile=tile))
                rec(f"gemm {tag} M={M} N={N} K={K} tile={tile}", s, flops=2.0 * M * N * K, bytes_=(A.numel() + out.numel()) * 2)
            del A, Wt, out
        # the residual-add epilogue (attention out-projection, FF out, proj_out: + x)
        for (M, N, K, tag) in [(tok, 320, 320, "L0 proj+res"), (tok, 320, 1280, "L0 ff-out+res"), (tok // 4, 640, 640, "L1 proj+res"),
                               (tok // 4, 640, 2560, "L1 ff-out+res"), (tok // 16 + 1536, 1280, 1280, "L2 proj+res")]:
            A = torch.randn(M, K, device=dev, dtype=dt)
            Wt = torch.randn(N, K, device=dev, dtype=dt) * 0.05
            b = torch.randn(N, device=dev)
            R = torch.randn(M, N, device=dev, dtype=dt)
            out = torch.empty(M, N, device=dev, dtype=dt)
            for tile in ([0] if not args.tiles else [int(t) for t in args.tiles.split(",")]):
                s = timeit(lambda: ctx.gemm(A, Wt, bias=b, res=R, out=out, force_tile=tile))
                rec(f"gemm {tag} M={M} N={N} K={K} tile={tile}", s, flops=2.0 * M * N * K, bytes_=(A.numel() + 2 * out.numel()) * 2)
            del A, Wt, out, R
    if want("conv"):
        for (NB, Cin, Hh, Ww, Cout, tag) in [(32, 320, 122, 216, 320, "L0 320->320"), (32, 640, 62, 108, 640, "L1 640->640"),
                                             (32, 1280, 32, 54, 1280, "L2 1280->1280"), (32, 2560, 17, 27, 1280, "L3 2560->1280")]:
            x = torch.randn(NB * Hh * Ww, Cin, device=dev, dtype=dt)
            w = torch.randn(Cout, 9 * Cin, device=dev, dtype=dt) * 0.02
            b = torch.randn(Cout, device=dev)
            out = torch.empty(NB * Hh * Ww, Cout, device=dev, dtype=dt)
            for tile in ([0] if not args.tiles else [int(t) for t in args.tiles.split(",")]):
                s = timeit(lambda: ctx.gemm(x, w, bias=b, out=out, mode=L.A_CONV3X3, conv=(NB, Hh, Ww, Cin, Hh, Ww, 1, 1, 1), force_tile=tile), iters=3, warmup=1)
                rec(f"conv3x3 {tag} tile={tile}", s, flops=2.0 * NB * Hh * Ww * Cout * 9 * Cin)
            del x, w, out
        x = torch.randn(tok, 320, device=dev, dtype=dt)
        w = torch.randn(320, 960, device=dev, dtype=dt) * 0.02
        out = torch.empty(tok, 320, device=dev, dtype=dt)
        for tile in ([0] if not args.tiles else [int(t) for t in args.tiles.split(",")]):
            s = timeit(lambda: ctx.gemm(x, w, out=out, res=x, mode=L.A_TCONV3, temporal=(F_, HW, 320), force_tile=tile))
            rec(f"tconv L0 320 tile={tile}", s, flops=2.0 * tok * 320 * 960, bytes_=3 * tok * 320 * 2)
        del x, w, out
        x = torch.randn(tok // 16, 1280, device=dev, dtype=dt)
        w = torch.randn(1280, 3840, device=dev, dtype=dt) * 0.02
        out = torch.empty(tok // 16, 1280, device=dev, dtype=dt)
        for tile in ([0] if not args.tiles else [int(t) for t in args.tiles.split(",")]):
            s = timeit(lambda: ctx.gemm(x, w, out=out, res=x, mode=L.A_TCONV3, temporal=(F_, HW // 16, 1280), force_tile=tile))
            rec(f"tconv L2 1280 tile={tile}", s, flops=2.0 * (tok // 16) * 1280 * 3840)
        del x, w, out
    if want("norm"):
        for C in (320, 1280):
            rows = tok if C == 320 else tok // 16
            x = torch.randn(rows, C, device=dev, dtype=dt)
            g = torch.randn(C, device=dev); b = torch.randn(C, device=dev)
            out = torch.empty_like(x)
            s = timeit(lambda: ctx.group_norm(x, g, b, rows // 32, silu=True, out=out))
            rec(f"group_norm+silu C={C} rows={rows} (stats+apply)", s, bytes_=3 * x.numel() * 2)
            s = timeit(lambda: ctx.layer_norm(x, g, b, out=out))
            rec(f"layer_norm C={C}", s, bytes_=2 * x.numel() * 2)
            if C == 320:
                w7 = torch.randn(98, device=dev)
                maps = torch.empty(rows, 2, device=dev)
                def f():
                    ctx.layer_norm(x, None, None, mode=L.LN_STATS_ONLY, maps=maps)
                    ctx.layer_norm(x, g, b, mode=L.LN_GATE_MAP, gate_w=w7, maps=maps, H=H, W=W, out=out)
                s = timeit(f)
                rec("liem maps + gated layer_norm C=320", s, bytes_=3 * x.numel() * 2)
            del x, out
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
