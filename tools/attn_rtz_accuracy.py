"""Accuracy of the spatial self-attention with the probabilities packed round-toward-zero (bench variant 35, attn5.h RTZ) beside the
nearest-even product kernel (variant 9): both against fp64 softmax(QK^T/8)V on the SAME f16 operands, at the level-0 length of cfg2
(N = 26352), on N(0, 1.5) operands and on the peaked trained-like logits of tests/test_fullsize.py.
  python tools/attn_rtz_accuracy.py [variant ...]        (bench build: tools/bench/libstar_hip_bench.so)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from star_amd import lib as L  # noqa: E402


def reference(qkv):
    q, k, v = (qkv[:, i * 64:(i + 1) * 64].double() for i in range(3))
    ref = torch.empty(q.shape[0], 64, dtype=torch.float64, device=q.device)
    for s in range(0, q.shape[0], 2048):
        ref[s:s + 2048] = torch.softmax(q[s:s + 2048] @ k.T / 8.0, dim=-1) @ v
    return ref


def main():
    variants = [int(a) for a in sys.argv[1:]] or [9, 35]
    lib = L.Library(os.path.join(ROOT, "tools", "bench", "libstar_hip_bench.so"))
    ctx = L.Context(0, torch.float16, lib)
    dev = ctx.torch_device
    g = torch.Generator().manual_seed(3)
    H, W = 122, 216
    N = H * W
    cases = {"N(0, 1.5)": (torch.randn(N, 192, generator=g) * 1.5)}
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    pos = torch.stack([torch.sin(yy / 9.0), torch.cos(yy / 9.0), torch.sin(xx / 13.0), torch.cos(xx / 13.0)], dim=-1).reshape(N, 4)
    basis = torch.randn(4, 64, generator=g)
    q = pos @ basis * 2.5 + torch.randn(N, 64, generator=g) * 0.7
    k = pos @ basis * 2.5 + torch.randn(N, 64, generator=g) * 0.7
    k[N - 3000:] *= 1.6
    cases["peaked"] = torch.cat([q, k, torch.randn(N, 64, generator=g)], dim=-1)
    for name, x in cases.items():
        qkv = x.to(torch.float16).to(dev)
        ref = reference(qkv)
        rr = float(ref.pow(2).mean().sqrt())
        outs = {}
        for v in variants:
            o = ctx.attention(qkv[None, :, :64], qkv[None, :, 64:128], qkv[None, :, 128:], 1, variant=v)[0]
            outs[v] = o
            d = o.double() - ref
            # the f16 rounding of the OUTPUT alone (the floor any kernel sits on)
            floor = float((ref.to(torch.float16).double() - ref).pow(2).mean().sqrt())
            print(f"{name:10s} variant {v:2d}: max|d| {float(d.abs().max()):.3e}  rms {float(d.pow(2).mean().sqrt()):.3e}  "
                  f"mean {float(d.mean()):+.2e}  (ref rms {rr:.3e}, output-rounding floor {floor:.3e})")
        if len(variants) > 1:
            a, b = outs[variants[0]], outs[variants[1]]
            print(f"{name:10s} variant {variants[1]} vs {variants[0]}: {int((a != b).sum())} of {a.numel()} outputs differ, max|d| {float((a.float() - b.float()).abs().max()):.3e}")
    ctx.sync()
    ctx.close()


if __name__ == "__main__":
    main()
