"""Ablation probes of the 256x256 GEMM main loop (results are wrong by construction for tiles 11-13).
   python tools/ablate_gemm.py [M N K]      default 8192^3; e.g. 843264 2560 320 for the L0 feed-forward shape"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_amd import lib as L
dt = torch.float16
ctx = L.Context(0, dt, L.Library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench", "libstar_hip_bench.so")))   # bench build: make bench
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8192, 8192, 8192)
A = torch.randn(M, K, device="cuda", dtype=dt); W = torch.randn(N, K, device="cuda", dtype=dt) * 0.01
b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=dt)
outg = torch.empty(M, N // 2, device="cuda", dtype=dt)
names = {1: "full kernel", 11: "no LDS fragment reads (constant frags)", 12: "no MFMA (reads kept live)",
         13: "no re-staging after tile 0 (no LDS-DMA in the loop)", 15: "no K loop at all (launch + epilogue only)",
         16: "no global stores (everything else kept)", -1: "full kernel, GEGLU epilogue (half the output)"}
def run(tile):
    if tile == -1:
        ctx.gemm(A, W, bias=b, out=outg, geglu=True, force_tile=1)
    else:
        ctx.gemm(A, W, bias=b, out=out, force_tile=tile)
for rnd in range(2):
    for tile, nm in names.items():
        for _ in range(2): run(tile)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run(tile)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"tile {tile:2d} {nm:55s} {ms:7.3f} ms  ({2.0 * M * N * K / ms / 1e9:7.1f} TF/s-equivalent)", flush=True)
