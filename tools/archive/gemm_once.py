"""Launch a few GEMMs of given shape / tile (for rocprofv3 --pmc passes).  usage: gemm_once.py M N K tile[,tile...] [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from star_amd import lib as L
M, N, K = (int(x) for x in sys.argv[1:4])
tiles = [int(t) for t in sys.argv[4].split(",")]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dt = torch.float16
ctx = L.Context(0, dt, L.Library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench", "libstar_hip_bench.so")))   # bench build: make bench
A = torch.randn(M, K, device="cuda", dtype=dt)
W = torch.randn(N, K, device="cuda", dtype=dt) * 0.05
out = torch.empty(M, N, device="cuda", dtype=dt)
for t in tiles:
    for _ in range(iters):
        ctx.gemm(A, W, out=out, force_tile=t)
torch.cuda.synchronize()
