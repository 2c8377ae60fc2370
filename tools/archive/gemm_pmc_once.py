"""three launches of one GEMM variant (product library force_tile id, or "vendor" = torch.matmul) for rocprofv3 --pmc passes
   usage: gemm_pmc_once.py M N K tile|vendor"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from star_amd import lib as L
M, N, K = (int(x) for x in sys.argv[1:4])
dt = torch.float16
A = torch.randn(M, K, device="cuda", dtype=dt)
W = torch.randn(N, K, device="cuda", dtype=dt) * 0.05
out = torch.empty(M, N, device="cuda", dtype=dt)
if sys.argv[4] == "vendor":
    for _ in range(3): torch.matmul(A, W.t(), out=out)
else:
    ctx = L.Context(0, dt)
    for _ in range(3): ctx.gemm(A, W, out=out, force_tile=int(sys.argv[4]))
torch.cuda.synchronize()
