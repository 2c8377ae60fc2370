"""one torch.matmul per shape, for rocprofv3 --kernel-trace: which vendor kernel (name, registers, LDS, grid) serves the shape"""
import sys, torch
dt = torch.float16
for (M, N, K) in [(8192, 8192, 8192), (55296, 3840, 1280), (55296, 1280, 11520), (843264, 960, 320)]:
    A = torch.randn(M, K, device="cuda", dtype=dt); W = torch.randn(N, K, device="cuda", dtype=dt) * 0.05
    for _ in range(3): C = torch.matmul(A, W.t())
    torch.cuda.synchronize()
    del A, W, C
