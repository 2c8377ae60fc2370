"""Timing ablations of the phase-interleaved GEMM (bench build: tools/bench/libstar_hip_bench.so, make bench).
usage: ablate_gemm8.py [tiles]   -- interleaved rounds, median ms per tile id"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tiles = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "1,20,31,32,33,34,35").split(",")]
fill = sys.argv[2] if len(sys.argv) > 2 else "randn"   # randn | zeros | small (integers -2..2)
NAMES = {1: "2-stage 256x256 (round 1)", 2: "2-stage 256x320", 20: "gemm8", 31: "gemm8 no DMA in loop", 32: "gemm8 no fragment reads",
         33: "gemm8 no DMA, no reads", 34: "gemm8 MFMA only (no barriers)", 35: "gemm8 no DMA/reads, 1 barrier per phase",
         36: "gemm8 DMA from a hot 64 KB region", 37: "gemm8 no A DMA", 38: "gemm8 no B DMA", 39: "gemm8 without stagger (correct)"}
dt = torch.float16
ctx = L.Context(0, dt, L.Library(os.path.join(ROOT, "tools/bench/libstar_hip_bench.so")))
shapes = [(8192, 8192, 8192), (843264, 2560, 320), (55296, 3840, 1280)]
if len(sys.argv) > 3:
    shapes = [tuple(int(v) for v in sh.split('x')) for sh in sys.argv[3].split(',')]
for (M, N, K) in shapes:
    if fill == "zeros":
        A = torch.zeros(M, K, device="cuda", dtype=dt); W = torch.zeros(N, K, device="cuda", dtype=dt)
    elif fill == "small":
        A = torch.randint(-2, 3, (M, K), device="cuda").to(dt); W = torch.randint(-2, 3, (N, K), device="cuda").to(dt)
    else:
        A = torch.randn(M, K, device="cuda", dtype=dt)
        W = torch.randn(N, K, device="cuda", dtype=dt) * 0.05
    out = torch.empty(M, N, device="cuda", dtype=dt)
    ms = {t: [] for t in tiles}
    for rnd in range(7):
        for t in tiles:
            for _ in range(1 if rnd else 2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ctx.gemm(A, W, out=out, force_tile=t); e1.record(); torch.cuda.synchronize()
            ms[t].append(e0.elapsed_time(e1))
    print(f"# M={M} N={N} K={K} fill={fill}")
    for t in tiles:
        m = statistics.median(ms[t][1:])
        print(f"tile {t:3d} {NAMES.get(t, ''):42s} {m:8.3f} ms  {2.0 * M * N * K / m / 1e9:8.1f} TF/s-equivalent  (min {min(ms[t]):.3f})", flush=True)
    del A, W, out
