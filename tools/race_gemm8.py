"""Race screen for the phase-interleaved GEMM (gemm8.h) on hardware: its LDS-DMA hazards are covered by counted waits and
barrier distances, which a single passing run does not prove.  Every shape is run `reps` times against the bit pattern of
the 2-stage kernel (same MFMA shape, k order and accumulation, so the results must be identical), with an HBM-heavy
elementwise stream running beside it on a second stream for half of the repetitions (uneven memory load)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_amd import lib as L

dt = torch.float16
ctx = L.Context(0, dt, L.Library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench", "libstar_hip_bench.so")))   # bench build: make bench
dev = ctx.torch_device
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
shapes = [(843264 // 8, 960, 320, 0), (843264 // 8, 2560, 320, 1), (214272 // 2, 1920, 640, 0), (55296, 3840, 1280, 0), (8192, 8192, 2048, 0),
          (55296 + 77, 1280, 5120, 2), (4099, 520, 64, 0), (100000, 264, 128, 2)]
side = torch.cuda.Stream()
junk = torch.empty(1 << 28, device=dev, dtype=torch.float16)
bad_total = 0
for (M, N, K, kind) in shapes:
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dt).to(dev)
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(dt).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    R = torch.randn(M, N, generator=g).to(dt).to(dev) if kind == 2 else None
    kw = dict(bias=b, geglu=(kind == 1), res=R)
    ref = ctx.gemm(A, W, force_tile=1, **kw)
    bad = 0
    for r in range(reps):
        if r % 2:
            with torch.cuda.stream(side):
                junk.mul_(1.0001)
        out = ctx.gemm(A, W, force_tile=20, **kw)
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1
            d = (out.float() - ref.float()).abs()
            print(f"  MISMATCH rep {r}: {int((d > 0).sum())} elements, max {float(d.max()):.4g}", flush=True)
    print(f"race M={M} N={N} K={K} kind={kind}: {bad}/{reps} mismatching runs", flush=True)
    bad_total += bad
    del A, W, ref
print("RACE_SCREEN", "FAIL" if bad_total else "PASS", bad_total)
