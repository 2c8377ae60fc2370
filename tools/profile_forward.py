"""One full-size UNet+ControlNet forward (cfg2: 32 f, latent 122x216) with per-launch HIP-event timing.
   python tools/profile_forward.py [--dtype f16] [--frames 32] [--h 122] [--w 216]
Writes gpurun_out/forward_detail.csv and prints a per-shape table."""
import argparse
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_amd import lib as L
from star_amd.modules.unet_v2v import ControlledV2VUNet
from star_amd.topology import UNetConfig, random_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f16")
ap.add_argument("--frames", type=int, default=32)
ap.add_argument("--h", type=int, default=122)
ap.add_argument("--w", type=int, default=216)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--lib", default=None, help="another build of the library (tools/bench/libstar_hip_bench.so for in-situ A/Bs of bench-only switches)")
a = ap.parse_args()
torch.set_grad_enabled(False)
dt = {"f16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
cfg = UNetConfig()
net = ControlledV2VUNet(cfg, dtype=dt, library=L.Library(a.lib) if a.lib else None)
net.load_state_dict(random_state_dict(cfg, seed=0))
net.release_host_weights()
g = torch.Generator().manual_seed(1)
x = torch.randn(1, 4, a.frames, a.h, a.w, generator=g).cuda()
hint = torch.randn(1, 4, a.frames, a.h, a.w, generator=g).cuda() * 0.5
y = torch.randn(1, 77, 1024, generator=g).cuda()
t = torch.tensor([500])
net(x, t=t, y=y, hint=hint)   # warm the pool
torch.cuda.synchronize()
os.makedirs("gpurun_out", exist_ok=True)
path = "gpurun_out/forward_detail.csv"
if os.path.exists(path):
    os.remove(path)
os.environ["STAR_PROF_DETAIL"] = path
net.ctx.profile_begin()
t0 = time.perf_counter()
for _ in range(a.reps):
    net(x, t=t, y=y, hint=hint)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / a.reps
prof = net.ctx.profile_end()
y2 = torch.randn(1, 77, 1024, generator=g).cuda()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    net.forward_cfg_pair(x, t, y, y2, hint=hint)
torch.cuda.synchronize()
print(f"CFG pair (2 branches, shared prefix) wall {(time.perf_counter() - t0) / a.reps * 1e3:.1f} ms vs 2 x single {2 * wall * 1e3:.1f} ms")
print(f"forward wall {wall * 1e3:.1f} ms; sum of kernel ms {sum(v['ms'] for v in prof.values()) / a.reps:.1f}")
agg = collections.OrderedDict()
for line in open(path):
    k, d0, d1, d2, d3, ms, fl = line.strip().split(",")
    key = (L.PROF_KINDS[int(k)], int(d0), int(d1), int(d2), int(d3))
    e = agg.setdefault(key, [0, 0.0, 0.0])
    e[0] += 1; e[1] += float(ms); e[2] += float(fl)
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print(f"{'kind':14s} {'d0':>8s} {'d1':>6s} {'d2':>6s} {'d3':>6s} {'n':>5s} {'ms/fwd':>9s} {'ms/launch':>10s} {'TFLOP/s':>8s}")
for (kind, d0, d1, d2, d3), (n, ms, fl) in rows[:60]:
    print(f"{kind:14s} {d0:8d} {d1:6d} {d2:6d} {d3:6d} {n // a.reps:5d} {ms / a.reps:9.2f} {ms / n:10.3f} {fl / ms / 1e9 if ms else 0:8.1f}")
