"""The VAE at BASELINE configs[1]'s frame size (976 x 1728 padded pixels, latent 122 x 216) with per-launch HIP-event timing: encode of
3 frames (one per call, video_to_video_model.py:153-161) and decode of one 3-frame group (:144-151); prints a per-shape table like
tools/profile_forward.py.   python tools/profile_vae.py      (measurement tooling)"""
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_amd import lib as L
from star_amd.vae import AutoencoderKLTemporalDecoder
from star_amd.vae_topology import VaeConfig, random_vae_state_dict

torch.set_grad_enabled(False)
cfg = VaeConfig()
vae = AutoencoderKLTemporalDecoder(cfg, dtype=torch.float16).load_state_dict(random_vae_state_dict(cfg, seed=0))
g = torch.Generator().manual_seed(1)
n = 3
x = torch.randn(n, 3, 976, 1728, generator=g).clamp(-1, 1).cuda()
z = (torch.randn(n, 4, 122, 216, generator=g) * 0.2).cuda()


def run(what):
    if what in ("enc", "both"):
        for i in range(n):
            vae.encode(x[i:i + 1]).latent_dist.parameters
    if what in ("dec", "both"):
        vae.decode(z, num_frames=n).sample


run("both"); torch.cuda.synchronize()
os.makedirs("gpurun_out", exist_ok=True)
for what in ("enc", "dec"):
    path = f"gpurun_out/vae_detail_{what}.csv"
    if os.path.exists(path):
        os.remove(path)
    os.environ["STAR_PROF_DETAIL"] = path
    vae.ctx.profile_begin()
    t0 = time.perf_counter()
    run(what)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    prof = vae.ctx.profile_end()
    print(f"== VAE {what} of {n} frames at 976x1728: wall {wall * 1e3:.1f} ms = {wall / n * 1e3:.1f} ms per frame; sum of kernel ms {sum(v['ms'] for v in prof.values()):.1f}")
    agg = collections.OrderedDict()
    tot_fl = 0.0
    for line in open(path):
        k, d0, d1, d2, d3, ms, fl = line.strip().split(",")
        key = (L.PROF_KINDS[int(k)], int(d0), int(d1), int(d2), int(d3))
        e = agg.setdefault(key, [0, 0.0, 0.0])
        e[0] += 1; e[1] += float(ms); e[2] += float(fl)
        tot_fl += float(fl)
    print(f"   algorithmic {tot_fl / 1e12:.2f} TFLOP = {tot_fl / wall / 1e12:.0f} TFLOP/s over the wall time")
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    print(f"{'kind':14s} {'d0':>8s} {'d1':>6s} {'d2':>6s} {'d3':>6s} {'n':>5s} {'ms':>9s} {'ms/launch':>10s} {'TFLOP/s':>8s}")
    for (kind, d0, d1, d2, d3), (cnt, ms, fl) in rows[:40]:
        print(f"{kind:14s} {d0:8d} {d1:6d} {d2:6d} {d3:6d} {cnt:5d} {ms:9.2f} {ms / cnt:10.3f} {fl / ms / 1e9 if ms else 0:8.1f}")
