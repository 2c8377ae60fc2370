"""Time the VAE alone at BASELINE configs[1]'s frame size (976 x 1728 padded pixels, latent 122 x 216): encode of N frames (one per call,
video_to_video_model.py:153-161) and decode of N frames in 3-frame groups (:144-151).  STAR_NO_GNEPI=1 runs every GroupNorm with its own
statistics pass (the round-5 VAE) for a same-box A/B.   python tools/vae_time.py [frames=6]      (measurement tooling)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_amd.vae import AutoencoderKLTemporalDecoder
from star_amd.vae_topology import VaeConfig, random_vae_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
torch.set_grad_enabled(False)
cfg = VaeConfig()
vae = AutoencoderKLTemporalDecoder(cfg, dtype=torch.float16).load_state_dict(random_vae_state_dict(cfg, seed=0))
g = torch.Generator().manual_seed(1)
x = torch.randn(n, 3, 976, 1728, generator=g).clamp(-1, 1).cuda()
z = (torch.randn(n, 4, 122, 216, generator=g) * 0.2).cuda()


def run():
    for i in range(n):
        vae.encode(x[i:i + 1]).latent_dist.parameters
    for i in range(0, n, 3):
        vae.decode(z[i:i + 3], num_frames=min(3, n - i)).sample


run(); torch.cuda.synchronize()
f0 = vae.ctx.lib.gn_fused_count(vae.ctx.h)
t0 = time.perf_counter()
run(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"VAE encode + decode of {n} frames at 976x1728 (STAR_NO_GNEPI={'1' if os.environ.get('STAR_NO_GNEPI') else '0'}): {dt * 1e3:.1f} ms = {dt / n * 1e3:.1f} ms per frame "
      f"(x 32 frames = {dt / n * 32:.2f} s per cfg2 clip); GroupNorms finalized from producer statistics: {vae.ctx.lib.gn_fused_count(vae.ctx.h) - f0}")
