"""What would running the two guidance branches of a CFG pair CONCURRENTLY (two HIP streams) buy?  An upper-bound probe without touching
the executor: two independent UNet + ControlNet instances (own context, own pool, own stream) run one full forward each at cfg2 geometry,
(a) one after the other, (b) at the same time from two host threads (ctypes drops the GIL inside star_unet_forward).  If (b) is not
clearly faster than (a) there is nothing to gain from a two-stream executor: the MFMA kernels saturate every CU's registers, so kernels
of two streams can only overlap in each other's tails.       python tools/concurrent_forward.py [reps=3]      (measurement tooling)"""
import os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_amd.modules.unet_v2v import ControlledV2VUNet
from star_amd.topology import UNetConfig, random_state_dict

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
torch.set_grad_enabled(False)
cfg = UNetConfig()
sd = random_state_dict(cfg, seed=0)
nets = []
for i in range(2):
    n = ControlledV2VUNet(cfg, dtype=torch.float16, device=0)
    n.load_state_dict(sd)
    n.release_host_weights()
    nets.append(n)
del sd
g = torch.Generator().manual_seed(1)
f, h, w = 32, 122, 216
x = torch.randn(1, 4, f, h, w, generator=g).cuda(); hint = (torch.randn(1, 4, f, h, w, generator=g) * 0.5).cuda()
ys = [torch.randn(1, 77, 1024, generator=g).cuda() for _ in range(2)]
t = torch.tensor([500]).cuda()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
outs = [None, None]


def one(i):
    with torch.cuda.stream(streams[i]):
        outs[i] = nets[i](x, t=t, y=ys[i], hint=hint)
        streams[i].synchronize()


for i in range(2):
    one(i)                                  # warm-up: pools sized, LDS attributes set
torch.cuda.synchronize()
ref = [o.clone() for o in outs]
seq, conc = [], []
for _ in range(reps):
    t0 = time.perf_counter(); one(0); one(1); torch.cuda.synchronize(); seq.append(time.perf_counter() - t0)
    th = [threading.Thread(target=one, args=(i,)) for i in range(2)]
    t0 = time.perf_counter()
    for k in th: k.start()
    for k in th: k.join()
    torch.cuda.synchronize(); conc.append(time.perf_counter() - t0)
    assert torch.equal(outs[0], ref[0]) and torch.equal(outs[1], ref[1]), "concurrent forwards must be bit-identical to the sequential ones"
print(f"two full forwards (32 f, 122x216, f16), one after the other: {min(seq) * 1e3:.1f} ms (min of {reps}); on two streams at once: {min(conc) * 1e3:.1f} ms "
      f"({(min(conc) / min(seq) - 1) * 100:+.1f} %); outputs bit-identical")
