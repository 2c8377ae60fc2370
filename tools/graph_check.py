"""hipGraph replay of the UNet forward (star_unet_graph) against the eager path: bit-equality over several timesteps / inputs (single
forward and the CFG pair), then wall time of both at full size.   python tools/graph_check.py [frames h w]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_amd.modules.unet_v2v import ControlledV2VUNet
from star_amd.topology import UNetConfig, random_state_dict
torch.set_grad_enabled(False)
f, h, w = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 122, 216)
cfg = UNetConfig()
net = ControlledV2VUNet(cfg, dtype=torch.float16); net.load_state_dict(random_state_dict(cfg, seed=0)); net.release_host_weights()
g = torch.Generator().manual_seed(1)
def inputs(seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(1, 4, f, h, w, generator=g).cuda(), torch.randn(1, 4, f, h, w, generator=g).cuda() * 0.5,
            torch.randn(1, 77, 1024, generator=g).cuda(), torch.randn(1, 77, 1024, generator=g).cuda())
cases = [(1, 500), (2, 300), (3, 7), (1, 500)]
eager_1, eager_2 = [], []
for seed, t in cases:
    x, hint, yc, yu = inputs(seed)
    eager_1.append(net(x, t=torch.tensor([t]), y=yc, hint=hint).clone())
    eager_2.append(tuple(o.clone() for o in net.forward_cfg_pair(x, torch.tensor([t]), yc, yu, hint=hint)))
torch.cuda.synchronize()
net.use_graph(True)
for rep in range(2):          # first pass: eager warm + capture + replays; second pass: replays only
    for i, (seed, t) in enumerate(cases):
        x, hint, yc, yu = inputs(seed)
        o = net(x, t=torch.tensor([t]), y=yc, hint=hint)
        assert torch.equal(o, eager_1[i]), ("single", rep, i, (o - eager_1[i]).abs().max().item())
        oc, ou = net.forward_cfg_pair(x, torch.tensor([t]), yc, yu, hint=hint)
        assert torch.equal(oc, eager_2[i][0]) and torch.equal(ou, eager_2[i][1]), ("pair", rep, i)
torch.cuda.synchronize()
print("graph replay bit-identical to the eager path: %d single forwards, %d CFG pairs" % (2 * len(cases), 2 * len(cases)), flush=True)
x, hint, yc, yu = inputs(1)
def wall(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3
for rnd in range(2):
    net.use_graph(False); e1 = wall(lambda: net(x, t=torch.tensor([500]), y=yc, hint=hint)); e2 = wall(lambda: net.forward_cfg_pair(x, torch.tensor([500]), yc, yu, hint=hint))
    net.use_graph(True); g1 = wall(lambda: net(x, t=torch.tensor([500]), y=yc, hint=hint)); g2 = wall(lambda: net.forward_cfg_pair(x, torch.tensor([500]), yc, yu, hint=hint))
    print(f"round {rnd}: single forward eager {e1:.1f} ms  graph {g1:.1f} ms | CFG pair eager {e2:.1f} ms  graph {g2:.1f} ms", flush=True)
