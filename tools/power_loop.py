"""Run one workload in a loop for a few seconds while sampling socket power / shader clock of GPU 0 from sysfs hwmon twice a
second (never run rocm-smi beside a kernel on this pool: it faulted every run, profiles/r03_attn7_ab.txt).
   python tools/power_loop.py gemm M N K [tile | vendor]     |  attn VARIANT  |  forward [frames h w]"""
import glob, os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_amd import lib as L

def hwmon0():
    c = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"), key=lambda p: int(p.split("/card")[1].split("/")[0]))
    return c[0] if c else None

samples, stop = [], False
def sampler():
    h = hwmon0()
    while not stop and h:
        try:
            p = int(open(h + "/power1_average").read()) / 1e6
        except Exception:
            p = int(open(h + "/power1_input").read()) / 1e6
        f = int(open(h + "/freq1_input").read()) / 1e6
        samples.append((p, f)); time.sleep(0.25)

kind = sys.argv[1]
dt = torch.float16
torch.set_grad_enabled(False)
if kind == "gemm":
    M, N, K = (int(x) for x in sys.argv[2:5])
    ctx = L.Context(0, dt)
    A = torch.randn(M, K, device="cuda", dtype=dt); W = torch.randn(N, K, device="cuda", dtype=dt) * 0.05
    out = torch.empty(M, N, device="cuda", dtype=dt)
    tile = sys.argv[5] if len(sys.argv) > 5 else "0"
    fn = (lambda: torch.matmul(A, W.t(), out=out)) if tile == "vendor" else (lambda: ctx.gemm(A, W, out=out, force_tile=int(tile)))
    work = 2.0 * M * N * K
elif kind == "attn":
    ctx = L.Context(0, dt)
    B, heads, Nn = 32, 5, 26352; C = 320
    qkv = torch.randn(B, Nn, 3 * C, device="cuda", dtype=dt); out = torch.empty(B, Nn, C, device="cuda", dtype=dt)
    v = int(sys.argv[2])
    fn = lambda: ctx.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, out=out, variant=v)
    work = 4.0 * B * heads * Nn * Nn * 64
else:
    from star_amd.modules.unet_v2v import ControlledV2VUNet
    from star_amd.topology import UNetConfig, random_state_dict
    f, h, w = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (32, 122, 216)
    cfg = UNetConfig(); net = ControlledV2VUNet(cfg, dtype=dt); net.load_state_dict(random_state_dict(cfg, seed=0)); net.release_host_weights()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 4, f, h, w, generator=g).cuda(); hint = torch.randn(1, 4, f, h, w, generator=g).cuda() * 0.5
    y = torch.randn(1, 77, 1024, generator=g).cuda(); t = torch.tensor([500])
    fn = lambda: net(x, t=t, y=y, hint=hint)
    work = 572.3e12 * (f / 32.0) * (h * w) / (122 * 216.0)
fn(); torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); n = 0
while time.time() - t0 < 4.0:
    for _ in range(1 if kind == "forward" else 20): fn()
    torch.cuda.synchronize(); n += 1 if kind == "forward" else 20
el = time.time() - t0
stop = True; th.join()
ss = samples[4:] or samples
print(f"{' '.join(sys.argv[1:])}: {el / n * 1e3:.3f} ms, {work / (el / n) / 1e12:.0f} TF/s | power {sum(p for p, _ in ss) / len(ss):.0f} W, sclk {sum(f for _, f in ss) / len(ss):.0f} MHz ({len(ss)} samples)", flush=True)
