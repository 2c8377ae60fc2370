import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
from make_golden import unet_inputs
from star_amd.modules.unet_v2v import ControlledV2VUNet
from star_amd.topology import SMALL_TEST_CONFIG, random_state_dict
torch.set_grad_enabled(False)
net = ControlledV2VUNet(SMALL_TEST_CONFIG, dtype=torch.float16); net.load_state_dict(random_state_dict(SMALL_TEST_CONFIG, seed=0))
x, t, y, hint = unet_inputs(SMALL_TEST_CONFIG, 5, 10, 8, 11)
outs = [net(x.cuda(), t=t, y=y.cuda(), hint=hint.cuda()).clone() for _ in range(4)]
rr = lambda a, b: float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())
print("run-to-run rel rms (single forward, same inputs):", [rr(outs[i], outs[0]) for i in range(1, 4)])
pa, pb = net.forward_cfg_pair(x.cuda(), t, y.cuda(), y.cuda(), hint=hint.cuda())
print("pair(cond==uncond) branch0 vs branch1:", rr(pa, pb), " pair vs single:", rr(pa, outs[0]))
