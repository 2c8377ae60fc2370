"""Parity fixture on BASELINE config[3]'s OWN latent geometry (TEST INFRASTRUCTURE ONLY; needs /root/reference).

    nice python oracle/make_golden_cfg4.py        # ~90 TFLOP of fp32 on the CPU: 5-20 minutes, < 8 GB

cfg4 = 540x960 -> x4 = 2160x3840, padded by `pad_to_fit` to 2192x3904: latent 274 x 488 = 133 712 tokens per frame, level sizes
274 -> 138 -> 70 -> 36 rows and 488 -> 244 -> 122 -> 61 columns (the stride-2 convs' asymmetric (2,1) pad on the way down, the
nearest-x2 + row crop [1:-1] on the way up; an odd width at level 3), self-attention over 133 712 keys INSIDE a forward.  Until round 5 that geometry was only
checked on the level-0 attention unit (tests/test_fullsize.py) and run, never compared, as a whole forward.

What runs here, in fp32 on the CPU: the REFERENCE's own `ControlledV2VUNet` + `VideoControlNet`
(video_to_video/modules/unet_v2v.py:563-564,709-722,1717-1809, imported by oracle/ref_loader.py) at REDUCED WIDTH
(`SMALL_TEST_CONFIG`: dim 64, 2 heads of 64 -- the same reduced model as tests/golden/unet_small_*.pt, built by
make_golden.build_reference_unet), f = 2 frames, weights `random_state_dict(SMALL_TEST_CONFIG, seed=0)`; one forward.
Inputs are re-derived from the seed by tests/test_parity_cfg4.py (make_golden.unet_inputs); only the output is stored (fp32, 4.3 MB).

Round 6 -- the same geometry at FULL WIDTH:

    nice python oracle/make_golden_cfg4.py full   # 0.47 PFLOP of fp32 on the CPU (78 % of it the level-0 attention over
                                                  # 133 712 keys): hours on 8 cores, < 30 GB

`UNetConfig()` (dim 320, 2.04 B parameters, weights `random_state_dict(UNetConfig(), seed=0)`), f = 2, the same latent: this is
what pins the 320 / 640 / 1280-wide tile choices, the tail splits and the tile-17 convs at the level sizes 274 -> 138 -> 70 -> 36
against the reference (missing #3).  Stored: the output (fp32, 4.3 MB) -> tests/golden/cfg4_full_f2_274x488.pt.
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

GOLD = os.path.join(ROOT, "tests", "golden")

CFG4 = dict(frames=2, latent=(274, 488), seed=4104, wseed=0)
CFG4_FULL = dict(frames=2, latent=(274, 488), seed=4106, wseed=0, width="full")


def main(full=False):
    from make_golden import build_reference_unet, unet_inputs
    from star_amd.geometry import pad_to_fit
    from star_amd.topology import SMALL_TEST_CONFIG, UNetConfig, random_state_dict
    CFG4, SMALL_TEST_CONFIG = (CFG4_FULL, UNetConfig()) if full else (globals()["CFG4"], SMALL_TEST_CONFIG)
    torch.set_grad_enabled(False)
    torch.set_num_threads(int(os.environ.get("STAR_GOLDEN_THREADS", os.cpu_count())))
    # the latent size IS configs[3]'s: 540x960 upscaled x4, padded as VideoToVideo.test() does (video_to_video_model.py:86-87,164-186)
    h, w = 540 * 4, 960 * 4
    pads = pad_to_fit(h, w)
    assert ((h + pads[2] + pads[3]) // 8, (w + pads[0] + pads[1]) // 8) == CFG4["latent"], pads
    t0 = time.time()
    net = build_reference_unet(SMALL_TEST_CONFIG)
    net.load_state_dict(random_state_dict(SMALL_TEST_CONFIG, seed=CFG4["wseed"]), strict=True)
    f, (lh, lw) = CFG4["frames"], CFG4["latent"]
    x, t, y, hint = unet_inputs(SMALL_TEST_CONFIG, f, lh, lw, CFG4["seed"])
    print("model built", time.time() - t0, "t =", int(t), flush=True)
    out = net(x, t=t, y=y, hint=hint)
    print("forward", time.time() - t0, tuple(out.shape), float(out.abs().mean()), flush=True)
    path = os.path.join(GOLD, "cfg4_full_f2_274x488.pt" if full else "cfg4_small_f2_274x488.pt")
    torch.save({"out": out.clone(), "cfg": CFG4, "t": int(t)}, path)
    print("wrote", path)


if __name__ == "__main__":
    main(full=len(sys.argv) > 1 and sys.argv[1] == "full")
