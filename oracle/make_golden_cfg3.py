"""Parity fixture for BASELINE config[2]'s per-step CHUNK LOOP at its own geometry (TEST INFRASTRUCTURE ONLY; needs /root/reference).

    STAR_GOLDEN_THREADS=6 nice python oracle/make_golden_cfg3.py      # 4.5 PFLOP of fp32 on the CPU: ~4 h on 6 cores, ~25 GB of RAM

cfg3 = 72 frames 240x426 -> x4 (latent 122x216), `--max_chunk_len 16`: the reference's `make_chunks(72, 0, 16)`
(video_to_video/video_to_video_model.py:188-210) gives 8 overlapping 16-frame chunks (0,16), (8,24) ... (56,72) and
`GaussianDiffusion.sample_sr`'s `model_chunk_fn` (video_to_video/diffusion/diffusion_sdedit.py:330-353) denoises each chunk with its
own `hint_chunk` slice, trims the overlaps (first chunk keeps [:12], inner chunks [4:12], last chunk [4:]) and concatenates.  Until
round 5 that loop had been compared with the reference at 11 frames / latent 90x160 only.

What runs here, in fp32 on the CPU: ONE solver step (steps = 1: sigmas [sigma_899, 0], i.e. x0 = model_chunk_fn(x c_in, sigma_899)) of the
REFERENCE's own `sample_sr` (timestep ladder, sigma ladder, `model_chunk_fn`, `denoise`) around the REFERENCE's own `ControlledV2VUNet`
(unet_v2v.py:1717-1809) at full width, weights `random_state_dict(UNetConfig(), seed=0)`, CFG 7.5 with the 0.2 rescale: 8 chunks x 2
forwards of 16 frames.  The reference's `sample_dpmpp_2m_sde` itself cannot run a one-step ladder -- after the `sigmas[i + 1] == 0` branch
it executes `h_last = h` with `h` unassigned (solvers_sdedit.py:173-175,197-198: UnboundLocalError; found the hard way, at the end of the
first 2.8-hour run) -- so its single step is restated in `one_step_solver` below (three lines: x = noise sigma_0, c_in from
`get_scalings`, x0 = model(x c_in, sigma_0)) and installed in place of it; everything that this fixture is about (the chunk loop) is the
reference's code.  Every forward's output is also cached under /tmp so that a failure behind the forwards does not cost them again.
Inputs as in oracle/make_golden_cfg2.py (`cfg2_inputs` with 72 frames); stored: the stitched x0 (fp16: quantisation 72 dB below its range).
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

GOLD = os.path.join(ROOT, "tests", "golden")

CFG3 = dict(frames=72, latent=(122, 216), max_chunk_len=16, t=899, steps=1, guide_scale=7.5, guide_rescale=0.2, seed=3777, wseed=0)


def cfg3_inputs():
    from make_golden_cfg2 import cfg2_inputs
    return cfg2_inputs(CFG3)


def reference_make_chunks():
    """the reference's own make_chunks / sliding_windows_1d, executed from its source (the module imports diffusers / open_clip)"""
    import ref_loader
    src = open(os.path.join(ref_loader.REF_ROOT, "video_to_video/video_to_video_model.py")).read()
    ns = {}
    exec(compile(src[src.index("def pad_to_fit"):], "ref_geometry", "exec"), ns)
    return ns["make_chunks"]


def main():
    import ref_loader
    from star_amd.topology import UNetConfig, random_state_dict
    torch.set_grad_enabled(False)
    torch.set_num_threads(int(os.environ.get("STAR_GOLDEN_THREADS", os.cpu_count())))
    assert ref_loader.reference_available()
    t0 = time.time()
    m = ref_loader.load_unet_module()
    dif, sol, sch = ref_loader.load_diffusion_modules()
    net = m.ControlledV2VUNet().eval()
    net.load_state_dict(random_state_dict(UNetConfig(), seed=CFG3["wseed"]), strict=True)
    chunks = reference_make_chunks()(CFG3["frames"], 0, CFG3["max_chunk_len"])
    print("model built", time.time() - t0, "chunks", chunks, flush=True)
    assert len(chunks) == 8 and chunks[0] == (0, 16) and chunks[1] == (8, 24) and chunks[-1] == (56, 72)

    z, eps, y, neg = cfg3_inputs()
    sig = sch.noise_schedule(schedule="logsnr_cosine_interp", n=1000, zero_terminal_snr=True, scale_min=2.0, scale_max=4.0)
    gd = dif.GaussianDiffusion(sigmas=sig)
    noised = gd.diffuse(z, torch.LongTensor([CFG3["t"]]), noise=eps)

    class NoNoise:   # BrownianTreeNoiseSampler is constructed (solvers_sdedit.py:160) but a one-step trajectory never samples it
        def __init__(self, x, smin, smax, seed=None, transform=None):
            pass

        def __call__(self, s, s_next):
            raise AssertionError("a one-step trajectory draws no noise")

    sol.BrownianTreeNoiseSampler = NoNoise

    def one_step_solver(noise, model, sigmas, variant_info=None, show_progress=False, **kw):
        # solvers_sdedit.py:157,171-175 on the ladder [sigma_0, 0]: x = noise sigma_0; denoised = model(x c_in, sigma_0); x = denoised
        assert len(sigmas) == 2 and float(sigmas[1]) == 0.0
        _, c_in = sol.get_scalings(sigmas[0])
        return model(noise * sigmas[0] * c_in, sigmas[0], variant_info=variant_info)

    dif.sample_dpmpp_2m_sde = one_step_solver   # sample_sr looks the solver up in its module's namespace at call time (:296-299)
    n = [0]
    cache = os.environ.get("STAR_GOLDEN_CACHE", "/tmp/star_cfg3_fwd")
    os.makedirs(cache, exist_ok=True)

    def model(x, t=None, y=None, hint=None, hint_chunk=None, variant_info=None):
        n[0] += 1
        path = os.path.join(cache, f"fwd_{n[0]:02d}.pt")
        if os.path.isfile(path):
            rec = torch.load(path)
            if torch.equal(rec["x"], x):   # same inputs (deterministic run): the cached output of an interrupted run
                print("forward", n[0], "from cache", flush=True)
                return rec["out"]
        out = net(x, t=t, y=y, hint=hint, hint_chunk=hint_chunk, variant_info=variant_info)
        torch.save({"x": x.clone(), "out": out.clone()}, path)
        print("forward", n[0], tuple(x.shape), time.time() - t0, flush=True)
        return out

    x0 = gd.sample_sr(noise=noised, model=model, model_kwargs=[{"y": y}, {"y": neg}, {"hint": z}], guide_scale=CFG3["guide_scale"],
                      guide_rescale=CFG3["guide_rescale"], solver="dpmpp_2m_sde", solver_mode="normal", steps=CFG3["steps"], t_max=CFG3["t"],
                      t_min=0, discretization="trailing", chunk_inds=chunks, show_progress=False)
    assert n[0] == 16 and tuple(x0.shape) == (1, 4, CFG3["frames"], *CFG3["latent"])
    path = os.path.join(GOLD, "cfg3_chunkloop.pt")
    torch.save({"cfg": CFG3, "chunks": chunks, "x0_f16": x0.to(torch.float16), "x0_range": (float(x0.min()), float(x0.max())),
                "noised_sum": float(noised.double().sum())}, path)
    print("wrote", path, tuple(x0.shape), time.time() - t0)


if __name__ == "__main__":
    main()
