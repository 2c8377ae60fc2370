"""bench.py -- upscaled frames/s of the STAR hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]           (N > 1: launched by torch.distributed.run)

  python bench.py --config cfg1|cfg2|cfg3|cfg4 ...               (other BASELINE configs; the default line is cfg2)

One "step" = one full `VideoToVideo_sr.test()` on a synthetic 32-frame 240x426 clip -> 4x (BASELINE config[1]):
bilinear upsample + pad to 976x1728, VAE encode of 32 frames, 50 DPM-Solver++ evaluations x 2 denoiser forwards
(CFG), VAE decode in 3-frame groups; the LR clip is already resident in HBM when the timed region starts.
N GPUs = N independent clips, one per rank (weak scaling), with an RCCL all-gather of the decoded frames (C1).
--config cfg3 (72 frames as 8 overlapping 16-frame chunks) shards ONE video over the ranks instead: every solver step's
chunks by ChunkSharder (all-gather of the x0 cores, C2), the VAE decode groups by FrameSharder (strong scaling).
Random-init weights of the full architecture (2.04 B-parameter UNet+ControlNet, 97.7 M-parameter SVD VAE) and
synthetic data: no checkpoint or dataset is reachable offline.

Besides the headline line, the JSON carries `roofline` (the spatial self-attention kernel, timed live with HIP
events on the launch stream) and `cpu_baseline` (the CPU oracle timed on this box's host cores on a bounded
sample, extrapolated by FLOP ratio -- never presented as measured end-to-end).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic FLOPs (2 per MAC; matmul + conv + attention), SURVEY.md section 8(d) / BASELINE.md section 3
UNET_FWD_TFLOP_CFG2 = 572.3         # one UNet+ControlNet forward, 32 f, latent 122x216
VAE_TFLOP_PER_FRAME = 8.4 + 20.8    # encode + decode at 976x1728 (estimate)
PEAK_BF16_MFMA = 2.5e15             # dense, MI355X_MICROARCH.md
PEAK_HBM = 8.0e12                   # HBM3E, MI355X_MICROARCH.md
ROOFLINE_KERNEL = "flash_attn_v5_kernel"   # the kernel the top-level roofline object describes (star_amd/csrc/attn5.h, AttnArgs::variant 9)

# BASELINE.json configs[0..3] (configs[4], CogVideoX, is out of scope: SURVEY.md section 8f).  fwd_tflop = one UNet+ControlNet
# forward on one chunk (FlopCounterMode on the reference); vae_scale = padded pixels / (976*1728).
CONFIGS = {
    "cfg1": dict(frames=8, height=128, width=128, denoise_steps=5, solver_mode="normal", max_chunk_len=32, fwd_tflop=64.4, chunks=1, vae_scale=720 * 1280 / (976 * 1728)),
    "cfg2": dict(frames=32, height=240, width=426, denoise_steps=50, solver_mode="normal", max_chunk_len=32, fwd_tflop=572.3, chunks=1, vae_scale=1.0),
    "cfg3": dict(frames=72, height=240, width=426, denoise_steps=50, solver_mode="fast", max_chunk_len=16, fwd_tflop=286.2, chunks=8, vae_scale=1.0),
    "cfg4": dict(frames=32, height=540, width=960, denoise_steps=50, solver_mode="normal", max_chunk_len=32, fwd_tflop=7582.0, chunks=1, vae_scale=2192 * 3904 / (976 * 1728)),
}


def steps_detail(t0, ends, power):
    """wall time of every timed step with the socket power / shader clock sampled during it: a clip is ~75 kJ at the socket cap, and
    what a box sustains over many clips is not what its first one shows"""
    out, a = [], t0
    for b in ends:
        d = {"ms": round((b - a) * 1e3, 1)}
        ss = [x for x in (power.samples if power else []) if a <= x[0] < b]
        if ss:
            d["socket_W"] = round(sum(x[1] for x in ss) / len(ss), 1)
            d["sclk_MHz"] = round(sum(x[2] for x in ss) / len(ss), 1)
        out.append(d)
        a = b
    return out


class PowerSampler:
    """socket power / shader clock of ONE GPU from sysfs hwmon, sampled on a host thread while kernels run (never rocm-smi beside a
    kernel on this pool: profiles/r03_attn7_ab.txt).  The hwmon directory is matched to the torch device through its PCI address."""

    def __init__(self, device_index=0, period=0.1):
        import glob
        self.period, self.samples, self._stop, self._th = period, [], False, None
        cards = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"), key=lambda p: int(p.split("/card")[1].split("/")[0]))
        self.path = None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            for c in cards:
                if os.path.basename(os.path.realpath(os.path.join(c, "..", ".."))).lower().startswith(want):
                    self.path = c
        except Exception:
            pass
        if self.path is None and cards:
            self.path = cards[min(device_index, len(cards) - 1)]

    def _read(self):
        try:
            try:
                w = int(open(self.path + "/power1_average").read()) / 1e6
            except Exception:
                w = int(open(self.path + "/power1_input").read()) / 1e6
            return time.perf_counter(), w, int(open(self.path + "/freq1_input").read()) / 1e6
        except Exception:
            return None

    def _run(self):
        while not self._stop:
            r = self._read()
            if r:
                self.samples.append(r)
            time.sleep(self.period)

    def start(self):
        import threading
        self.samples, self._stop = [], False
        if self.path:
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        return self

    def stop(self):
        self._stop = True
        if self._th:
            self._th.join()
        return self

    def summary(self, skip=0.0):
        """mean / p5 / p95 of the samples taken later than `skip` seconds after the first one"""
        ss = [x for x in self.samples if x[0] - self.samples[0][0] >= skip] if self.samples else []
        if not ss:
            return None

        def stats(v):
            v = sorted(v)
            pick = lambda q: v[min(len(v) - 1, int(q * (len(v) - 1) + 0.5))]
            return {"mean": round(sum(v) / len(v), 1), "p5": round(pick(0.05), 1), "p95": round(pick(0.95), 1)}
        return {"socket_W": stats([x[1] for x in ss]), "sclk_MHz": stats([x[2] for x in ss]), "samples": len(ss), "period_s": self.period}


def attention_operand_sweep(ctx, dev, device_index, heads=5, frames=32, hw=(122, 216), seconds=2.5):
    """The dominant kernel alone, in a loop, on three operand sets of the cfg2 level-0 shape -- (i) N(0,1) (what random-init weights
    feed it in the clip), (ii) the peaked-logit statistics of a TRAINED layer (tests/test_fullsize.py: shared low-rank component,
    logits over +-40, a few keys carrying each row), (iii) zeros -- with the socket power and shader clock sampled beside each.
    Outside the timed region; answers what the kernel's power-bound operating point is on each kind of data."""
    H, W = hw
    N, C = H * W, heads * 64
    g = torch.Generator(device=dev).manual_seed(7)
    sets = {}
    sets["normal"] = torch.randn(frames, N, 3 * C, generator=g, device=dev, dtype=torch.float32).to(ctx.dtype)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=dev), torch.arange(W, dtype=torch.float32, device=dev), indexing="ij")
    pos = torch.stack([torch.sin(yy / 9.0), torch.cos(yy / 9.0), torch.sin(xx / 13.0), torch.cos(xx / 13.0)], dim=-1).reshape(N, 4)
    pk = torch.empty(frames, N, 3 * C, device=dev, dtype=ctx.dtype)
    for h in range(heads):
        basis = torch.randn(4, 64, generator=g, device=dev)
        for part in range(2):
            x = (pos @ basis * 2.5)[None] + torch.randn(frames, N, 64, generator=g, device=dev) * 0.7
            if part == 1:
                x[:, N - 3000:] *= 1.6
            pk[:, :, part * C + h * 64: part * C + (h + 1) * 64] = x.to(ctx.dtype)
        pk[:, :, 2 * C + h * 64: 2 * C + (h + 1) * 64] = torch.randn(frames, N, 64, generator=g, device=dev).to(ctx.dtype)
    sets["peaked_trained_like"] = pk
    sets["zeros"] = torch.zeros(frames, N, 3 * C, device=dev, dtype=ctx.dtype)
    out = torch.empty(frames, N, C, device=dev, dtype=ctx.dtype)
    flops = 4.0 * frames * heads * float(N) * N * 64
    res = {}
    runs = [(name, ctx, 9, qkv) for name, qkv in sets.items()]
    # the SAME kernel without its softmax VALU (QK^T, P V, LDS reads, staging and barriers kept; P is a constant: "variant 16", a timing
    # ablation that exists in the bench build of the library only) on the N(0,1) operands, in this very run: the MFMA + staging skeleton's
    # power-bound rate on this socket = the ceiling any softmax placement could reach on this data
    bench_so = os.path.join(ROOT, "tools", "bench", "libstar_hip_bench.so")
    bctx = None
    if os.path.isfile(bench_so):
        try:
            from star_amd import lib as L_
            bctx = L_.Context(device_index, ctx.dtype, L_.Library(bench_so))
            runs.append(("mfma_skeleton_no_softmax_normal", bctx, 16, sets["normal"]))
        except Exception as e:   # measurement tooling only: the bench line does not depend on it
            print(f"bench: skeleton ceiling not measured ({e})", file=sys.stderr)
    for name, c_, variant, qkv in runs:
        fn = lambda: c_.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, out=out, variant=variant)
        fn(); c_.sync(); torch.cuda.synchronize()
        ps = PowerSampler(device_index, 0.05).start()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:      # 10 launches (~0.25 s) per host synchronisation: the wall clock is the kernel time
            for _ in range(10):
                fn()
            c_.sync(); torch.cuda.synchronize(); n += 10
        ms = (time.perf_counter() - t0) * 1e3 / n
        ps.stop()
        pw = ps.summary(skip=0.5)
        res[name] = {"ms": round(ms, 3), "TFLOP/s": round(flops / ms / 1e9, 1), "frac_of_peak": round(flops / ms / 1e9 / (PEAK_BF16_MFMA / 1e12), 4),
                     "socket_W": pw["socket_W"]["mean"] if pw else None, "sclk_MHz": pw["sclk_MHz"]["mean"] if pw else None, "launches": n}
    del sets, pk, out
    if bctx is not None:
        bctx.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"],
                    help="storage/MFMA input type; f16 = the reference's own (generator.half()+autocast) and the one that meets the PSNR>=50 dB parity bar")
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS), help="BASELINE.json configs[0..3]; cfg2 is the metric's configuration")
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--denoise-steps", type=int, default=None)
    ap.add_argument("--solver-mode", default=None, choices=["normal", "fast"])
    ap.add_argument("--max-chunk-len", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", default="auto", choices=["auto", "on", "off"],
                    help="per-family kernel times (HIP events around EVERY launch cost ~2 %% of a clip, so they are taken on an UNTIMED step: the last "
                         "warm-up step (auto, when --warmup >= 1) or one extra step after the timed region (on)")
    ap.add_argument("--no-operand-sweep", action="store_true", help="skip the post-run power / clock sweep of the attention kernel (N = 1 only)")
    ap.add_argument("--small", action="store_true", help="reduced-width smoke configuration (NOT the metric)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl = RCCL over xGMI (default); gloo only for single-GPU plumbing tests")
    ap.add_argument("--share-gpu0", action="store_true", help="TEST ONLY: every rank uses cuda:0 (1-GPU box, gloo backend)")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    custom = False
    for k in ("frames", "height", "width", "denoise_steps", "solver_mode", "max_chunk_len"):
        v = getattr(args, k)
        if v is None:
            setattr(args, k, cfg[k])
        elif v != cfg[k]:
            custom = True
    shard_one_video = args.config == "cfg3"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: launch ourselves under torch.distributed.run, one rank per GPU on this node (the driver's
        # own `python -m torch.distributed.run ... bench.py --gpus N` form arrives with WORLD_SIZE set and skips this)
        # (--standalone: torchrun's own c10d rendezvous picks a free port itself -- no bind-then-close race with another process)
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
               f"--nproc-per-node={args.gpus}", os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_gpu0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    torch.set_grad_enabled(False)
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
    dev = torch.device("cuda", local_rank)

    from star_amd.topology import SMALL_TEST_CONFIG, UNetConfig, random_state_dict
    from star_amd.vae_topology import SMALL_VAE_CONFIG, VaeConfig, random_vae_state_dict
    from star_amd.video_to_video_model import VideoToVideo_sr
    ucfg = SMALL_TEST_CONFIG if args.small else UNetConfig()
    vcfg = VaeConfig(block_out_channels=(64, 64, 128, 128)) if args.small else VaeConfig()

    t0 = time.time()
    sd = random_state_dict(ucfg, seed=0)
    t_weights = time.time() - t0

    # ---------------- CPU baseline: the oracle (a port of the reference arithmetic) on the host cores, bounded sample
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(sd, ucfg, args)

    gneg = torch.Generator().manual_seed(668)
    opt = dict(state_dict=sd, vae_state_dict=random_vae_state_dict(vcfg, seed=0), unet_config=ucfg, vae_config=vcfg, dtype=dtype,
               negative_y=torch.randn(1, 77, ucfg.context_dim, generator=gneg))
    if shard_one_video and world > 1:       # one long video: identical noise on every rank, chunks / decode groups sharded
        from star_amd.parallel import ChunkSharder, FrameSharder
        opt.update(rng=torch.Generator().manual_seed(666), chunk_executor=ChunkSharder(), frame_sharder=FrameSharder())
    t0 = time.time()
    model = VideoToVideo_sr(opt, device=dev)
    del sd, opt
    t_load = time.time() - t0

    # synthetic LR clip + text embedding (seeds: SURVEY.md section 8d), resident on the device
    g = torch.Generator().manual_seed(666 + (0 if shard_one_video else rank))
    video = (torch.randn(args.frames, 3, args.height, args.width, generator=g) * 0.5).clamp(-1, 1).to(dev)
    y = torch.randn(1, 77, ucfg.context_dim, generator=torch.Generator().manual_seed(667)).to(dev)
    data = {"video_data": video, "y": y, "target_res": (args.height * 4, args.width * 4)}

    def step():
        torch.manual_seed(666 + rank)
        if model.rng is not None:
            model.rng.manual_seed(666)
        out = model.test(data, total_noise_levels=900, steps=args.denoise_steps, solver_mode=args.solver_mode,
                         guide_scale=7.5, max_chunk_len=args.max_chunk_len, return_device=True)
        if world > 1 and not shard_one_video:
            from star_amd.parallel import gather_frames
            step.gathered = gather_frames(out, source=video, ctx=model.generator.ctx)   # uint8 frames (colour-fixed on the device): 1/4 of the xGMI bytes
        return out          # this rank's own fp32 clip (finiteness is checked on it)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    uctx, vctx = model.generator.ctx, model.vae.ctx
    prof_u_all = prof_v_all = None
    for i in range(args.warmup):
        last = i == args.warmup - 1 and args.breakdown != "off"
        if last:                                   # the per-family breakdown: every launch bracketed, on an UNTIMED step
            uctx.profile_begin(); vctx.profile_begin()
        step()
        if last:
            prof_u_all, prof_v_all = uctx.profile_end(), vctx.profile_end()
    barrier()
    # inside the timed region only the dominant kernel (spatial self-attention) is bracketed with HIP events: two event records per
    # launch cost stream time, and a clip has ~340 000 launches (measured: 1.4 s of 61 s with all of them bracketed)
    uctx.profile_begin(kinds=["attn_self"])
    power = PowerSampler(local_rank, 0.1).start() if rank == 0 else None
    t0 = time.perf_counter()
    step_ends = []
    for _ in range(args.steps):
        out = step()
        torch.cuda.synchronize()            # per-step wall times (steps_detail): one host sync per ~minute-long clip
        step_ends.append(time.perf_counter())
    barrier()
    elapsed = time.perf_counter() - t0
    if power:
        power.stop()
    prof_u = uctx.profile_end()
    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([elapsed], device=dev if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    final = out[0] if isinstance(out, list) else out
    finite = bool(torch.isfinite(final).all())
    # the reference's test() ends with `.cpu()` of the fp32 output (video_to_video_model.py:139); the timed region hands the frames
    # over on the device (return_device=True: the colour fix / uint8 conversion that follows runs there) -- measured here, reported
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    host_copy = final.float().cpu()
    d2h_ms = (time.perf_counter() - t1) * 1e3
    d2h_bytes = host_copy.numel() * 4
    del host_copy
    if prof_u_all is None and args.breakdown == "on":   # one extra untimed step with every launch bracketed
        uctx.profile_begin(); vctx.profile_begin()
        step()
        prof_u_all, prof_v_all = uctx.profile_end(), vctx.profile_end()
    sweep = None
    if rank == 0 and world == 1 and not args.small and not args.no_operand_sweep:
        del out, final
        uctx.trim(); vctx.trim()
        sweep = attention_operand_sweep(uctx, dev, local_rank)

    if rank == 0:
        frames_total = args.frames * (1 if shard_one_video else world) * args.steps
        evals = 14 if args.solver_mode == "fast" else args.denoise_steps
        a = prof_u["attn_self"]
        l0_flops = a["max_flops"]                       # the largest launches = the L0 layers (N = H*W of the padded latent)
        # HBM-side traffic of the dominant kernel: the NEWEST committed PMC collection (profiles/rNN_attn*_traffic.json, highest round
        # first) whose `kernel` names the kernel this run launched; a file of another kernel is not quoted (traffic = null)
        import glob as _glob
        traffic, tname = None, None
        for tp in sorted(_glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_attn*_traffic.json")), reverse=True):
            try:
                tj = json.load(open(tp))
            except Exception:
                continue
            if ROOFLINE_KERNEL in tj.get("kernel", "") and "traffic_bytes_per_launch" in tj:
                tname = os.path.basename(tp)
                if not args.small and args.config in ("cfg2", "cfg3") and not custom:   # PMC pass of the same kernel at the same shape
                    traffic = tj["traffic_bytes_per_launch"]
                break
        roof = {"bound": "mfma", "kernel": f"{ROOFLINE_KERNEL} (spatial self-attention, d=64)", "unit": "TFLOP/s",
                "achieved": (l0_flops / (a["max_flops_ms"] * 1e-3) / 1e12) if a["max_flops_ms"] else None,
                "peak": PEAK_BF16_MFMA / 1e12, "traffic": traffic,
                "traffic_source": f"profiles/{tname} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, bytes per launch)" if traffic else None,
                "algorithmic_flops_per_launch": l0_flops, "avg_launch_ms": a["max_flops_ms"],   # true mean over the L0 launches of the timed region
                "all_self_attn_launches": {"launches": a["launches"], "ms": a["ms"], "TFLOP/s": a["flops"] / max(a["ms"], 1e-9) / 1e9}}
        roof["frac"] = roof["achieved"] / roof["peak"] if roof["achieved"] else None
        # power: sampled LIVE in this run (sysfs hwmon beside the kernels) -- over the whole timed region, and, after it, beside the
        # dominant kernel alone on three operand sets (N(0,1) / trained-like peaked logits / zeros).  DESIGN.md section 3.6: on
        # full-entropy operands this kernel holds the socket at its power cap and the shader clock falls; `peak` stays the nominal
        # dense MFMA figure the metric is defined against.
        roof["power"] = {"measured_live": bool(power and power.samples), "timed_region": power.summary() if power else None,
                         "source": "sysfs hwmon (power1_average, freq1_input) sampled every 0.1 s on a host thread during the timed region",
                         "dominant_kernel_alone": sweep,
                         "dominant_kernel_alone_note": "L0 spatial self-attention of cfg2 (32 x 5 heads, N = 26352) looped ~2.5 s per operand set "
                                                       "AFTER the timed region, hwmon sampled every 0.05 s beside it" if sweep else None}
        if sweep and sweep.get("peaked_trained_like"):
            pk_ = sweep["peaked_trained_like"]
            roof["second_operating_point"] = {"operands": "trained-like peaked logits (tests/test_fullsize.py statistics) -- NOT the headline: the clip's own "
                                                          "operands are the N(0,1)-like activations of random-init weights",
                                              "achieved": pk_["TFLOP/s"], "frac": pk_["frac_of_peak"], "socket_W": pk_["socket_W"], "sclk_MHz": pk_["sclk_MHz"]}
        skel = sweep.pop("mfma_skeleton_no_softmax_normal", None) if sweep else None
        if skel:   # measured in THIS run, beside the operand sweep (bench build of the library, variant 16)
            roof["mfma_skeleton_ceiling_on_random_operands"] = {"TFLOP/s": skel["TFLOP/s"], "socket_W": skel["socket_W"], "sclk_MHz": skel["sclk_MHz"],
                "source": "live: flash_attn_v3_kernel<..., ABL 5> (variant 16 of tools/bench/libstar_hip_bench.so: the kernel without its softmax VALU) looped on the "
                          "N(0,1) operand set right after the timed region, hwmon sampled beside it",
                "frac_of_nominal": skel["frac_of_peak"]}
            ceil_tf = skel["TFLOP/s"]
        else:
            roof["mfma_skeleton_ceiling_on_random_operands"] = {"TFLOP/s": 1568.0, "source": "static: profiles/r02_power_limit.txt (variant 16, N(0,1) f16), a round-2 "
                                                                "measurement -- the bench build of the library was not found / not loadable in this run"}
            ceil_tf = 1568.0
        # what the kernel reaches of the power-bound ceiling of its own MFMA + staging skeleton on the same data, socket and run; the 0.70 of the
        # NOMINAL peak that north_star names lies above that ceiling
        roof["power_ceiling_frac"] = roof["achieved"] / ceil_tf if roof["achieved"] else None
        roof["power_ceiling_frac_same_operands_alone"] = (sweep["normal"]["TFLOP/s"] / ceil_tf) if sweep and sweep.get("normal") else None
        breakdown = None if prof_u_all is None else {k: {"ms": round(v["ms"], 1), "launches": v["launches"],
                         "TFLOP/s": round(v["flops"] / v["ms"] / 1e9, 1) if v["ms"] and v["flops"] else None,
                         "GB/s": round(v["bytes"] / v["ms"] / 1e6, 1) if v["ms"] else None} for k, v in prof_u_all.items()}
        # every kernel family of the UNet against the roof that bounds it (the per-family table of the untimed breakdown step): MFMA
        # families in TFLOP/s of their algorithmic FLOPs against the nominal dense peak, HBM families in GB/s of their algorithmic bytes
        # against 8 TB/s.  `share` = the family's part of the summed kernel time of that step.
        if prof_u_all is not None:
            fam_bound = {"attn_self": "mfma", "gemm": "mfma", "conv3x3": "mfma", "tconv": "mfma", "attn_cross": "hbm", "temporal_attn": "hbm",
                         "group_norm": "hbm", "layer_norm": "hbm", "misc": "hbm"}
            tot_ms = sum(v["ms"] for v in prof_u_all.values()) or 1.0
            fams = {}
            for k, v in prof_u_all.items():
                if not v["ms"]:
                    continue
                if fam_bound.get(k) == "mfma":
                    ach, pk, unit = v["flops"] / v["ms"] / 1e9, PEAK_BF16_MFMA / 1e12, "TFLOP/s"
                else:
                    ach, pk, unit = v["bytes"] / v["ms"] / 1e6, PEAK_HBM / 1e9, "GB/s"
                fams[k] = {"bound": fam_bound.get(k, "hbm"), "achieved": round(ach, 1), "peak": pk, "unit": unit, "frac": round(ach / pk, 4),
                           "ms": round(v["ms"], 1), "share": round(v["ms"] / tot_ms, 4), "launches": v["launches"]}
            roof["families"] = fams
            roof["families_note"] = ("one UNTIMED clip with HIP events around every launch; achieved = the family's algorithmic FLOPs (or bytes: every "
                                     "operand read once, the output written once) / its summed launch time; weakest MFMA family first to fix")
        line = {
            "metric": "upscaled frames/sec (4x, 32f 240x426 chunk)" if args.config == "cfg2" and not custom else
                      f"upscaled frames/sec (4x, {args.frames}f {args.height}x{args.width}; BASELINE {args.config}{' modified' if custom else ''})",
            "value": frames_total / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if shard_one_video else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"baseline_config": args.config + (" (modified)" if custom else ""),
                       "workload": f"I2VGen-XL STAR path, {args.frames}f {args.height}x{args.width} -> 4x "
                                   f"({args.height * 4}x{args.width * 4}, padded latent), solver_mode={args.solver_mode}, "
                                   f"{evals} DPM++(2M)SDE evals x 2 CFG forwards, VAE enc 1f/call + dec 3f/group, "
                                   "random-init weights" + (" [REDUCED-WIDTH SMOKE CONFIG]" if args.small else ""),
                       "frames_per_gpu": args.frames, "evals": evals, "max_chunk_len": args.max_chunk_len,
                       "parallelism": (f"one video, solver-step chunks + VAE groups sharded over {world} ranks (RCCL all-gather of x0 cores / frames)"
                                       if shard_one_video else f"chunk-replicas x{world} + RCCL all-gather of frames")},
            "roofline": roof, "cpu_baseline": cpu_baseline,
            "unet_kernel_ms": breakdown, "vae_kernel_ms": None if prof_v_all is None else {k: round(v["ms"], 1) for k, v in prof_v_all.items()},
            "kernel_ms_note": "per-family times of ONE UNTIMED step with HIP events around every launch (the last warm-up step, or --breakdown on); "
                              "inside the timed region only the spatial self-attention launches are bracketed" if breakdown else
                              "no breakdown in this run (--warmup 0 without --breakdown on)",
            "setup_s": {"weights": round(t_weights, 1), "load": round(t_load, 1)},
            "steps_detail": steps_detail(t0, step_ends, power),
            "algorithmic_pflop_per_step": (2 * evals * cfg["chunks"] * cfg["fwd_tflop"] + args.frames * VAE_TFLOP_PER_FRAME * cfg["vae_scale"]) / 1e3
                                          if not args.small and not custom else None,
            "timed_region_excludes": f"the final .cpu() of the fp32 output ({d2h_bytes / 1e6:.0f} MB, video_to_video_model.py:139): the frames are handed over "
                                     f"on the device; that copy measured once after the timed region: {d2h_ms:.0f} ms (pageable host memory)",
            "output_d2h_ms": round(d2h_ms, 1),
            "output_finite": finite, "hbm_pool_gb": {"unet": round(uctx.lib.pool_peak_bytes(uctx.h) / 2 ** 30, 1), "vae": round(vctx.lib.pool_peak_bytes(vctx.h) / 2 ** 30, 1)},
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def run_cpu_baseline(sd, ucfg, args):
    """Time the CPU oracle (oracle/unet_oracle.py: a PyTorch fp32 port of the reference forward, pinned to the reference's own
    code by tests/test_oracle.py; /root/reference itself does not exist on the GPU box) on this box's host cores:
      1. a thread sweep (16 / 32 / 64 / 128) on ONE frame of the sample's own shape (latent 90x160, full width, 5.9 TFLOP) picks
         the thread count (torch's CPU kernels collapse on these shapes when given all 256 hardware threads), and
      2. ONE full UNet+ControlNet forward at the real level-0 shape of BASELINE config[0] (latent 90x160 = 14 400 tokens per
         frame, full 2.04 B-parameter width) on 4 frames is the measured sample (tens of TFLOP, tens of seconds).
    The frames/s figure is that measured rate EXTRAPOLATED by FLOP ratio to the benchmark workload and says so; the
    workload itself is tens of PFLOP -- days on a CPU."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import unet_oracle as O
    from make_golden import unet_inputs
    from torch.utils.flop_counter import FlopCounterMode
    ncpu = os.cpu_count() or 1
    small = args.small

    def timed(f, h, w, seed):
        x, t, y, hint = unet_inputs(ucfg, f, h, w, seed)
        with FlopCounterMode(display=False) as fc:
            t0 = time.perf_counter()
            O.unet_forward(sd, ucfg, x, t, y, hint)
            secs = time.perf_counter() - t0
        return float(fc.get_total_flops()), secs

    # thread sweep ON THE REAL LEVEL-0 SHAPE (one frame of the sample: 14 400 tokens, full width; round-2 review: the sweep used
    # to run on a 2-frame 26x24 toy and its winner was then applied to a 2 500x larger forward)
    sweep = {}
    sf, sh, sw = (1, 10, 8) if small else (1, 90, 160)
    timed(1, 10, 8, 3)
    for n in [c for c in (16, 32, 64, 128) if c <= ncpu] or [ncpu]:
        torch.set_num_threads(n)
        fl, secs = timed(sf, sh, sw, 5)
        sweep[n] = fl / secs / 1e9
    cores = max(sweep, key=sweep.get)
    torch.set_num_threads(cores)
    f, h, w = (2, 26, 24) if small else (4, 90, 160)
    flops, secs = timed(f, h, w, 7)
    cpu_flops = flops / secs
    evals = 14 if args.solver_mode == "fast" else args.denoise_steps
    cfg = CONFIGS[args.config]
    total = (2 * evals * cfg["chunks"] * cfg["fwd_tflop"] + args.frames * VAE_TFLOP_PER_FRAME * cfg["vae_scale"]) * 1e12
    return {"value": args.frames / (total / cpu_flops), "unit": "frames/s", "cores": cores, "kind": "port",
            "host_threads_available": ncpu, "thread_sweep_gflops": {str(k): round(v, 1) for k, v in sweep.items()},
            "thread_sweep_shape": f"f={sf}, latent {sh}x{sw}, full width" if not small else "reduced-width smoke",
            "sample": f"one full-width UNet+ControlNet forward of the fp32 CPU oracle at f={f}, latent {h}x{w} (the level-0 shape of BASELINE "
                      f"config[0]): {flops / 1e12:.1f} TFLOP in {secs:.1f} s = {cpu_flops / 1e9:.0f} GFLOP/s on {cores} threads (best of the sweep); "
                      f"frames/s EXTRAPOLATED by FLOP ratio to the {total / 1e15:.1f} PFLOP workload (= {total / cpu_flops / 3600:.1f} h on this host)",
            "measured_gflops": cpu_flops / 1e9, "sample_seconds": secs, "sample_tflop": flops / 1e12}


if __name__ == "__main__":
    main()
