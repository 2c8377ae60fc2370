"""world_size-2 gloo tests (CPU) of the multi-GPU sharding logic: the sharded chunk loop must reproduce the
single-process sampler bit-for-bit, and the frame sharder / C1 gather must return frames in order."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from star_amd.diffusion import GaussianDiffusion, noise_schedule
        from star_amd.geometry import make_chunks
        from star_amd.parallel import ChunkSharder, FrameSharder, gather_frames
        sig = noise_schedule(schedule="logsnr_cosine_interp", n=1000, zero_terminal_snr=True, scale_min=2.0, scale_max=4.0)
        gd = GaussianDiffusion(sig)
        g = torch.Generator().manual_seed(5)
        A = torch.randn(4, 4, generator=g) * 0.3

        def model(x, t, y=None, hint=None, hint_chunk=None, variant_info=None):
            h_ = hint_chunk if hint_chunk is not None else hint
            return torch.einsum("oc,bcfhw->bofhw", A, x) * (1.0 + 0.1 * float(y.mean())) + 0.05 * h_ + 0.001 * float(t[0])

        class Noise:
            def __init__(self, x, a, b, seed=None):
                self.g = torch.Generator().manual_seed(99)
                self.shape = x.shape

            def __call__(self, s, sn):
                return torch.randn(self.shape, generator=self.g)

        F_ = 72
        noise = torch.randn(1, 4, F_, 10, 8, generator=g)
        hint = torch.randn(1, 4, F_, 10, 8, generator=g)
        y1, y2 = torch.randn(1, 77, 16, generator=g), torch.randn(1, 77, 16, generator=g)
        res = {}
        for mx in (16, 32):   # 8 chunks (cfg3 layout) and 3 chunks (ragged over 2 ranks)
            chunks = make_chunks(F_, 0, mx)
            kw = dict(noise=noise, model=model, model_kwargs=[{"y": y1}, {"y": y2}, {"hint": hint}], guide_scale=7.5, guide_rescale=0.2,
                      solver_mode="normal", steps=3, t_max=899, t_min=0, discretization="trailing", chunk_inds=chunks, noise_sampler_cls=Noise)
            single = gd.sample_sr(**kw)
            sharded = gd.sample_sr(chunk_executor=ChunkSharder(), **kw)
            res[f"chunks{mx}"] = bool(torch.equal(single, sharded))
        z = torch.arange(11 * 2 * 3, dtype=torch.float32).reshape(11, 2, 3)
        groups = [(i, min(i + 3, 11)) for i in range(0, 11, 3)]
        out = FrameSharder().map_groups(groups, lambda a, b: z[a:b] * 2)
        res["frames"] = bool(torch.equal(out, z * 2))
        zl = torch.arange(400 * 2, dtype=torch.float32).reshape(400, 2)      # 134 groups -> 67 parts per rank (more than one metadata row held before)
        gl = [(i, min(i + 3, 400)) for i in range(0, 400, 3)]
        res["frames_many_groups"] = bool(torch.equal(FrameSharder().map_groups(gl, lambda a, b: zl[a:b] + 1), zl + 1))
        clip = torch.full((1, 3, 2, 4, 4), float(rank))
        allc = gather_frames(clip)
        res["gather"] = all(float(allc[r].mean()) == r for r in range(world))
        # cfg5 (CogVideoX variant): replicas only -- the reference's read_from_file striding, rank / world from the process group
        from star_amd.parallel import replica_items
        lines = [f"prompt {i}\n" for i in range(7)]
        res["replica_items"] = list(replica_items(lines)) == [(f"prompt {i}", i) for i in range(rank, 7, world)]
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_chunk_and_frame_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res in results:
        assert all(res.values()), (rank, res)


def _worker_cfg3(rank, world, port, q):
    """BASELINE configs[2]: 72 frames, max_chunk_len = 16 -> 8 overlapping chunks, ONE per rank; 24 VAE decode groups of 3 frames,
    three per rank; C1 gather of uint8-sized frame tensors."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from star_amd.diffusion import GaussianDiffusion, noise_schedule
        from star_amd.geometry import make_chunks
        from star_amd.parallel import ChunkSharder, FrameSharder, gather_frames
        sig = noise_schedule(schedule="logsnr_cosine_interp", n=1000, zero_terminal_snr=True, scale_min=2.0, scale_max=4.0)
        gd = GaussianDiffusion(sig)
        g = torch.Generator().manual_seed(11)
        A = torch.randn(4, 4, generator=g) * 0.3
        calls = []

        def model(x, t, y=None, hint=None, hint_chunk=None, variant_info=None):
            calls.append(x.shape[2])
            h_ = hint_chunk if hint_chunk is not None else hint
            # couples the frames of a chunk (as the 5-D GroupNorm / temporal attention do): a chunk cannot be split further
            return torch.einsum("oc,bcfhw->bofhw", A, x) * (1.0 + 0.1 * float(y.mean())) + 0.05 * h_ + 0.01 * x.mean(dim=2, keepdim=True) + 0.001 * float(t[0])

        class Noise:
            def __init__(self, x, a, b, seed=None):
                self.g = torch.Generator().manual_seed(99)
                self.shape = x.shape

            def __call__(self, s, sn):
                return torch.randn(self.shape, generator=self.g)

        F_, h, w = 72, 10, 8
        noise = torch.randn(1, 4, F_, h, w, generator=g)
        hint = torch.randn(1, 4, F_, h, w, generator=g)
        y1, y2 = torch.randn(1, 77, 16, generator=g), torch.randn(1, 77, 16, generator=g)
        chunks = make_chunks(F_, 0, 16)
        res = {"eight_chunks": len(chunks) == 8 and chunks[0] == (0, 16) and chunks[-1] == (56, 72)}
        kw = dict(noise=noise, model=model, model_kwargs=[{"y": y1}, {"y": y2}, {"hint": hint}], guide_scale=7.5, guide_rescale=0.2,
                  solver_mode="fast", steps=50, t_max=899, t_min=0, discretization="trailing", chunk_inds=chunks, noise_sampler_cls=Noise)
        single = gd.sample_sr(**kw)
        n_single = len(calls)
        calls.clear()
        sharder = ChunkSharder()
        sharded = gd.sample_sr(chunk_executor=sharder, **kw)
        res["bit_identical"] = bool(torch.equal(single, sharded))
        # 14 evaluations x 2 CFG forwards: every rank ran ONE chunk per forward, the single process all eight
        res["one_chunk_per_rank_per_forward"] = len(calls) * world == n_single and n_single == 14 * 2 * 8 and set(calls) == {16}
        res["plan_exchanged_once"] = len(sharder._plans) == 1
        # VAE decode: 24 groups of 3 frames, three per rank, every rank ends with all 72 frames in order
        z = torch.arange(F_ * 3 * 4 * 4, dtype=torch.float32).reshape(F_, 3, 4, 4)
        groups = [(i, min(i + 3, F_)) for i in range(0, F_, 3)]
        mine = []
        out = FrameSharder().map_groups(groups, lambda a, b: (mine.append((a, b)), z[a:b] * 2 + 1)[1])
        res["decode_groups"] = len(groups) == 24 and len(mine) == 3 and mine == groups[rank::world] and bool(torch.equal(out, z * 2 + 1))
        # C1: uint8 frames [F, H, W, 3] gathered as bytes
        clip = torch.full((F_ // world, 8, 8, 3), rank, dtype=torch.uint8)
        allc = gather_frames(clip)
        res["gather_u8"] = len(allc) == world and all(a.dtype == torch.uint8 and int(a.max()) == r and int(a.min()) == r for r, a in enumerate(allc))
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_cfg3_layout_on_eight_ranks():
    """world_size 8 (gloo, CPU): the exact cfg3 layout -- 72 frames, 8 chunks of 16, one per rank in every solver evaluation, 24 decode
    groups over 8 ranks -- reproduces the single-process sampler bit for bit (the reference's per-step chunk
    loop: diffusion_sdedit.py:330-353)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_cfg3, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r for r, _ in results) == list(range(8))
    for rank, res in results:
        assert all(res.values()), (rank, res)


def test_replica_items_is_the_references_prompt_striding():
    """cogvideox-based/sat/sample_sr.py:38-45: line cnt of the input file goes to rank cnt % world_size, stripped, with its index;
    every line is owned by exactly one rank; a single process owns everything."""
    from star_amd.parallel import replica_items
    lines = ["a\n", " b \n", "c", "d\n", "e\n"]
    assert list(replica_items(lines)) == [("a", 0), ("b", 1), ("c", 2), ("d", 3), ("e", 4)]
    got = [list(replica_items(lines, rank=r, world=3)) for r in range(3)]
    assert got[0] == [("a", 0), ("d", 3)] and got[1] == [("b", 1), ("e", 4)] and got[2] == [("c", 2)]
    assert sorted(c for g in got for _, c in g) == list(range(5))
    assert list(replica_items([], rank=0, world=2)) == []
    with pytest.raises(ValueError):
        list(replica_items(lines, rank=3, world=3))


def _worker_rccl(rank, world, port, q):
    """the sharders on the REAL backend: `nccl` (= RCCL) with device tensors.  The GPU box has one GPU, so the world is 1: the
    communicator is created, the metadata exchange and the payload all-gather run as RCCL collectives on cuda:0 (no host staging
    branch is taken), and the results must equal the unsharded ones bit for bit."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from star_amd.diffusion import GaussianDiffusion, noise_schedule
        from star_amd.geometry import make_chunks
        from star_amd.parallel import ChunkSharder, FrameSharder, gather_frames
        res = {"backend_is_nccl": dist.get_backend() == "nccl"}
        sig = noise_schedule(schedule="logsnr_cosine_interp", n=1000, zero_terminal_snr=True, scale_min=2.0, scale_max=4.0)
        gd = GaussianDiffusion(sig)
        g = torch.Generator().manual_seed(5)
        A = (torch.randn(4, 4, generator=g) * 0.3).to(dev)

        def model(x, t, y=None, hint=None, hint_chunk=None, variant_info=None):
            h_ = hint_chunk if hint_chunk is not None else hint
            return torch.einsum("oc,bcfhw->bofhw", A, x) * (1.0 + 0.1 * float(y.mean())) + 0.05 * h_ + 0.001 * float(t[0])

        class Noise:
            def __init__(self, x, a, b, seed=None):
                self.g = torch.Generator().manual_seed(99)
                self.shape, self.device = x.shape, x.device

            def __call__(self, s, sn):
                return torch.randn(self.shape, generator=self.g).to(self.device)

        F_ = 72
        noise = torch.randn(1, 4, F_, 10, 8, generator=g).to(dev)
        hint = torch.randn(1, 4, F_, 10, 8, generator=g).to(dev)
        y1, y2 = torch.randn(1, 77, 16, generator=g).to(dev), torch.randn(1, 77, 16, generator=g).to(dev)
        for mx in (16, 32):
            chunks = make_chunks(F_, 0, mx)
            kw = dict(noise=noise, model=model, model_kwargs=[{"y": y1}, {"y": y2}, {"hint": hint}], guide_scale=7.5, guide_rescale=0.2,
                      solver_mode="normal", steps=3, t_max=899, t_min=0, discretization="trailing", chunk_inds=chunks, noise_sampler_cls=Noise)
            single = gd.sample_sr(**kw)
            sharded = gd.sample_sr(chunk_executor=ChunkSharder(), **kw)
            res[f"chunks{mx}"] = bool(sharded.is_cuda and torch.equal(single, sharded))
        z = torch.arange(11 * 2 * 3, dtype=torch.float32, device=dev).reshape(11, 2, 3)
        groups = [(i, min(i + 3, 11)) for i in range(0, 11, 3)]
        out = FrameSharder().map_groups(groups, lambda a, b: z[a:b] * 2)
        res["frames"] = bool(out.is_cuda and torch.equal(out, z * 2))
        clip = torch.full((9, 8, 8, 3), 7, dtype=torch.uint8, device=dev)
        allc = gather_frames(clip)
        res["gather_u8"] = len(allc) == world and allc[0].is_cuda and bool(torch.equal(allc[0], clip))
        # the bench's max-over-ranks reduction and barrier, as bench.py issues them
        t = torch.tensor([1.25], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier(device_ids=[0])
        torch.cuda.synchronize()
        res["allreduce_max"] = float(t) == 1.25
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharders_on_rccl_device_tensors():
    """RCCL on hardware: world size 1 is all a 1-GPU box offers -- communicator creation, the plan exchange and the
    payload all-gathers run through the `nccl` backend on device tensors and reproduce the unsharded results bit for bit."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_rccl, args=(0, 1, _free_port(), q))
    p.start()
    rank, res = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert all(res.values()), res
