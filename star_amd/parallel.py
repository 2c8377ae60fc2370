"""Multi-GPU execution of the hot path: one process per GPU, torch.distributed over RCCL/xGMI
(backend "nccl" on ROCm; "gloo" in the CPU tests).  SURVEY.md section 8(e).

The path shards along the frame-chunk axis:
  * independent clips (the 1/2/4/8-GPU frames/s metric): replicas, no data-path collective; `gather_frames`
    (C1) all-gathers the decoded frames at the end;
  * one long video (cfg3): inside every solver step the overlapping chunks are independent (diffusion_sdedit.py:
    337-350), so `ChunkSharder` gives chunk i to rank i % world and all-gathers the trimmed x0 cores (C2, a few MB)
    so that every rank applies the identical solver update to the full-length latent.  VAE decode groups of 3 frames
    are sharded the same way by `FrameSharder`.
  * CogVideoX variant (BASELINE config #5): REPLICAS ONLY.  The reference shards its 8 GPUs by prompt / input video -- line `cnt` of
    the input file goes to data-parallel rank `cnt % world_size` (cogvideox-based/sat/sample_sr.py:38-45,139-141) -- and forces
    model- and context-parallel size 1 at sampling time (:263-264): the 42-layer DiT over 9676 tokens is never split.
    `replica_items` is that striding; no collective exists on that path (none is invented here).
The only collectives are all-gathers: the payload (RCCL over xGMI) and, once per (latent shape, chunk list), two small
all-gathers of part counts / sizes; no reduce-type collective exists anywhere on the path.
"""
import torch
import torch.distributed as dist


def _world(group):
    return dist.get_world_size(group), dist.get_rank(group)


_DTYPES = [torch.float32, torch.float16, torch.bfloat16, torch.float64]


def _gather_ints(vec, group, device):
    """all-gather of a small int64 vector (metadata only; runs once per plan, see _RaggedPlan)."""
    world, _ = _world(group)
    v = vec.to(device if dist.get_backend(group) == "nccl" else "cpu")
    outs = [torch.empty_like(v) for _ in range(world)]
    dist.all_gather(outs, v, group=group)
    return torch.stack(outs).cpu()


def _all_gather_one(t, world, group):
    """all_gather_into_tensor of equally shaped contiguous tensors (the concatenated-along-dim-0 output form, which both RCCL and
    gloo accept) -> list of per-rank views of one allocation"""
    flat = torch.empty([world * t.shape[0]] + list(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(flat, t, group=group)
    return list(flat.chunk(world, 0))


class _RaggedPlan:
    """Who sends how many parts of which size: static for a given (latent shape, chunk list) / (frame count, groups), so it
    is exchanged ONCE (two small all-gathers, no reduce, no pickling, any number of parts) and cached under a key every rank
    computes identically; every later call is a single RCCL all-gather of the payload."""

    def __init__(self, parts, dim, group, device):
        world, rank = _world(group)
        head = torch.zeros(12, dtype=torch.int64)
        head[0] = len(parts)
        if parts:
            ref = parts[0]
            assert ref.dim() <= 8
            head[1], head[2] = ref.dim(), _DTYPES.index(ref.dtype)
            head[3:3 + ref.dim()] = torch.tensor(ref.shape)
        heads = _gather_ints(head, group, device)
        nmax = max(1, int(heads[:, 0].max()))
        mine = torch.zeros(nmax, dtype=torch.int64)
        for j, q in enumerate(parts):
            mine[j] = q.shape[dim]
        sizes = _gather_ints(mine, group, device)
        self.sizes = [[int(sizes[r, j]) for j in range(int(heads[r, 0]))] for r in range(world)]
        owner = next((r for r in range(world) if int(heads[r, 0]) > 0), None)
        if owner is None:
            raise ValueError("ragged all-gather: no rank owns a part (empty chunk / frame list on every rank)")
        nd = int(heads[owner, 1])
        self.dtype = _DTYPES[int(heads[owner, 2])]
        self.pad_shape = [int(v) for v in heads[owner, 3:3 + nd]]
        self.pad_shape[dim] = max(sum(sz) for sz in self.sizes)
        self.own = [q.shape[dim] for q in parts]


def _all_gather_ragged(parts, dim, group, cache=None, key=None):
    """all-gather per-rank lists of tensors whose sizes differ along `dim` -> list (per rank) of lists of tensors."""
    world, rank = _world(group)
    if parts:
        device = parts[0].device
    else:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    plan = cache.get(key) if cache is not None and key is not None else None
    if plan is None:
        plan = _RaggedPlan(parts, dim, group, device)
        if cache is not None and key is not None:
            cache[key] = plan
    assert plan.own == [q.shape[dim] for q in parts], "ragged-gather plan does not match this call (key collision)"
    # every rank sends ONE buffer padded along `dim` to the largest per-rank total.  When all ranks hold the same total (the cfg3
    # layout: 8 chunks on 8 ranks, 24 decode groups on 8 ranks) nothing is padded, and the pad region is never read otherwise, so
    # the buffer is not zero-filled either way.
    total = sum(plan.own)
    if parts and len(parts) == 1 and total == plan.pad_shape[dim] and parts[0].is_contiguous():
        buf = parts[0]                                   # one part of full size: send it as it is
    else:
        buf = torch.empty(plan.pad_shape, dtype=plan.dtype, device=device)
        off = 0
        for q in parts:
            buf.narrow(dim, off, q.shape[dim]).copy_(q)
            off += q.shape[dim]
    if dist.get_backend(group) == "gloo" and buf.is_cuda:   # plumbing tests on a 1-GPU box: stage through the host
        host = buf.cpu()
        houts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(houts, host, group=group)
        outs = [o.to(device) for o in houts]
    else:
        outs = _all_gather_one(buf.contiguous(), world, group)      # ONE RCCL all-gather over xGMI (C1 / C2) into one allocation
    result = []
    for r in range(world):
        off, lst = 0, []
        for sz in plan.sizes[r]:
            lst.append(outs[r].narrow(dim, off, sz))
            off += sz
        result.append(lst)
    return result


class ChunkSharder:
    """chunk_executor for GaussianDiffusion.sample_sr: rank r denoises chunks r, r+world, ... of every solver step."""

    def __init__(self, group=None):
        self.group = group
        self._plans = {}

    def __call__(self, run_chunk, n_chunks, key=None):
        """key: anything every rank computes identically and that fixes the part sizes (sample_sr passes the latent shape and
        the chunk list); the size exchange then happens in the first solver step only."""
        world, rank = _world(self.group)
        mine = [run_chunk(i) for i in range(rank, n_chunks, world)]
        per_rank = _all_gather_ragged(mine, dim=2, group=self.group, cache=self._plans, key=key)
        cores = [None] * n_chunks
        for r in range(world):
            for j, i in enumerate(range(r, n_chunks, world)):
                cores[i] = per_rank[r][j]
        return cores


class FrameSharder:
    """shards independent frame groups (VAE decode groups of <= 3 frames) over ranks, then all-gathers the frames."""

    def __init__(self, group=None):
        self.group = group

    def map_groups(self, groups, fn):
        world, rank = _world(self.group)
        mine = [fn(a, b) for (a, b) in groups[rank::world]]
        per_rank = _all_gather_ragged(mine, dim=0, group=self.group)   # once per video: no plan cache needed
        out = [None] * len(groups)
        for r in range(world):
            for j, i in enumerate(range(r, len(groups), world)):
                out[i] = per_rank[r][j]
        return torch.cat(out)


def replica_items(items, rank=None, world=None, group=None):
    """Prompt-level data parallelism of the CogVideoX sampler (cogvideox-based/sat/sample_sr.py:38-45: `read_from_file(p, rank,
    world_size)`): yields `(item, cnt)` for every `cnt` with `cnt % world == rank`, in order; `items` is any iterable (the lines
    of the prompt file, input videos).  rank / world default to the process group's (1 process: everything).  No collective."""
    if world is None or rank is None:
        if dist.is_available() and dist.is_initialized():
            world, rank = _world(group)
        else:
            world, rank = 1, 0
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"replica_items: rank {rank} outside world {world}")
    for cnt, item in enumerate(items):
        if cnt % world == rank:
            yield (item.strip() if isinstance(item, str) else item), cnt


def gather_frames(frames, group=None, source=None, ctx=None):
    """C1: all-gather of each rank's decoded clip -> list of world tensors (same shape on all ranks).

    frames: the pipeline output [1, 3, F, H, W] fp32 (any dtype / shape is gathered as it is).  With `source` (the rank's
    low-resolution clip [F, 3, h, w] in [-1, 1]) the frames first go through tensor2vid + adain_color_fix + `.astype('uint8')`
    ON THE DEVICE (star_color_fix_u8; inference_sr.py:47-48 + inference_utils.py:92 -- what is done to them before they are saved
    anyway) and the uint8 [F, H, W, 3] frames are gathered: a quarter of the bytes over xGMI (157 MB instead of 628 MB per rank at
    cfg2, 1.26 GB instead of 5 GB landed per rank at N = 8)."""
    world, _ = _world(group)
    if source is not None:
        from . import frames as _frames
        c = ctx if ctx is not None else _frames.context_for(frames.device)
        c.use_current_stream()   # the collective below is ordered against torch's CURRENT stream: the colour fix must run on it too
        frames = c.color_fix(frames.float(), source.float(), as_uint8=True)
    frames = frames.contiguous()
    if dist.get_backend(group) == "gloo" and frames.is_cuda:   # plumbing tests on a 1-GPU box: stage through the host
        host = frames.cpu()
        outs = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(outs, host, group=group)
        return [o.to(frames.device) for o in outs]
    return _all_gather_one(frames, world, group)
