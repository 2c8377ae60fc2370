"""Multi-GPU execution of the hot path: one process per GPU, torch.distributed over RCCL/xGMI
(backend "nccl" on ROCm; "gloo" in the CPU tests).  SURVEY.md section 8(e).

The path shards along the frame-chunk axis:
  * independent clips (the 1/2/4/8-GPU frames/s metric): replicas, no data-path collective; `gather_frames`
    (C1) all-gathers the decoded frames at the end;
  * one long video (cfg3): inside every solver step the overlapping chunks are independent (diffusion_sdedit.py:
    337-350), so `ChunkSharder` gives chunk i to rank i % world and all-gathers the trimmed x0 cores (C2, a few MB)
    so that every rank applies the identical solver update to the full-length latent.  VAE decode groups of 3 frames
    are sharded the same way by `FrameSharder`.
No reduce-type collective exists anywhere on the path.
"""
import torch
import torch.distributed as dist


def _world(group):
    return dist.get_world_size(group), dist.get_rank(group)


def _all_gather_ragged(parts, dim, group):
    """all-gather per-rank lists of tensors whose sizes differ along `dim` -> list (per rank) of lists of tensors."""
    world, rank = _world(group)
    device = parts[0].device if parts else None
    counts = torch.zeros(world, 64, dtype=torch.int64)
    assert len(parts) <= 63
    counts[rank, 0] = len(parts)
    for j, p in enumerate(parts):
        counts[rank, 1 + j] = p.shape[dim]
    cdev = counts.to(device) if device is not None and device.type == "cuda" else counts
    dist.all_reduce(cdev, group=group)          # tiny metadata exchange (sizes only)
    counts = cdev.cpu()
    total = counts[:, 1:].sum(dim=1)
    mx = int(total.max())
    ref = None
    for p in parts:
        ref = p
    # every rank needs a template for shape/dtype even if it owns no part
    meta = [None]
    if ref is not None:
        shp = list(ref.shape)
        shp[dim] = 0
        meta = [(shp, ref.dtype)]
    gathered_meta = [None] * world
    dist.all_gather_object(gathered_meta, meta[0], group=group)
    shp, dtype = next(m for m in gathered_meta if m is not None)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    pad_shape = list(shp)
    pad_shape[dim] = mx
    buf = torch.zeros(pad_shape, dtype=dtype, device=device)
    if parts:
        cat = torch.cat(parts, dim=dim)
        buf.narrow(dim, 0, cat.shape[dim]).copy_(cat)
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)      # RCCL all-gather over xGMI (C1 / C2)
    result = []
    for r in range(world):
        n = int(counts[r, 0])
        sizes = [int(counts[r, 1 + j]) for j in range(n)]
        off, lst = 0, []
        for s in sizes:
            lst.append(outs[r].narrow(dim, off, s))
            off += s
        result.append(lst)
    return result


class ChunkSharder:
    """chunk_executor for GaussianDiffusion.sample_sr: rank r denoises chunks r, r+world, ... of every solver step."""

    def __init__(self, group=None):
        self.group = group

    def __call__(self, run_chunk, n_chunks):
        world, rank = _world(self.group)
        mine = [run_chunk(i) for i in range(rank, n_chunks, world)]
        per_rank = _all_gather_ragged(mine, dim=2, group=self.group)
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
        per_rank = _all_gather_ragged(mine, dim=0, group=self.group)
        out = [None] * len(groups)
        for r in range(world):
            for j, i in enumerate(range(r, len(groups), world)):
                out[i] = per_rank[r][j]
        return torch.cat(out)


def gather_frames(frames, group=None):
    """C1: all-gather of each rank's decoded clip [1, 3, F, H, W] -> list of world tensors (same shape on all ranks)."""
    world, _ = _world(group)
    if dist.get_backend(group) == "gloo" and frames.is_cuda:   # plumbing tests on a 1-GPU box: stage through the host
        host = frames.contiguous().cpu()
        outs = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(outs, host, group=group)
        return [o.to(frames.device) for o in outs]
    outs = [torch.empty_like(frames) for _ in range(world)]
    dist.all_gather(outs, frames.contiguous(), group=group)
    return outs
