"""Weight repacking helpers shared by the loader and the tests."""
import torch


def geglu_interleave(n_half):
    """Row permutation of a GEGLU projection [2*n_half, K] (value rows then gate rows, unet_v2v.py:500-504)
    into alternating 32-row (value, gate) blocks, the layout the GEMM's GEGLU epilogue pairs in registers."""
    assert n_half % 32 == 0
    idx = []
    for blk in range(n_half // 32):
        idx += list(range(blk * 32, blk * 32 + 32))
        idx += list(range(n_half + blk * 32, n_half + blk * 32 + 32))
    return torch.tensor(idx, dtype=torch.long)
