"""Randomised shape sweep of the hand-scheduled kernels on the SIMT emulator (tools/fuzz_emu.py): a fixed seed, 8 cases per kernel
family -- persistent tile 18 and strided operand views (bit for bit against the plain 8-wave tile), the fused temporal projection +
attention (bit for bit against the two-kernel path), the tail split under the three gather modes, the A-stationary kernel (<= 1 ulp),
the two attention kernels on ragged lengths (fp32 softmax, <= 4 ulp).  The hand-picked edges live in test_kernels.py; this draws the
shapes (700 cases of the same generator: profiles/r04_fuzz_emu.txt)."""
import os
import random
import sys

import pytest
import torch

from util import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_emu as FZ  # noqa: E402

KINDS = {"persist": FZ.fuzz_persist, "strided": FZ.fuzz_strided_persist, "tq": FZ.fuzz_tq, "conv": FZ.fuzz_conv_split,
         "astat": FZ.fuzz_astat, "attn": FZ.fuzz_attn, "tattn": FZ.fuzz_tattn}


@pytest.mark.parametrize("kind", sorted(KINDS))
def test_random_shapes_on_the_emulator(kind, emu_lib):
    from star_amd import lib as L
    rng = random.Random(4000 + sorted(KINDS).index(kind))
    n = 3 if kind == "conv" else 8            # a conv case is > 4096 rows x 576 k on the emulator: seconds each
    for i in range(n):
        dtype = rng.choice([torch.float16, torch.bfloat16])
        ctx = L.Context(0, dtype, emu_lib)
        ok, what = KINDS[kind](ctx, dtype, rng, 1e8)
        assert ok, what
