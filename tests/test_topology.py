"""The data-driven architecture description must reproduce the reference's state dict exactly."""
import json
import os

import pytest
import torch

from star_amd.topology import SMALL_TEST_CONFIG, UNetConfig, param_shapes, random_state_dict

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "unet_state_dict_shapes.json")


def test_param_shapes_match_reference_state_dict():
    """golden = {key: shape} of reference ControlledV2VUNet().state_dict() (oracle/make_golden.py, meta device)."""
    gold = json.load(open(GOLDEN))
    mine = param_shapes(UNetConfig())
    assert len(gold) == 2247
    assert set(mine.keys()) == set(gold.keys())
    for k, shp in gold.items():
        assert tuple(shp) == tuple(mine[k]), k
    n = sum(int(torch.Size(s).numel()) for s in mine.values())
    assert abs(n / 1e6 - 2041.1) < 0.1


def test_random_state_dict_is_order_independent_and_seeded():
    a = random_state_dict(SMALL_TEST_CONFIG, seed=3)
    b = random_state_dict(SMALL_TEST_CONFIG, seed=3)
    c = random_state_dict(SMALL_TEST_CONFIG, seed=4)
    k = "middle_block.1.transformer_blocks.0.attn1.to_q.weight"
    assert torch.equal(a[k], b[k]) and not torch.equal(a[k], c[k])
    assert all(torch.isfinite(v).all() for v in a.values())
