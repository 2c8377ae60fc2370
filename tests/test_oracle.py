"""Pin the CPU oracle (oracle/unet_oracle.py) against fixtures produced by the reference's own code
(oracle/make_golden.py), and -- where /root/reference exists -- against the live reference modules."""
import glob
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import unet_oracle as O  # noqa: E402
from make_golden import unet_inputs  # noqa: E402
from star_amd.topology import SMALL_TEST_CONFIG, random_state_dict  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def small_sd():
    return random_state_dict(SMALL_TEST_CONFIG, seed=0)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "unet_small_f*.pt"))), ids=os.path.basename)
def test_oracle_unet_matches_reference_golden(path, small_sd):
    gold = torch.load(path)
    f, h, w, seed = gold["case"]
    if f > 16 and os.environ.get("STAR_SLOW") != "1":
        pytest.skip("long case; STAR_SLOW=1")
    x, t, y, hint = unet_inputs(SMALL_TEST_CONFIG, f, h, w, seed)
    out = O.unet_forward(small_sd, SMALL_TEST_CONFIG, x, t, y, hint)
    err = float((out - gold["out"]).abs().max())
    assert err < 2e-4 * max(1.0, float(gold["out"].abs().max())), err


def test_oracle_control_residuals_match_reference_golden(small_sd):
    """the 13 residuals of the reference's VideoControlNet.forward (row a2)."""
    gold = torch.load(os.path.join(GOLD, "unet_small_control_f3_18x16.pt"))
    f, h, w, seed = gold["case"]
    x, t, y, hint = unet_inputs(SMALL_TEST_CONFIG, f, h, w, seed)
    from star_amd.topology import build_blocks
    res = O.control_net_forward(small_sd, SMALL_TEST_CONFIG, build_blocks(SMALL_TEST_CONFIG, control=True), x, t, y, hint)
    assert len(res) == len(gold["residuals"]) == 13
    for i, (a, b) in enumerate(zip(res, gold["residuals"])):
        assert a.shape == b.shape, i
        assert float((a - b).abs().max()) < 2e-4 * max(1.0, float(b.abs().max())), i


def test_oracle_blocks_match_reference_golden():
    blocks = torch.load(os.path.join(GOLD, "blocks.pt"))
    for name, b in blocks.items():
        sd = {"m." + k: v for k, v in b["sd"].items()}
        x = b["x"]
        if b["kind"] == "res":
            y = O.res_block(sd, "m", x, b["emb"], 1)
        elif b["kind"] == "st":
            y = O.spatial_transformer(sd, "m", x, b["context"], 2)
        elif b["kind"] == "tt":
            f, c, h, w = x.shape
            y = O.temporal_transformer(sd, "m", x.reshape(1, f, c, h, w).permute(0, 2, 1, 3, 4), 2)
            y = y.permute(0, 2, 1, 3, 4).reshape(f, c, h, w)
        elif b["kind"] == "down":
            y = O.downsample(sd, "m", x)
        else:
            y = O.upsample(sd, "m", x)
        err = float((y - b["y"]).abs().max())
        assert err < 1e-4 * max(1.0, float(b["y"].abs().max())), (name, err)
