"""bench.py's N > 1 launch contract on a 1-GPU box: two ranks launched by torch.distributed.run (the form the task contract gives
for the driver) and by plain `python bench.py --gpus 2` (bench.py then launches itself), both on cuda:0, gloo instead of RCCL (plumbing only: rendezvous, barriers, max-over-ranks timing, the C1 gather of the replica
mode and the chunk / frame sharding of --config cfg3), reduced-width model."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(extra, self_launch=False):
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--small", "--share-gpu0", "--dist-backend", "gloo",
            "--no-cpu-baseline"] + extra
    cmd = [sys.executable] + tail if self_launch else \
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + tail
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints ONE json line
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_rank_replicas_weak_scaling_line():
    d = _run(["--frames", "4", "--height", "32", "--width", "32", "--denoise-steps", "2"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["output_finite"] and d["value"] > 0
    assert abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"] + 1e-9        # whole-job frames / max-over-ranks time


@pytest.mark.gpu
def test_two_rank_sharded_long_video_cfg3_layout():
    d = _run(["--config", "cfg3", "--frames", "20", "--max-chunk-len", "8", "--height", "32", "--width", "32",
              "--solver-mode", "normal", "--denoise-steps", "2"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["output_finite"]
    assert abs(d["value"] - 20 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"] + 1e-9       # ONE video: frames are not multiplied by ranks


@pytest.mark.gpu
def test_plain_python_bench_gpus_2_launches_itself():
    """`python bench.py --gpus 2` without a launcher (round-2 review: it used to die on an assert before touching a GPU)."""
    d = _run(["--frames", "4", "--height", "32", "--width", "32", "--denoise-steps", "2"], self_launch=True)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["output_finite"] and d["value"] > 0
