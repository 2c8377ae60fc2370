"""tools/cbench (the torch-free timing harness over the C ABI) against the emulator build: the spec parser, every kernel family it
drives, and its on-device bit comparison of tiles.  The GPU build of the same source is what the profiles/r04_cbench_*.txt tables come
from."""
import os
import subprocess

from util import ROOT

EMU = os.path.join(ROOT, "tools", "hostemu", "libstar_emu.so")
BIN = os.path.join(ROOT, "tools", "cbench", "cbench_emu")


def test_cbench_runs_every_kind_on_the_emulator(emu_lib, tmp_path):
    if not os.path.isfile(BIN):
        subprocess.check_call(["make", "tools/cbench/cbench_emu"], cwd=ROOT)
    spec = tmp_path / "spec.txt"
    spec.write_text("# comment\n"
                    "gemm 70 128 320 37 30,1,38\n"          # folded LayerNorm + GEGLU: A-stationary kernel, 8-wave tile, a scheduling variant
                    "gemm 300 256 128 3 1,18,3\n"           # bias + residual: 8-wave tile, persistent tile, 128 x 128
                    "conv 1 6 5 64 64 0\n"
                    "tconv 2 9 64 0\n"
                    "attn 1 1 40 70\n"
                    "tq 3 2\n"
                    "bogus 1 2 3\n")
    out = subprocess.run([BIN, EMU, "f16", str(spec), "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().split("\n")
    assert sum("bit-identical" in l for l in lines) == 4 and not any("differ" in l and "bit-identical" not in l for l in lines), out.stdout
    assert not any("FAILED" in l for l in lines), out.stdout
    for kind in ("gemm 70x128x320", "conv3x3 1x6x5", "tconv F=2", "attn B=1", "tq F=3"):
        assert any(l.startswith(kind) and "mean" in l for l in lines), (kind, out.stdout)
    assert any(l.startswith("unknown spec") for l in lines)
    # a missing library / the wrong build are refused
    assert subprocess.run([BIN, "/nonexistent.so", "f16", str(spec)], capture_output=True).returncode == 2
