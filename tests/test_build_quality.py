"""Static checks on the built product library (no GPU): every gfx950 kernel in star_amd/libstar_hip.so is free of register
spills and scratch memory, and the hot kernels keep the occupancy their design assumes (two waves per SIMD = at most 256
registers per lane).  Parses the clang offload bundles of the .so and the AMDGPU metadata notes of each code object."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

from util import ROOT

LIB = os.path.join(ROOT, "star_amd", "libstar_hip.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _kernels():
    blob = open(LIB, "rb").read()
    out = {}
    for m in re.finditer(re.escape(MAGIC), blob):
        p = m.start()
        n = struct.unpack_from("<Q", blob, p + 24)[0]
        o = p + 32
        for _ in range(n):
            off, size, ts = struct.unpack_from("<QQQ", blob, o)
            o += 24
            triple = blob[o:o + ts].decode()
            o += ts
            if "gfx950" not in triple or size == 0:
                continue
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(blob[p + off:p + off + size])
                f.flush()
                notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
            for km in re.finditer(r"- \.agpr_count:.*?(?=\n\s+- \.agpr_count:|\namdhsa\.target|\Z)", notes, re.S):
                txt = km.group(0)
                name = re.search(r"\.name:\s+(\S+)", txt).group(1)
                get = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", txt).group(1))
                out[name] = dict(vgpr=get("vgpr_count"), spill=get("vgpr_spill_count"), sspill=get("sgpr_spill_count"),
                                 scratch=get("private_segment_fixed_size"), lds=get("group_segment_fixed_size"))
    return out


@pytest.fixture(scope="module")
def kernels():
    if not (os.path.isfile(LIB) and os.path.isfile(READELF)):
        pytest.skip("product library or llvm-readelf not available")
    k = _kernels()
    assert len(k) > 50, f"only {len(k)} kernels found in {LIB}"
    return k


def test_no_spills_no_scratch(kernels):
    bad = {n: v for n, v in kernels.items() if v["spill"] or v["scratch"]}
    assert not bad, f"kernels with register spills / scratch memory: {bad}"


def test_hot_kernels_keep_two_waves_per_simd(kernels):
    """the 8-wave GEMM tiles and the attention kernel are designed for two waves per SIMD: <= 256 registers per lane"""
    hot = {n: v for n, v in kernels.items() if "flash_attn_v5_kernel" in n or re.search(r"gemm_kernelID.*Li256ELi(256|320)ELi4ELi2E", n)}
    assert len(hot) >= 20
    over = {n: v["vgpr"] for n, v in hot.items() if v["vgpr"] > 256}
    assert not over, over
