import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EMU_LIB = os.path.join(ROOT, "tools", "hostemu", "libstar_emu.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running emulator test (opt in with STAR_SLOW=1)")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("STAR_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="slow emulator test; set STAR_SLOW=1")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def emu_lib():
    """The SIMT-emulator build of the kernels (test tooling; checks index logic without a GPU)."""
    from star_amd import lib as L
    if not os.path.isfile(EMU_LIB):
        import subprocess
        subprocess.check_call(["make", "-j8", "emu"], cwd=ROOT)
    return L.Library(EMU_LIB)


@pytest.fixture(scope="session")
def hip_lib():
    from star_amd import lib as L
    return L.default_library()
