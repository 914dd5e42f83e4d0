import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.build()
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def bm():
    """The product package; loading it binds the HIP library (raises if it is missing)."""
    import brickmap_amd
    brickmap_amd.load()
    return brickmap_amd


@pytest.fixture(scope="session")
def world256(orc):
    w = orc.World(256, 256)
    w.reset_device(True)
    return w


def build_fake_rccl(directory):
    """tests/fake_rccl.cpp -> <directory>/libfake_rccl.so: the host-staged stand-in for the RCCL entry points csrc/comm.hip binds
    (BM_RCCL_LIBRARY), which lets several ranks share one GPU in the multi-rank tests of the C-ABI exchange."""
    import subprocess
    out = os.path.join(str(directory), "libfake_rccl.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", "-Wno-unused-value", "-Wno-unused-result",
                           "-x", "hip", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "fake_rccl.cpp"), "-o", out, "-lrt", "-lpthread"])
    return out
