import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mappy-rs_amd"))

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The product leaves mm_update_extra's walk and the cs string to the device (k_extra) for batches of >= 1024 reads and walks on the host below
# that (latency).  The parity tests map small batches: run them through the device form (every mapping test then checks k_extra's mlen / blen /
# NM / MAPQ / cs against the oracle); test_gpu_map.py::test_update_extra_host_walk switches back to cover the host walk.
os.environ.setdefault("MM355_EXTRA_MIN_READS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def built():
    """make sure the oracle and the HIP library are built (hipcc cross-compiles without a GPU)"""
    import __graft_entry__ as ge
    ge.build()
    return True
