import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


def pytest_collection_finish(session):
    """One GPU test (tests/test_gpu_halo.py) moves device buffers with torch.  torch bundles its own
    HIP runtime; it has to be the FIRST runtime initialised in the process (libeqlb_amd.so then binds
    to the loaded one, as in bench.py) - a runtime loaded after ours sees no device."""
    if any(item.get_closest_marker("gpu") for item in session.items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:  # no torch / no device: the tests that need them report it themselves
            pass
