"""libeqlb_amd.so loaded before torch.  cpp.lib() maps torch's bundled HIP runtime first,
so both orders give one runtime; checked in a fresh interpreter (this process already has torch)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("entry", ["ctypes", "pybind"])
def test_library_before_torch(entry):
    """entry = pybind: the library comes in through dolfinx_eqlb_amd._cpp (front ends), ADVICE r2."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_load_order.py")]
                       + (["pybind"] if entry == "pybind" else []),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "load order ok" in r.stdout
