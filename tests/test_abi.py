"""The C-ABI library loads and exports every symbol include/eqlb.h declares (no compute
calls: there is no GPU on the CPU test box)."""

import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "eqlb.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(eqlb_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    from dolfinx_eqlb_amd import cpp
    assert _declared_symbols() == sorted(cpp.EXPORTED_SYMBOLS)


def test_library_exports_all_symbols():
    from dolfinx_eqlb_amd import cpp
    if not os.path.exists(cpp.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = cpp.lib()
    for s in _declared_symbols():
        assert hasattr(lib, s), s


def test_reference_tables_from_library_match_generator():
    """Host-only entry point: the tensors compiled into the library are the generated ones."""
    import numpy as np
    from dolfinx_eqlb_amd import cpp
    from gen_tables import tables_float
    for (k, deg) in [(1, 0), (2, 1), (3, 2)]:
        t = tables_float(k, deg)
        for name in "SFHD":
            assert np.array_equal(cpp.get_reference_table(k, deg, name), t[name])


def test_no_device_fails_loudly():
    """Without a HIP device the product path must raise, never fall back to the CPU."""
    from dolfinx_eqlb_amd import cpp
    from dolfinx_eqlb_amd.mesh import create_unit_square
    if cpp.device_count() > 0:
        pytest.skip("a device is visible")
    with pytest.raises(RuntimeError):
        cpp.DeviceMesh(create_unit_square(2))


def test_product_does_not_import_oracle():
    """Nothing under dolfinx_eqlb_amd/ (the product) may reference the oracle."""
    pkg = os.path.join(ROOT, "dolfinx_eqlb_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in txt.lower() or fn == "eqlb_tables_gen.h", (dirpath, fn)
