"""libeqlb_amd.so loaded BEFORE torch is imported: both must see the device and share pointers.
Run on a GPU box: python tools/check_load_order.py [pybind]  (exit code 0 = ok).  With `pybind` the library
comes in through the pybind11 module dolfinx_eqlb_amd._cpp (the front ends FluxEqlbSE / FluxEqlbEV /
local_projection) instead of the ctypes binding."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402

from dolfinx_eqlb_amd import cpp  # noqa: E402
from dolfinx_eqlb_amd.mesh import create_unit_square  # noqa: E402
from synthetic import facet_types, make_compatible_data  # noqa: E402

assert "torch" not in sys.modules
if len(sys.argv) > 1 and sys.argv[1] == "pybind":
    # front end first: reconstruct_fluxes_semiexplt through _cpp, before cpp.lib() was ever called
    from dolfinx_eqlb_amd.eqlb import FluxEqlbSE
    m0 = create_unit_square(4)
    ft0 = facet_types(m0, None)
    G0, f0 = make_compatible_data(m0, 2, ft0, seed=3)
    eq0 = FluxEqlbSE(2, m0, [f0], [G0])
    eq0.set_boundary_conditions([m0.boundary_facets()], [[]])
    eq0.equilibrate_fluxes()
    assert "torch" not in sys.modules
assert cpp.device_count() >= 1, "library sees no device"
mesh = create_unit_square(8)
ft = facet_types(mesh, None)
G, f = make_compatible_data(mesh, 2, ft, seed=3)
eq = cpp.SemiExplicitEquilibrator(cpp.DeviceMesh(mesh), 2, 1)
eq.set_boundary(ft)
ref = eq.equilibrate_host(G[None], f[None])

import torch  # noqa: E402  (after the library on purpose)

assert torch.cuda.is_available(), "torch sees no device after libeqlb_amd.so was loaded first"
dev = torch.device("cuda:0")
tG = torch.from_numpy(np.ascontiguousarray(G[None])).to(dev)
tf = torch.from_numpy(np.ascontiguousarray(f[None])).to(dev)
tx = torch.zeros(ref.shape, dtype=torch.float64, device=dev)
eq.equilibrate_device(tG.data_ptr(), tf.data_ptr(), tx.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
assert np.array_equal(tx.cpu().numpy(), ref), "device-pointer call differs from the host-array call"
print("load order ok")
