"""Soak test: repeated handle creation / set_boundary / equilibrate on one device; device memory must
return to its starting level and results must stay bitwise identical.  usage: python tools/soak.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from dolfinx_eqlb_amd import cpp
from dolfinx_eqlb_amd.mesh import create_unit_square
from synthetic import facet_types, make_compatible_data

torch.cuda.init()
mesh = create_unit_square(60, shuffle_seed=1, perturb=0.2)
ft = facet_types(mesh, lambda x: x[:, 0] < 1e-12)
G, f = make_compatible_data(mesh, 2, ft)
free0 = torch.cuda.mem_get_info()[0]
ref = None
for it in range(60):
    dm = cpp.DeviceMesh(mesh)
    import os
    kinds = [int(c) for c in os.environ.get("SOAK_KINDS", "012")]
    for kind in kinds:
        if kind == 0:
            eq = cpp.SemiExplicitEquilibrator(dm, 2, 1)
        elif kind == 1:
            eq = cpp.SemiExplicitEquilibrator(dm, 2, 1)
            eq.set_option("scatter", 0)
        else:
            eq = cpp.ConstrainedMinEquilibrator(dm, 2, 1)
        for rep in range(2):
            eq.set_boundary(ft)
            x = eq.equilibrate_host(G[None], f[None])
        if kind == kinds[0]:
            if ref is None:
                ref = x.copy()
            assert np.array_equal(ref, x), "result changed between iterations"
        eq.close()
    dm.close()
    if it == 0:  # code objects, runtime heaps of the first use are a fixed cost
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("free before %.1f MB, after %.1f MB, delta %.2f MB" % (free0 / 2**20, free1 / 2**20, (free0 - free1) / 2**20))
assert free0 - free1 < 64 * 2**20, "device memory leak"
print("soak OK")
