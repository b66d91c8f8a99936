"""PCIe-inclusive rate of the drop-in call with host buffers (EQLB_MEM_HOST): H2D of G, f, sigma,
kernel, D2H of sigma.  Not the headline (bench.py keeps the data resident).  usage: python tools/host_rate.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from dolfinx_eqlb_amd import cpp
from dolfinx_eqlb_amd import distributed as dd
from synthetic import make_compatible_data
part = dd.StripPartition(500, 0, 1)
mesh, ft = part.mesh, part.facet_types()
G, f = make_compatible_data(mesh, 2, ft)
eq = cpp.SemiExplicitEquilibrator(cpp.DeviceMesh(mesh), 2, 1)
eq.set_boundary(ft)
x = np.zeros((1, mesh.ncells * 8))
for _ in range(3):
    eq.equilibrate_host(G[None], f[None], x)
t = []
for _ in range(10):
    t0 = time.perf_counter()
    eq.equilibrate_host(G[None], f[None], x)
    t.append(time.perf_counter() - t0)
best = min(t)
print("host-buffer call: %.3f ms (best of 10) = %.3e patches/s; bytes moved over PCIe: %.0f MB" %
      (best * 1e3, mesh.nnodes / best, (G.nbytes + f.nbytes + 2 * x.nbytes) / 1e6))
