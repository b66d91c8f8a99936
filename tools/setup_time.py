import time, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dolfinx_eqlb_amd import cpp
from dolfinx_eqlb_amd import distributed as dd
part = dd.StripPartition(500, 0, 1)
mesh = part.mesh
ft = part.facet_types()
t0 = time.perf_counter(); dm = cpp.DeviceMesh(mesh); t1 = time.perf_counter()
eq = cpp.SemiExplicitEquilibrator(dm, 2, 1); t2 = time.perf_counter()
eq.set_boundary(ft); t3 = time.perf_counter()
eq.set_boundary(ft); t4 = time.perf_counter()
print("mesh upload %.1f ms, create %.1f ms, set_boundary %.1f ms (2nd %.1f ms)" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3))
