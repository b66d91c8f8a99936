import sys, numpy as np
sys.path[:0]=['/root/repo','/root/repo/tests']
from cases import make_case
from dolfinx_eqlb_amd import cpp
from oracle import oracle
mesh, ft, G, f = make_case(2, 2, "dirichlet", shuffle=None, perturb=0.0)
dm = cpp.DeviceMesh(mesh); eq = cpp.SemiExplicitEquilibrator(dm, 2, 1); eq.set_boundary(ft)
try:
    x = eq.equilibrate_host(G, f)
    ref = oracle.se_reconstruct(mesh, 2, ft, G, f)
    print("err", np.abs(x-ref).max(), np.abs(ref).max())
except Exception as e:
    print("EXC", e)
