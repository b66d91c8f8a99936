import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from dolfinx_eqlb_amd import cpp
from dolfinx_eqlb_amd.mesh import create_mesh, create_unit_square
from synthetic import facet_types
torch.cuda.init()
base = create_unit_square(250)
for aspect in (1.0, 30.0, 1000.0):
    xy = base.x[:, :2].copy(); xy[:, 0] *= aspect
    mesh = create_mesh(xy, base.cell_nodes)
    ft = facet_types(mesh)
    eq = cpp.SemiExplicitEquilibrator(cpp.DeviceMesh(mesh), 2, 1)
    t0 = time.perf_counter(); eq.set_boundary(ft); dt = time.perf_counter() - t0
    ti = eq.tiling_info()
    print(f"aspect {aspect}: patches {eq.num_patches}, instances {ti['patch_instances']} ({ti['patch_instances']/eq.num_patches:.3f}x), lane slots {ti['lane_slots']}, set_boundary {1e3*dt:.0f} ms")
