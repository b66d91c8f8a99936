#!/usr/bin/env python3
"""Data point off the headline: unstructured Delaunay mesh of random points (mean valence 6, valences 3 ... 12 mixed: the P = 8
lane groups run with two idle lanes and the generic patch body), RT_2, ~1M triangles."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from dolfinx_eqlb_amd import cpp
    from dolfinx_eqlb_amd.eqlb import check_eqlb_conditions as chk
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_unstructured import delaunay_mesh
    from synthetic import facet_types, make_compatible_data
    k, n = 2, 500000
    mesh = delaunay_mesh(n, seed=1)
    ft = facet_types(mesh)
    G, f = make_compatible_data(mesh, k, ft, seed=1)
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    eq = cpp.SemiExplicitEquilibrator(cpp.DeviceMesh(mesh), k, 1)
    eq.set_boundary(ft)
    dG, df = torch.from_numpy(G).to(dev), torch.from_numpy(f).to(dev)
    x = torch.zeros(mesh.ncells * 8, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    eq.equilibrate_device(dG.data_ptr(), df.data_ptr(), x.data_ptr(), stream)
    torch.cuda.synchronize()
    res, nrm = chk.divergence_residual(mesh, k, x.cpu().numpy(), G, f)
    for _ in range(3):
        eq.equilibrate_device(dG.data_ptr(), df.data_ptr(), x.data_ptr(), stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 50
    for _ in range(steps):
        eq.equilibrate_device(dG.data_ptr(), df.data_ptr(), x.data_ptr(), stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"Delaunay mesh of {n} random points: {mesh.ncells} cells, {eq.num_patches} patches, {1e3 * dt:.4f} ms/step, "
          f"{eq.num_patches / dt:.3e} patches/s, rel. divergence residual {res / nrm:.2e}, tiling {eq.tiling_info()}")


if __name__ == "__main__":
    main()
