#!/usr/bin/env python3
"""Data point off the headline: unstructured Delaunay mesh of random points (mean valence 6, valences 3 ... 12 mixed: the P = 8
lane groups run with two idle lanes and the generic patch body), ~1M triangles.
python tools/bench_unstructured.py [--k K] [--stress] [--points N]"""
import argparse
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=2)
    ap.add_argument("--stress", action="store_true")
    ap.add_argument("--points", type=int, default=500000)
    args = ap.parse_args()
    k, n = args.k, args.points
    mesh = delaunay_mesh(n, seed=1)
    ft = facet_types(mesh)
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    if args.stress:
        from synthetic import make_compatible_stress_data
        ft = np.repeat(ft, 2, axis=0)
        G, f = make_compatible_stress_data(mesh, k, ft)
        eq = cpp.SemiExplicitEquilibrator(cpp.DeviceMesh(mesh), k, 2, reconstruct_stress=True)
    else:
        G, f = make_compatible_data(mesh, k, ft, seed=1)
        eq = cpp.SemiExplicitEquilibrator(cpp.DeviceMesh(mesh), k, 1)
    eq.set_boundary(ft)
    dG, df = torch.from_numpy(np.ascontiguousarray(G)).to(dev), torch.from_numpy(np.ascontiguousarray(f)).to(dev)
    nr = 2 if args.stress else 1
    x = torch.zeros(nr * mesh.ncells * k * (k + 2), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    eq.equilibrate_device(dG.data_ptr(), df.data_ptr(), x.data_ptr(), stream)
    torch.cuda.synchronize()
    x0 = x.cpu().numpy().reshape(nr, -1)
    res, nrm = chk.divergence_residual(mesh, k, x0[0], np.asarray(G).reshape(nr, -1)[0], np.asarray(f).reshape(nr, -1)[0])
    x.zero_()
    for _ in range(3):
        eq.equilibrate_device(dG.data_ptr(), df.data_ptr(), x.data_ptr(), stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 50
    for _ in range(steps):
        eq.equilibrate_device(dG.data_ptr(), df.data_ptr(), x.data_ptr(), stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    what = ("stress " if args.stress else "") + f"RT_{k}"
    print(f"{what}: Delaunay mesh of {n} random points: {mesh.ncells} cells, {eq.num_patches} patches, {1e3 * dt:.4f} ms/step, "
          f"{eq.num_patches / dt:.3e} patches/s, rel. divergence residual {res / nrm:.2e}, tiling {eq.tiling_info()}")


if __name__ == "__main__":
    main()
