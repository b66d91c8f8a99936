#!/usr/bin/env python3
"""CPU-side cost of enqueueing one sweep (ctypes call into libeqlb_amd.so) and of the halo kernels'
calls, measured without waiting for the device: is a multi-GPU step host-bound?"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from dolfinx_eqlb_amd import cpp
    from dolfinx_eqlb_amd import distributed as dd
    from synthetic import make_compatible_data
    n, k = 100, 2
    part = dd.StripPartition(n, 0, 1)
    mesh = part.mesh
    ft = part.facet_types()
    G, f = make_compatible_data(mesh, k, ft, seed=1)
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    eq = cpp.SemiExplicitEquilibrator(cpp.DeviceMesh(mesh), k, 1)
    eq.set_boundary(ft)
    dG, df = torch.from_numpy(G).to(dev), torch.from_numpy(f).to(dev)
    x = torch.zeros(mesh.ncells * 8, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    idx = torch.arange(0, 300, dtype=torch.int64, device=dev)
    buf = torch.zeros(300 * 8, dtype=torch.float64, device=dev)
    for _ in range(10):
        eq.equilibrate_device(dG.data_ptr(), df.data_ptr(), x.data_ptr(), stream)
    torch.cuda.synchronize()
    for name, fn in (
        ("equilibrate_device", lambda: eq.equilibrate_device(dG.data_ptr(), df.data_ptr(), x.data_ptr(), stream)),
        ("halo_pack", lambda: cpp.halo_pack(x.data_ptr(), idx.data_ptr(), buf.data_ptr(), 1, 300, 8, mesh.ncells, True, stream)),
        ("halo_unpack_add", lambda: cpp.halo_unpack_add(x.data_ptr(), idx.data_ptr(), buf.data_ptr(), 1, 300, 8, mesh.ncells, stream)),
        ("torch index_add (reference)", lambda: x.view(-1, 8).index_add_(0, idx, buf.view(-1, 8))),
    ):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name:32s} enqueue {1e6 * (t1 - t0) / 300:7.1f} us/call   (drain {1e3 * (t2 - t1):.2f} ms)")


if __name__ == "__main__":
    main()
