#!/usr/bin/env python3
"""The N > 1 step of bench.py on ONE device without RCCL: strip 0 of a 2-rank partition, priority
tiles + halo pack + remaining tiles + unpack, against the single-launch step + pack + unpack.
Prices the split into two launches (the transfer itself is what it is meant to hide)."""
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
    n, k, nrt = 500, 2, 8
    part = dd.StripPartition(n, 0, 2)
    mesh, ft = part.mesh, part.facet_types()
    G, f = make_compatible_data(mesh, k, ft, seed=1)
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    dG, df = torch.from_numpy(G).to(dev), torch.from_numpy(f).to(dev)
    sidx = torch.from_numpy(part.send_cells).to(dev)
    buf = torch.zeros(sidx.numel() * nrt, dtype=torch.float64, device=dev)
    out = {}
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    for mode in ("single", "two_phase", "two_phase_512", "two_stream"):
        eq = cpp.SemiExplicitEquilibrator(cpp.DeviceMesh(mesh), k, 1)
        if mode != "single":
            eq.set_priority_cells(part.send_cells)
        eq.set_boundary(ft, node_mask=part.node_mask)
        nprio = eq.num_priority_tiles
        if mode == "two_phase_512":  # a full first round of workgroups instead of the priority tiles alone
            nprio = max(nprio, min(512, eq.tiling_info()["ntiles"] // 2))
        x = torch.zeros(mesh.ncells * nrt, dtype=torch.float64, device=dev)

        def step():
            if mode == "two_stream":  # priority tiles + pack on a side stream, beside the remaining tiles
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    eq.equilibrate_device_tiles(dG.data_ptr(), df.data_ptr(), x.data_ptr(), 0, nprio, side.cuda_stream)
                    cpp.halo_pack(x.data_ptr(), sidx.data_ptr(), buf.data_ptr(), 1, sidx.numel(), nrt, mesh.ncells, True,
                                  side.cuda_stream)
                eq.equilibrate_device_tiles(dG.data_ptr(), df.data_ptr(), x.data_ptr(), nprio, -1, stream)
                main.wait_stream(side)
                return
            if mode != "single":
                eq.set_option("tile_first", 0)
                eq.set_option("tile_count", nprio)
                eq.equilibrate_device(dG.data_ptr(), df.data_ptr(), x.data_ptr(), stream)
                cpp.halo_pack(x.data_ptr(), sidx.data_ptr(), buf.data_ptr(), 1, sidx.numel(), nrt, mesh.ncells, True, stream)
                eq.set_option("tile_first", nprio)
                eq.set_option("tile_count", -1)
                eq.equilibrate_device(dG.data_ptr(), df.data_ptr(), x.data_ptr(), stream)
            else:
                eq.equilibrate_device(dG.data_ptr(), df.data_ptr(), x.data_ptr(), stream)
                cpp.halo_pack(x.data_ptr(), sidx.data_ptr(), buf.data_ptr(), 1, sidx.numel(), nrt, mesh.ncells, True, stream)

        step()
        torch.cuda.synchronize()
        out[mode] = (x.cpu().numpy().copy(), buf.cpu().numpy().copy())
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
        print(f"{mode:10s} {1e3 * dt:.4f} ms/step  (tiles {eq.tiling_info()['ntiles']}, priority tiles {nprio})")
    dx = max(np.abs(out["single"][0] - out[m][0]).max() for m in ("two_phase", "two_phase_512", "two_stream")) / np.abs(out["single"][0]).max()
    db = max(np.abs(out["single"][1] - out[m][1]).max() for m in ("two_phase", "two_phase_512", "two_stream")) / np.abs(out["single"][1]).max()
    print(f"rel. difference of the sweeps: x {dx:.2e}, ghost rows {db:.2e}")


if __name__ == "__main__":
    main()
