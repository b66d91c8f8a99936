"""Worker of tests/test_distributed.py (launched by torch.distributed.run, gloo, CPU):
strip partition + reverse halo reduction, with the CPU oracle as the per-rank patch solver,
must reproduce the single-process result on the union mesh."""

import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]

from dolfinx_eqlb_amd import distributed as dd  # noqa: E402
from dolfinx_eqlb_amd.mesh import create_rectangle  # noqa: E402
from synthetic import facet_types, make_compatible_data  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n, k = int(sys.argv[1]), int(sys.argv[2])
    nrt, nd = k * (k + 2), k * (k + 1) // 2

    gmesh = create_rectangle(world * n, n, 0.0, float(world))
    gft = facet_types(gmesh)
    gG, gf = make_compatible_data(gmesh, k, gft, seed=5)
    gref = oracle.se_reconstruct(gmesh, k, gft, gG[None], gf[None])[0].reshape(gmesh.ncells, nrt)

    part = dd.StripPartition(n, rank, world)
    gi, gj, gt = part.grid_ids
    gcell = (gj * (world * n) + gi + rank * n) * 4 + gt  # global id of every local cell
    m = part.mesh
    # shared cells must have the same local vertex order (coordinates of the 3 vertices agree)
    assert np.allclose(m.x[m.cell_nodes], gmesh.x[gmesh.cell_nodes[gcell]])
    G = gG.reshape(gmesh.ncells, -1)[gcell].ravel()
    f = gf.reshape(gmesh.ncells, -1)[gcell].ravel()
    ft = part.facet_types()

    x = np.zeros((1, m.ncells * nrt))
    for step in range(2):  # two accumulating sweeps: ghost rows must not be double counted
        for node in np.nonzero(part.node_mask)[0]:
            oracle.se_reconstruct(m, k, ft, G[None], f[None], flux_hdiv=x,
                                  node_range=(int(node), int(node) + 1))
        xt = torch.from_numpy(x.ravel())
        dd.HaloExchange(part, nrt, torch.device("cpu")).reduce(xt)
    got = x.reshape(m.ncells, nrt)[part.cell_owned]
    ref = 2.0 * gref[gcell[part.cell_owned]]
    err = np.abs(got - ref).max() / np.abs(ref).max()
    counts = torch.tensor([float(part.node_mask.sum()), float(part.ncells_owned)])
    dist.all_reduce(counts)
    ok = err < 1e-11 and int(counts[0]) == gmesh.nnodes and int(counts[1]) == gmesh.ncells
    print(f"rank {rank}: err {err:.2e} nodes {int(counts[0])}/{gmesh.nnodes} "
          f"cells {int(counts[1])}/{gmesh.ncells} {'OK' if ok else 'FAIL'}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
