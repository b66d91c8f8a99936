"""Worker of tests/test_distributed.py (torch.distributed.run, gloo, CPU): general node-ownership
partition of an UNSTRUCTURED mesh (Partition: owner of a node by angular sector of a Delaunay cloud, so
that every rank has several neighbours) + reverse halo reduction.  mode "se": broken RT rows of the
ghost cells; mode "ev": conforming DOFs (facet DOFs travel with the ghost cells).  The CPU oracle is
the per-rank patch solver; the union of the owned parts must equal the single-domain result."""

import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]

from dolfinx_eqlb_amd import distributed as dd  # noqa: E402
from dolfinx_eqlb_amd.eqlb.conforming import conforming_dofmap  # noqa: E402
from dolfinx_eqlb_amd.mesh import create_mesh  # noqa: E402
from synthetic import facet_types, make_compatible_data  # noqa: E402
from oracle import oracle  # noqa: E402


def delaunay_mesh(npts, seed):
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    pts = rng.random((npts, 2))
    tri = Delaunay(pts)
    cells = tri.simplices.astype(np.int32)
    # drop needle cells of the convex hull
    x = pts[cells]
    area = 0.5 * np.abs((x[:, 1, 0] - x[:, 0, 0]) * (x[:, 2, 1] - x[:, 0, 1])
                        - (x[:, 2, 0] - x[:, 0, 0]) * (x[:, 1, 1] - x[:, 0, 1]))
    cells = cells[area > 1e-4]
    used = np.unique(cells)
    remap = -np.ones(npts, dtype=np.int32)
    remap[used] = np.arange(used.size, dtype=np.int32)
    return create_mesh(pts[used], remap[cells])


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    mode, k = sys.argv[1], int(sys.argv[2])
    nrt = k * (k + 2)
    gmesh = delaunay_mesh(260, 3)
    gft = facet_types(gmesh)
    gG, gf = make_compatible_data(gmesh, k, gft, seed=5)
    # owner of a node: angular sector around the centre of the cloud - every rank meets every other
    ang = np.arctan2(gmesh.x[:, 1] - 0.5, gmesh.x[:, 0] - 0.5)
    owner = np.minimum(((ang + np.pi) / (2 * np.pi) * world).astype(int), world - 1)
    part = dd.Partition(gmesh, owner, rank, world)
    if mode == "se_local":
        # the same decomposition from the rank's OWN arrays only (Partition.from_local): local cells, their
        # coordinates, owner of every local node / cell, global cell ids - what a distributed mesh provides
        cown = owner[gmesh.cell_nodes].max(axis=1)
        loc = part.cell_global
        gn = part.node_global
        g2l = -np.ones(gmesh.nnodes, dtype=np.int64)
        g2l[gn] = np.arange(gn.size)
        part2 = dd.Partition.from_local(gmesh.x[gn, :2], g2l[gmesh.cell_nodes[loc]], owner[gn], cown[loc], loc,
                                        rank, world, node_global=gn)
        assert np.array_equal(part2.mesh.cell_nodes, part.mesh.cell_nodes)
        assert np.array_equal(part2.node_mask, part.node_mask) and np.array_equal(part2.cell_owned, part.cell_owned)
        assert sorted(part2.send) == sorted(part.send) and sorted(part2.recv) == sorted(part.recv)
        for q in part.send:
            assert np.array_equal(part2.send[q], part.send[q])
        for q in part.recv:
            assert np.array_equal(part2.recv[q], part.recv[q])
        assert part2.halo_bytes(nrt) == (8 * nrt * sum(len(v) for v in part.send.values()),
                                         8 * nrt * sum(len(v) for v in part.recv.values()))
        part = part2
        mode = "se"
    m = part.mesh
    assert np.allclose(m.x[m.cell_nodes], gmesh.x[gmesh.cell_nodes[part.cell_global]])
    nd = gG.size // gmesh.ncells
    G = gG.reshape(gmesh.ncells, -1)[part.cell_global].ravel()
    f = gf.reshape(gmesh.ncells, -1)[part.cell_global].ravel()
    ft = part.facet_types(gft) if part._gmesh_nfacets else facet_types(m)  # all-Dirichlet data: the local table
    nodes = np.nonzero(part.node_mask)[0]
    nsteps = 2  # accumulating sweeps: ghost rows / DOFs must not be double counted
    if mode == "se":
        gref = oracle.se_reconstruct(gmesh, k, gft, gG[None], gf[None])[0].reshape(gmesh.ncells, nrt)
        x = np.zeros((1, m.ncells * nrt))
        halo = dd.HaloExchange(part, nrt, torch.device("cpu"))
        for step in range(nsteps):
            for node in nodes:
                oracle.se_reconstruct(m, k, ft, G[None], f[None], flux_hdiv=x, node_range=(int(node), int(node) + 1))
            halo.reduce(torch.from_numpy(x.ravel()))
        got = x.reshape(m.ncells, nrt)[part.cell_owned]
        ref = nsteps * gref[part.cell_global[part.cell_owned]]
        err = np.abs(got - ref).max() / np.abs(ref).max()
        ghost_cleared = bool(np.all(x.reshape(m.ncells, nrt)[~part.cell_owned] == 0.0))
    else:
        gcd, gnd = conforming_dofmap(gmesh, k)
        gref = oracle.ev_reconstruct(gmesh, k, gft, gG[None], gf[None], gcd, gnd)[0]
        cd, ndofs = conforming_dofmap(m, k)
        x = np.zeros((1, ndofs))
        send, recv = part.conforming_halo(k)
        halo = dd.HaloExchange(part, 1, torch.device("cpu"), lists=(send, recv), nentries=ndofs)
        for step in range(nsteps):
            for node in nodes:
                oracle.ev_reconstruct(m, k, ft, G[None], f[None], cd, ndofs, flux_hdiv=x,
                                      node_range=(int(node), int(node) + 1))
            halo.reduce(torch.from_numpy(x.ravel()))
        # every conforming DOF ends up on exactly one rank, its owner: all local DOFs that are not sent away
        # and belong to a facet / cell owned here (local facet frame == global facet frame: the local
        # node numbering keeps the order of the global one)
        sent = np.concatenate([v for v in send.values()]) if send else np.zeros(0, np.int64)
        held = np.setdiff1d(np.unique(cd.ravel()), sent)
        l2g = np.zeros(ndofs, dtype=np.int64)
        l2g[cd.ravel()] = gcd[part.cell_global].ravel()
        err = np.abs(x[0][held] - nsteps * gref[l2g[held]]).max() / np.abs(gref).max()
        ghost_cleared = bool(np.all(x[0][np.setdiff1d(np.arange(ndofs), held)] == 0.0))
        nheld = torch.tensor([float(held.size)])
        dist.all_reduce(nheld)
        assert int(nheld[0]) == gnd, (int(nheld[0]), gnd)
    counts = torch.tensor([float(part.node_mask.sum()), float(part.ncells_owned), float(len(part.send) + len(part.recv))])
    dist.all_reduce(counts)
    ok = err < 1e-10 and ghost_cleared and int(counts[0]) == gmesh.nnodes and int(counts[1]) == gmesh.ncells
    print(f"rank {rank}: {mode} err {err:.2e} peers {sorted(set(part.send) | set(part.recv))} ghost cleared "
          f"{ghost_cleared} nodes {int(counts[0])}/{gmesh.nnodes} cells {int(counts[1])}/{gmesh.ncells} "
          f"{'OK' if ok else 'FAIL'}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
