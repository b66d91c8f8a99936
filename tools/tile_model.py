"""Cost model of the tiled launch: wave-blocks per tile and rounds on 8 waves for several tile
sizes (RCB tiles as in eqlb_api.hip::build_tiles).  usage: python tools/tile_model.py [n]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from dolfinx_eqlb_amd.mesh import create_unit_square


def rcb(cx, cy, idx, ntile, tc, out):
    n = idx.size
    if ntile <= 1 or n <= tc:
        out.append(idx)
        return
    tl = ntile // 2
    nl = min(n, tl * tc)
    x, y = cx[idx], cy[idx]
    key = x if (x.max() - x.min()) >= (y.max() - y.min()) else y
    part = np.argpartition(key, nl - 1) if nl < n else np.arange(n)
    rcb(cx, cy, idx[part[:nl]], tl, tc, out)
    rcb(cx, cy, idx[part[nl:]], ntile - tl, tc, out)


def model(mesh, tc):
    cen = mesh.x[mesh.cell_nodes, :2].mean(axis=1)
    out = []
    ntile = (mesh.ncells + tc - 1) // tc
    sys.setrecursionlimit(10000)
    rcb(cen[:, 0], cen[:, 1], np.arange(mesh.ncells), ntile, tc, out)
    nf = np.diff(mesh.node_facets_offsets)
    P = np.where(nf <= 4, 4, np.where(nf <= 8, 8, 16))
    rounds = 0
    wbs = []
    inst = 0
    for cells in out:
        nodes = np.unique(mesh.cell_nodes[cells])
        p = P[nodes]
        wb = sum((int((p == q).sum()) * q + 63) // 64 for q in (4, 8, 16))
        wbs.append(wb)
        rounds += -(-wb // 8)
        inst += nodes.size
    wbs = np.array(wbs)
    return len(out), inst / mesh.nnodes, wbs.mean(), rounds, wbs.sum()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    mesh = create_unit_square(n)
    for tc in (160, 176, 192, 208, 224, 240, 256):
        nt, halo, wb, rounds, tot = model(mesh, tc)
        print(f"TC={tc}: tiles {nt}, instances x{halo:.3f}, wave-blocks/tile {wb:.1f}, "
              f"rounds {rounds} ({rounds * 8 / tot:.3f} x work), rounds+0.5/tile {rounds + 0.5 * nt:.0f}")
