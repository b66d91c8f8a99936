"""Rank of the patch-wise weak-symmetry system (test helper, independent of the oracle).

For a node a the symmetry step looks for corrections u_r in the patch-wise H(div=0) space of row r
(own boundary types) with (u_0[1] - u_1[0], v) = -(sigma_0[1] - sigma_1[0], v) for the P1 functions v of the
patch.  With DIFFERENT boundary types per row the operator z -> S z can lose rank beyond the constant
multiplier of interior patches; the reference's pivoted LU and the device solve then return different
members of a non-unique (or inconsistent) family, and only the row-wise conditions can be compared."""

import numpy as np
import scipy.linalg as sla

import kkt_reference as kr
from dolfinx_eqlb_amd.elmtlib import e_raviart_thomas as ert
from dolfinx_eqlb_amd.elmtlib.lagrange import Lagrange
from dolfinx_eqlb_amd.elmtlib.quadrature import make_quadrature_triangle


def symmetry_operator(mesh, k, node, ft):
    """(Sz [npatchnodes, nz0 + nz1], patch nodes): the symmetry functionals on the reduced unknowns."""
    rt = ert.HierarchicRT(k)
    nrt = rt.ndofs
    qp, qw = make_quadrature_triangle(2 * k + 2)
    phi = rt.tabulate(qp)
    hv = Lagrange(1).tabulate(qp)[0]
    cells = mesh.node_cells[mesh.node_cells_offsets[node]:mesh.node_cells_offsets[node + 1]]
    n = cells.size
    pos = {int(c): i for i, c in enumerate(cells)}
    Ns = [sla.null_space(kr.constraint_matrix(mesh, k, node, ft[r])[0], rcond=1e-11) for r in range(2)]
    pnodes = sorted(set(mesh.cell_nodes[cells].ravel().tolist()))
    S = np.zeros((len(pnodes), 2, n * nrt))
    for c in cells:
        x = mesh.x[mesh.cell_nodes[c], :2]
        J = np.stack([x[1] - x[0], x[2] - x[0]], axis=1)
        detJ = np.linalg.det(J)
        phys = np.einsum("ab,qib->qia", J, phi) / detJ
        for v in range(3):
            j = pnodes.index(int(mesh.cell_nodes[c, v]))
            wv = qw * abs(detJ) * hv[:, v]
            S[j, 0, pos[int(c)] * nrt:(pos[int(c)] + 1) * nrt] += wv @ phys[:, :, 1]
            S[j, 1, pos[int(c)] * nrt:(pos[int(c)] + 1) * nrt] -= wv @ phys[:, :, 0]
    return np.hstack([S[:, 0] @ Ns[0], S[:, 1] @ Ns[1]]), pnodes


def deficient_nodes(mesh, k, ft, rtol=1e-9):
    """Nodes whose symmetry operator has rank below (patch nodes) - (1 on interior patches)."""
    bnodes = np.zeros(mesh.nnodes, dtype=bool)
    bnodes[mesh.facet_nodes[mesh.boundary_facets()].ravel()] = True
    out = []
    # only patches that touch a flux-BC facet of some row can differ from the all-Dirichlet / interior case
    cand = np.unique(mesh.facet_nodes[np.nonzero((np.asarray(ft) == 2).any(axis=0))[0]].ravel())
    for node in cand:
        node = int(node)
        Sz, pn = symmetry_operator(mesh, k, node, ft)
        sv = np.linalg.svd(Sz, compute_uv=False) if Sz.shape[1] else np.zeros(0)
        rank = int((sv > rtol * max(sv.max(), 1e-300)).sum()) if sv.size else 0
        if rank < len(pn) - (0 if bnodes[node] else 1):
            out.append((node, rank, len(pn)))
    return out
