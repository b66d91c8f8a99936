"""TEST INFRASTRUCTURE - CPU restatement of the cell-local projector
(cpp/dolfinx_eqlb/base/local_solver.hpp:38-187 with the forms of
python/dolfinx_eqlb/lsolver/projection.py:17-77): for every cell assemble A_e = (psi_i, psi_j)_T
and L_e = (f, psi_i)_T by quadrature, factorise A_e (Cholesky, the reference's Eigen::LLT),
solve, and write the cell's DOFs.  Deliberately cell-by-cell like the reference."""

import numpy as np

from dolfinx_eqlb_amd.elmtlib.lagrange import Lagrange
from dolfinx_eqlb_amd.elmtlib.quadrature import make_quadrature_triangle


def local_projection(mesh, degree, qpoints, qweights, qvalues, bs=1):
    """qvalues [nrhs, ncells, nq, bs] -> [nrhs, ncells*nd*bs]."""
    el = Lagrange(degree)
    nd = el.ndofs
    qv = np.asarray(qvalues, dtype=np.float64).reshape(-1, mesh.ncells, len(qweights), bs)
    nrhs = qv.shape[0]
    psi_l = el.tabulate(qpoints)[0]            # load rule (the caller's)
    qa, wa = make_quadrature_triangle(2 * degree)  # mass matrix: exact rule, like FFCx
    psi_a = el.tabulate(qa)[0]
    out = np.zeros((nrhs, mesh.ncells, nd, bs))
    x = mesh.x[:, :2]
    for c in range(mesh.ncells):
        xc = x[mesh.cell_nodes[c]]
        detJ = (xc[1, 0] - xc[0, 0]) * (xc[2, 1] - xc[0, 1]) - (xc[2, 0] - xc[0, 0]) * (xc[1, 1] - xc[0, 1])
        A_e = np.einsum("q,qi,qj->ij", wa * abs(detJ), psi_a, psi_a)
        Lc = np.linalg.cholesky(A_e)
        for r in range(nrhs):
            for cb in range(bs):
                L_e = psi_l.T @ (qweights * abs(detJ) * qv[r, c, :, cb])
                y = np.linalg.solve(Lc, L_e)
                out[r, c, :, cb] = np.linalg.solve(Lc.T, y)
    return out.reshape(nrhs, -1)
