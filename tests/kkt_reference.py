"""Independent dense statement of the patch problem of the semi-explicit equilibration.

For one mesh node a the reference computes (se/solve_patch_semiexplt.hpp:212-1163) an explicit
particular solution plus the L2-minimal correction in a patch-wise H(div=0) space.  The result
is the unique solution of

    min  sum_T || sigma_a ||^2_{L2(T)}      over broken RT_k on the patch cells,  subject to
      (i)   sigma_a . n = 0 on the patch-boundary facets opposite to a,
      (ii)  (div sigma_a, q)_T = (hat_a (f - div G), q)_T           for all q in P_{k-1}(T),
      (iii) [(sigma_a + hat_a G) . n] = 0 (moments < k) on interior patch facets,
      (iv)  (sigma_a + hat_a G) . n = 0 (moments < k) on flux-BC (homogeneous) facets at a,

so it can be pinned without the reference's explicit construction: build the constraints
generically with quadrature, solve the KKT system by a null-space method in dense numpy.
Nothing here shares code with the oracle (oracle/eqlb_oracle.c) except the element library.
"""

import numpy as np
import scipy.linalg as sla

from dolfinx_eqlb_amd.elmtlib import e_raviart_thomas as ert
from dolfinx_eqlb_amd.elmtlib.lagrange import Lagrange
from dolfinx_eqlb_amd.elmtlib.quadrature import make_quadrature_interval, make_quadrature_triangle

_PF = np.where(np.array(ert.FACET_NORMAL_IS_OUTWARD), 1.0, -1.0)


def _geom(mesh, c):
    x = mesh.x[mesh.cell_nodes[c], :2]
    J = np.stack([x[1] - x[0], x[2] - x[0]], axis=1)
    detJ = np.linalg.det(J)
    return J, detJ, np.linalg.inv(J)


def _system(mesh, k, node, facet_type, flux_dg, rhs_dg, degree_dg=None, neumann_flux=None,
            ev=False):
    """Constraint rows B, mass matrix M, right-hand side d of the patch problem.

    ev=True: the constrained-minimisation (Ern-Vohralik) patch problem instead: conforming
    sigma_a (no jump data), (div sigma_a, q) = (hat_a f + grad hat_a . G, q), and the functional
    || sigma_a - hat_a G ||^2, whose linear term (phi_i, hat_a G) is returned as 7th entry."""
    degree_dg = k - 1 if degree_dg is None else degree_dg
    rt = ert.HierarchicRT(k)
    dg = Lagrange(degree_dg)
    hat = Lagrange(1)
    nd, ndofs = dg.ndofs, rt.ndofs
    cells = mesh.node_cells[mesh.node_cells_offsets[node]:mesh.node_cells_offsets[node + 1]]
    fcts = mesh.node_facets[mesh.node_facets_offsets[node]:mesh.node_facets_offsets[node + 1]]
    n = cells.size
    pos = {int(c): i for i, c in enumerate(cells)}
    G = flux_dg.reshape(mesh.ncells, nd, 2)
    f = rhs_dg.reshape(mesh.ncells, nd)
    ft = np.asarray(facet_type).reshape(-1, mesh.nfacets)[0]

    qp, qw = make_quadrature_triangle(2 * k + 2)
    phi = rt.tabulate(qp)
    tabdg = dg.tabulate(qp, 1)
    hq = hat.tabulate(qp)[0]
    s, w = make_quadrature_interval(2 * k + 1)

    M = np.zeros((n * ndofs, n * ndofs))
    blin = np.zeros(n * ndofs)
    rows, rhs = [], []

    def add(row, val):
        rows.append(row)
        rhs.append(val)

    # exponents of a basis of P_{k-1}: 1 first, then the div-moment monomials
    expo = [(0, 0)] + rt.div_exponents
    for i, c in enumerate(cells):
        J, detJ, K = _geom(mesh, c)
        sl = slice(i * ndofs, (i + 1) * ndofs)
        phys = np.einsum("ab,qib->qia", J, phi) / detJ
        M[sl, sl] = np.einsum("q,qia,qja->ij", qw * abs(detJ), phys, phys)
        ln = int(np.nonzero(mesh.cell_nodes[c] == node)[0][0])
        # (i) outer facet = local facet opposite to the patch node
        for j in range(k):
            row = np.zeros(n * ndofs)
            row[i * ndofs + ln * k + j] = 1.0
            add(row, 0.0)
        # (ii) divergence moments, in reference coordinates (div sigma = div_ref / detJ)
        divphi = rt.tabulate_div(qp)  # [q, i]
        gpsi = np.einsum("Xd,Xqj->qjd", K, tabdg[1:3])  # K^T grad_ref psi
        fq = tabdg[0] @ f[c]
        divG = np.einsum("jd,qjd->q", G[c], gpsi)
        res = (fq - divG) * hq[:, ln]
        if ev:
            ghat = K.T @ np.array([[-1.0, -1.0], [1.0, 0.0], [0.0, 1.0]][ln])
            Gq_c = tabdg[0] @ G[c]
            res = fq * hq[:, ln] + Gq_c @ ghat
            blin[sl] = np.einsum("q,qia,qa->i", qw * abs(detJ) * hq[:, ln], phys, Gq_c)
        for (l, m) in expo:
            mono = qp[:, 0] ** l * qp[:, 1] ** m
            row = np.zeros(n * ndofs)
            row[sl] = np.einsum("q,qi->i", qw * mono * np.sign(detJ), divphi)
            add(row, float(np.sum(qw * abs(detJ) * res * mono)))

    # facet conditions
    for F in fcts:
        fc = mesh.facet_cells[mesh.facet_cells_offsets[F]:mesh.facet_cells_offsets[F + 1]]
        if fc.size == 1 and ft[F] != 2:
            continue  # primal-Dirichlet boundary facet: flux is free
        for j in range(k):
            row = np.zeros(n * ndofs)
            val = 0.0
            for c in fc:
                i = pos[int(c)]
                J, detJ, K = _geom(mesh, c)
                lf = int(np.nonzero(mesh.cell_facets[c] == F)[0][0])
                ln = int(np.nonzero(mesh.cell_nodes[c] == node)[0][0])
                perm = mesh.facet_perm[c, lf]
                s_loc = (1.0 - s) if perm else s
                pts = ert.facet_points(s_loc)[lf]
                nref = np.array(ert.FACET_NORMALS[lf], dtype=float)
                sg = np.sign(detJ) * _PF[lf]
                dens = (rt.tabulate(pts) @ nref) * sg  # outward flux density of phi_i
                row[i * ndofs:(i + 1) * ndofs] += (w * s ** j) @ dens
                Gq = dg.tabulate(pts)[0] @ G[c]  # [q, 2]
                pb = detJ * (Gq @ K.T)
                hatv = hat.tabulate(pts)[0][:, ln]
                densG = (pb @ nref) * sg * hatv
                if not ev:
                    val -= float(np.sum(w * s ** j * densG))
                if fc.size == 1 and neumann_flux is not None:
                    # prescribed total normal flux hat_a * (w . n) on the flux-BC facet
                    x0 = mesh.x[mesh.cell_nodes[c][0], :2]
                    xq = x0[None, :] + pts @ J.T
                    wx, wy = neumann_flux(xq[:, 0], xq[:, 1])
                    pbw = detJ * (np.stack([wx, wy], axis=1) @ K.T)
                    val += float(np.sum(w * s ** j * (pbw @ nref) * sg * hatv))
            add(row, val)

    if ev:
        return np.array(rows), M, np.array(rhs), cells, n, ndofs, blin
    return np.array(rows), M, np.array(rhs), cells, n, ndofs


def solve_patch_ev(mesh, k, node, facet_type, flux_dg, rhs_dg, neumann_flux=None):
    """Constrained minimiser of the EV patch problem (ev/solve_patch.hpp:58-238 solves its
    saddle-point form): (cells, broken coefficients [n, ndofs], constraint residual)."""
    B, M, d, cells, n, ndofs, b = _system(mesh, k, node, facet_type, flux_dg, rhs_dg, None,
                                          neumann_flux, ev=True)
    cp, *_ = np.linalg.lstsq(B, d, rcond=None)
    resid = np.linalg.norm(B @ cp - d)
    N = sla.null_space(B, rcond=1e-11)
    if N.shape[1]:
        y = np.linalg.solve(N.T @ M @ N, N.T @ (b - M @ cp))
        cp = cp + N @ y
    return cells, cp.reshape(n, ndofs), resid


def solve_patch(mesh, k, node, facet_type, flux_dg, rhs_dg, degree_dg=None, neumann_flux=None):
    """Returns (cells, coefficients[n, ndofs]) of the constrained minimiser on the patch."""
    B, M, d, cells, n, ndofs = _system(mesh, k, node, facet_type, flux_dg, rhs_dg, degree_dg,
                                       neumann_flux)
    # particular solution + null space
    cp, *_ = np.linalg.lstsq(B, d, rcond=None)
    resid = np.linalg.norm(B @ cp - d)
    N = sla.null_space(B, rcond=1e-11)
    if N.shape[1]:
        y = np.linalg.solve(N.T @ M @ N, N.T @ M @ cp)
        cp = cp - N @ y
    return cells, cp.reshape(n, ndofs), resid, N.shape[1]


def constraint_matrix(mesh, k, node, facet_type_row):
    """(B, M): homogeneous constraint rows (i)-(iv) of the patch space and the block-diagonal
    mass matrix, for one row of BC types (used by the weak-symmetry check)."""
    nd = k * (k + 1) // 2
    zG = np.zeros(mesh.ncells * nd * 2)
    zf = np.zeros(mesh.ncells * nd)
    return _system(mesh, k, node, facet_type_row, zG, zf)[:2]
