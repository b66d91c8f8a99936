"""TEST INFRASTRUCTURE - a small conforming P_k (k = 1, 2, 3) Galerkin solver for -Laplace u = f
with homogeneous Dirichlet data on the flat mesh container, standing in for the DOLFINx primal
solves of the reference's tests and demos (python/test/unit/utils.py, demo/poisson/
demo_reconstruction.py:44-54): it provides REAL discrete fluxes sigma_h = -grad u_h, so that the
equilibration can be validated end to end (Prager-Synge bound, convergence rates)."""

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from dolfinx_eqlb_amd.elmtlib.lagrange import Lagrange
from dolfinx_eqlb_amd.elmtlib.quadrature import make_quadrature_triangle
from dolfinx_eqlb_amd.eqlb.check_eqlb_conditions import cell_geometry


def dofmap(mesh, k):
    """cell -> global DOF of the conforming P_k space (Basix-like local numbering of
    dolfinx_eqlb_amd.elmtlib.lagrange): vertices, then k-1 DOFs per edge ordered along the global
    edge direction (low -> high node), then the cell-interior DOFs."""
    nn, nf, nc = mesh.nnodes, mesh.nfacets, mesh.ncells
    ne = k - 1
    ni = (k - 1) * (k - 2) // 2
    cd = np.empty((nc, (k + 1) * (k + 2) // 2), dtype=np.int64)
    cd[:, :3] = mesh.cell_nodes
    col = 3
    for f in range(3):
        base = nn + mesh.cell_facets[:, f].astype(np.int64) * ne
        for j in range(ne):
            jj = np.where(mesh.facet_perm[:, f] == 1, ne - 1 - j, j)
            cd[:, col] = base + jj
            col += 1
    for j in range(ni):
        cd[:, col] = nn + nf * ne + np.arange(nc, dtype=np.int64) * ni + j
        col += 1
    return cd, nn + nf * ne + nc * ni


def solve_poisson(mesh, k, f_exact, qdeg=None, f_dg=None, dirichlet_facets=None):
    """u_h in P_k with u_h = 0 on the boundary (or on `dirichlet_facets` only, natural homogeneous
    Neumann condition on the other boundary facets); returns (u [ndofs], cell dofmap).
    f_dg: DG_{k-1} nodal values used INSTEAD of f_exact as right-hand side (for k = 1 the
    equilibration needs the primal problem solved with Pi_0 f, demo_reconstruction.py:504)."""
    el = Lagrange(k)
    cd, ndofs = dofmap(mesh, k)
    J, detJ, K = cell_geometry(mesh)
    qp, qw = make_quadrature_triangle(qdeg or 2 * k + 4)
    tab = el.tabulate(qp, 1)
    dphi = np.stack([tab[1], tab[2]], axis=2)                # [q, i, X]
    g = np.einsum("cXd,qiX->cqid", K, dphi)                  # physical gradients
    w = qw[None, :] * np.abs(detJ)[:, None]
    Ke = np.einsum("cq,cqid,cqjd->cij", w, g, g)
    x0 = mesh.x[mesh.cell_nodes[:, 0], :2]
    xq = x0[:, None, :] + np.einsum("cij,qj->cqi", J, qp)
    if f_dg is not None:
        fq = np.asarray(f_dg).reshape(mesh.ncells, -1) @ Lagrange(k - 1).tabulate(qp)[0].T
    else:
        fq = f_exact(xq[..., 0], xq[..., 1])
    fe = np.einsum("cq,cq,qi->ci", w, fq, tab[0])
    nl = cd.shape[1]
    A = sp.csr_matrix((Ke.ravel(), (np.repeat(cd, nl, axis=1).ravel(), np.tile(cd, (1, nl)).ravel())),
                      shape=(ndofs, ndofs))
    b = np.zeros(ndofs)
    np.add.at(b, cd.ravel(), fe.ravel())
    # homogeneous Dirichlet: vertex and edge DOFs of the boundary facets
    bf = mesh.boundary_facets() if dirichlet_facets is None else np.asarray(dirichlet_facets)
    fixed = np.zeros(ndofs, dtype=bool)
    fixed[mesh.facet_nodes[bf].ravel()] = True
    for j in range(k - 1):
        fixed[mesh.nnodes + bf * (k - 1) + j] = True
    free = np.nonzero(~fixed)[0]
    u = np.zeros(ndofs)
    u[free] = spla.spsolve(A[free][:, free].tocsc(), b[free])
    return u, cd


def discrete_flux(mesh, k, u, cd):
    """G = -grad u_h as DG_{k-1}^2 nodal values [ncells*nd*2] (exact: grad u_h is in P_{k-1}^2)."""
    el = Lagrange(k)
    dg = Lagrange(k - 1)
    J, detJ, K = cell_geometry(mesh)
    nodes = np.array([[float(a), float(b)] for a, b in dg.nodes])
    tab = el.tabulate(nodes, 1)
    dphi = np.stack([tab[1], tab[2]], axis=2)                # [node, i, X]
    gref = np.einsum("ci,niX->cnX", u[cd], dphi)
    G = -np.einsum("cXd,cnX->cnd", K, gref)
    return np.ascontiguousarray(G.reshape(-1))


def energy_error(mesh, k, u, cd, grad_exact):
    """|| grad(u - u_h) ||_L2."""
    el = Lagrange(k)
    J, detJ, K = cell_geometry(mesh)
    qp, qw = make_quadrature_triangle(2 * k + 6)
    tab = el.tabulate(qp, 1)
    dphi = np.stack([tab[1], tab[2]], axis=2)
    gh = np.einsum("cXd,ci,qiX->cqd", K, u[cd], dphi)
    x0 = mesh.x[mesh.cell_nodes[:, 0], :2]
    xq = x0[:, None, :] + np.einsum("cij,qj->cqi", J, qp)
    gx, gy = grad_exact(xq[..., 0], xq[..., 1])
    e2 = (gx - gh[..., 0]) ** 2 + (gy - gh[..., 1]) ** 2
    return float(np.sqrt(np.sum(qw[None, :] * np.abs(detJ)[:, None] * e2)))


def project_rhs(mesh, k, f_exact):
    """Pi_{k-1} f as DG_{k-1} nodal values [ncells*nd] and the data for the oscillation term:
    (f_h, || f - Pi f ||_T^2 per cell, cell diameters)."""
    dg = Lagrange(k - 1)
    J, detJ, K = cell_geometry(mesh)
    qp, qw = make_quadrature_triangle(2 * k + 6)
    psi = dg.tabulate(qp)[0]
    M = np.einsum("q,qi,qj->ij", qw, psi, psi)
    x0 = mesh.x[mesh.cell_nodes[:, 0], :2]
    xq = x0[:, None, :] + np.einsum("cij,qj->cqi", J, qp)
    fq = f_exact(xq[..., 0], xq[..., 1])
    rhs = np.einsum("q,cq,qi->ci", qw, fq, psi)
    fh = np.linalg.solve(M, rhs.T).T                         # |detJ| cancels
    osc2 = np.einsum("q,cq->c", qw, (fq - fh @ psi.T) ** 2) * np.abs(detJ)
    xc = mesh.x[mesh.cell_nodes, :2]
    e = np.stack([xc[:, 1] - xc[:, 0], xc[:, 2] - xc[:, 1], xc[:, 0] - xc[:, 2]], axis=1)
    h = np.sqrt((e ** 2).sum(axis=2)).max(axis=1)
    return np.ascontiguousarray(fh.reshape(-1)), osc2, h


# --- linear elasticity with component-wise boundary conditions -------------------------------------
# Stand-in for solve_primal_problem_general_usquare of the reference's
# python/test/unit/test_stressqlb_bcond.py:27-144: P_k^2 displacement, sigma(u) = 2 eps(u) + div u I,
# per side of the unit square and per displacement component either u_r = 0 or a prescribed traction
# component t_r (random DG_{k-1} data as :80-83), random DG_{k-1}^2 body force (testcase_general.py:
# set_arbitrary_rhs).  The equilibration sees sigma_h = -sigma(u_h) row by row and -t_r as flux BC.

SIDES = (lambda mp: np.abs(mp[:, 0]) < 1e-12, lambda mp: np.abs(mp[:, 1]) < 1e-12,
         lambda mp: np.abs(mp[:, 0] - 1.0) < 1e-12, lambda mp: np.abs(mp[:, 1] - 1.0) < 1e-12)


def side_facets(mesh):
    """Boundary facets of the four sides x = 0, y = 0, x = 1, y = 1 (boundary ids 1..4 of utils.py:74-79)."""
    bf = mesh.boundary_facets()
    mp = mesh.facet_midpoints()[bf]
    return [bf[s(mp)] for s in SIDES]


def elasticity_facet_types(mesh, layout):
    """facet_type [2, nfacets] of the two stress rows: layout[side][row] True -> traction (flux BC, 2) on that
    side for that row, else primal Dirichlet (1); sides beyond len(layout) are Dirichlet."""
    ft = np.zeros((2, mesh.nfacets), dtype=np.int8)
    ft[:, mesh.boundary_facets()] = 1
    sides = side_facets(mesh)
    for s, flags in enumerate(layout):
        for r in range(2):
            if flags[r]:
                ft[r, sides[s]] = 2
    return ft


def _boundary_cells(mesh, facets):
    cells = mesh.facet_cells[mesh.facet_cells_offsets[facets]]
    lf = np.argmax(mesh.cell_facets[cells] == facets[:, None], axis=1)
    return cells, lf


def solve_elasticity(mesh, k, ft, seed=0, traction=None, body_force=True):
    """u_h in P_k^2 with u_r = 0 on the facets ft[r] == 1 and traction component t_r (trace of a random
    DG_{k-1} function; `traction(r, x, y)` if given: prescribed values, polynomial of degree < k per facet)
    on ft[r] == 2; body force random DG_{k-1}^2 (zero with body_force=False).
    Returns (G [2, ncells*nd*2] rows of -sigma(u_h) as DG_{k-1}^2 nodal values, f [2, ncells*nd],
    boundary_values [2, ncells*k(k+2)] = global boundary DOFs of the prescribed normal flux -t_r)."""
    from dolfinx_eqlb_amd.elmtlib import e_raviart_thomas as ert
    from dolfinx_eqlb_amd.elmtlib.quadrature import make_quadrature_interval
    rng = np.random.default_rng(seed)
    el, dg = Lagrange(k), Lagrange(k - 1)
    nd = len(dg.nodes)
    cd, ns = dofmap(mesh, k)
    J, detJ, K = cell_geometry(mesh)
    qp, qw = make_quadrature_triangle(2 * k + 2)
    tab = el.tabulate(qp, 1)
    dphi = np.stack([tab[1], tab[2]], axis=2)
    g = np.einsum("cXd,qiX->cqid", K, dphi)                  # [c, q, i, d] physical gradients
    w = qw[None, :] * np.abs(detJ)[:, None]
    nl = cd.shape[1]
    # local stiffness on (i, r), (j, s): int 2 eps(phi_j e_s):eps(phi_i e_r) + div div
    gg = np.einsum("cq,cqid,cqjd->cij", w, g, g)
    gx = np.einsum("cq,cqir,cqjs->cirjs", w, g, g)           # d_r phi_i d_s phi_j
    Ke = np.zeros((mesh.ncells, nl, 2, nl, 2))
    for r in range(2):
        Ke[:, :, r, :, r] += gg
    Ke += np.einsum("cirjs->cisjr", gx)                      # d_s phi_i d_r phi_j  (grad u^T term)
    Ke += gx                                                 # div u div v
    vd = (2 * cd[:, :, None] + np.arange(2)[None, None, :]).reshape(mesh.ncells, 2 * nl)
    Ke = Ke.reshape(mesh.ncells, 2 * nl, 2 * nl)
    A = sp.csr_matrix((Ke.ravel(), (np.repeat(vd, 2 * nl, axis=1).ravel(), np.tile(vd, (1, 2 * nl)).ravel())),
                      shape=(2 * ns, 2 * ns))
    f = 2.0 * (rng.random((2, mesh.ncells, nd)) + 0.1)
    if not body_force:
        f[:] = 0.0
    psi = dg.tabulate(qp)[0]                                 # [q, nd]
    b = np.zeros(2 * ns)
    for r in range(2):
        fe = np.einsum("cq,cn,qn,qi->ci", w, f[r], psi, tab[0])
        np.add.at(b, (2 * cd + r).ravel(), fe.ravel())
    # tractions: trace of a random DG_{k-1} function per row
    tdg = 2.0 * (rng.random((2, mesh.ncells, nd)) + 0.1)
    s, wq = make_quadrature_interval(2 * k + 2)
    pfo = np.where(np.array(ert.FACET_NORMAL_IS_OUTWARD), 1.0, -1.0)
    bv = np.zeros((2, mesh.ncells * k * (k + 2)))
    xc = mesh.x[mesh.cell_nodes, :2]
    for r in range(2):
        facets = np.nonzero(ft[r] == 2)[0]
        if facets.size == 0:
            continue
        cells, lf = _boundary_cells(mesh, facets)
        for fl in range(3):
            sel = np.nonzero(lf == fl)[0]
            if sel.size == 0:
                continue
            pts = ert.facet_points(s)[fl]
            va, vb = [(1, 2), (0, 2), (0, 1)][fl]
            length = np.linalg.norm(xc[cells[sel], vb] - xc[cells[sel], va], axis=1)
            tq = tdg[r][cells[sel]] @ dg.tabulate(pts)[0].T  # [c, q]
            if traction is not None:
                xq = xc[cells[sel], 0][:, None, :] + np.einsum(
                    "qX,cXd->cqd", pts, xc[cells[sel]][:, 1:] - xc[cells[sel]][:, :1])
                tq = np.broadcast_to(np.asarray(traction(r, xq[..., 0], xq[..., 1]), dtype=float), tq.shape)
            fe = np.einsum("c,q,cq,qi->ci", length, wq, tq, el.tabulate(pts)[0])
            np.add.at(b, (2 * cd[cells[sel]] + r).ravel(), fe.ravel())
            # D_{f,j} = int_0^1 (detJ K w) . N_f s^j ds with w . n_out = -t_r
            sg = np.sign(detJ[cells[sel]]) * pfo[fl] * length
            for j in range(k):
                bv[r, cells[sel] * k * (k + 2) + fl * k + j] = -sg * (tq @ (wq * s ** j))
    fixed = np.zeros(2 * ns, dtype=bool)
    for r in range(2):
        df = np.nonzero(ft[r] == 1)[0]
        fixed[2 * mesh.facet_nodes[df].ravel() + r] = True
        for j in range(k - 1):
            fixed[2 * (mesh.nnodes + df * (k - 1) + j) + r] = True
    free = np.nonzero(~fixed)[0]
    u = np.zeros(2 * ns)
    u[free] = spla.spsolve(A[free][:, free].tocsc(), b[free])
    # rows of -sigma(u_h) at the DG_{k-1} nodes (exact: grad u_h in P_{k-1})
    nodes = np.array([[float(a), float(b_)] for a, b_ in dg.nodes])
    tn = el.tabulate(nodes, 1)
    dn = np.stack([tn[1], tn[2]], axis=2)                    # [n, i, X]
    uc = u.reshape(ns, 2)[cd]                                # [c, i, r]
    gu = np.einsum("cXd,cir,niX->cnrd", K, uc, dn)           # d_d u_r
    div = gu[..., 0, 0] + gu[..., 1, 1]
    sig = gu + np.swapaxes(gu, 2, 3)
    sig[..., 0, 0] += div
    sig[..., 1, 1] += div
    G = np.stack([np.ascontiguousarray(-sig[:, :, r, :].reshape(-1)) for r in range(2)])
    return G, np.ascontiguousarray(f.reshape(2, -1)), bv
