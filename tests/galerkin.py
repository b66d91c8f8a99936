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
