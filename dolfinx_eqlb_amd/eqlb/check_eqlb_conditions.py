"""Acceptance predicates of an equilibrated flux, restated on flat arrays.

Mirrors python/dolfinx_eqlb/eqlb/check_eqlb_conditions.py of the reference (which
needs DOLFINx Functions): divergence condition (:183-291), jump / H(div)-conformity
condition (:294-473), flux boundary condition (:90-179) and the reversed-edge
detector (:19-86).  Inputs are the arrays the C ABI works on:

  sigma_eq [ncells*k(k+2)]  corrector in the discontinuous hierarchic RT_k space
  flux_dg  [ncells*nd*2]    projected flux G (DG_{k-1}^2, components interleaved)
  rhs_dg   [ncells*nd]      projected right-hand side

The reconstructed flux is sigma_eq + G (FluxEqlbSE.get_reconstructed_fluxes).
"""

import numpy as np

from ..elmtlib import e_raviart_thomas as ert
from ..elmtlib.lagrange import Lagrange
from ..elmtlib.quadrature import make_quadrature_interval, make_quadrature_triangle


def cell_geometry(mesh):
    """J[c] = dx/dX (2x2), detJ[c], K[c] = J^-1 of the affine cell maps."""
    x = mesh.x[:, :2]
    cn = mesh.cell_nodes
    x0, x1, x2 = x[cn[:, 0]], x[cn[:, 1]], x[cn[:, 2]]
    J = np.stack([x1 - x0, x2 - x0], axis=2)  # J[c, i, j] = dx_i/dX_j
    detJ = J[:, 0, 0] * J[:, 1, 1] - J[:, 0, 1] * J[:, 1, 0]
    K = np.empty_like(J)
    K[:, 0, 0] = J[:, 1, 1] / detJ
    K[:, 0, 1] = -J[:, 0, 1] / detJ
    K[:, 1, 0] = -J[:, 1, 0] / detJ
    K[:, 1, 1] = J[:, 0, 0] / detJ
    return J, detJ, K


def mesh_has_reversed_edges(mesh) -> bool:
    """True if some interior facet is traversed in opposite directions by its two cells
    (check_eqlb_conditions.py:19-86)."""
    off = mesh.facet_cells_offsets
    interior = np.nonzero(np.diff(off) == 2)[0]
    c0 = mesh.facet_cells[off[interior]]
    c1 = mesh.facet_cells[off[interior] + 1]
    l0 = np.argmax(mesh.cell_facets[c0] == interior[:, None], axis=1)
    l1 = np.argmax(mesh.cell_facets[c1] == interior[:, None], axis=1)
    return bool(np.any(mesh.facet_perm[c0, l0] != mesh.facet_perm[c1, l1]))


def _eval_flux(mesh, k, degree_dg, sigma_eq, flux_dg, ref_points, geom):
    """(sigma_eq + G) at reference points of every cell: [ncells, npts, 2]."""
    J, detJ, K = geom
    rt = ert.HierarchicRT(k)
    dg = Lagrange(degree_dg)
    phi = rt.tabulate(ref_points)  # [q, i, 2]
    psi = dg.tabulate(ref_points)[0]  # [q, j]
    c = sigma_eq.reshape(mesh.ncells, rt.ndofs)
    G = flux_dg.reshape(mesh.ncells, dg.ndofs, 2)
    ref = np.einsum("ci,qid->cqd", c, phi)
    val = np.einsum("cij,cqj->cqi", J, ref) / detJ[:, None, None]
    val += np.einsum("cjd,qj->cqd", G, psi)
    return val


def divergence_residual(mesh, k, sigma_eq, flux_dg, rhs_dg, degree_dg=None):
    """(|| div(sigma_eq + G) - f ||_L2, || f ||_L2): the 'L2 flux-divergence residual' of the
    headline metric (check_divergence_condition compares the same two fields pointwise)."""
    degree_dg = k - 1 if degree_dg is None else degree_dg
    J, detJ, K = cell_geometry(mesh)
    rt = ert.HierarchicRT(k)
    dg = Lagrange(degree_dg)
    qp, qw = make_quadrature_triangle(2 * k)
    divphi = rt.tabulate_div(qp)  # [q, i]
    tab = dg.tabulate(qp, 1)  # [3, q, j]
    c = sigma_eq.reshape(mesh.ncells, rt.ndofs)
    G = flux_dg.reshape(mesh.ncells, dg.ndofs, 2)
    f = rhs_dg.reshape(mesh.ncells, dg.ndofs)
    div_sig = np.einsum("ci,qi->cq", c, divphi) / detJ[:, None]
    # grad psi_j = K^T grad_ref psi_j
    dpsi = np.stack([tab[1], tab[2]], axis=2)  # [q, j, X]
    gpsi = np.einsum("cXd,qjX->cqjd", K, dpsi)
    div_G = np.einsum("cjd,cqjd->cq", G, gpsi)
    fq = np.einsum("cj,qj->cq", f, tab[0])
    res = div_sig + div_G - fq
    w = qw[None, :] * np.abs(detJ)[:, None]
    return float(np.sqrt(np.sum(w * res ** 2))), float(np.sqrt(np.sum(w * fq ** 2)))


def check_divergence_condition(mesh, k, sigma_eq, flux_dg, rhs_dg, degree_dg=None,
                               rtol=1e-5, atol=1e-8) -> bool:
    """Pointwise np.allclose of div(sigma_eq + G) and f (reference :280)."""
    degree_dg = k - 1 if degree_dg is None else degree_dg
    J, detJ, K = cell_geometry(mesh)
    rt = ert.HierarchicRT(k)
    dg = Lagrange(degree_dg)
    rng = np.random.default_rng(0)
    n_points = (k + 3) * (k + 4) // 2
    xs = np.sort(rng.random((2, n_points)), axis=0)
    pts = np.stack([xs[1] - xs[0], 1.0 - xs[1]], axis=1)  # barycentric weights of v1, v2
    divphi = rt.tabulate_div(pts)
    tab = dg.tabulate(pts, 1)
    c = sigma_eq.reshape(mesh.ncells, rt.ndofs)
    G = flux_dg.reshape(mesh.ncells, dg.ndofs, 2)
    f = rhs_dg.reshape(mesh.ncells, dg.ndofs)
    div_sig = np.einsum("ci,qi->cq", c, divphi) / detJ[:, None]
    dpsi = np.stack([tab[1], tab[2]], axis=2)
    gpsi = np.einsum("cXd,qjX->cqjd", K, dpsi)
    div_G = np.einsum("cjd,cqjd->cq", G, gpsi)
    fq = np.einsum("cj,qj->cq", f, tab[0])
    return bool(np.allclose(div_sig + div_G, fq, rtol=rtol, atol=atol))


def _facet_traces(mesh, k, degree_dg, sigma_eq, flux_dg, facets, side):
    """Outward normal flux density (w.r.t. ds of the unit parameter) of sigma_eq + G on the
    given facets, seen from cell `side` (0/1) of each facet, at matching physical points."""
    J, detJ, K = cell_geometry(mesh)
    s, w = make_quadrature_interval(2 * k)
    off = mesh.facet_cells_offsets
    cells = mesh.facet_cells[off[facets] + side]
    lf = np.argmax(mesh.cell_facets[cells] == facets[:, None], axis=1)
    perm = mesh.facet_perm[cells, lf]
    rt = ert.HierarchicRT(k)
    dg = Lagrange(degree_dg)
    c = sigma_eq.reshape(mesh.ncells, rt.ndofs)
    G = flux_dg.reshape(mesh.ncells, dg.ndofs, 2)
    out = np.zeros((facets.size, s.size))
    pf = np.where(np.array(ert.FACET_NORMAL_IS_OUTWARD), 1.0, -1.0)
    for f in range(3):
        for rev in (0, 1):
            sel = np.nonzero((lf == f) & (perm == rev))[0]
            if sel.size == 0:
                continue
            # parameter measured along the GLOBAL low->high direction of the facet
            sl = (1.0 - s) if rev else s
            pts = ert.facet_points(sl)[f]
            phi = rt.tabulate(pts)
            psi = dg.tabulate(pts)[0]
            cc = cells[sel]
            nref = np.array(ert.FACET_NORMALS[f], dtype=float)
            # sigma_eq: reference normal flux n_ref . phi_ref is the DOF density; outward
            # physical flux density = sign(detJ) * pf_f * that
            dens = np.einsum("ci,qi->cq", c[cc], phi @ nref) * (np.sign(detJ[cc]) * pf[f])[:, None]
            # G: pull back detJ K G, then the same functional density
            Gq = np.einsum("cjd,qj->cqd", G[cc], psi)
            pb = np.einsum("cXd,cqd->cqX", K[cc], Gq) * detJ[cc][:, None, None]
            dens += (pb @ nref) * (np.sign(detJ[cc]) * pf[f])[:, None]
            out[sel] = dens
    return out, w


def jump_residual(mesh, k, sigma_eq, flux_dg, degree_dg=None):
    """max over interior facets and points of |[(sigma_eq + G) . n]| (facet-length weighted
    flux density), cf. the per-facet variant check_eqlb_conditions.py:362-473."""
    degree_dg = k - 1 if degree_dg is None else degree_dg
    interior = np.nonzero(np.diff(mesh.facet_cells_offsets) == 2)[0]
    t0, _ = _facet_traces(mesh, k, degree_dg, sigma_eq, flux_dg, interior, 0)
    t1, _ = _facet_traces(mesh, k, degree_dg, sigma_eq, flux_dg, interior, 1)
    return float(np.max(np.abs(t0 + t1))) if interior.size else 0.0


def check_jump_condition(mesh, k, sigma_eq, flux_dg, degree_dg=None, atol=1e-10) -> bool:
    return jump_residual(mesh, k, sigma_eq, flux_dg, degree_dg) < atol


def boundary_flux_residual(mesh, k, sigma_eq, flux_dg, facets, degree_dg=None, boundary_values=None):
    """max |(sigma_eq + G) . n| on the given (flux-BC, homogeneous) boundary facets.
    boundary_values [ncells * k(k+2)] (the global boundary DOFs BoundaryData holds): max deviation of the
    facet DOFs of sigma_eq + G from them instead - check_boundary_conditions of the reference
    (check_eqlb_conditions.py:90-179) in the hierarchic basis, whose facet DOFs are the moments themselves."""
    degree_dg = k - 1 if degree_dg is None else degree_dg
    facets = np.asarray(facets, dtype=np.int64)
    if facets.size == 0:
        return 0.0
    if boundary_values is not None:
        J, detJ, K = cell_geometry(mesh)
        s, w = make_quadrature_interval(2 * k)
        cells = mesh.facet_cells[mesh.facet_cells_offsets[facets]]
        lf = np.argmax(mesh.cell_facets[cells] == facets[:, None], axis=1)
        nrt = k * (k + 2)
        dg = Lagrange(degree_dg)
        c = np.asarray(sigma_eq).reshape(mesh.ncells, nrt)
        G = np.asarray(flux_dg).reshape(mesh.ncells, dg.ndofs, 2)
        bv = np.asarray(boundary_values).reshape(mesh.ncells, nrt)
        worst = 0.0
        for f in range(3):
            sel = np.nonzero(lf == f)[0]
            if sel.size == 0:
                continue
            cc = cells[sel]
            psi = dg.tabulate(ert.facet_points(s)[f])[0]
            Gq = np.einsum("cjd,qj->cqd", G[cc], psi)
            pb = np.einsum("cXd,cqd->cqX", K[cc], Gq) * detJ[cc][:, None, None]
            dens = pb @ np.array(ert.FACET_NORMALS[f], dtype=float)
            for j in range(k):
                dof = c[cc, f * k + j] + dens @ (w * s ** j)
                worst = max(worst, float(np.max(np.abs(dof - bv[cc, f * k + j]))))
        return worst
    t0, _ = _facet_traces(mesh, k, degree_dg, sigma_eq, flux_dg, facets, 0)
    return float(np.max(np.abs(t0)))


def hdiv_seminorm_error(mesh, k, sigma_eq, flux_dg, rhs_exact, degree_dg=None):
    """|| div(sigma_eq + G) - f_exact ||_L2 for convergence-rate tests
    (python/test/unit/test_fluxeqlb_convrate.py:131-135)."""
    degree_dg = k - 1 if degree_dg is None else degree_dg
    J, detJ, K = cell_geometry(mesh)
    rt = ert.HierarchicRT(k)
    dg = Lagrange(degree_dg)
    qp, qw = make_quadrature_triangle(2 * k + 4)
    divphi = rt.tabulate_div(qp)
    tab = dg.tabulate(qp, 1)
    c = sigma_eq.reshape(mesh.ncells, rt.ndofs)
    G = flux_dg.reshape(mesh.ncells, dg.ndofs, 2)
    div_sig = np.einsum("ci,qi->cq", c, divphi) / detJ[:, None]
    dpsi = np.stack([tab[1], tab[2]], axis=2)
    gpsi = np.einsum("cXd,qjX->cqjd", K, dpsi)
    div_G = np.einsum("cjd,cqjd->cq", G, gpsi)
    x0 = mesh.x[mesh.cell_nodes[:, 0], :2]
    xq = x0[:, None, :] + np.einsum("cij,qj->cqi", J, qp)
    fq = rhs_exact(xq[..., 0], xq[..., 1])
    w = qw[None, :] * np.abs(detJ)[:, None]
    return float(np.sqrt(np.sum(w * (div_sig + div_G - fq) ** 2)))


def weak_symmetry_residual(mesh, k, sigma):
    """Assembled vector L_a = (sigma_01 - sigma_10, hat_a) over the P1 test space for a stress with
    the rows sigma [2, ncells*k(k+2)] (broken hierarchic RT_k); returns (max_a |L_a|, L).
    The linear form of check_weak_symmetry_condition, check_eqlb_conditions.py:476-521."""
    J, detJ, K = cell_geometry(mesh)
    rt = ert.HierarchicRT(k)
    qp, qw = make_quadrature_triangle(k + 2)
    phi = rt.tabulate(qp)
    hv = Lagrange(1).tabulate(qp)[0]
    c = np.asarray(sigma).reshape(2, mesh.ncells, rt.ndofs)
    # physical rows: J phi / detJ; only the components (row 0, y) and (row 1, x) are needed
    r0y = np.einsum("cj,ci,qij->cq", J[:, 1, :], c[0], phi) / detJ[:, None]
    r1x = np.einsum("cj,ci,qij->cq", J[:, 0, :], c[1], phi) / detJ[:, None]
    loc = np.einsum("cq,cq,qn->cn", qw[None] * np.abs(detJ)[:, None], r0y - r1x, hv)
    L = np.zeros(mesh.nnodes)
    np.add.at(L, mesh.cell_nodes.ravel(), loc.ravel())
    return float(np.abs(L).max()), L


def check_weak_symmetry_condition(mesh, k, sigma, rtol=1e-5, atol=1e-8) -> bool:
    """np.allclose(L, 0) as the reference (check_eqlb_conditions.py:517)."""
    return bool(np.allclose(weak_symmetry_residual(mesh, k, sigma)[1], 0.0, rtol=rtol, atol=atol))


def stress_estimator_terms(mesh, k, sigma, korn=None, pi_1=1.0):
    """Cell-wise terms of the stress estimator (demo/elasticity/demo_error_estimation.py:100-121), numpy
    statement with quadrature: (int dsig : A dsig, int (C_K (dsig_01 - dsig_10) / 2)^2) for the rows
    sigma [2, ncells*k(k+2)]; A tau = (tau - pi_1 / (2 + 2 pi_1) tr(tau) I) / 2."""
    J, detJ, K = cell_geometry(mesh)
    rt = ert.HierarchicRT(k)
    qp, qw = make_quadrature_triangle(2 * k)
    phi = rt.tabulate(qp)
    c = np.asarray(sigma).reshape(2, mesh.ncells, rt.ndofs)
    val = np.einsum("cxa,rci,qia->rcqx", J, c, phi) / detJ[None, :, None, None]   # [row, cell, q, comp]
    w = qw[None] * np.abs(detJ)[:, None]
    fro = np.einsum("cq,rcqx,rcqx->c", w, val, val)
    tr = val[0, ..., 0] + val[1, ..., 1]
    asym = val[0, ..., 1] - val[1, ..., 0]
    ck = np.ones(mesh.ncells) if korn is None else np.asarray(korn)
    energy = 0.5 * (fro - pi_1 / (2.0 + 2.0 * pi_1) * np.einsum("cq,cq,cq->c", w, tr, tr))
    wsym = 0.25 * ck ** 2 * np.einsum("cq,cq,cq->c", w, asym, asym)
    return energy, wsym


def cell_diameter(mesh):
    """Longest edge per cell (dolfinx::mesh::h on triangles)."""
    x = mesh.x[mesh.cell_nodes, :2]
    e = np.stack([x[:, 1] - x[:, 0], x[:, 2] - x[:, 0], x[:, 2] - x[:, 1]], axis=1)
    return np.sqrt((e ** 2).sum(axis=2)).max(axis=1)


def oscillation_term(mesh, k, sigma, flux_dg, f, qdegree, korn=None):
    """C_K^2 (h_T/pi)^2 || f - div(sigma + G) ||^2_T per cell (demo/poisson/demo_error_estimation.py:96-98),
    numpy statement: f callable f(x, y), sigma [ncells*k(k+2)] broken hierarchic RT_k, flux_dg the DG_{k-1}^2
    part G of the total flux or None."""
    J, detJ, K = cell_geometry(mesh)
    rt = ert.HierarchicRT(k)
    qp, qw = make_quadrature_triangle(qdegree)
    c = np.asarray(sigma).reshape(mesh.ncells, rt.ndofs)
    div = np.einsum("ci,qi->cq", c, rt.tabulate_div(qp)) / detJ[:, None]
    if flux_dg is not None:
        dg = Lagrange(k - 1)
        gr = dg.tabulate(qp, 1)[1:]                               # [2, q, nd] reference derivatives
        G = np.asarray(flux_dg).reshape(mesh.ncells, dg.ndofs, 2)
        # d/dx_i = sum_a K[a, i] d/dX_a
        div = div + np.einsum("cai,aqd,cdi->cq", K, gr, G)
    x0 = mesh.x[mesh.cell_nodes[:, 0], :2]
    xq = x0[:, None, :] + np.einsum("cij,qj->cqi", J, qp)
    r = f(xq[..., 0], xq[..., 1]) - div
    ck = np.ones(mesh.ncells) if korn is None else np.asarray(korn)
    return ck ** 2 * (cell_diameter(mesh) / np.pi) ** 2 * np.abs(detJ) * np.einsum("q,cq->c", qw, r * r)
