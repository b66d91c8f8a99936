"""Weak symmetry of equilibrated stresses: pinning of the oracle restatement of
se/solve_patch_weaksym.hpp.  As for the flux path the reference cannot run here, so:
  * acceptance predicates of python/test/unit/test_stressqlb_conditions.py:21-181: divergence,
    jump and BC per row and the weak symmetry condition (sigma_01 - sigma_10, v) = 0 for all
    v in P1 (check_eqlb_conditions.py:476-521), evaluated for the corrector (the synthetic G is
    not symmetric, see synthetic.make_compatible_stress_data);
  * every patch correction equals the unique minimiser of |u_0|^2 + |u_1|^2 over the patch-wise
    H(div=0) spaces subject to the symmetry constraints, computed by an independent dense
    null-space solve on broken RT coefficients.
Not restated: grouped boundary patches for RT_2 with flux BCs on the stress
(se/reconstruction.hpp:170-234) - such configurations violate the symmetry predicate, as the
reference's own expected failures do (test_stressqlb_bcond.py:166)."""

import numpy as np
import pytest
import scipy.linalg as sla

import kkt_reference as kr
from cases import BCS
from dolfinx_eqlb_amd.eqlb import check_eqlb_conditions as chk
from dolfinx_eqlb_amd.elmtlib import e_raviart_thomas as ert
from dolfinx_eqlb_amd.elmtlib.lagrange import Lagrange
from dolfinx_eqlb_amd.elmtlib.quadrature import make_quadrature_triangle
from dolfinx_eqlb_amd.mesh import create_unit_square
from synthetic import facet_types, make_compatible_stress_data


def asym_moments(mesh, k, sig):
    """(sigma_01 - sigma_10, hat_n)_T per cell and vertex [ncells, 3] and the assembled vector."""
    J, detJ, K = chk.cell_geometry(mesh)
    rt = ert.HierarchicRT(k)
    qp, qw = make_quadrature_triangle(k + 2)
    phi = rt.tabulate(qp)
    hv = Lagrange(1).tabulate(qp)[0]
    c = sig.reshape(2, mesh.ncells, rt.ndofs)
    ref = np.einsum("rci,qid->rcqd", c, phi)
    val = np.einsum("cij,rcqj->rcqi", J, ref) / detJ[None, :, None, None]
    asym = val[0, ..., 1] - val[1, ..., 0]
    loc = np.einsum("cq,cq,qn->cn", qw[None] * np.abs(detJ)[:, None], asym, hv)
    r = np.zeros(mesh.nnodes)
    np.add.at(r, mesh.cell_nodes.ravel(), loc.ravel())
    return loc, r


def stress_case(n, k, bc, shuffle=5):
    mesh = create_unit_square(n, shuffle_seed=shuffle, perturb=0.3)
    ft = np.repeat(facet_types(mesh, BCS[bc]), 2, axis=0)
    G, f = make_compatible_stress_data(mesh, k, ft)
    return mesh, ft, G, f


@pytest.mark.parametrize("k,bc", [(2, "dirichlet"), (3, "dirichlet"), (3, "neumann_lt"), (3, "neumann_bottom"),
                                  (2, "neumann_bottom"), (2, "neumann_lt"), (4, "dirichlet"), (4, "neumann_lt")])
def test_stress_conditions(oracle_mod, k, bc):
    """(2, neumann_lt): the corner node between the two flux-BC sides has two cells -> grouped with
    the adjacent internal patch (se/reconstruction.hpp:170-234)."""
    mesh, ft, G, f = stress_case(5, k, bc)
    x0 = oracle_mod.se_reconstruct(mesh, k, ft, G, f)
    xs = oracle_mod.se_reconstruct(mesh, k, ft, G, f, stress=True)
    before = np.abs(asym_moments(mesh, k, x0)[1]).max()
    after = np.abs(asym_moments(mesh, k, xs)[1]).max()
    assert before > 1e-4 and after < 1e-12
    for r in range(2):
        assert chk.check_divergence_condition(mesh, k, xs[r], G[r], f[r])
        assert chk.check_jump_condition(mesh, k, xs[r], G[r], atol=1e-11)
        assert chk.boundary_flux_residual(mesh, k, xs[r], G[r], np.nonzero(ft[r] == 2)[0]) < 1e-11


@pytest.mark.parametrize("k,bc", [(2, "dirichlet"), (3, "neumann_lt"), (4, "neumann_bottom")])
def test_patch_corrections_are_constrained_minimisers(oracle_mod, k, bc):
    mesh, ft, G, f = stress_case(3, k, bc)
    rt = ert.HierarchicRT(k)
    nrt = rt.ndofs
    qp, qw = make_quadrature_triangle(2 * k + 2)
    phi = rt.tabulate(qp)
    hv = Lagrange(1).tabulate(qp)[0]
    worst = 0.0
    for node in range(mesh.nnodes):
        rng = (node, node + 1)
        x0 = oracle_mod.se_reconstruct(mesh, k, ft, G, f, node_range=rng)
        xs = oracle_mod.se_reconstruct(mesh, k, ft, G, f, node_range=rng, stress=True)
        cells = mesh.node_cells[mesh.node_cells_offsets[node]:mesh.node_cells_offsets[node + 1]]
        n = cells.size
        pos = {int(c): i for i, c in enumerate(cells)}
        # homogeneous constraint matrices of both rows (own BC types) and the mass matrix
        Ns, M = [], None
        for r in range(2):
            Bh, Mh = kr.constraint_matrix(mesh, k, node, ft[r])
            Ns.append(sla.null_space(Bh, rcond=1e-11))
            M = Mh
        # symmetry functionals on the patch P1 space: nodes of the patch
        pnodes = sorted(set(mesh.cell_nodes[cells].ravel().tolist()))
        S = np.zeros((len(pnodes), 2, n * nrt))
        cvec = np.zeros(len(pnodes))
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
                cvec[j] += np.sum(wv)
        sig0 = np.stack([x0[r].reshape(mesh.ncells, nrt)[cells].ravel() for r in range(2)])
        ell = -(S[:, 0] @ sig0[0] + S[:, 1] @ sig0[1])
        # reduced unknowns z = (z0, z1), u_r = N_r z_r
        Sz = np.hstack([S[:, 0] @ Ns[0], S[:, 1] @ Ns[1]])
        # constraints are enforced modulo the mean-value multiplier where Sz has the constant
        # in its left null space: project the right-hand side onto range(Sz)
        U, sv, _ = np.linalg.svd(Sz, full_matrices=True)
        rank = int((sv > 1e-10 * sv.max()).sum())
        Ur = U[:, :rank]
        if rank < len(pnodes):  # interior patches: remove the c-direction of the residual
            lam = (np.ones(len(pnodes)) @ ell) / (np.ones(len(pnodes)) @ cvec)
            ell = ell - lam * cvec
        Mz = sla.block_diag(Ns[0].T @ M @ Ns[0], Ns[1].T @ M @ Ns[1])
        # min z^T Mz z  s.t.  Ur^T Sz z = Ur^T ell
        Cz = Ur.T @ Sz
        Mi = np.linalg.inv(Mz)
        z = Mi @ Cz.T @ np.linalg.solve(Cz @ Mi @ Cz.T, Ur.T @ ell)
        nz0 = Ns[0].shape[1]
        u = np.stack([Ns[0] @ z[:nz0], Ns[1] @ z[nz0:]])
        got = np.stack([(xs[r] - x0[r]).reshape(mesh.ncells, nrt)[cells].ravel() for r in range(2)])
        worst = max(worst, np.abs(got - u).max() / max(np.abs(u).max(), 1e-30) if np.abs(u).max() > 1e-12
                    else np.abs(got - u).max())
    assert worst < 1e-9


def _all_traction_case(mesh, k=2):
    ft = np.repeat(facet_types(mesh, lambda mp: np.ones(len(mp), dtype=bool)), 2, axis=0)
    G, f = make_compatible_stress_data(mesh, k, ft)
    return ft, G, f


def test_groups_of_boundary_patches_on_an_unstructured_mesh(oracle_mod):
    """Two groups that share no cell (cases.fan_chain_mesh): all of the reference's conditions hold."""
    from cases import fan_chain_mesh
    k = 2
    mesh = fan_chain_mesh()
    ft, G, f = _all_traction_case(mesh)
    xs = oracle_mod.se_reconstruct(mesh, k, ft, G, f, stress=True)
    assert np.abs(asym_moments(mesh, k, xs)[1]).max() < 1e-12
    for r in range(2):
        assert chk.check_divergence_condition(mesh, k, xs[r], G[r], f[r])
        assert chk.check_jump_condition(mesh, k, xs[r], G[r], atol=1e-11)
        assert chk.boundary_flux_residual(mesh, k, xs[r], G[r], np.nonzero(ft[r] == 2)[0]) < 1e-11


def test_overlapping_groups_of_boundary_patches(oracle_mod):
    """RT_2, tractions on the whole boundary, two groups whose internal patches share cells
    (cases.double_fan_mesh): se/reconstruction.hpp:170-234 treats the groups one after the other, the second
    group's symmetry step sees - and changes - what the first one left on the shared cells.  The row-wise
    conditions hold; weak symmetry does NOT (the second correction undoes the first group's symmetry on the
    shared cells - a limit of the reference's algorithm, cf. its "TODO - Extend patch grouping",
    :168); the result depends on the node order."""
    from cases import double_fan_mesh
    k = 2
    res = []
    for order in (0, 1):
        mesh = double_fan_mesh(order)
        assert sorted(np.diff(mesh.node_cells_offsets).tolist()) == [2, 2, 2, 2, 2, 2, 3, 3, 6, 6]
        ft, G, f = _all_traction_case(mesh)
        xs = oracle_mod.se_reconstruct(mesh, k, ft, G, f, stress=True)
        for r in range(2):
            assert chk.check_divergence_condition(mesh, k, xs[r], G[r], f[r])
            assert chk.check_jump_condition(mesh, k, xs[r], G[r], atol=1e-11)
            assert chk.boundary_flux_residual(mesh, k, xs[r], G[r], np.nonzero(ft[r] == 2)[0]) < 1e-11
        res.append(np.abs(asym_moments(mesh, k, xs)[1]).max())
    assert min(res) > 1e-4
