"""Inhomogeneous flux (Neumann) boundary conditions, SURVEY 8(f)-1: per-patch boundary DOFs
hat_a * g (BoundaryData::calculate_patch_bc, base/BoundaryData.cpp:687-745).  Oracle pinned by
the independent KKT solve with the prescribed normal flux; HIP path against the oracle."""

import numpy as np
import pytest

import kkt_reference as kr
from cases import BCS
from dolfinx_eqlb_amd.eqlb import check_eqlb_conditions as chk
from dolfinx_eqlb_amd.mesh import create_unit_square
from synthetic import boundary_dofs_from_field, facet_types, make_compatible_data


def w_lin(x, y):
    return 1.0 + 0.5 * x - 0.3 * y, -0.7 + 0.2 * x + 0.4 * y


def w_const(x, y):
    return 0 * x + 0.8, 0 * x - 0.6


def case(n, k, bc):
    w = w_const if k == 1 else w_lin
    mesh = create_unit_square(n, shuffle_seed=5, perturb=0.3)
    ft = facet_types(mesh, BCS[bc])
    G, f = make_compatible_data(mesh, k, ft, neumann_flux=w)
    bv = boundary_dofs_from_field(mesh, k, ft[0], w)
    return mesh, ft, G, f, bv, w


@pytest.mark.parametrize("k", [1, 2, 3])
@pytest.mark.parametrize("bc", ["neumann_lt", "neumann_bottom"])
def test_oracle_inhomogeneous_bc(oracle_mod, k, bc):
    mesh, ft, G, f, bv, w = case(3, k, bc)
    x = oracle_mod.se_reconstruct(mesh, k, ft, G[None], f[None], boundary_values=bv[None])[0]
    res, nrm = chk.divergence_residual(mesh, k, x, G, f)
    assert res < 1e-10 * nrm and chk.check_jump_condition(mesh, k, x, G, atol=1e-11)
    worst = 0.0
    for node in range(mesh.nnodes):
        cells, st, sol, u = oracle_mod.se_patch(mesh, k, ft, G[None], f[None], node,
                                                boundary_values=bv[None])
        kc, kcoef, resid, nn = kr.solve_patch(mesh, k, node, ft, G, f, neumann_flux=w)
        order = [list(kc).index(c) for c in cells]
        worst = max(worst, np.abs(sol[0] - kcoef[order]).max(), resid)
    assert worst < 1e-11
    # boundary condition: facet DOFs of sigma_eq + G equal the boundary DOFs on the flux-BC facets
    nrt = k * (k + 2)
    tot = x + boundary_dofs_from_field(mesh, k, ft[0], _as_field(mesh, k, G))
    sel = np.nonzero(bv != 0)[0]
    assert sel.size and np.allclose(tot[sel], bv[sel], atol=1e-11)


def _as_field(mesh, k, G):
    """The DG field G as a callable evaluated cell-wise is not available through (x, y) alone;
    boundary_dofs_from_field only samples boundary facets of single cells, so wrap a lookup."""
    from dolfinx_eqlb_amd.elmtlib.lagrange import Lagrange
    from dolfinx_eqlb_amd.eqlb.check_eqlb_conditions import cell_geometry
    dg = Lagrange(k - 1)
    J, detJ, K = cell_geometry(mesh)
    Gc = G.reshape(mesh.ncells, dg.ndofs, 2)
    bf = mesh.boundary_facets()
    bcells = mesh.facet_cells[mesh.facet_cells_offsets[bf]]
    x0 = mesh.x[mesh.cell_nodes[:, 0], :2]

    def field(x, y):
        # x, y: [ncells_sel, nq] points on boundary facets; find the owning cell by locating the
        # reference coordinates among the boundary cells
        out = np.zeros(x.shape + (2,))
        for i in range(x.shape[0]):
            for c in bcells:
                X = (np.stack([x[i], y[i]], axis=1) - x0[c]) @ K[c].T
                if np.all(X > -1e-9) and np.all(X.sum(axis=1) < 1 + 1e-9):
                    psi = dg.tabulate(X)[0]
                    out[i] = psi @ Gc[c]
                    break
        return out[..., 0], out[..., 1]

    return field


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 2, 3])
@pytest.mark.parametrize("bc", ["neumann_lt", "neumann_bottom"])
@pytest.mark.parametrize("solver", [0, 1])
def test_gpu_inhomogeneous_bc(oracle_mod, k, bc, solver):
    from dolfinx_eqlb_amd import cpp
    mesh, ft, G, f, bv, w = case(7, k, bc)
    eq = cpp.SemiExplicitEquilibrator(cpp.DeviceMesh(mesh), k, 1)
    eq.set_option("solver", solver)
    eq.set_boundary(ft, boundary_values=bv)
    x = eq.equilibrate_host(G[None], f[None])[0]
    ref = oracle_mod.se_reconstruct(mesh, k, ft, G[None], f[None], boundary_values=bv[None])[0]
    assert np.abs(x - ref).max() <= 1e-11 * np.abs(ref).max()


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 2])
def test_class_api_with_inhomogeneous_fluxbc(oracle_mod, k):
    """FluxEqlbSE / FluxEqlbEV with `fluxbc(value=callable, facets)` through `boundarydata`, the
    reference's call sequence (FluxEqlbSE.py:118-174, bcs.py:25-217)."""
    from dolfinx_eqlb_amd.eqlb import FluxEqlbEV, FluxEqlbSE, fluxbc
    from dolfinx_eqlb_amd.eqlb.conforming import broken_to_conforming, conforming_dofmap
    mesh, ft, G, f, bv, w = case(6, k, "neumann_lt")
    bf = mesh.boundary_facets()
    prime, dual = bf[ft[0][bf] == 1], bf[ft[0][bf] == 2]
    se = FluxEqlbSE(k, mesh, [f], [G])
    se.set_boundary_conditions([prime], [[fluxbc(w, dual)]])
    assert np.allclose(se.list_bfunctions[0], bv)
    se.equilibrate_fluxes()
    ref = oracle_mod.se_reconstruct(mesh, k, ft, G[None], f[None], boundary_values=bv[None])[0]
    assert np.abs(se.get_reconstructed_fluxes(0)[0] - ref).max() <= 1e-11 * np.abs(ref).max()
    ev = FluxEqlbEV(k, mesh, [f], [G])
    ev.set_boundary_conditions([prime], [[fluxbc(w, dual)]])
    ev.equilibrate_fluxes()
    cd, nd = conforming_dofmap(mesh, k)
    bvc = broken_to_conforming(mesh, k, bv)
    refe = oracle_mod.ev_reconstruct(mesh, k, ft, G[None], f[None], cd, nd, boundary_values=bvc[None])[0]
    assert np.abs(ev.get_reconstructed_fluxes(0) - refe).max() <= 1e-11 * max(1.0, np.abs(refe).max())
    with pytest.raises(RuntimeError, match="does not match"):
        from dolfinx_eqlb_amd.eqlb import boundarydata
        boundarydata([[]], [], (mesh, k), True, [prime], False)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [2, 3])
def test_class_api_ev_in_another_basis(oracle_mod, k):
    """FluxEqlbEV with BoundaryData.set_basis_transform (the hook for Basix' RT_k): the
    boundary DOFs the BoundaryData computes from the FluxBC stay facet moments (option "boundary_basis" = 1
    inside the module), the output arrives in the target basis.  Reference: C / R applied in numpy to the
    broken hierarchic result."""
    from test_gpu_ev import _transform_reference
    from dolfinx_eqlb_amd import cpp
    from dolfinx_eqlb_amd.eqlb import FluxEqlbEV, fluxbc
    from dolfinx_eqlb_amd.eqlb.conforming import broken_to_conforming, conforming_dofmap, reversal_matrix
    mesh, ft, G, f, bv, w = case(6, k, "neumann_lt")
    nrt = k * (k + 2)
    bf = mesh.boundary_facets()
    prime, dual = bf[ft[0][bf] == 1], bf[ft[0][bf] == 2]
    cd, nd = conforming_dofmap(mesh, k)
    ev = FluxEqlbEV(k, mesh, [f], [G])
    ev.set_boundary_conditions([prime], [[fluxbc(w, dual)]])
    ev.equilibrate_fluxes()
    x_hier = np.array(ev.get_reconstructed_fluxes(0), copy=True)
    base = cpp.ConstrainedMinEquilibrator(cpp.DeviceMesh(mesh), k, 1)
    base.set_option("output", 1)
    base.set_boundary(ft, boundary_values=broken_to_conforming(mesh, k, bv)[None])
    xb = base.equilibrate_host(G[None], f[None])[0]
    rng = np.random.default_rng(3)
    C = rng.standard_normal((nrt, nrt))
    for fl in range(3):
        C[fl * k:(fl + 1) * k, :] = 0.0
        C[fl * k:(fl + 1) * k, fl * k:(fl + 1) * k] = rng.standard_normal((k, k)) + 2.0 * np.eye(k)
    R = rng.standard_normal((k, k)) + 2.0 * np.eye(k)
    ev2 = FluxEqlbEV(k, mesh, [f], [G])
    ev2.set_boundary_conditions([prime], [[fluxbc(w, dual)]])
    ev2.boundary_data.set_basis_transform(C, R)
    ev2.equilibrate_fluxes()
    ref = _transform_reference(mesh, k, xb, C, R, cd, nd)
    assert np.abs(ev2.get_reconstructed_fluxes(0) - ref).max() <= 1e-11 * np.abs(ref).max()
    # the hierarchic basis written as a transform
    Ci = np.diag(np.concatenate([-np.ones(3 * k), np.ones(nrt - 3 * k)]))
    ev3 = FluxEqlbEV(k, mesh, [f], [G])
    ev3.set_boundary_conditions([prime], [[fluxbc(w, dual)]])
    ev3.boundary_data.set_basis_transform(Ci, -reversal_matrix(k))
    ev3.equilibrate_fluxes()
    assert np.abs(ev3.get_reconstructed_fluxes(0) - x_hier).max() <= 1e-12 * np.abs(x_hier).max()
    with pytest.raises(RuntimeError, match="k\\(k\\+2\\)"):
        ev3.boundary_data.set_basis_transform(np.eye(3), None)
