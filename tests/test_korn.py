"""Korn constant estimate.  The reference has no test for it and no executable
reference exists here: the oracle is a line-by-line restatement of
OrientedPatch::estimate_squared_korn_constant (se/Patch.cpp:130-334), checked for the
invariances the formula must have; the HIP kernel is compared with the oracle."""

import numpy as np
import pytest

from cases import make_case
from dolfinx_eqlb_amd.mesh import create_mesh, create_unit_square
from synthetic import facet_types


def test_oracle_korn_invariances(oracle_mod):
    mesh = create_unit_square(5, perturb=0.3)
    ft = facet_types(mesh)
    k0 = oracle_mod.se_korn(mesh, ft)
    assert np.all(k0 >= 3 * 3 * 2.0)  # 2/sin^2 >= 2 per patch, 3 patches per cell, factor gdim+1
    # scaling + rotation of the geometry leave the angles unchanged
    th = 0.7
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    m2 = create_mesh(3.5 * mesh.x[:, :2] @ R.T + 1.0, mesh.cell_nodes)
    assert np.allclose(oracle_mod.se_korn(m2, ft), k0, rtol=1e-10)
    # interior patches do not depend on the local vertex order of the cells
    m3 = create_unit_square(5, perturb=0.3, shuffle_seed=9)
    k3 = oracle_mod.se_korn(m3, facet_types(m3))
    interior_cells = np.all((mesh.x[mesh.cell_nodes, :2] > 0.21) & (mesh.x[mesh.cell_nodes, :2] < 0.79),
                            axis=(1, 2))
    assert interior_cells.any()
    assert np.allclose(k3[interior_cells], k0[interior_cells], rtol=1e-10)


def test_oracle_korn_regular_patch_value(oracle_mod):
    """Uniform crossed mesh: an interior corner node sees right triangles with 45 degree angles at
    the ring -> theta_min = pi/4 -> c_K^2 = 2/sin^2(pi/8)."""
    mesh = create_unit_square(4)
    k = oracle_mod.se_korn(mesh, facet_types(mesh), node_range=(12, 13))  # node (2,2), interior
    assert np.isclose(k.max(), 3 * 2 / np.sin(np.pi / 8) ** 2, rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("bc", ["dirichlet", "neumann_lt"])
def test_gpu_korn_equals_oracle(oracle_mod, bc):
    from dolfinx_eqlb_amd import cpp
    mesh, ft, G, f = make_case(9, 2, bc, shuffle=5)
    eq = cpp.SemiExplicitEquilibrator(cpp.DeviceMesh(mesh), 2, 1, estimate_korn=True)
    eq.set_boundary(ft)
    x, korn = eq.equilibrate_host_with_kornconst(G, f)
    ref = oracle_mod.se_korn(mesh, ft)
    assert np.abs(korn - ref).max() <= 1e-11 * ref.max()
    xr = oracle_mod.se_reconstruct(mesh, 2, ft, G, f)
    assert np.abs(x - xr).max() <= 1e-11 * np.abs(xr).max()
    # accumulation like the reference
    _, korn2 = eq.equilibrate_host_with_kornconst(G, f, korn=korn.copy())
    assert np.allclose(korn2, 2 * korn, rtol=1e-14)


@pytest.mark.gpu
def test_flux_eqlb_se_class_with_korn(oracle_mod):
    from dolfinx_eqlb_amd.eqlb.FluxEqlbSE import FluxEqlbSE
    mesh, ft, G, f = make_case(6, 2)
    eq = FluxEqlbSE(2, mesh, [f[0]], [G[0]], estimate_korn_constant=True)
    eq.set_boundary_conditions([np.nonzero(ft[0] == 1)[0]], [[]])
    eq.equilibrate_fluxes()
    assert np.allclose(eq.get_korn_constants(), np.sqrt(oracle_mod.se_korn(mesh, ft)), rtol=1e-11)
