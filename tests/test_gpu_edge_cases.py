"""Edge cases of the device path: nothing to equilibrate (empty node mask), the smallest crossed square (one
interior patch of four cells, four two-cell corner patches), launches with empty bins."""

import numpy as np
import pytest

from cases import BCS, make_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", [1, 2, 3])
@pytest.mark.parametrize("scatter", [0, 2])
def test_empty_node_mask_leaves_the_output_alone(k, scatter):
    from dolfinx_eqlb_amd import cpp
    mesh, ft, G, f = make_case(6, k, "neumann_lt")
    eq = cpp.SemiExplicitEquilibrator(cpp.DeviceMesh(mesh), k, 1)
    eq.set_option("scatter", scatter)
    eq.set_boundary(ft, node_mask=np.zeros(mesh.nnodes, dtype=np.uint8))
    assert eq.num_patches == 0
    x0 = np.random.default_rng(1).standard_normal((1, mesh.ncells * k * (k + 2)))
    assert np.array_equal(eq.equilibrate_host(G, f, x0.copy()), x0)          # += nothing
    eq.set_option("accumulate", 0)
    assert not eq.equilibrate_host(G, f, x0.copy()).any()                     # store: every DOF written (zero)


def test_empty_node_mask_ev_and_stress():
    from dolfinx_eqlb_amd import cpp
    from dolfinx_eqlb_amd.eqlb.conforming import conforming_dofmap
    k = 2
    mesh, ft, G, f = make_case(5, k, "dirichlet")
    dm = cpp.DeviceMesh(mesh)
    none = np.zeros(mesh.nnodes, dtype=np.uint8)
    ev = cpp.ConstrainedMinEquilibrator(dm, k, 1)
    ev.set_boundary(ft, node_mask=none)
    _, nd = conforming_dofmap(mesh, k)
    x0 = np.random.default_rng(2).standard_normal((1, nd))
    assert np.array_equal(ev.equilibrate_host(G, f, x0.copy()), x0)
    st = cpp.SemiExplicitEquilibrator(dm, k, 2, reconstruct_stress=True)
    ft2 = np.repeat(ft, 2, axis=0)
    st.set_boundary(ft2, node_mask=none)
    G2, f2 = np.repeat(G, 2, axis=0), np.repeat(f, 2, axis=0)
    y0 = np.random.default_rng(3).standard_normal((2, mesh.ncells * 8))
    assert np.array_equal(st.equilibrate_host(G2, f2, y0.copy()), y0)


@pytest.mark.parametrize("k", [1, 2, 3, 4])
@pytest.mark.parametrize("bc", ["dirichlet", "neumann_bottom"])
def test_smallest_crossed_square(oracle_mod, k, bc):
    """1 x 1 crossed square: 4 cells, one interior patch of 4 cells and four corner patches of 2 cells."""
    from dolfinx_eqlb_amd import cpp
    from dolfinx_eqlb_amd.eqlb.conforming import conforming_dofmap
    from dolfinx_eqlb_amd.mesh import create_unit_square
    from synthetic import facet_types, make_compatible_data
    mesh = create_unit_square(1, shuffle_seed=3)
    ft = facet_types(mesh, BCS[bc])
    G, f = make_compatible_data(mesh, k, ft, seed=5)
    dm = cpp.DeviceMesh(mesh)
    eq = cpp.SemiExplicitEquilibrator(dm, k, 1)
    eq.set_boundary(ft)
    x = eq.equilibrate_host(G[None], f[None])
    ref = oracle_mod.se_reconstruct(mesh, k, ft, G[None], f[None])
    assert np.abs(x - ref).max() <= 1e-10 * np.abs(ref).max()
    ev = cpp.ConstrainedMinEquilibrator(dm, k, 1)
    ev.set_boundary(ft)
    cd, nd = conforming_dofmap(mesh, k)
    refe = oracle_mod.ev_reconstruct(mesh, k, ft, G[None], f[None], cd, nd)
    assert np.abs(ev.equilibrate_host(G[None], f[None]) - refe).max() <= 1e-10 * max(1.0, np.abs(refe).max())
