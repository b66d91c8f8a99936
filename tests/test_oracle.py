"""Pinning of the CPU oracle (oracle/eqlb_oracle.c).

The reference cannot be executed in this pipeline and holds no golden vectors:
"parity unpinned by execution".  The oracle is pinned by mathematics instead:
  * every patch solution equals the unique constrained minimiser computed by an independent
    dense KKT solve (tests/kkt_reference.py),
  * the reference's acceptance predicates hold (divergence, jump, flux BC;
    python/test/unit/test_fluxeqlb_conditions.py:115-136),
  * multi-RHS == single-RHS (python/test/unit/test_fluxeqlb_multirhs.py:149-158),
  * committed golden vectors (tests/golden) guard against regressions of the oracle itself.
"""

import os

import numpy as np
import pytest

import kkt_reference as kr
from cases import make_case
from dolfinx_eqlb_amd.eqlb import check_eqlb_conditions as chk


@pytest.mark.parametrize("k", [1, 2, 3])
@pytest.mark.parametrize("bc", ["dirichlet", "neumann_lt"])
def test_patches_equal_kkt_minimiser(oracle_mod, k, bc):
    mesh, ft, G, f = make_case(3, k, bc)
    worst = 0.0
    for node in range(mesh.nnodes):
        cells, st, sol, u = oracle_mod.se_patch(mesh, k, ft, G, f, node)
        kc, kcoef, resid, nn = kr.solve_patch(mesh, k, node, ft, G[0], f[0])
        assert resid < 1e-10  # constraints are consistent (compatible data)
        order = [list(kc).index(c) for c in cells]
        worst = max(worst, np.abs(sol[0] - kcoef[order]).max())
    assert worst < 5e-12


@pytest.mark.parametrize("k", [1, 2, 3])
@pytest.mark.parametrize("bc", ["dirichlet", "neumann_lt", "neumann_bottom"])
@pytest.mark.parametrize("shuffle", [None, 5])
def test_equilibration_conditions(oracle_mod, k, bc, shuffle):
    """BC, divergence and jump conditions (test_fluxeqlb_conditions.py:115-136); the meshes
    contain reversed edges like the reference's gmsh case (utils.py:136-139)."""
    mesh, ft, G, f = make_case(6, k, bc, shuffle=shuffle)
    assert chk.mesh_has_reversed_edges(mesh)
    sig = oracle_mod.se_reconstruct(mesh, k, ft, G, f)[0]
    assert chk.check_divergence_condition(mesh, k, sig, G[0], f[0])
    res, nrm = chk.divergence_residual(mesh, k, sig, G[0], f[0])
    assert res < 1e-10 * nrm
    assert chk.check_jump_condition(mesh, k, sig, G[0], atol=1e-11)
    assert chk.boundary_flux_residual(mesh, k, sig, G[0], np.nonzero(ft[0] == 2)[0]) < 1e-11


@pytest.mark.parametrize("k", [1, 2, 3])
def test_multirhs_equals_single_rhs(oracle_mod, k):
    """Different BC sets per RHS equilibrated together == one by one
    (test_fluxeqlb_multirhs.py:149-158); exercises the mixed / reversed patch types."""
    from cases import BCS
    from dolfinx_eqlb_amd.mesh import create_unit_square
    from synthetic import facet_types, make_compatible_data
    mesh = create_unit_square(4, shuffle_seed=3, perturb=0.2)
    names = ["neumann_lt", "dirichlet", "neumann_bottom"]
    fts = [facet_types(mesh, BCS[n])[0] for n in names]
    data = [make_compatible_data(mesh, k, ft[None], seed=11 + i) for i, ft in enumerate(fts)]
    ft = np.stack(fts)
    G = np.stack([d[0] for d in data])
    f = np.stack([d[1] for d in data])
    together = oracle_mod.se_reconstruct(mesh, k, ft, G, f)
    for i in range(len(names)):
        single = oracle_mod.se_reconstruct(mesh, k, ft[i:i + 1], G[i:i + 1], f[i:i + 1])[0]
        assert np.allclose(together[i], single, rtol=1e-10, atol=1e-12)
        assert chk.check_divergence_condition(mesh, k, together[i], G[i], f[i])
        assert chk.check_jump_condition(mesh, k, together[i], G[i], atol=1e-11)
        assert chk.boundary_flux_residual(mesh, k, together[i], G[i],
                                          np.nonzero(ft[i] == 2)[0]) < 1e-11


def test_accumulates_like_reference(oracle_mod):
    mesh, ft, G, f = make_case(3, 2)
    a = oracle_mod.se_reconstruct(mesh, 2, ft, G, f)
    b = oracle_mod.se_reconstruct(mesh, 2, ft, G, f, flux_hdiv=a.copy())
    assert np.allclose(b, 2 * a)


def test_one_cell_patch_is_an_error(oracle_mod):
    """se/Patch.cpp:353-359: a right-diagonal mesh has corner patches with a single cell."""
    from dolfinx_eqlb_amd.mesh import create_unit_square
    from synthetic import facet_types
    mesh = create_unit_square(2, diagonal="right")
    ft = facet_types(mesh)
    with pytest.raises(RuntimeError):
        oracle_mod.se_reconstruct(mesh, 1, ft, np.zeros((1, mesh.ncells * 2)),
                                  np.zeros((1, mesh.ncells)))


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz") and not f.startswith(("ev_", "stress_bcond_")))
                         if os.path.isdir(GOLDEN) else [])
def test_golden_vectors(oracle_mod, name):
    from golden_util import load_case
    mesh, k, ft, G, f, expected = load_case(os.path.join(GOLDEN, name))
    got = oracle_mod.se_reconstruct(mesh, k, ft, G, f)
    assert np.allclose(got, expected, rtol=1e-11, atol=1e-13)
