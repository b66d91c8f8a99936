"""EV (constrained minimisation) oracle, SURVEY rows a14-a16: pinned by the independent
null-space minimiser of tests/kkt_reference.py and by the predicates that apply to the EV flux
(div sigma = Pi f, H(div) conformity by construction, flux BC)."""

import numpy as np
import pytest

import kkt_reference as kr
from cases import BCS
from dolfinx_eqlb_amd.eqlb import check_eqlb_conditions as chk
from dolfinx_eqlb_amd.eqlb.conforming import (broken_to_conforming, conforming_dofmap,
                                              conforming_to_broken)
from dolfinx_eqlb_amd.mesh import create_unit_square
from synthetic import boundary_dofs_from_field, facet_types, make_compatible_data


def kkt_sweep(mesh, k, ft, G, f, neumann_flux=None):
    nrt = k * (k + 2)
    x = np.zeros((mesh.ncells, nrt))
    worst = 0.0
    for node in range(mesh.nnodes):
        cells, coef, resid = kr.solve_patch_ev(mesh, k, node, ft, G, f, neumann_flux)
        x[cells] += coef
        worst = max(worst, resid)
    return x.reshape(-1), worst


@pytest.mark.parametrize("k", [1, 2, 3, 4])
@pytest.mark.parametrize("bc", ["dirichlet", "neumann_lt", "neumann_bottom"])
def test_ev_oracle_matches_independent_minimiser(oracle_mod, k, bc):
    mesh = create_unit_square(3, shuffle_seed=11, perturb=0.25)
    ft = facet_types(mesh, BCS[bc])
    G, f = make_compatible_data(mesh, k, ft)
    cd, nd = conforming_dofmap(mesh, k)
    x = oracle_mod.ev_reconstruct(mesh, k, ft, G[None], f[None], cd, nd)[0]
    xb = conforming_to_broken(mesh, k, x)
    ref, resid = kkt_sweep(mesh, k, ft, G, f)
    assert resid < 1e-11
    assert np.abs(xb - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())
    # predicates: div sigma = Pi f (no sigma_proj part), conforming, zero flux on flux-BC facets
    zG = np.zeros_like(G)
    res, nrm = chk.divergence_residual(mesh, k, xb, zG, f)
    assert res < 1e-10 * nrm
    assert chk.check_jump_condition(mesh, k, xb, zG, atol=1e-11)
    assert np.allclose(broken_to_conforming(mesh, k, xb), x, atol=1e-13)
    bf = np.nonzero(ft[0] == 2)[0]
    if bf.size:
        assert np.abs(x[(bf[:, None] * k + np.arange(k)[None, :]).reshape(-1)]).max() < 1e-12


def w_lin(x, y):
    return 1.0 + 0.5 * x - 0.3 * y, -0.7 + 0.2 * x + 0.4 * y


def w_const(x, y):
    return 0 * x + 0.8, 0 * x - 0.6


@pytest.mark.parametrize("k", [1, 2, 3])
@pytest.mark.parametrize("bc", ["neumann_lt", "neumann_bottom"])
def test_ev_oracle_inhomogeneous_bc(oracle_mod, k, bc):
    w = w_const if k == 1 else w_lin
    mesh = create_unit_square(3, shuffle_seed=5, perturb=0.3)
    ft = facet_types(mesh, BCS[bc])
    # EV: sigma_eq.n = w.n on the flux-BC facets, data compatible with that total flux
    G, f = make_compatible_data(mesh, k, ft, neumann_flux=w)
    cd, nd = conforming_dofmap(mesh, k)
    bv = broken_to_conforming(mesh, k, boundary_dofs_from_field(mesh, k, ft[0], w))
    x = oracle_mod.ev_reconstruct(mesh, k, ft, G[None], f[None], cd, nd,
                                  boundary_values=bv[None])[0]
    xb = conforming_to_broken(mesh, k, x)
    ref, resid = kkt_sweep(mesh, k, ft, G, f, neumann_flux=w)
    assert resid < 1e-11
    assert np.abs(xb - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())
    sel = np.nonzero(bv != 0)[0]
    assert sel.size and np.allclose(x[sel], bv[sel], atol=1e-11)


def test_ev_multirhs_equals_single(oracle_mod):
    k = 2
    mesh = create_unit_square(3, shuffle_seed=2, perturb=0.2)
    ft = np.concatenate([facet_types(mesh, BCS["dirichlet"]), facet_types(mesh, BCS["neumann_lt"]),
                         facet_types(mesh, BCS["neumann_bottom"])])
    data = [make_compatible_data(mesh, k, ft[i:i + 1], seed=100 + i) for i in range(3)]
    G = np.stack([d[0] for d in data])
    f = np.stack([d[1] for d in data])
    cd, nd = conforming_dofmap(mesh, k)
    xm = oracle_mod.ev_reconstruct(mesh, k, ft, G, f, cd, nd)
    for i in range(3):
        xs = oracle_mod.ev_reconstruct(mesh, k, ft[i:i + 1], G[i:i + 1], f[i:i + 1], cd, nd)[0]
        assert np.abs(xm[i] - xs).max() < 1e-12 * max(1.0, np.abs(xs).max())


def test_conforming_frame_convention():
    """g_{E,0} is the flux through E in direction n_E = (t_y, -t_x), t = x_hi - x_lo."""
    from dolfinx_eqlb_amd.elmtlib import e_raviart_thomas as ert
    from dolfinx_eqlb_amd.eqlb.check_eqlb_conditions import cell_geometry
    k = 1
    mesh = create_unit_square(3, shuffle_seed=4, perturb=0.3)
    J, detJ, K = cell_geometry(mesh)
    v = np.array([0.7, -1.3])
    vhat = detJ[:, None] * np.einsum("cXd,d->cX", K, v)
    xb = np.stack([vhat @ np.array(ert.FACET_NORMALS[f], dtype=float) for f in range(3)], axis=1)
    g = broken_to_conforming(mesh, k, xb.reshape(-1))[:mesh.nfacets]
    lo = np.minimum(mesh.facet_nodes[:, 0], mesh.facet_nodes[:, 1])
    hi = np.maximum(mesh.facet_nodes[:, 0], mesh.facet_nodes[:, 1])
    t = mesh.x[hi, :2] - mesh.x[lo, :2]
    assert np.allclose(g, v[0] * t[:, 1] - v[1] * t[:, 0], atol=1e-13)
    # both cells of an interior facet give the same global DOF
    assert np.allclose(conforming_to_broken(mesh, k, np.r_[g, np.zeros(0)]), xb.reshape(-1),
                       atol=1e-13)


EV_GOLDEN = ["ev_crossed2_k2_dirichlet", "ev_crossed4_k1_shuffled_neumann",
             "ev_crossed4_k2_shuffled_neumann", "ev_crossed4_k3_shuffled_neumann"]


@pytest.mark.parametrize("name", EV_GOLDEN)
def test_ev_oracle_reproduces_golden(oracle_mod, name):
    import os
    from golden_util import load_case
    mesh, k, ft, G, f, expected = load_case(
        os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    cd, nd = conforming_dofmap(mesh, k)
    x = oracle_mod.ev_reconstruct(mesh, k, ft, G, f, cd, nd)
    assert np.abs(x - expected).max() <= 1e-12 * max(1.0, np.abs(expected).max())
