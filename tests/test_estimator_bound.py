"""End-to-end validation with REAL discrete solutions (the reference's demos and convergence
tests, demo/poisson/demo_error_estimation.py:52-124, python/test/unit/test_fluxeqlb_convrate.py:
131-135): P_k Galerkin solve -> sigma_h = -grad u_h -> equilibration -> the Prager-Synge estimate
    || grad(u - u_h) || <= || sigma_eq || + (h/pi) || f - Pi f ||
must be a guaranteed upper bound with an effectivity index close to one, and
|| div(sigma_eq + sigma_h) - f || must converge with order k.  A wrong sign or orientation
convention anywhere in the chain breaks the bound."""

import numpy as np
import pytest

import galerkin as gk
from dolfinx_eqlb_amd.eqlb import check_eqlb_conditions as chk
from dolfinx_eqlb_amd.mesh import create_unit_square
from synthetic import facet_types


def u_ex(x, y):
    return np.sin(np.pi * x) * np.sin(np.pi * y) * (1.0 + x)


def grad_ex(x, y):
    s, c = np.sin, np.cos
    return (np.pi * c(np.pi * x) * s(np.pi * y) * (1 + x) + s(np.pi * x) * s(np.pi * y),
            np.pi * s(np.pi * x) * c(np.pi * y) * (1 + x))


def f_ex(x, y):
    s, c = np.sin, np.cos
    return 2 * np.pi ** 2 * s(np.pi * x) * s(np.pi * y) * (1 + x) - 2 * np.pi * c(np.pi * x) * s(np.pi * y)


def problem(n, k, shuffle=3):
    mesh = create_unit_square(n, shuffle_seed=shuffle, perturb=0.15)
    fh, osc2, h = gk.project_rhs(mesh, k, f_ex)
    u, cd = gk.solve_poisson(mesh, k, f_ex, f_dg=fh if k == 1 else None)
    G = gk.discrete_flux(mesh, k, u, cd)
    err = gk.energy_error(mesh, k, u, cd, grad_ex)
    return mesh, facet_types(mesh, None), G, fh, osc2, h, err


def flux_norm2(mesh, k, x):
    from test_gpu_estimate import flux_norm2_cells
    return flux_norm2_cells(mesh, k, x)


def estimate(eta_sig2, osc2, h):
    eta_osc2 = (h / np.pi) ** 2 * osc2
    return float(np.sqrt(np.sum(eta_sig2 + eta_osc2 + 2 * np.sqrt(eta_sig2 * eta_osc2))))


@pytest.mark.parametrize("k", [1, 2, 3])
def test_guaranteed_upper_bound_and_rates(oracle_mod, k):
    eff, errs, hdiv = [], [], []
    ns = (4, 8, 16)
    for n in ns:
        mesh, ft, G, fh, osc2, h, err = problem(n, k)
        x = oracle_mod.se_reconstruct(mesh, k, ft, G[None], fh[None])[0]
        res, nrm = chk.divergence_residual(mesh, k, x, G, fh)
        assert res < 1e-9 * nrm and chk.check_jump_condition(mesh, k, x, G, atol=1e-9)
        eta = estimate(flux_norm2(mesh, k, x), osc2, h)
        eff.append(eta / err)
        errs.append(err)
        hdiv.append(chk.hdiv_seminorm_error(mesh, k, x, G, f_ex))
    assert all(e >= 1.0 - 1e-10 for e in eff), eff          # guaranteed bound
    assert eff[-1] < 1.6, eff                                # and a sharp one
    rate_err = np.log2(errs[-2] / errs[-1])
    rate_div = np.log2(hdiv[-2] / hdiv[-1])
    assert rate_err > k - 0.2 and rate_div > k - 0.1, (rate_err, rate_div)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 2, 3])
def test_gpu_pipeline_bound(k):
    """Same chain on the device: projector -> equilibration -> eqlb_se_estimate."""
    from dolfinx_eqlb_amd import cpp
    from dolfinx_eqlb_amd.elmtlib.quadrature import make_quadrature_triangle
    mesh, ft, G, fh, osc2, h, err = problem(12, k)
    dm = cpp.DeviceMesh(mesh)
    # the projected right-hand side from point values, on the device
    qp, qw = make_quadrature_triangle(2 * k + 6)
    J, detJ, K = chk.cell_geometry(mesh)
    x0 = mesh.x[mesh.cell_nodes[:, 0], :2]
    xq = x0[:, None, :] + np.einsum("cij,qj->cqi", J, qp)
    fh_dev = cpp.project_dg(dm, k - 1, qp, qw, f_ex(xq[..., 0], xq[..., 1])[None, :, :, None])[0]
    assert np.allclose(fh_dev, fh, rtol=1e-10, atol=1e-10)
    eq = cpp.SemiExplicitEquilibrator(dm, k, 1)
    eq.set_boundary(ft)
    x = eq.equilibrate_host(G[None], fh_dev[None])
    div2, sig2, jump = cpp.estimate(dm, k, x, G[None], fh_dev[None])
    assert np.sqrt(div2.sum()) < 1e-9 * np.sqrt(np.sum(fh ** 2)) and jump.max() < 1e-9
    eta = estimate(sig2[0], osc2, h)
    assert 1.0 - 1e-10 <= eta / err < 1.6
    # the oscillation term from the device as well (eqlb_oscillation: exact f at the quadrature points,
    # div(sigma_eq + G) evaluated exactly): the whole estimator of demo_error_estimation.py:93-120
    eta_osc2 = cpp.oscillation(dm, k, x, G[None], qp, qw, f_ex(xq[..., 0], xq[..., 1])[None])[0]
    assert np.allclose(eta_osc2, (h / np.pi) ** 2 * osc2, rtol=1e-8, atol=1e-12 * eta_osc2.max())
    eta_dev = float(np.sqrt(np.sum(sig2[0] + eta_osc2 + 2 * np.sqrt(sig2[0] * eta_osc2))))
    assert abs(eta_dev - eta) <= 1e-9 * eta


def ev_flux_error2(mesh, k, xb, G):
    """|| sigma_EV - G ||^2_T per cell (err_sig = grad(u_h) + sigma_eqlb of the reference's
    estimator for a conforming flux, demo_error_estimation.py:97-100), xb: broken coefficients."""
    from dolfinx_eqlb_amd.elmtlib import e_raviart_thomas as ert
    from dolfinx_eqlb_amd.elmtlib.lagrange import Lagrange
    from dolfinx_eqlb_amd.elmtlib.quadrature import make_quadrature_triangle
    J, detJ, K = chk.cell_geometry(mesh)
    rt = ert.HierarchicRT(k)
    dg = Lagrange(k - 1)
    qp, qw = make_quadrature_triangle(2 * k + 2)
    phi = rt.tabulate(qp)
    c = xb.reshape(mesh.ncells, rt.ndofs)
    sig = np.einsum("cdX,ci,qiX->cqd", J, c, phi) / detJ[:, None, None]
    Gq = np.einsum("cjd,qj->cqd", G.reshape(mesh.ncells, dg.ndofs, 2), dg.tabulate(qp)[0])
    return np.einsum("q,cqd,cqd->c", qw, sig - Gq, sig - Gq) * np.abs(detJ)


@pytest.mark.parametrize("k", [1, 2, 3])
def test_ev_guaranteed_upper_bound(oracle_mod, k):
    """The constrained-minimisation flux gives a guaranteed bound as well.  With the flux degree
    equal to the primal degree it is less sharp than the semi-explicit flux for k >= 2 (hat_a G is
    a P_k field that RT_k does not contain; measured effectivity at n = 4/8/16: k = 2
    1.30/1.34/1.51, k = 3 1.36/1.71/2.75 on the flux part), so only the bound is asserted tightly."""
    from dolfinx_eqlb_amd.eqlb.conforming import conforming_dofmap, conforming_to_broken
    mesh, ft, G, fh, osc2, h, err = problem(8, k)
    cd, nd = conforming_dofmap(mesh, k)
    x = oracle_mod.ev_reconstruct(mesh, k, ft, G[None], fh[None], cd, nd)[0]
    xb = conforming_to_broken(mesh, k, x)
    res, nrm = chk.divergence_residual(mesh, k, xb, np.zeros_like(G), fh)
    assert res < 1e-9 * nrm
    eta_ev = estimate(ev_flux_error2(mesh, k, xb, G), osc2, h)
    xs = oracle_mod.se_reconstruct(mesh, k, ft, G[None], fh[None])[0]
    eta_se = estimate(flux_norm2(mesh, k, xs), osc2, h)
    assert 1.0 - 1e-10 <= eta_ev / err < 2.0
    assert eta_ev < 1.5 * eta_se


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 2, 3])
def test_gpu_ev_pipeline_bound(k):
    """EV on the device: equilibrate (broken output) -> eqlb_ev_estimate; guaranteed bound."""
    from dolfinx_eqlb_amd import cpp
    mesh, ft, G, fh, osc2, h, err = problem(10, k)
    dm = cpp.DeviceMesh(mesh)
    ev = cpp.ConstrainedMinEquilibrator(dm, k, 1)
    ev.set_option("output", 1)
    ev.set_boundary(ft)
    xb = ev.equilibrate_host(G[None], fh[None])
    div2, sig2, jump = cpp.estimate(dm, k, xb, G[None], fh[None], conforming_flux=True)
    assert np.sqrt(div2.sum()) < 1e-9 * np.sqrt(np.sum(fh ** 2)) and jump.max() < 1e-9
    ref = ev_flux_error2(mesh, k, xb[0], G)
    # || sigma - G ||^2 is evaluated as (sigma, sigma) - 2 (sigma, G) + (G, G): the difference of
    # O(|G|^2 |T|) terms, so the tolerance is relative to that scale
    scale = np.abs(G).max() ** 2 * np.abs(chk.cell_geometry(mesh)[1]).max()
    assert np.allclose(sig2[0], ref, rtol=1e-6, atol=1e-13 * scale)
    eta = estimate(sig2[0], osc2, h)
    assert 1.0 - 1e-10 <= eta / err < 2.2
