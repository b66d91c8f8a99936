"""On-device acceptance predicates / estimator quantities (eqlb_se_estimate)
against their numpy statements in dolfinx_eqlb_amd/eqlb/check_eqlb_conditions.py."""

import numpy as np
import pytest

from cases import make_case
from dolfinx_eqlb_amd.elmtlib import e_raviart_thomas as ert
from dolfinx_eqlb_amd.elmtlib.quadrature import make_quadrature_triangle
from dolfinx_eqlb_amd.eqlb import check_eqlb_conditions as chk

pytestmark = pytest.mark.gpu


def flux_norm2_cells(mesh, k, x):
    J, detJ, K = chk.cell_geometry(mesh)
    rt = ert.HierarchicRT(k)
    qp, qw = make_quadrature_triangle(2 * k + 1)
    phi = rt.tabulate(qp)  # [q, i, X]
    c = x.reshape(mesh.ncells, rt.ndofs)
    sig_ref = np.einsum("ci,qiX->cqX", c, phi)
    sig = np.einsum("cdX,cqX->cqd", J, sig_ref) / detJ[:, None, None]
    return np.einsum("q,cqd,cqd->c", qw, sig, sig) * np.abs(detJ)


@pytest.mark.parametrize("k", [1, 2, 3])
def test_estimate_on_equilibrated_flux(oracle_mod, k):
    from dolfinx_eqlb_amd import cpp
    mesh, ft, G, f = make_case(6, k, "neumann_lt")
    x = oracle_mod.se_reconstruct(mesh, k, ft, G, f)
    dm = cpp.DeviceMesh(mesh)
    div2, sig2, jump = cpp.estimate(dm, k, x, G, f)
    scale = np.abs(f).max() ** 2 * np.abs(chk.cell_geometry(mesh)[1]).max()
    assert div2.max() < 1e-20 * max(scale, 1.0) + 1e-22      # equilibrated: div condition holds
    assert jump.max() < 1e-11                                  # and the flux is H(div) conforming
    ref = flux_norm2_cells(mesh, k, x[0])
    assert np.allclose(sig2[0], ref, rtol=1e-12, atol=1e-14 * ref.max())


@pytest.mark.parametrize("k", [1, 2, 3])
def test_estimate_detects_violations(k):
    """Random (non-equilibrated) coefficients: the residuals equal the numpy predicates."""
    from dolfinx_eqlb_amd import cpp
    mesh, ft, G, f = make_case(5, k, "dirichlet")
    rng = np.random.default_rng(4)
    x = rng.standard_normal((1, mesh.ncells * k * (k + 2)))
    dm = cpp.DeviceMesh(mesh)
    div2, sig2, jump = cpp.estimate(dm, k, x, G, f)
    res, nrm = chk.divergence_residual(mesh, k, x[0], G[0], f[0])
    assert abs(np.sqrt(div2.sum()) - res) < 1e-11 * res
    ref = flux_norm2_cells(mesh, k, x[0])
    assert np.allclose(sig2[0], ref, rtol=1e-12)
    # jumps: interior facets only, positive, and zero exactly on the boundary
    bnd = np.diff(mesh.facet_cells_offsets) == 1
    assert np.all(jump[0][bnd] == 0.0) and np.all(jump[0][~bnd] > 0.0)
    # a conforming field has no jump: sigma_eq = 0, G = constant vector
    Gc = np.tile(np.array([0.3, -1.1]), mesh.ncells * k * (k + 1) // 2)[None]
    _, _, j0 = cpp.estimate(dm, k, np.zeros_like(x), Gc, f)
    assert j0.max() < 1e-13
