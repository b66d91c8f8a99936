"""Hierarchic RT element and tables without Basix (properties of
python/test/unit/test_hierarchic_rt.py:35-110 that do not need Basix)."""

from fractions import Fraction

import numpy as np
import pytest

from dolfinx_eqlb_amd.elmtlib import e_raviart_thomas as ert
from dolfinx_eqlb_amd.elmtlib import polynomials as P
from dolfinx_eqlb_amd.elmtlib.lagrange import Lagrange, facet_closure_dofs
from dolfinx_eqlb_amd.elmtlib.quadrature import make_quadrature_interval, make_quadrature_triangle


@pytest.mark.parametrize("k", [1, 2, 3, 4])
def test_duality_exact(k):
    e = ert.HierarchicRT(k)
    for i in range(e.ndofs):
        d = e.apply_functionals(e.basis[i])
        assert d == [Fraction(int(i == j)) for j in range(e.ndofs)]


@pytest.mark.parametrize("k", [1, 2, 3, 4])
def test_space_is_rt(k):
    """Normal traces on every facet are polynomials of degree k-1, div in P_{k-1}, and the
    non-facet functions have vanishing normal traces."""
    e = ert.HierarchicRT(k)
    for i, (px, py) in enumerate(e.basis):
        div = e.divergence(i)
        assert all(a + b <= k - 1 for (a, b) in div)
        for f in range(3):
            xs, ys = ert.FACET_PARAM[f]
            n = ert.FACET_NORMALS[f]
            tr = P.restrict_to_line(P.add(P.scale(px, n[0]), P.scale(py, n[1])), xs, ys)
            assert all(c == 0 for c in tr[k:])
            if i >= 3 * k or i // k != f:
                assert all(c == 0 for c in tr)


@pytest.mark.parametrize("k", [2, 3, 4])
def test_higher_facet_and_interior_functions_divergence_free(k):
    """Only the zero-order facet functions carry mean divergence; facet functions j>=1 and
    the e2 functions are divergence free."""
    e = ert.HierarchicRT(k)
    for i in range(e.ndofs):
        kind = e.functional_kind(i)
        div = e.divergence(i)
        if (kind[0] == "facet" and kind[2] >= 1) or kind[0] == "e2":
            assert div == {}
        if kind[0] == "facet" and kind[2] == 0:
            assert abs(P.integrate_triangle(div)) == 1


def test_zero_order_divergence_k2():
    e = ert.HierarchicRT(2)
    assert e.divergence(0) == {(0, 0): -18, (1, 0): 24, (0, 1): 24}


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_lagrange_nodal(deg):
    L = Lagrange(deg)
    nodes = np.array([[float(a), float(b)] for a, b in L.nodes])
    assert np.allclose(L.tabulate(nodes)[0], np.eye(L.ndofs), atol=1e-13)
    if deg:
        for f, dofs in enumerate(facet_closure_dofs(deg)):
            s = np.linspace(0, 1, 7)
            pts = ert.facet_points(s)[f]
            tab = L.tabulate(pts)[0]
            others = [i for i in range(L.ndofs) if i not in dofs]
            assert np.allclose(tab[:, others], 0, atol=1e-13)


@pytest.mark.parametrize("deg", [1, 2, 5, 7])
def test_quadrature_exact(deg):
    from math import factorial
    pts, w = make_quadrature_triangle(deg)
    for a in range(deg + 1):
        for b in range(deg + 1 - a):
            ex = factorial(a) * factorial(b) / factorial(a + b + 2)
            assert abs((w * pts[:, 0] ** a * pts[:, 1] ** b).sum() - ex) < 1e-14
    s, ws = make_quadrature_interval(deg)
    assert np.allclose(s, 1 - s[::-1])  # symmetric (reversed facets rely on it)
    for a in range(deg + 1):
        assert abs((ws * s ** a).sum() - 1 / (a + 1)) < 1e-14


def test_reversal_transformation():
    T = ert.reversal_transformation(3)
    assert np.array_equal(T, np.array([[-1, -1, -1], [0, 1, 2], [0, 0, -1]]))


@pytest.mark.parametrize("k,deg", [(1, 0), (2, 1), (3, 2)])
def test_generated_tensors_match_quadrature(k, deg):
    """tools/gen_tables.py (exact) vs. quadrature over the float tabulation."""
    from gen_tables import tables_float
    t = tables_float(k, deg)
    rt = ert.HierarchicRT(k)
    qp, qw = make_quadrature_triangle(2 * k)
    phi = rt.tabulate(qp)
    S0 = np.einsum("q,qi,qj->ij", qw, phi[:, :, 0], phi[:, :, 0])
    S2 = np.einsum("q,qi,qj->ij", qw, phi[:, :, 1], phi[:, :, 1])
    S1 = np.einsum("q,qi,qj->ij", qw, phi[:, :, 0], phi[:, :, 1])
    assert np.allclose(t["S"][0], S0, atol=1e-13)
    assert np.allclose(t["S"][2], S2, atol=1e-13)
    assert np.allclose(t["S"][1], S1 + S1.T, atol=1e-13)
    dg, hat = Lagrange(deg), Lagrange(1)
    s, w = make_quadrature_interval(2 * k + 2)
    for f in range(3):
        pts = ert.facet_points(s)[f]
        Fq = np.einsum("q,qi,qn,qj->nij", w, dg.tabulate(pts)[0], hat.tabulate(pts)[0],
                       np.stack([s ** j for j in range(k)], axis=1))
        assert np.allclose(t["F"][f], Fq, atol=1e-13)
