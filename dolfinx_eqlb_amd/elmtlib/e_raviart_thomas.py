"""Hierarchic Raviart-Thomas element RT_k on the reference triangle, without Basix.

Mirrors `create_hierarchic_rt` of the reference
(python/dolfinx_eqlb/elmtlib/e_raviart_thomas.py:14-196): the basis is dual to

  * facet moments   l_{f,j}(v) = int_0^1 v(x_f(s)) . n_f s^j ds,  j < k, with the
    facet parametrisations (1-s, s), (0, s), (s, 0) and the un-normalised
    reference "normals" [-1,-1], [-1,0], [0,1]        (reference :74-90),
  * divergence moments int_T div v x^l y^m, l+m >= 1   (reference :105-112),
  * e_2 moments       int_T v_y x^l y^m               (reference :116-122),

in exactly that DOF order, on the space RT_k = P_{k-1}^2 + (x,y) P~_{k-1}
(degree convention of the reference: lowest order is k = 1).  The reference
delegates the dual-basis inversion to `basix.create_custom_element`; here it is
done in exact rational arithmetic, so the coefficient table carries no rounding.

The discontinuous variant of the reference only moves the functionals into the
cell interior (no DOF transformations); DOF order and basis are identical, so
one class serves both.
"""

from fractions import Fraction

import numpy as np

from . import polynomials as P

# reference :77-82
FACET_PARAM = (((1, -1), (0, 1)), ((0, 0), (0, 1)), ((0, 1), (0, 0)))  # (x(s), y(s)) as (c0, c1)
FACET_NORMALS = ((-1, -1), (-1, 0), (0, 1))
# functional of facet f measures the OUTWARD flux iff True (facets 0 and 2 measure inward flux)
FACET_NORMAL_IS_OUTWARD = (False, True, False)
# vertices of facet f, low local index first (direction of the parameter s)
FACET_VERTICES = ((1, 2), (0, 2), (0, 1))


def _spanning_set(k):
    """Vector polynomials spanning RT_k (as (px, py) pairs)."""
    span = []
    for deg in range(k):
        for a in range(deg, -1, -1):
            b = deg - a
            span.append((P.monomial(a, b), {}))
            span.append(({}, P.monomial(a, b)))
    for a in range(k - 1, -1, -1):
        b = k - 1 - a
        span.append((P.monomial(a + 1, b), P.monomial(a, b + 1)))
    return span


def div_moment_exponents(k):
    """(l, m) of the divergence-moment DOFs, reference loop order (:105-112)."""
    return [(l, m) for l in range(k) for m in range(k - l) if l + m >= 1]


def e2_moment_exponents(k):
    """(l, m) of the e_2-moment DOFs, reference loop order (:116-122)."""
    return [(l, m) for l in range(1, k - 1) for m in range(k - 1 - l)]


class HierarchicRT:
    """RT_k with the hierarchic (Boffi-Brezzi-Fortin type) moment basis."""

    def __init__(self, degree: int):
        if degree < 1:
            raise ValueError("Degree must be at least 1")
        k = self.degree = degree
        self.ndofs = k * (k + 2)
        self.ndofs_fct = k
        self.ndofs_div = k * (k + 1) // 2 - 1
        self.ndofs_add = (k - 1) * (k - 2) // 2
        self.div_exponents = div_moment_exponents(k)
        self.e2_exponents = e2_moment_exponents(k)
        assert 3 * k + len(self.div_exponents) + len(self.e2_exponents) == self.ndofs

        span = _spanning_set(k)
        assert len(span) == self.ndofs
        D = [[self._functional(i, w) for w in span] for i in range(self.ndofs)]
        Dinv = P.solve_exact(D, [[Fraction(int(i == j)) for j in range(self.ndofs)]
                                 for i in range(self.ndofs)])
        # phi_i = sum_j Dinv[j][i] w_j
        self.basis = []
        for i in range(self.ndofs):
            px, py = {}, {}
            for j, (wx, wy) in enumerate(span):
                c = Dinv[j][i]
                if c != 0:
                    px = P.add(px, P.scale(wx, c))
                    py = P.add(py, P.scale(wy, c))
            self.basis.append((px, py))

    # -- the dual functionals (exact) -----------------------------------------------------------
    def functional_kind(self, i):
        k = self.degree
        if i < 3 * k:
            return ("facet", i // k, i % k)
        i -= 3 * k
        if i < len(self.div_exponents):
            return ("div",) + self.div_exponents[i]
        return ("e2",) + self.e2_exponents[i - len(self.div_exponents)]

    def _functional(self, i, w):
        kind = self.functional_kind(i)
        wx, wy = w
        if kind[0] == "facet":
            _, f, j = kind
            xs, ys = FACET_PARAM[f]
            n = FACET_NORMALS[f]
            tr = P.restrict_to_line(P.add(P.scale(wx, n[0]), P.scale(wy, n[1])), xs, ys)
            return P.integrate_unit_interval(tr, j)
        if kind[0] == "div":
            _, l, m = kind
            return P.integrate_triangle(P.mul(P.add(P.ddx(wx), P.ddy(wy)), P.monomial(l, m)))
        _, l, m = kind
        return P.integrate_triangle(P.mul(wy, P.monomial(l, m)))

    def apply_functionals(self, w):
        """All DOFs of an exact vector polynomial w = (px, py)."""
        return [self._functional(i, w) for i in range(self.ndofs)]

    # -- float tabulation -----------------------------------------------------------------------
    def tabulate(self, points):
        """phi[q, i, c]: component c of basis function i at reference point q."""
        pts = np.atleast_2d(np.asarray(points, dtype=np.float64))
        out = np.zeros((pts.shape[0], self.ndofs, 2))
        for i, (px, py) in enumerate(self.basis):
            out[:, i, 0] = P.evaluate(px, pts)
            out[:, i, 1] = P.evaluate(py, pts)
        return out

    def tabulate_div(self, points):
        """div phi[q, i] on the reference cell."""
        pts = np.atleast_2d(np.asarray(points, dtype=np.float64))
        out = np.zeros((pts.shape[0], self.ndofs))
        for i, (px, py) in enumerate(self.basis):
            out[:, i] = P.evaluate(P.add(P.ddx(px), P.ddy(py)), pts)
        return out

    def divergence(self, i):
        px, py = self.basis[i]
        return P.add(P.ddx(px), P.ddy(py))

    def facet_interpolation_matrix(self, s_points, weights):
        """M[f, j, d, q] = n_f[d] s_q^j w_q  (reference :82-90, base/KernelData.cpp:191-268)."""
        s = np.asarray(s_points, dtype=np.float64)
        w = np.asarray(weights, dtype=np.float64)
        M = np.zeros((3, self.degree, 2, s.size))
        for f in range(3):
            for j in range(self.degree):
                for d in range(2):
                    M[f, j, d, :] = FACET_NORMALS[f][d] * s ** j * w
        return M


def facet_points(s_points):
    """Reference-cell coordinates of the parameter values s on the three facets: [3, nq, 2]."""
    s = np.asarray(s_points, dtype=np.float64)
    out = np.zeros((3, s.size, 2))
    for f in range(3):
        (x0, x1), (y0, y1) = FACET_PARAM[f]
        out[f, :, 0] = x0 + x1 * s
        out[f, :, 1] = y0 + y1 * s
    return out


def reversal_transformation(k):
    """T(i, line) = -(-1)^i C(line, i): DOFs of a facet seen with reversed parameter and
    opposite normal (se/KernelData.cpp:49-64)."""
    T = np.zeros((k, k))
    for line in range(k):
        val = 1
        for i in range(line + 1):
            T[i, line] = -val if i % 2 == 0 else val
            val = val * (line - i) // (i + 1)
    return T


def create_hierarchic_rt(degree: int, discontinuous: bool = True) -> HierarchicRT:
    """Signature-compatible stand-in (cell is always the triangle)."""
    return HierarchicRT(degree)
