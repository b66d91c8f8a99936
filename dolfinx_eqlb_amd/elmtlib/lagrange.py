"""Lagrange P_d on the reference triangle with Basix DOF numbering (equispaced).

Stand-in for `basix::element::create_lagrange` as used by the reference for the
projected flux / RHS (DG_{k-1}) and the hat function (P1)
(cpp/dolfinx_eqlb/se/reconstruction.hpp:108-117).  Node order: vertices, then
edges e0:(v1,v2), e1:(v0,v2), e2:(v0,v1) (interior edge points running from the
low to the high local vertex), then the cell interior.  DOLFINx' default
gll_warped variant coincides with equispaced for d <= 2 (i.e. RT_k, k <= 3);
d = 3, 4 are equispaced here (d = 4 only as the primal P_4 of the test solvers).
"""

from fractions import Fraction

import numpy as np

from . import polynomials as P

_VERT = ((0, 0), (1, 0), (0, 1))
_EDGES = ((1, 2), (0, 2), (0, 1))


def lagrange_nodes(d: int):
    """Nodes as exact Fractions, Basix order."""
    if d == 0:
        return [(Fraction(1, 3), Fraction(1, 3))]
    nodes = [tuple(map(Fraction, v)) for v in _VERT]
    for (a, b) in _EDGES:
        for i in range(1, d):
            t = Fraction(i, d)
            nodes.append((_VERT[a][0] + t * (_VERT[b][0] - _VERT[a][0]),
                          _VERT[a][1] + t * (_VERT[b][1] - _VERT[a][1])))
    if d > 4:
        raise NotImplementedError("Lagrange degree > 4 not needed (RT_k, k <= 4)")
    for j in range(1, d - 1):
        for i in range(1, d - j):
            nodes.append((Fraction(i, d), Fraction(j, d)))
    return nodes


def facet_closure_dofs(d: int):
    """entity_closure_dofs[1][f] of the continuous element (se/Patch.hpp:420,836-850)."""
    if d == 0:
        return [[0], [0], [0]]
    out = []
    for f, (a, b) in enumerate(_EDGES):
        out.append([a, b] + [3 + f * (d - 1) + i for i in range(d - 1)])
    return out


class Lagrange:
    def __init__(self, degree: int):
        self.degree = d = degree
        self.nodes = lagrange_nodes(d)
        self.ndofs = len(self.nodes)
        monos = [(a, deg - a) for deg in range(d + 1) for a in range(deg, -1, -1)]
        assert len(monos) == self.ndofs
        V = [[x ** a * y ** b for (a, b) in monos] for (x, y) in self.nodes]
        C = P.solve_exact(V, [[Fraction(int(i == j)) for j in range(self.ndofs)]
                              for i in range(self.ndofs)])
        self.basis = []
        for i in range(self.ndofs):
            p = {}
            for j, (a, b) in enumerate(monos):
                if C[j][i] != 0:
                    p = P.add(p, P.monomial(a, b, C[j][i]))
            self.basis.append(p)

    def tabulate(self, points, nderiv: int = 0):
        """[1 + 2*nderiv, nq, ndofs]: values, d/dX, d/dY (Basix layout)."""
        pts = np.atleast_2d(np.asarray(points, dtype=np.float64))
        out = np.zeros((1 + 2 * nderiv, pts.shape[0], self.ndofs))
        for i, p in enumerate(self.basis):
            out[0, :, i] = P.evaluate(p, pts)
            if nderiv:
                out[1, :, i] = P.evaluate(P.ddx(p), pts)
                out[2, :, i] = P.evaluate(P.ddy(p), pts)
        return out
