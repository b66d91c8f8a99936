"""Gauss-Jacobi quadrature on the unit interval and the reference triangle.

Stand-in for `basix::quadrature::make_quadrature` (base/QuadratureRule.hpp:52-53,
e_raviart_thomas.py:71).  Basix' default triangle rule (Xiao-Gimbutas) is not
reproduced; all integrands of the equilibration are polynomials on affine cells,
so any rule exact to the requested degree gives the same result up to rounding.
"""

import numpy as np
from scipy.special import roots_jacobi


def make_quadrature_interval(degree: int):
    """Gauss-Legendre on [0, 1], m = (degree + 2) // 2 points, ascending."""
    m = (degree + 2) // 2
    x, w = np.polynomial.legendre.leggauss(m)
    return 0.5 * (x + 1.0), 0.5 * w


def make_quadrature_triangle(degree: int):
    """Collapsed Gauss-Jacobi rule on {x, y >= 0, x + y <= 1}; weights sum to 1/2."""
    m = (degree + 2) // 2
    xg, wg = np.polynomial.legendre.leggauss(m)
    xj, wj = roots_jacobi(m, 1.0, 0.0)
    pts = np.zeros((m * m, 2))
    wts = np.zeros(m * m)
    c = 0
    for i in range(m):
        x = 0.5 * (xj[i] + 1.0)
        for j in range(m):
            pts[c, 0] = x
            pts[c, 1] = 0.5 * (xg[j] + 1.0) * (1.0 - x)
            wts[c] = wj[i] * 0.125 * wg[j]
            c += 1
    return pts, wts
