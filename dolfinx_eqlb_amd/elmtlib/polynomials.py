"""Exact bivariate polynomial arithmetic on the reference triangle.

Small helper used to build element tables without Basix (not installed here):
polynomials are dicts {(a, b): Fraction} meaning sum c_ab x^a y^b.  Everything
the element construction needs (products, derivatives, restriction to an edge,
exact integrals over the reference triangle / unit interval) is rational, so
tables are generated without any rounding and converted to float once.
"""

from fractions import Fraction
from math import factorial

import numpy as np

Poly = dict  # {(a, b): Fraction}


def poly(terms) -> Poly:
    out = {}
    for (a, b), c in terms.items():
        c = Fraction(c)
        if c != 0:
            out[(a, b)] = out.get((a, b), Fraction(0)) + c
    return {k: v for k, v in out.items() if v != 0}


def monomial(a: int, b: int, c=1) -> Poly:
    return {(a, b): Fraction(c)}


def add(p: Poly, q: Poly, sq=1) -> Poly:
    out = dict(p)
    for k, v in q.items():
        out[k] = out.get(k, Fraction(0)) + sq * v
    return {k: v for k, v in out.items() if v != 0}


def scale(p: Poly, s) -> Poly:
    s = Fraction(s)
    return {k: v * s for k, v in p.items()} if s != 0 else {}


def mul(p: Poly, q: Poly) -> Poly:
    out = {}
    for (a, b), c in p.items():
        for (d, e), f in q.items():
            k = (a + d, b + e)
            out[k] = out.get(k, Fraction(0)) + c * f
    return {k: v for k, v in out.items() if v != 0}


def ddx(p: Poly) -> Poly:
    return {(a - 1, b): c * a for (a, b), c in p.items() if a > 0}


def ddy(p: Poly) -> Poly:
    return {(a, b - 1): c * b for (a, b), c in p.items() if b > 0}


def integrate_triangle(p: Poly) -> Fraction:
    """int over {x,y>=0, x+y<=1} of p  (int x^a y^b = a! b! / (a+b+2)!)."""
    s = Fraction(0)
    for (a, b), c in p.items():
        s += c * Fraction(factorial(a) * factorial(b), factorial(a + b + 2))
    return s


def _poly1d_mul(p, q):
    out = [Fraction(0)] * (len(p) + len(q) - 1)
    for i, a in enumerate(p):
        for j, b in enumerate(q):
            out[i + j] += a * b
    return out


def _poly1d_pow(p, n):
    out = [Fraction(1)]
    for _ in range(n):
        out = _poly1d_mul(out, p)
    return out


def restrict_to_line(p: Poly, x_of_s, y_of_s):
    """Substitute x = x0 + x1 s, y = y0 + y1 s; returns 1-D coefficient list in s."""
    out = [Fraction(0)]
    for (a, b), c in p.items():
        t = _poly1d_mul(_poly1d_pow(list(map(Fraction, x_of_s)), a),
                        _poly1d_pow(list(map(Fraction, y_of_s)), b))
        if len(t) > len(out):
            out = out + [Fraction(0)] * (len(t) - len(out))
        for i, v in enumerate(t):
            out[i] += c * v
    return out


def integrate_unit_interval(p1d, power: int = 0) -> Fraction:
    """int_0^1 p(s) s^power ds."""
    return sum((c * Fraction(1, i + power + 1) for i, c in enumerate(p1d)), Fraction(0))


def evaluate(p: Poly, pts: np.ndarray) -> np.ndarray:
    """Evaluate at float points [n, 2]."""
    pts = np.asarray(pts, dtype=np.float64)
    out = np.zeros(pts.shape[0])
    for (a, b), c in p.items():
        out += float(c) * pts[:, 0] ** a * pts[:, 1] ** b
    return out


def solve_exact(A, B):
    """Solve A X = B exactly (Fractions), A [n][n], B [n][m] -> X [n][m]."""
    n = len(A)
    m = len(B[0])
    M = [list(map(Fraction, A[i])) + list(map(Fraction, B[i])) for i in range(n)]
    for col in range(n):
        piv = next((r for r in range(col, n) if M[r][col] != 0), None)
        if piv is None:
            raise ValueError("singular matrix")
        M[col], M[piv] = M[piv], M[col]
        inv = 1 / M[col][col]
        M[col] = [v * inv for v in M[col]]
        for r in range(n):
            if r != col and M[r][col] != 0:
                f = M[r][col]
                M[r] = [vr - f * vc for vr, vc in zip(M[r], M[col])]
    return [row[n:n + m] for row in M]
