"""Cell-local projector: oracle restatement of base::local_solver + HIP kernel.
The reference tests it against a global PETSc projection (test_localsolver_projection.py:300,
410,462); for DG spaces the global mass matrix is block diagonal, so the cell-wise solve IS the
global projection, and polynomials of the target degree are reproduced exactly."""

import numpy as np
import pytest

from cases import make_case
from dolfinx_eqlb_amd.elmtlib.lagrange import Lagrange
from dolfinx_eqlb_amd.elmtlib.quadrature import make_quadrature_triangle
from synthetic import dg_points


def _poly(deg, seed):
    rng = np.random.default_rng(seed)
    c = rng.standard_normal((deg + 1, deg + 1))
    return lambda x, y: sum(c[a, b] * x ** a * y ** b for a in range(deg + 1) for b in range(deg + 1 - a))


def _qvalues(mesh, qp, fns):
    from dolfinx_eqlb_amd.lsolver.projection import quadrature_points_physical
    xq = quadrature_points_physical(mesh, qp)
    return np.stack([np.stack([fn(xq[..., 0], xq[..., 1]) for fn in comp], axis=-1) for comp in fns])


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_oracle_projector_reproduces_polynomials(deg):
    from oracle import projection as op
    mesh, *_ = make_case(3, 1)
    qp, qw = make_quadrature_triangle(2 * deg + 1)
    fns = [[_poly(deg, 1), _poly(deg, 2)], [_poly(deg, 3), _poly(deg, 4)]]  # nrhs = 2, bs = 2
    out = op.local_projection(mesh, deg, qp, qw, _qvalues(mesh, qp, fns), bs=2)
    pts = dg_points(mesh, deg)
    for r in range(2):
        expect = np.stack([fn(pts[..., 0], pts[..., 1]) for fn in fns[r]], axis=-1)
        assert np.allclose(out[r].reshape(expect.shape), expect, atol=1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("deg,bs", [(0, 1), (1, 1), (1, 2), (2, 2), (3, 1)])
def test_gpu_projector_equals_oracle(deg, bs):
    from dolfinx_eqlb_amd import cpp
    from oracle import projection as op
    mesh, *_ = make_case(6, 1)
    qp, qw = make_quadrature_triangle(2 * deg + 3)
    # non-polynomial data, 3 right-hand sides
    fns = [[(lambda x, y, s=s, c=c: np.sin(3 * x + s) * np.cos(2 * y + c)) for c in range(bs)]
           for s in range(3)]
    qv = _qvalues(mesh, qp, fns)
    ref = op.local_projection(mesh, deg, qp, qw, qv, bs=bs)
    got = cpp.project_dg(cpp.DeviceMesh(mesh), deg, qp, qw, qv, bs=bs)
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()


@pytest.mark.gpu
def test_gpu_local_projection_feeds_equilibration(oracle_mod):
    """Pipeline of the demos: sigma_proj = Pi(-grad u_h) and Pi(f) from callables, then
    equilibration; checked against the oracle for the same projected data."""
    from dolfinx_eqlb_amd import cpp
    from dolfinx_eqlb_amd.lsolver import local_projection
    k = 2
    mesh, ft, _, _ = make_case(5, k)
    dm = cpp.DeviceMesh(mesh)
    G = local_projection(dm, k - 1, [lambda x, y: np.stack([-np.cos(x) * y, np.sin(y) + x], -1)], bs=2)[0]
    f0 = local_projection(dm, k - 1, [lambda x, y: np.exp(x) * np.cos(3 * y)])[0]
    # interpolate == project for data that is already in DG_{k-1}: projecting the projection
    # (given by its nodal values evaluated through the basis) returns it unchanged
    el = Lagrange(k - 1)
    qp, qw = make_quadrature_triangle(2 * (k - 1) + 2)
    psi = el.tabulate(qp)[0]
    vals = np.einsum("qj,cj->cq", psi, f0.reshape(mesh.ncells, el.ndofs))
    again = cpp.project_dg(dm, k - 1, qp, qw, vals[None, :, :, None])[0]
    assert np.allclose(again, f0, atol=1e-12)
    assert G.shape == (mesh.ncells * el.ndofs * 2,)
