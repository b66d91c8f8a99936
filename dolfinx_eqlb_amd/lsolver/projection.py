"""Cell-local L2 projection into DG spaces on the GPU - mirror of
python/dolfinx_eqlb/lsolver/projection.py:17-77 (`local_projection`) on flat arrays.

The reference takes UFL expressions and JIT-compiles the load kernels; here the data are Python
callables f(x, y) -> array [..., bs] (or [...] for bs = 1) evaluated at the physical quadrature
points, or precomputed point values.
"""

import typing

import numpy as np

from ..elmtlib.quadrature import make_quadrature_triangle
from ..eqlb.check_eqlb_conditions import cell_geometry


def quadrature_points_physical(mesh, qpoints):
    """x_c(X_q) for all cells: [ncells, nq, 2]."""
    J, _, _ = cell_geometry(mesh)
    x0 = mesh.x[mesh.cell_nodes[:, 0], :2]
    return x0[:, None, :] + np.einsum("cij,qj->cqi", J, qpoints)


def local_projection(dmesh, degree: int, data: typing.List[typing.Any], bs: int = 1,
                     quadrature_degree: typing.Optional[int] = None,
                     solver: str = "cholesky") -> typing.List[np.ndarray]:
    """Project every entry of `data` into DG_degree (block size bs); returns the DOF arrays
    [ncells*nd*bs] (cell-major, x[bs*dof+cb]).  data[i]: callable(x, y) or array [ncells, nq, bs].
    `dmesh`: flat mesh container (or a `cpp.DeviceMesh` of one).  The solve runs through
    `local_solver_<solver>` of the compiled module (names of python/dolfinx_eqlb/wrappers.cpp:52-80):
    a = (u, v) on DG_degree, l_i = (f_i, v) given by the point values of f_i."""
    from ..eqlb import _adapter
    c = _adapter.module()
    mesh = getattr(dmesh, "mesh", dmesh)
    qdeg = 2 * degree + 2 if quadrature_degree is None else quadrature_degree
    qp, qw = make_quadrature_triangle(qdeg)
    xq = None
    V = _adapter.dg_space(mesh, degree, bs)
    sols, forms = [], []
    for d in data:
        if callable(d):
            if xq is None:
                xq = quadrature_points_physical(mesh, qp)
            v = np.asarray(d(xq[..., 0], xq[..., 1]), dtype=np.float64)
        else:
            v = np.asarray(d, dtype=np.float64)
        if v.size != mesh.ncells * qw.size * bs:
            raise RuntimeError("Local solver: Input sizes does not match")
        forms.append(c.Form.from_point_values(qp, qw, np.ascontiguousarray(v.reshape(mesh.ncells, qw.size, bs))))
        sols.append(c.Function(V))
    fn = {"cholesky": c.local_solver_cholesky, "lu": c.local_solver_lu, "cg": c.local_solver_cg}[solver]
    fn(sols, c.Form([]), forms)
    return [s_.array for s_ in sols]


def embed_dg(values, ncells: int, degree_from: int, degree_to: int, bs: int = 1):
    """Exact embedding DG_{degree_from} -> DG_{degree_to} (degree_to >= degree_from) of nodal
    values [ncells*nd_from*bs]: the reference accepts projected data of any degree <= k-1
    (se/reconstruction.hpp:363-373); the device kernels take DG_{k-1}, which contains them."""
    import numpy as np

    from ..elmtlib.lagrange import Lagrange
    if degree_to < degree_from:
        raise RuntimeError("Equilibration: Wrong polynomial degree of the projected RHS")
    lo, hi = Lagrange(degree_from), Lagrange(degree_to)
    nodes = np.array([[float(a), float(b)] for a, b in hi.nodes])
    E = lo.tabulate(nodes)[0]  # [nd_to, nd_from]
    v = np.asarray(values, dtype=np.float64).reshape(ncells, lo.ndofs, bs)
    return np.ascontiguousarray(np.einsum("ij,cjb->cib", E, v).reshape(-1))
