"""Boundary conditions for flux equilibration - host-side mirror of
python/dolfinx_eqlb/eqlb/bcs.py (`fluxbc` :25-162, `boundarydata` :165-217) over the classes
`FluxBC` / `BoundaryData` of the compiled module dolfinx_eqlb_amd._cpp (same constructor arguments as
the reference's, python/dolfinx_eqlb/wrappers.cpp:144-256).

DOLFINx objects are replaced by flat arrays: `V` is the pair (mesh, degree_flux) of the flux space
or a `_cpp.FunctionSpace`, a boundary function is a numpy vector in the layout of that space
(discontinuous hierarchic RT_k for `custom_rt=True`, the conforming version of
dolfinx_eqlb_amd/eqlb/conforming.py otherwise) or a `_cpp.Function`.  The value of a `fluxbc` is 0 /
None, a callable (x, y) -> (w_x, w_y) whose normal component is the prescribed flux, or a callable
(x, y) -> g returning the normal flux itself (`scalar=True`).  The reference JIT-compiles a UFL
expression evaluated at the facet points (bcs.py:66-118) and hands its address to FluxBC; here the
same role is played by a C callback of the same signature (ufcx tabulate_tensor_float64) around the
Python callable.
"""

import ctypes
import typing

import numpy as np

from . import _adapter

_KERNEL = ctypes.CFUNCTYPE(None, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                           ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                           ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint8))


def _facet_points(s):
    """Reference points of the three facets for the parameters s (bcs.py:92-106):
    facet 0: (1 - s, s), facet 1: (0, s), facet 2: (s, 0); [3, nq, 2]."""
    z = np.zeros_like(s)
    return np.stack([np.stack([1 - s, s], 1), np.stack([z, s], 1), np.stack([s, z], 1)])


class fluxbc:
    """Essential boundary condition for one flux on a set of facets (bcs.py:25-162).

    value: 0 / None (homogeneous) or a callable, see the module docstring.  requires_projection:
    evaluate at the points of a Gauss rule of `quadrature_degree` (default 2 (k - 1), bcs.py:69-71)
    and project, instead of at the interpolation points of the flux element."""

    def __init__(self, value: typing.Any, facets, V=None,
                 requires_projection: typing.Optional[bool] = False,
                 quadrature_degree: typing.Optional[int] = None, scalar: bool = False):
        if value is None or (not callable(value) and value in (0, 0.0)):
            value = None
        if value is not None and not callable(value):
            raise NotImplementedError("flux BC values are 0 or a callable (x, y) -> (wx, wy)")
        self.value = value
        self.scalar = bool(scalar)
        self.facets = np.asarray(facets, dtype=np.int32)
        self.V = V
        self.requires_projection = bool(requires_projection)
        self._quadrature_degree = quadrature_degree
        self._keep = None  # the ctypes callback must outlive the FluxBC that holds its address

    @property
    def quadrature_degree(self):
        return 0 if self._quadrature_degree is None else int(self._quadrature_degree)

    def to_cpp(self, V):
        """The `_cpp.FluxBC` of this condition for the flux space V (a `_cpp.FunctionSpace`)."""
        c = _adapter.module()
        k = V.degree
        if self.requires_projection:
            qdeg = 2 * (k - 1) if self._quadrature_degree is None else int(self._quadrature_degree)
        else:
            qdeg = c.interpolation_quadrature_degree(k)
        s, _ = c.facet_quadrature(qdeg)
        nq = s.size
        if self.value is None:
            ptr = 0
        else:
            pts = _facet_points(s)  # [3, nq, 2]
            value, scalar = self.value, self.scalar

            def kernel(values, w, cst, coords, entity, perm):
                x = np.ctypeslib.as_array(coords, shape=(3, 3))[:, :2]
                J = np.stack([x[1] - x[0], x[2] - x[0]], axis=1)
                xq = x[0] + pts @ J.T  # [3, nq, 2]
                out = np.ctypeslib.as_array(values, shape=(3, nq))
                if scalar:
                    out[:] = np.asarray(value(xq[..., 0], xq[..., 1]), dtype=np.float64)
                    return
                wx, wy = value(xq[..., 0], xq[..., 1])
                # outward unit normals of the three facets: edge (a, b) of facet f, outward = away from
                # the opposite vertex
                for f, (a, b) in enumerate(((1, 2), (0, 2), (0, 1))):
                    t = x[b] - x[a]
                    n = np.array([t[1], -t[0]]) / np.hypot(*t)
                    if n @ (x[f] - x[a]) > 0:
                        n = -n
                    out[f] = wx[f] * n[0] + wy[f] * n[1]

            self._keep = _KERNEL(kernel)
            ptr = ctypes.cast(self._keep, ctypes.c_void_p).value
        facets = [int(f) for f in self.facets]
        if self.requires_projection:
            return c.FluxBC(V, facets, ptr, int(nq), int(qdeg), [], [], [])
        return c.FluxBC(V, facets, ptr, int(nq), [], [], [])


def boundarydata(flux_conditions: typing.List[typing.List[fluxbc]],
                 boundary_data: typing.List[typing.Any], V, custom_rt: bool,
                 dirichlet_facets: typing.List[np.ndarray], equilibrate_stress: bool):
    """The collected essential boundary conditions of a set of reconstructed fluxes
    (bcs.py:165-217) -> `_cpp.BoundaryData`.  `boundary_data[i]` (numpy vector or `_cpp.Function`)
    receives the global boundary DOFs of flux i in place, like the reference's boundary Functions
    (base/BoundaryData.cpp:423,609)."""
    c = _adapter.module()
    n_rhs = len(flux_conditions)
    if n_rhs != len(boundary_data) or n_rhs != len(dirichlet_facets):
        raise RuntimeError("Size of input data does not match!")  # bcs.py:193-194
    if not isinstance(V, c.FunctionSpace):
        mesh, degree_flux = V
        V = _adapter.flux_space(mesh, degree_flux, custom_rt)
    # (default) quadrature degree, bcs.py:196-203
    qdegree = 2 * (V.degree - 1)
    for bcs in flux_conditions:
        for bc in bcs:
            qdegree = max(qdegree, bc.quadrature_degree)
    # every projected condition has to use that one rule (the C++ side checks the number of points,
    # base/BoundaryData.cpp:437-445)
    cpp_bcs = [[bc.to_cpp(V) for bc in bcs] for bcs in flux_conditions]
    fns = [_adapter.function(V, b) for b in boundary_data]
    prime = [[int(f) for f in np.asarray(d).ravel()] for d in dirichlet_facets]
    bd = c.BoundaryData(cpp_bcs, fns, V, bool(custom_rt), int(qdegree), prime, bool(equilibrate_stress))
    bd._keep = (flux_conditions, fns)  # callbacks and Functions stay alive with the boundary data
    return bd
