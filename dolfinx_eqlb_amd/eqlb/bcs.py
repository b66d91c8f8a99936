"""Boundary conditions for flux equilibration - host-side mirror of
python/dolfinx_eqlb/eqlb/bcs.py (`fluxbc` :25-162, `boundarydata` :165-217) and of what
base::BoundaryData collects (cpp/dolfinx_eqlb/base/BoundaryData.cpp:279-633): the per-RHS facet
types (0 internal, 1 essential BC of the primal problem, 2 flux BC) and the GLOBAL boundary DOFs
of the flux.  The per-patch values hat_a * g are formed on the device.

DOLFINx objects are replaced by flat arrays: `V` is the pair (mesh, degree_flux) of the flux
space, a boundary function is a numpy vector in the layout of that space (discontinuous
hierarchic RT_k for `custom_rt=True`, the conforming version of dolfinx_eqlb_amd/eqlb/conforming.py
otherwise), and the value of a `fluxbc` is 0 / None or a callable (x, y) -> (w_x, w_y) whose
normal component is the prescribed flux (the UFL/JIT evaluation of the reference, bcs.py:66-118,
stays with DOLFINx).
"""

import typing

import numpy as np


class fluxbc:
    """Essential boundary condition for one flux on a set of facets (bcs.py:25-162)."""

    def __init__(self, value: typing.Any, facets, V=None,
                 requires_projection: typing.Optional[bool] = False,
                 quadrature_degree: typing.Optional[int] = None):
        if value in (0, 0.0):
            value = None
        if value is not None and not callable(value):
            raise NotImplementedError("flux BC values are 0 or a callable (x, y) -> (wx, wy)")
        self.value = value
        self.facets = np.asarray(facets, dtype=np.int32)
        self.requires_projection = bool(requires_projection)
        self.quadrature_degree = 0 if quadrature_degree is None else int(quadrature_degree)


class BoundaryData:
    """facet_type [nrhs, nfacets] int8, boundary_values [nrhs, ndofs] float64 or None."""

    def __init__(self, facet_type, boundary_values, custom_rt, equilibrate_stress):
        self.facet_type = facet_type
        self.boundary_values = boundary_values
        self.custom_rt = bool(custom_rt)
        self.equilibrate_stress = bool(equilibrate_stress)


def boundarydata(flux_conditions: typing.List[typing.List[fluxbc]],
                 boundary_data: typing.List[np.ndarray], V, custom_rt: bool,
                 dirichlet_facets: typing.List[np.ndarray], equilibrate_stress: bool) -> BoundaryData:
    """The collected essential boundary conditions of a set of reconstructed fluxes
    (bcs.py:165-217).  `boundary_data[i]` receives the global boundary DOFs of flux i in place,
    like the reference's boundary Functions (base/BoundaryData.cpp:423,609)."""
    from ..synthetic import boundary_dofs_from_field
    from .conforming import broken_to_conforming
    n_rhs = len(flux_conditions)
    if n_rhs != len(boundary_data) or n_rhs != len(dirichlet_facets):
        raise RuntimeError("Size of input data does not match!")  # bcs.py:193-194
    mesh, degree_flux = V
    ft = np.zeros((n_rhs, mesh.nfacets), dtype=np.int8)
    inhomogeneous = False
    for i in range(n_rhs):
        ft[i, np.asarray(dirichlet_facets[i], dtype=np.int64)] = 1
        for bc in flux_conditions[i]:
            ft[i, bc.facets] = 2
        for bc in flux_conditions[i]:
            if bc.value is None:
                continue
            inhomogeneous = True
            row = np.zeros(mesh.nfacets, dtype=np.int8)
            row[bc.facets] = 2
            vals = boundary_dofs_from_field(mesh, degree_flux, row, bc.value)
            if not custom_rt:
                vals = broken_to_conforming(mesh, degree_flux, vals)
            boundary_data[i] += vals
    bv = np.stack([np.asarray(b, dtype=np.float64) for b in boundary_data]) if inhomogeneous else None
    return BoundaryData(ft, bv, custom_rt, equilibrate_stress)
