"""Flux equilibration based on a semi-explicit strategy - host-side mirror of the reference's
`FluxEqlbSE` (python/dolfinx_eqlb/eqlb/FluxEqlbSE.py:26-198) on flat arrays.

Same constructor arguments, methods and error behaviour; DOLFINx Functions are replaced by
numpy arrays in the layouts of include/eqlb.h, wrapped (zero copy) into the Function stand-ins of
the compiled module `dolfinx_eqlb_amd._cpp`, whose functions carry the names and argument order of
the reference's `dolfinx_eqlb.cpp` (python/dolfinx_eqlb/wrappers.cpp:52-272).  All numerical work
happens in libeqlb_amd.so on the GPU.
"""

import typing

import numpy as np

from . import _adapter
from ..mesh import Mesh
from .bcs import boundarydata, fluxbc  # noqa: F401  (fluxbc is re-exported from here)


class FluxEqlbSE:
    """Equilibrate fluxes in a semi-explicit manner (steps 1 and 2 of the reference class)."""

    def __init__(self, degree_flux: int, msh: Mesh, list_rhs: typing.List[np.ndarray],
                 list_proj_flux: typing.List[np.ndarray],
                 equilibrate_stress: typing.Optional[bool] = False,
                 estimate_korn_constant: typing.Optional[bool] = False):
        self.degree_flux = degree_flux
        self.n_fluxes = len(list_rhs)
        self.equilibrate_stresses = bool(equilibrate_stress)
        self.estimate_korn_constant = bool(estimate_korn_constant)
        self.korn_constants = None
        if len(list_proj_flux) != self.n_fluxes:
            raise RuntimeError("Mismatching inputs!")  # FluxEqlbSE.py:74-75
        self.mesh = msh
        self.list_rhs = [np.ascontiguousarray(r, dtype=np.float64).ravel().copy() for r in list_rhs]
        self.list_proj_flux = [np.ascontiguousarray(g, dtype=np.float64).ravel().copy()
                               for g in list_proj_flux]
        nd = self.list_rhs[0].size // msh.ncells
        degree_dg = {1: 0, 3: 1, 6: 2, 10: 3}.get(nd)
        if degree_dg is None or nd * msh.ncells != self.list_rhs[0].size:
            raise RuntimeError("Equilibration: Input sizes does not match")
        if degree_dg > degree_flux - 1:  # se/reconstruction.hpp:363-373
            raise RuntimeError("Equilibration: Wrong polynomial degree of the projected RHS")
        if degree_dg < degree_flux - 1:
            # lower-degree data: embedded exactly into DG_{k-1}, the space the kernels work in
            from ..lsolver import embed_dg
            self.list_rhs = [embed_dg(r, msh.ncells, degree_dg, degree_flux - 1) for r in self.list_rhs]
            self.list_proj_flux = [embed_dg(g, msh.ncells, degree_dg, degree_flux - 1, bs=2)
                                   for g in self.list_proj_flux]
            degree_dg = degree_flux - 1
        self.degree_dg = degree_dg
        if equilibrate_stress:  # se/reconstruction.hpp:376-388
            if self.n_fluxes < 2:
                raise RuntimeError("Stress equilibration: Specify all rows of stress tensor")
            if degree_flux < 2:
                raise RuntimeError("Stress equilibration: RT_k with k>1 required!")
        # function spaces (FluxEqlbSE.py:94-105): discontinuous hierarchic RT_k, DG_{k-1} (x 2)
        self.V_flux = _adapter.flux_space(msh, degree_flux, True)
        self.V_flux_dg = _adapter.dg_space(msh, degree_dg, 2)
        self.V_rhs = _adapter.dg_space(msh, degree_dg, 1)
        ndofs = degree_flux * (degree_flux + 2)
        self.list_flux = np.zeros((self.n_fluxes, msh.ncells * ndofs))
        self._f_flux = [_adapter.function(self.V_flux, self.list_flux[i]) for i in range(self.n_fluxes)]
        self._f_proj = [_adapter.function(self.V_flux_dg, g) for g in self.list_proj_flux]
        self._f_rhs = [_adapter.function(self.V_rhs, r) for r in self.list_rhs]
        self._f_korn = None
        if estimate_korn_constant:
            self.korn_constants = np.zeros(msh.ncells)  # DG0 function of the reference
            self._f_korn = _adapter.function(_adapter.dg_space(msh, 0, 1), self.korn_constants)
        self.boundary_data = None

    def set_boundary_conditions(self, list_bfct_prime: typing.List[np.ndarray],
                                list_bcs_flux: typing.List[typing.List[fluxbc]]):
        """list_bfct_prime[i]: facets with essential BCs of the primal problem;
        list_bcs_flux[i]: flux BCs (FluxEqlbSE.py:118-147)."""
        if self.n_fluxes != len(list_bfct_prime) or self.n_fluxes != len(list_bcs_flux):
            raise RuntimeError("Mismatching inputs!")
        # boundary functions of the discontinuous hierarchic RT_k space (FluxEqlbSE.py:134-145)
        self.list_bfunctions = [np.zeros(self.list_flux.shape[1]) for _ in range(self.n_fluxes)]
        self.boundary_data = boundarydata(list_bcs_flux, self.list_bfunctions, self.V_flux, True,
                                          list_bfct_prime, self.equilibrate_stresses)
        self.facet_type = self.boundary_data.facet_type

    def equilibrate_fluxes(self):
        """Equilibrate the fluxes (accumulates into list_flux like the reference)."""
        if self.boundary_data is None:
            raise RuntimeError("Boundary conditions have not been set")
        c = _adapter.module()
        if self.estimate_korn_constant:
            # reconstruct_fluxes_semiexplt_with_kornconst + sqrt (FluxEqlbSE.py:152-166)
            c.reconstruct_fluxes_semiexplt_with_kornconst(self._f_flux, self._f_proj, self._f_rhs,
                                                          self.boundary_data, self.equilibrate_stresses,
                                                          self._f_korn)
            self.korn_constants[:] = np.sqrt(self.korn_constants)
        else:
            c.reconstruct_fluxes_semiexplt(self._f_flux, self._f_proj, self._f_rhs, self.boundary_data,
                                           self.equilibrate_stresses)

    def get_reconstructed_fluxes(self, subproblem: int):
        """(corrector in discontinuous hierarchic RT_k, projected flux in DG_{k-1}^2): the
        reconstructed flux is their sum (FluxEqlbSE.py:176-186)."""
        return self.list_flux[subproblem], self.list_proj_flux[subproblem]

    def get_korn_constants(self):
        if self.estimate_korn_constant:
            return self.korn_constants
        raise RuntimeError("Korn constants are not estimated!")
