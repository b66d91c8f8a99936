"""Flux equilibration by patch-wise constrained minimisation (Ern & Vohralik) - host-side mirror
of the reference's `FluxEqlbEV` (python/dolfinx_eqlb/eqlb/FluxEqlbEV.py:20-188) on flat arrays.

Same constructor arguments, methods and error behaviour.  The UFL forms the reference builds
(FluxEqlbEV.py:113-134) are fixed, so the Form stand-ins of the compiled module
`dolfinx_eqlb_amd._cpp` only carry their data; the reconstructed flux lives in the conforming
hierarchic RT_k (dolfinx_eqlb_amd/eqlb/conforming.py) instead of the Basix RT_k space.
All numerical work happens in libeqlb_amd.so on the GPU.
"""

import typing

import numpy as np

from . import _adapter
from ..mesh import Mesh
from .bcs import boundarydata, fluxbc
from .conforming import conforming_dofmap


class FluxEqlbEV:
    """Equilibrate fluxes by a series of constrained minimisation problems."""

    def __init__(self, degree_flux: int, msh: Mesh, list_rhs: typing.List[np.ndarray],
                 list_proj_flux: typing.List[np.ndarray]):
        self.degree_flux = degree_flux
        self.n_fluxes = len(list_rhs)
        self.equilibrate_stresses = False  # FluxEqlbEV.py:43
        if len(list_proj_flux) != self.n_fluxes:
            raise RuntimeError("Missmatching inputs!")  # FluxEqlbEV.py:69-70
        self.mesh = msh
        self.list_rhs = [np.ascontiguousarray(r, dtype=np.float64).ravel().copy() for r in list_rhs]
        self.list_proj_flux = [np.ascontiguousarray(g, dtype=np.float64).ravel().copy()
                               for g in list_proj_flux]
        nd = degree_flux * (degree_flux + 1) // 2
        if any(r.size != nd * msh.ncells for r in self.list_rhs) or \
                any(g.size != 2 * nd * msh.ncells for g in self.list_proj_flux):
            raise RuntimeError("Equilibration: Input sizes does not match")
        c = _adapter.module()
        # V_flux: conforming RT_k (FluxEqlbEV.py:100)
        self.cell_dofs, self.ndofs = conforming_dofmap(msh, degree_flux)
        self.V_flux = _adapter.flux_space(msh, degree_flux, False)
        V_g = _adapter.dg_space(msh, degree_flux - 1, 2)
        V_f = _adapter.dg_space(msh, degree_flux - 1, 1)
        self.list_flux = np.zeros((self.n_fluxes, self.ndofs))
        self._f_flux = [_adapter.function(self.V_flux, self.list_flux[i]) for i in range(self.n_fluxes)]
        # the forms of FluxEqlbEV.py:113-134: a, l_pen carry no data, l_i depends on (G_i, f_i)
        self._a, self._l_pen = c.Form([]), c.Form([])
        self._l = [c.Form([_adapter.function(V_g, g), _adapter.function(V_f, r)])
                   for g, r in zip(self.list_proj_flux, self.list_rhs)]
        self.boundary_data = None

    def set_boundary_conditions(self, list_bfct_prime: typing.List[np.ndarray],
                                list_bcs_flux: typing.List[typing.List[fluxbc]]):
        """FluxEqlbEV.py:136-165."""
        if self.n_fluxes != len(list_bfct_prime) or self.n_fluxes != len(list_bcs_flux):
            raise RuntimeError("Mismatching inputs!")
        # boundary functions of the conforming flux space (FluxEqlbEV.py:153-165)
        self.list_bfunctions = [np.zeros(self.ndofs) for _ in range(self.n_fluxes)]
        self.boundary_data = boundarydata(list_bcs_flux, self.list_bfunctions, self.V_flux, False,
                                          list_bfct_prime, self.equilibrate_stresses)
        self.facet_type = self.boundary_data.facet_type

    def equilibrate_fluxes(self):
        """Equilibrate the fluxes (accumulates into list_flux, FluxEqlbEV.py:167-176)."""
        if self.boundary_data is None:
            raise RuntimeError("Boundary conditions have not been set")
        _adapter.module().reconstruct_fluxes_minimisation(self._a, self._l_pen, self._l, self._f_flux,
                                                          self.boundary_data)

    def get_reconstructed_fluxes(self, subproblem: int):
        """The reconstructed flux (conforming RT_k DOFs), FluxEqlbEV.py:178-188."""
        return self.list_flux[subproblem]
