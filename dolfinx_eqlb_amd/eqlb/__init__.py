"""Host-side equilibrator front end of libeqlb_amd.so.

Public names follow the reference package (python/dolfinx_eqlb/eqlb) so that user scripts keep
their imports: the two equilibrator classes and the two boundary-condition helpers.  Everything
here works on flat numpy arrays (mesh container of dolfinx_eqlb_amd.mesh); the numerical work is
done by the HIP library behind dolfinx_eqlb_amd.cpp.
"""

from . import bcs as _bcs
from . import FluxEqlbEV as _ev
from . import FluxEqlbSE as _se

FluxEqlbSE = _se.FluxEqlbSE
FluxEqlbEV = _ev.FluxEqlbEV
fluxbc = _bcs.fluxbc
boundarydata = _bcs.boundarydata

__all__ = ("FluxEqlbSE", "FluxEqlbEV", "fluxbc", "boundarydata")
