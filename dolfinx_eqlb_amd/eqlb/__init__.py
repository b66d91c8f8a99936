"""Flux equilibrators: mirror of python/dolfinx_eqlb/eqlb/__init__.py (same public names)."""

from .bcs import boundarydata, fluxbc
from .FluxEqlbEV import FluxEqlbEV
from .FluxEqlbSE import FluxEqlbSE

__all__ = ["FluxEqlbEV", "FluxEqlbSE", "fluxbc", "boundarydata"]
