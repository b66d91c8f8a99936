"""Cell-local solvers: mirror of python/dolfinx_eqlb/lsolver (projection.py, lsolver.py)."""

from .projection import embed_dg, local_projection

__all__ = ["local_projection", "embed_dg"]
