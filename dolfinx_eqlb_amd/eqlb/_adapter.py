"""Glue between the flat-array front end and the pybind11 module `dolfinx_eqlb_amd._cpp` (the
stand-in of the reference's `dolfinx_eqlb.cpp`): device meshes, function spaces and Functions over
numpy arrays.  Where DOLFINx exists the same calls are made with the arrays of the DOLFINx objects
(INTEGRATION.md)."""

import numpy as np


def module():
    """The compiled module; raises if it has not been built (no fallback)."""
    try:
        # _cpp is linked against libeqlb_amd.so, which pulls in the system HIP runtime: map torch's bundled
        # runtime first (as cpp.lib() does), or a later `import torch` brings a second runtime that sees no device
        from .. import cpp
        cpp._bind_torch_hip_runtime()
        from .. import _cpp
    except ImportError as e:  # pragma: no cover - build problem
        raise RuntimeError(
            "dolfinx_eqlb_amd._cpp is not built: python -c 'import __graft_entry__ as g; g.build()'") from e
    return _cpp


def cpp_mesh(mesh):
    """_cpp.Mesh of a flat mesh container (created once per mesh object and kept on it)."""
    cached = getattr(mesh, "_cpp_mesh", None)
    if cached is not None:
        return cached
    m = module().Mesh(np.ascontiguousarray(mesh.x, dtype=np.float64), mesh.cell_nodes, mesh.cell_facets,
                      mesh.facet_nodes, mesh.facet_cells_offsets, mesh.facet_cells, mesh.node_cells_offsets,
                      mesh.node_cells, mesh.node_facets_offsets, mesh.node_facets, mesh.facet_perm)
    try:
        mesh._cpp_mesh = m
    except AttributeError:
        pass
    return m


def flux_space(mesh, degree_flux, custom_rt=True):
    """Discontinuous hierarchic RT_k (semi-explicit equilibrator) or its conforming version (EV)."""
    return module().FunctionSpace(cpp_mesh(mesh), "RT", degree_flux, 1, bool(custom_rt))


def dg_space(mesh, degree, bs=1):
    return module().FunctionSpace(cpp_mesh(mesh), "DG", degree, bs, True)


def function(V, array=None):
    """Function over `array` (zero copy: float64, C-contiguous, writeable) or a new zero vector."""
    c = module()
    if array is None:
        return c.Function(V)
    if isinstance(array, c.Function):
        return array
    a = np.asarray(array)
    if a.dtype != np.float64 or not a.flags.c_contiguous or not a.flags.writeable:
        raise RuntimeError("Function: a writeable C-contiguous float64 array is required")
    return c.Function(V, a.reshape(-1))
