"""TEST INFRASTRUCTURE - ctypes binding of the CPU oracle (oracle/eqlb_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""

import ctypes as C
import os
import subprocess

import numpy as np

from .tables import Tables, make_tables

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force: bool = False):
    """Compile the C restatement (gcc) into oracle/_build/."""
    src = os.path.join(_HERE, "eqlb_oracle.c")
    deps = [src, src[:-2] + ".h", src[:-2] + "_ev.c"]
    # the library is compiled with -march=native and travels with the repository snapshot: rebuild where the
    # host CPU is another one than the one it was built on
    stamp, host = os.path.join(os.path.dirname(_LIB_PATH), "host.txt"), _host_cpu()
    try:
        built_on = open(stamp).read()
    except OSError:
        built_on = None
    if force or not os.path.exists(_LIB_PATH) or built_on != host or \
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in deps):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
        with open(stamp, "w") as fh:
            fh.write(host)
    return _LIB_PATH


def _host_cpu():
    try:
        with open("/proc/cpuinfo") as fh:
            lines = [ln for ln in fh if ln.startswith(("model name", "flags"))]
        return "".join(lines[:2])
    except OSError:
        return "unknown"


class _Mesh(C.Structure):
    _fields_ = [("nnodes", C.c_int32), ("ncells", C.c_int32), ("nfacets", C.c_int32),
                ("x", C.c_void_p), ("cell_nodes", C.c_void_p), ("cell_facets", C.c_void_p),
                ("facet_nodes", C.c_void_p), ("facet_cells_off", C.c_void_p),
                ("facet_cells", C.c_void_p), ("node_cells_off", C.c_void_p),
                ("node_cells", C.c_void_p), ("node_facets_off", C.c_void_p),
                ("node_facets", C.c_void_p), ("facet_perm", C.c_void_p)]


class _Tables(C.Structure):
    _fields_ = [("k", C.c_int32), ("ndofs", C.c_int32), ("nd", C.c_int32), ("ndf", C.c_int32),
                ("nq", C.c_int32), ("nqf", C.c_int32),
                ("qpoints", C.c_void_p), ("qweights", C.c_void_p), ("flux_basis", C.c_void_p),
                ("rhs_cell", C.c_void_p), ("rhs_fct", C.c_void_p), ("hat_cell", C.c_void_p),
                ("hat_fct", C.c_void_p), ("M", C.c_void_p), ("doftrafo", C.c_void_p),
                ("fct_normal_out", C.c_void_p), ("fct_dofs", C.c_void_p),
                ("flux_basis_fct", C.c_void_p)]


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_se_reconstruct.restype = C.c_int
        _lib.oracle_se_patch.restype = C.c_int
        _lib.oracle_build_patches.restype = C.c_int
        _lib.oracle_se_korn.restype = C.c_int
        _lib.oracle_se_reconstruct_stress.restype = C.c_int
        _lib.oracle_ev_reconstruct.restype = C.c_int
        _lib.oracle_ev_patch.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


_MESH_FIELDS = (("x", np.float64), ("cell_nodes", np.int32), ("cell_facets", np.int32),
                ("facet_nodes", np.int32), ("facet_cells_offsets", np.int32), ("facet_cells", np.int32),
                ("node_cells_offsets", np.int32), ("node_cells", np.int32),
                ("node_facets_offsets", np.int32), ("node_facets", np.int32), ("facet_perm", np.uint8))


def _mesh_struct(mesh):
    """C view of the mesh.  The converted arrays are kept on the mesh object (sampled single-patch
    calls on a 1M-cell mesh would otherwise convert the index arrays on every call); the cache is
    dropped when one of the mesh's arrays has been replaced."""
    key = tuple(id(getattr(mesh, name)) for name, _ in _MESH_FIELDS)
    cached = getattr(mesh, "_oracle_struct", None)
    if cached is not None and cached[0] == key:
        return cached[1], cached[2]
    keep = [np.ascontiguousarray(getattr(mesh, name), dtype=dt) for name, dt in _MESH_FIELDS]
    s = _Mesh(mesh.nnodes, mesh.ncells, mesh.nfacets, *[_p(a) for a in keep])
    try:
        mesh._oracle_struct = (key, s, keep)
    except AttributeError:
        pass
    return s, keep


def _tables_struct(t: Tables):
    keep = [t.qpoints, t.qweights, t.flux_basis, t.rhs_cell, t.rhs_fct, t.hat_cell, t.hat_fct,
            t.M, t.doftrafo, t.fct_normal_out, t.fct_dofs, t.flux_basis_fct]
    s = _Tables(t.k, t.ndofs, t.nd, t.ndf, t.nq, t.nqf, *[_p(a) for a in keep])
    return s, keep


def _prep(mesh, k, facet_type, flux_dg, rhs_dg, degree_dg):
    flux_dg = np.ascontiguousarray(flux_dg, dtype=np.float64)
    rhs_dg = np.ascontiguousarray(rhs_dg, dtype=np.float64)
    nrhs = rhs_dg.shape[0]
    t = make_tables(k, degree_dg)
    assert flux_dg.shape == (nrhs, mesh.ncells * t.nd * 2), flux_dg.shape
    assert rhs_dg.shape == (nrhs, mesh.ncells * t.nd)
    facet_type = np.ascontiguousarray(facet_type, dtype=np.int8).reshape(nrhs, mesh.nfacets)
    return t, nrhs, facet_type, flux_dg, rhs_dg


def se_reconstruct(mesh, k, facet_type, flux_dg, rhs_dg, boundary_values=None, degree_dg=None,
                   flux_hdiv=None, node_range=None, stress=False):
    """Run the reference algorithm over all (or a range of) patches; returns flux_hdiv
    [nrhs, ncells*k(k+2)] (accumulated into `flux_hdiv` if given, like the reference)."""
    lib = _load()
    t, nrhs, facet_type, flux_dg, rhs_dg = _prep(mesh, k, facet_type, flux_dg, rhs_dg, degree_dg)
    ms, keep_m = _mesh_struct(mesh)
    ts, keep_t = _tables_struct(t)
    if flux_hdiv is None:
        flux_hdiv = np.zeros((nrhs, mesh.ncells * t.ndofs))
    assert flux_hdiv.flags.c_contiguous and flux_hdiv.dtype == np.float64
    if boundary_values is not None:
        boundary_values = np.ascontiguousarray(boundary_values, dtype=np.float64)
    nb, ne = node_range if node_range is not None else (0, mesh.nnodes)
    fn = lib.oracle_se_reconstruct_stress if stress else lib.oracle_se_reconstruct
    st = fn(C.byref(ms), C.byref(ts), C.c_int(nrhs), _p(facet_type),
            _p(boundary_values), _p(flux_dg), _p(rhs_dg), _p(flux_hdiv),
            C.c_int32(nb), C.c_int32(ne))
    if st == -4:
        raise RuntimeError("Stress equilibration: Specify all rows of stress tensor / RT_k with k>1")
    if st == -1:
        raise RuntimeError("Patch with only one cell")  # se/Patch.cpp:353-359
    if st == -5:  # se/reconstruction.hpp:195-197
        raise RuntimeError("Incompatible mesh! To many patches with 2 cells on neumann boundary.")
    if st != 0:
        raise RuntimeError(f"oracle failed with status {st}")
    return flux_hdiv


def se_patch(mesh, k, facet_type, flux_dg, rhs_dg, node, boundary_values=None, degree_dg=None):
    """Single patch: returns (cells, sigma_tilde[nrhs,n,ndofs], patch_solution[nrhs,n,ndofs], u)."""
    lib = _load()
    t, nrhs, facet_type, flux_dg, rhs_dg = _prep(mesh, k, facet_type, flux_dg, rhs_dg, degree_dg)
    ms, keep_m = _mesh_struct(mesh)
    ts, keep_t = _tables_struct(t)
    n = int(mesh.node_cells_offsets[node + 1] - mesh.node_cells_offsets[node])
    nmax = int(np.diff(mesh.node_cells_offsets).max())
    dim_max = 1 + (k - 1) * (nmax + 1) + (k - 1) * (k - 2) // 2 * nmax
    st_ = np.zeros((nrhs, n, t.ndofs))
    sol = np.zeros((nrhs, n, t.ndofs))
    cells = np.zeros(n, dtype=np.int32)
    u = np.zeros((nrhs, dim_max))
    if boundary_values is not None:
        boundary_values = np.ascontiguousarray(boundary_values, dtype=np.float64)
    st = lib.oracle_se_patch(C.byref(ms), C.byref(ts), C.c_int(nrhs), _p(facet_type),
                             _p(boundary_values), _p(flux_dg), _p(rhs_dg), C.c_int32(node),
                             _p(st_), _p(sol), _p(cells), _p(u))
    if st < 0:
        raise RuntimeError(f"oracle failed with status {st}")
    return cells, st_, sol, u


def se_korn(mesh, facet_type, node_range=None):
    """Accumulated squared Korn constants per cell (before the square root of FluxEqlbSE.py:165)."""
    lib = _load()
    facet_type = np.ascontiguousarray(facet_type, dtype=np.int8).reshape(-1, mesh.nfacets)
    ms, keep_m = _mesh_struct(mesh)
    nb, ne = node_range if node_range is not None else (0, mesh.nnodes)
    korn = np.zeros(mesh.ncells)
    st = lib.oracle_se_korn(C.byref(ms), C.c_int(facet_type.shape[0]), _p(facet_type), _p(korn),
                            C.c_int32(nb), C.c_int32(ne))
    if st != 0:
        raise RuntimeError(f"oracle failed with status {st}")
    return korn


def build_patches(mesh, facet_type, node_range=None):
    """Patch fans of all nodes (see eqlb_oracle.h: oracle_build_patches)."""
    lib = _load()
    facet_type = np.ascontiguousarray(facet_type, dtype=np.int8).reshape(-1, mesh.nfacets)
    nrhs = facet_type.shape[0]
    ms, keep_m = _mesh_struct(mesh)
    nb, ne = node_range if node_range is not None else (0, mesh.nnodes)
    nn = ne - nb
    stride = int(np.diff(mesh.node_cells_offsets).max()) + 2
    ncells = np.zeros(nn, dtype=np.int32)
    cells = np.zeros((nn, stride), dtype=np.int32)
    fcts = np.zeros((nn, stride), dtype=np.int32)
    fl = np.zeros((nn, 2 * stride), dtype=np.int8)
    il = np.zeros((nn, stride), dtype=np.int8)
    types = np.zeros((nn, nrhs), dtype=np.int8)
    st = lib.oracle_build_patches(C.byref(ms), C.c_int(nrhs), _p(facet_type), C.c_int32(nb),
                                  C.c_int32(ne), C.c_int32(stride), _p(ncells), _p(cells),
                                  _p(fcts), _p(fl), _p(il), _p(types))
    if st != 0:
        raise RuntimeError(f"oracle failed with status {st}")
    return dict(ncells=ncells, cells=cells, fcts=fcts, fcts_local=fl, inodes_local=il,
                types=types, stride=stride)


def ev_reconstruct(mesh, k, facet_type, flux_dg, rhs_dg, cell_dofs, ndofs_glob,
                   boundary_values=None, flux_hdiv=None, node_range=None):
    """Constrained-minimisation (EV) equilibration over all (or a range of) patches; returns
    flux_hdiv [nrhs, ndofs_glob] in the conforming hierarchic RT_k numbering `cell_dofs`."""
    lib = _load()
    t, nrhs, facet_type, flux_dg, rhs_dg = _prep(mesh, k, facet_type, flux_dg, rhs_dg, None)
    ms, keep_m = _mesh_struct(mesh)
    ts, keep_t = _tables_struct(t)
    cell_dofs = np.ascontiguousarray(cell_dofs, dtype=np.int32)
    assert cell_dofs.shape == (mesh.ncells, t.ndofs)
    if flux_hdiv is None:
        flux_hdiv = np.zeros((nrhs, ndofs_glob))
    assert flux_hdiv.flags.c_contiguous and flux_hdiv.shape == (nrhs, ndofs_glob)
    if boundary_values is not None:
        boundary_values = np.ascontiguousarray(boundary_values, dtype=np.float64)
        assert boundary_values.shape == (nrhs, ndofs_glob)
    nb, ne = node_range if node_range is not None else (0, mesh.nnodes)
    st = lib.oracle_ev_reconstruct(C.byref(ms), C.byref(ts), _p(t.flux_div), C.c_int(nrhs),
                                   _p(facet_type), _p(cell_dofs), C.c_int64(ndofs_glob),
                                   _p(boundary_values), _p(flux_dg), _p(rhs_dg), _p(flux_hdiv),
                                   C.c_int32(nb), C.c_int32(ne))
    if st == -1:
        raise RuntimeError("Patch with only one cell")
    if st != 0:
        raise RuntimeError(f"oracle failed with status {st}")
    return flux_hdiv


def ev_patch(mesh, k, facet_type, flux_dg, rhs_dg, cell_dofs, ndofs_glob, node,
             boundary_values=None):
    """Single EV patch: (cells, u [nrhs, N]) with N = k nf + n (k^2-k + nd) + 1 and the patch
    numbering [facet DOFs | cell-interior flux DOFs | DG DOFs | multiplier]."""
    lib = _load()
    t, nrhs, facet_type, flux_dg, rhs_dg = _prep(mesh, k, facet_type, flux_dg, rhs_dg, None)
    ms, keep_m = _mesh_struct(mesh)
    ts, keep_t = _tables_struct(t)
    cell_dofs = np.ascontiguousarray(cell_dofs, dtype=np.int32)
    n = int(mesh.node_cells_offsets[node + 1] - mesh.node_cells_offsets[node])
    nf = int(mesh.node_facets_offsets[node + 1] - mesh.node_facets_offsets[node])
    nmax = int(np.diff(mesh.node_cells_offsets).max())
    ndof_max = k * (nmax + 1) + nmax * (k * k - k + t.nd) + 1
    cells = np.zeros(n, dtype=np.int32)
    u = np.zeros((nrhs, ndof_max))
    if boundary_values is not None:
        boundary_values = np.ascontiguousarray(boundary_values, dtype=np.float64)
    nm = C.c_int32(0)
    st = lib.oracle_ev_patch(C.byref(ms), C.byref(ts), _p(t.flux_div), C.c_int(nrhs),
                             _p(facet_type), _p(cell_dofs), C.c_int64(ndofs_glob),
                             _p(boundary_values), _p(flux_dg), _p(rhs_dg), C.c_int32(node),
                             _p(cells), _p(u), C.byref(nm))
    if st < 0:
        raise RuntimeError(f"oracle failed with status {st}")
    assert nm.value == ndof_max
    N = k * nf + n * (k * k - k + t.nd) + 1
    return cells, u[:, :N]
