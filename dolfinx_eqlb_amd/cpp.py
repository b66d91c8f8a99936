"""ctypes binding of libeqlb_amd.so (include/eqlb.h) - the stand-in for the reference's
pybind11 module `dolfinx_eqlb.cpp` (python/dolfinx_eqlb/wrappers.cpp:259-272).

The product path has no CPU fallback: if the HIP library is missing or no device is visible,
the calls raise.  Errors of the C ABI are raised as RuntimeError, like the reference's
std::runtime_error -> RuntimeError translation.
"""

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EQLB_AMD_LIB", os.path.join(_HERE, "libeqlb_amd.so"))

MEM_HOST, MEM_DEVICE = 0, 1
SOLVER_LDS_CHOLESKY, SOLVER_SHUFFLE = 0, 1
SCATTER_SLOTS, SCATTER_ATOMIC, SCATTER_TILED = 0, 1, 2

# every symbol include/eqlb.h declares (tests check that the library exports all of them)
EXPORTED_SYMBOLS = [
    "eqlb_last_error", "eqlb_device_count", "eqlb_mesh_create", "eqlb_mesh_destroy",
    "eqlb_mesh_max_patch_cells", "eqlb_se_create", "eqlb_se_destroy", "eqlb_se_set_option",
    "eqlb_se_set_boundary", "eqlb_se_equilibrate", "eqlb_se_num_patches",
    "eqlb_se_export_patches", "eqlb_get_reference_table", "eqlb_se_last_kernel_ms",
    "eqlb_project_dg", "eqlb_se_equilibrate_with_kornconst",
    "eqlb_ev_create", "eqlb_ev_destroy", "eqlb_ev_set_option", "eqlb_ev_set_dofmap",
    "eqlb_ev_num_dofs", "eqlb_ev_set_boundary", "eqlb_ev_equilibrate", "eqlb_ev_num_patches",
    "eqlb_ev_last_kernel_ms", "eqlb_se_tiling_info", "eqlb_se_estimate",
    "eqlb_halo_pack", "eqlb_halo_unpack_add", "eqlb_ev_estimate",
    "eqlb_se_check_status", "eqlb_ev_check_status",
    "eqlb_se_set_priority_cells", "eqlb_se_num_priority_tiles", "eqlb_se_equilibrate_tiles",
    "eqlb_se_equilibrate_lists", "eqlb_ev_equilibrate_lists", "eqlb_se_kornconst",
    "eqlb_ev_set_basis_transform", "eqlb_se_estimate_stress", "eqlb_oscillation",
    "eqlb_halo_exchange", "eqlb_halo_reduce", "eqlb_rccl_get_unique_id", "eqlb_rccl_comm_create",
    "eqlb_rccl_comm_destroy", "eqlb_halo_create", "eqlb_halo_destroy", "eqlb_halo_bytes", "eqlb_halo_reduce_plan",
]

_lib = None


def _bind_torch_hip_runtime():
    """torch bundles its own HIP runtime (torch/lib/libamdhip64.so, same soname as /opt/rocm's).  Two
    runtimes in one process do not share devices: the one initialised second sees none.  Callers that
    hand torch tensors to this library (bench.py, distributed.HaloExchange) therefore need ONE runtime,
    torch's.  If torch is installed but not imported yet, its runtime is mapped first (without importing
    torch), so that libeqlb_amd.so binds to it and a later `import torch` finds it already loaded."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass  # a torch build without a usable bundled runtime: the system runtime is used


def lib():
    """Load libeqlb_amd.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()')")
        _bind_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.eqlb_last_error.restype = C.c_char_p
        L.eqlb_device_count.restype = C.c_int
        L.eqlb_se_num_patches.restype = C.c_int64
        L.eqlb_se_last_kernel_ms.restype = C.c_double
        L.eqlb_mesh_max_patch_cells.restype = C.c_int32
        L.eqlb_ev_num_dofs.restype = C.c_int64
        L.eqlb_ev_num_patches.restype = C.c_int64
        L.eqlb_ev_last_kernel_ms.restype = C.c_double
        for name in ("eqlb_mesh_destroy", "eqlb_se_destroy", "eqlb_ev_destroy", "eqlb_halo_destroy",
                     "eqlb_rccl_comm_destroy"):
            getattr(L, name).restype = None
        _lib = L
    return _lib


def _check(status):
    if status != 0:
        raise RuntimeError(lib().eqlb_last_error().decode() or f"eqlb error {status}")


def _hp(a):
    return a.ctypes.data_as(C.c_void_p)


def device_count() -> int:
    return int(lib().eqlb_device_count())


class DeviceMesh:
    """Device-resident copy of a flat mesh (eqlb_mesh_create)."""

    def __init__(self, mesh):
        self.mesh = mesh
        self._h = C.c_void_p()
        arrs = [np.ascontiguousarray(mesh.x, dtype=np.float64),
                np.ascontiguousarray(mesh.cell_nodes, dtype=np.int32),
                np.ascontiguousarray(mesh.cell_facets, dtype=np.int32),
                np.ascontiguousarray(mesh.facet_nodes, dtype=np.int32),
                np.ascontiguousarray(mesh.facet_cells_offsets, dtype=np.int32),
                np.ascontiguousarray(mesh.facet_cells, dtype=np.int32),
                np.ascontiguousarray(mesh.node_cells_offsets, dtype=np.int32),
                np.ascontiguousarray(mesh.node_cells, dtype=np.int32),
                np.ascontiguousarray(mesh.node_facets_offsets, dtype=np.int32),
                np.ascontiguousarray(mesh.node_facets, dtype=np.int32),
                np.ascontiguousarray(mesh.facet_perm, dtype=np.uint8)]
        _check(lib().eqlb_mesh_create(C.c_int32(mesh.nnodes), C.c_int32(mesh.ncells),
                                      C.c_int32(mesh.nfacets), *[_hp(a) for a in arrs],
                                      C.byref(self._h)))

    @property
    def max_patch_cells(self):
        return int(lib().eqlb_mesh_max_patch_cells(self._h))

    def close(self):
        if self._h:
            lib().eqlb_mesh_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SemiExplicitEquilibrator:
    """eqlb_se_* handle: RT_k equilibrator on a device mesh."""

    def __init__(self, dmesh: DeviceMesh, k: int, nrhs: int, degree_dg=None,
                 reconstruct_stress=False, estimate_korn=False):
        self.dmesh = dmesh
        self.k, self.nrhs = k, nrhs
        self.reconstruct_stress = bool(reconstruct_stress)
        self.degree_dg = k - 1 if degree_dg is None else degree_dg
        self.nrt = k * (k + 2)
        self.nd = (self.degree_dg + 1) * (self.degree_dg + 2) // 2
        self._h = C.c_void_p()
        _check(lib().eqlb_se_create(dmesh._h, C.c_int32(k), C.c_int32(self.degree_dg),
                                    C.c_int32(nrhs), C.c_int32(int(reconstruct_stress)),
                                    C.c_int32(int(estimate_korn)), C.byref(self._h)))

    def set_option(self, key: str, value: int):
        _check(lib().eqlb_se_set_option(self._h, key.encode(), C.c_int32(value)))

    def set_boundary(self, facet_type, boundary_values=None, node_mask=None):
        m = self.dmesh.mesh
        ft = np.ascontiguousarray(facet_type, dtype=np.int8).reshape(self.nrhs, m.nfacets)
        bv = None
        if boundary_values is not None:
            bv = np.ascontiguousarray(boundary_values, dtype=np.float64)
            assert bv.size == self.nrhs * m.ncells * self.nrt
        nm = None
        if node_mask is not None:
            nm = np.ascontiguousarray(node_mask, dtype=np.uint8)
            assert nm.size == m.nnodes
        _check(lib().eqlb_se_set_boundary(self._h, _hp(ft), _hp(bv) if bv is not None else None,
                                          _hp(nm) if nm is not None else None))

    @property
    def num_patches(self):
        return int(lib().eqlb_se_num_patches(self._h))

    def tiling_info(self):
        """dict(ntiles, cells_per_tile, patch_instances, lane_slots) of the tiled launch."""
        v = [C.c_int64(0) for _ in range(4)]
        _check(lib().eqlb_se_tiling_info(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("ntiles", "cells_per_tile", "patch_instances", "lane_slots"),
                        [int(x.value) for x in v]))

    def equilibrate_host(self, flux_dg, rhs_dg, flux_hdiv=None):
        """Host numpy arrays in/out; flux_hdiv is accumulated (+=) like the reference."""
        m = self.dmesh.mesh
        g = np.ascontiguousarray(flux_dg, dtype=np.float64).reshape(self.nrhs, -1)
        f = np.ascontiguousarray(rhs_dg, dtype=np.float64).reshape(self.nrhs, -1)
        if g.shape[1] != m.ncells * self.nd * 2 or f.shape[1] != m.ncells * self.nd:
            raise RuntimeError("Equilibration: Input sizes does not match")
        if flux_hdiv is None:
            flux_hdiv = np.zeros((self.nrhs, m.ncells * self.nrt))
        assert flux_hdiv.dtype == np.float64 and flux_hdiv.flags.c_contiguous
        assert flux_hdiv.size == self.nrhs * m.ncells * self.nrt
        _check(lib().eqlb_se_equilibrate(self._h, _hp(g), _hp(f), _hp(flux_hdiv),
                                         C.c_int32(MEM_HOST), None))
        return flux_hdiv

    def equilibrate_host_with_kornconst(self, flux_dg, rhs_dg, flux_hdiv=None, korn=None):
        """As equilibrate_host, plus the accumulated squared Korn constants [ncells]."""
        m = self.dmesh.mesh
        g = np.ascontiguousarray(flux_dg, dtype=np.float64).reshape(self.nrhs, -1)
        f = np.ascontiguousarray(rhs_dg, dtype=np.float64).reshape(self.nrhs, -1)
        if g.shape[1] != m.ncells * self.nd * 2 or f.shape[1] != m.ncells * self.nd:
            raise RuntimeError("Equilibration: Input sizes does not match")
        if flux_hdiv is None:
            flux_hdiv = np.zeros((self.nrhs, m.ncells * self.nrt))
        if korn is None:
            korn = np.zeros(m.ncells)
        _check(lib().eqlb_se_equilibrate_with_kornconst(self._h, _hp(g), _hp(f), _hp(flux_hdiv),
                                                        _hp(korn), C.c_int32(MEM_HOST), None))
        return flux_hdiv, korn

    def equilibrate_device(self, flux_dg_ptr: int, rhs_dg_ptr: int, flux_hdiv_ptr: int,
                           stream: int = 0):
        """Raw device pointers (e.g. torch tensor .data_ptr()) and a hipStream_t handle;
        asynchronous."""
        _check(lib().eqlb_se_equilibrate(self._h, C.c_void_p(flux_dg_ptr), C.c_void_p(rhs_dg_ptr),
                                         C.c_void_p(flux_hdiv_ptr), C.c_int32(MEM_DEVICE),
                                         C.c_void_p(stream)))

    def equilibrate_device_tiles(self, flux_dg_ptr: int, rhs_dg_ptr: int, flux_hdiv_ptr: int,
                                 tile_first: int, tile_count: int, stream: int = 0):
        """equilibrate_device for a range of tiles (tiled scatter; count -1 = to the end)."""
        _check(lib().eqlb_se_equilibrate_tiles(self._h, C.c_void_p(flux_dg_ptr), C.c_void_p(rhs_dg_ptr),
                                               C.c_void_p(flux_hdiv_ptr), C.c_int32(tile_first),
                                               C.c_int32(tile_count), C.c_void_p(stream)))

    def set_priority_cells(self, cells):
        """Cells whose tiles become the first tiles at the next set_boundary (two-phase sweeps)."""
        c = np.ascontiguousarray(cells, dtype=np.int32)
        _check(lib().eqlb_se_set_priority_cells(self._h, _hp(c), C.c_int32(c.size)))

    @property
    def num_priority_tiles(self) -> int:
        return int(lib().eqlb_se_num_priority_tiles(self._h))

    def check_status(self, stream: int = 0):
        """After device-memory calls: waits for the stream and raises if a patch system was not
        positive definite (degenerate cell geometry)."""
        _check(lib().eqlb_se_check_status(self._h, C.c_void_p(stream)))

    def last_kernel_ms(self, which=0):
        return float(lib().eqlb_se_last_kernel_ms(self._h, C.c_int32(which)))

    def export_patches(self):
        m = self.dmesh.mesh
        stride = self.dmesh.max_patch_cells + 2
        nn = m.nnodes
        out = dict(ncells=np.zeros(nn, np.int32), cells=np.zeros((nn, stride), np.int32),
                   fcts=np.zeros((nn, stride), np.int32),
                   fcts_local=np.zeros((nn, 2 * stride), np.int8),
                   inodes_local=np.zeros((nn, stride), np.int8),
                   reversed=np.zeros((nn, 2 * stride), np.int8), stride=stride)
        _check(lib().eqlb_se_export_patches(self._h, C.c_int32(stride), _hp(out["ncells"]),
                                            _hp(out["cells"]), _hp(out["fcts"]),
                                            _hp(out["fcts_local"]), _hp(out["inodes_local"]),
                                            _hp(out["reversed"])))
        return out

    def close(self):
        if self._h:
            lib().eqlb_se_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ConstrainedMinEquilibrator:
    """eqlb_ev_* handle: constrained-minimisation (Ern-Vohralik) equilibrator, flux in the
    conforming hierarchic RT_k (include/eqlb.h)."""

    def __init__(self, dmesh: DeviceMesh, k: int, nrhs: int, cell_dofs=None, ndofs=None):
        self.dmesh = dmesh
        self.k, self.nrhs = k, nrhs
        self.nrt = k * (k + 2)
        self.nd = k * (k + 1) // 2
        self.output = 0
        self._h = C.c_void_p()
        _check(lib().eqlb_ev_create(dmesh._h, C.c_int32(k), C.c_int32(nrhs), C.byref(self._h)))
        if cell_dofs is not None:
            cd = np.ascontiguousarray(cell_dofs, dtype=np.int32)
            assert cd.shape == (dmesh.mesh.ncells, self.nrt)
            _check(lib().eqlb_ev_set_dofmap(self._h, _hp(cd), C.c_int64(int(ndofs))))

    def set_basis_transform(self, C=None, R=None):
        """Change of basis of the conforming output (eqlb_ev_set_basis_transform): C [nrt, nrt], R [k, k]."""
        c = None if C is None else np.ascontiguousarray(C, dtype=np.float64)
        r = None if R is None else np.ascontiguousarray(R, dtype=np.float64)
        if c is not None:
            assert c.shape == (self.nrt, self.nrt) and (r is None or r.shape == (self.k, self.k))
        _check(lib().eqlb_ev_set_basis_transform(self._h, _hp(c) if c is not None else None,
                                                 _hp(r) if r is not None else None))

    @property
    def ndofs(self):
        return int(lib().eqlb_ev_num_dofs(self._h))

    @property
    def num_patches(self):
        return int(lib().eqlb_ev_num_patches(self._h))

    def set_option(self, key: str, value: int):
        _check(lib().eqlb_ev_set_option(self._h, key.encode(), C.c_int32(value)))
        if key == "output":
            self.output = value

    def set_boundary(self, facet_type, boundary_values=None, node_mask=None):
        m = self.dmesh.mesh
        ft = np.ascontiguousarray(facet_type, dtype=np.int8).reshape(self.nrhs, m.nfacets)
        bv = None
        if boundary_values is not None:
            bv = np.ascontiguousarray(boundary_values, dtype=np.float64)
            assert bv.size == self.nrhs * self.ndofs
        nm = None
        if node_mask is not None:
            nm = np.ascontiguousarray(node_mask, dtype=np.uint8)
            assert nm.size == m.nnodes
        _check(lib().eqlb_ev_set_boundary(self._h, _hp(ft), _hp(bv) if bv is not None else None,
                                          _hp(nm) if nm is not None else None))

    def _nout(self):
        return self.dmesh.mesh.ncells * self.nrt if self.output == 1 else self.ndofs

    def equilibrate_host(self, flux_dg, rhs_dg, flux_hdiv=None):
        """Host numpy arrays in/out; flux_hdiv [nrhs, ndofs] is accumulated (+=)."""
        m = self.dmesh.mesh
        g = np.ascontiguousarray(flux_dg, dtype=np.float64).reshape(self.nrhs, -1)
        f = np.ascontiguousarray(rhs_dg, dtype=np.float64).reshape(self.nrhs, -1)
        if g.shape[1] != m.ncells * self.nd * 2 or f.shape[1] != m.ncells * self.nd:
            raise RuntimeError("Equilibration: Input sizes does not match")
        if flux_hdiv is None:
            flux_hdiv = np.zeros((self.nrhs, self._nout()))
        assert flux_hdiv.dtype == np.float64 and flux_hdiv.flags.c_contiguous
        assert flux_hdiv.size == self.nrhs * self._nout()
        _check(lib().eqlb_ev_equilibrate(self._h, _hp(g), _hp(f), _hp(flux_hdiv),
                                         C.c_int32(MEM_HOST), None))
        return flux_hdiv

    def equilibrate_device(self, flux_dg_ptr: int, rhs_dg_ptr: int, flux_hdiv_ptr: int,
                           stream: int = 0):
        _check(lib().eqlb_ev_equilibrate(self._h, C.c_void_p(flux_dg_ptr), C.c_void_p(rhs_dg_ptr),
                                         C.c_void_p(flux_hdiv_ptr), C.c_int32(MEM_DEVICE),
                                         C.c_void_p(stream)))

    def check_status(self, stream: int = 0):
        _check(lib().eqlb_ev_check_status(self._h, C.c_void_p(stream)))

    def last_kernel_ms(self, which=0):
        return float(lib().eqlb_ev_last_kernel_ms(self._h, C.c_int32(which)))

    def close(self):
        if self._h:
            lib().eqlb_ev_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def reconstruct_fluxes_minimisation(flux_hdiv, flux_dg, rhs_dg, boundary_data):
    """Stand-in for `reconstruct_fluxes_minimisation(a, l_pen, l, flux_hdiv, boundary_data)`
    (wrappers.cpp:85-95): the forms a, l_pen, l of FluxEqlbEV.py:113-134 are fixed, their data
    (projected flux, projected RHS) is passed as flat arrays; `boundary_data` is a configured
    ConstrainedMinEquilibrator."""
    return boundary_data.equilibrate_host(flux_dg, rhs_dg, flux_hdiv)


def project_dg(dmesh: DeviceMesh, degree: int, qpoints, qweights, qvalues, bs: int = 1):
    """eqlb_project_dg on host arrays: qvalues [nrhs, ncells, nq, bs] -> DOFs [nrhs, ncells*nd*bs]."""
    m = dmesh.mesh
    qp = np.ascontiguousarray(qpoints, dtype=np.float64)
    qw = np.ascontiguousarray(qweights, dtype=np.float64)
    nq = qw.size
    qv = np.ascontiguousarray(qvalues, dtype=np.float64)
    if qv.size % (m.ncells * nq * bs) != 0:
        raise RuntimeError("Local solver: Input sizes does not match")
    nrhs = qv.size // (m.ncells * nq * bs)
    nd = (degree + 1) * (degree + 2) // 2
    out = np.zeros((nrhs, m.ncells * nd * bs))
    _check(lib().eqlb_project_dg(dmesh._h, C.c_int32(degree), C.c_int32(bs), C.c_int32(nrhs),
                                 C.c_int32(nq), _hp(qp), _hp(qw), _hp(qv), _hp(out),
                                 C.c_int32(MEM_HOST), None))
    return out


def estimate(dmesh: DeviceMesh, k: int, flux_hdiv, flux_dg, rhs_dg, conforming_flux=False):
    """eqlb_se_estimate (eqlb_ev_estimate with conforming_flux=True: the flux is an EV result in
    the broken layout) on host arrays [nrhs, ...]: returns (cell_div2 [nrhs, ncells],
    cell_sig2 [nrhs, ncells], facet_jump [nrhs, nfacets])."""
    m = dmesh.mesh
    nrt, nd = k * (k + 2), k * (k + 1) // 2
    x = np.ascontiguousarray(flux_hdiv, dtype=np.float64).reshape(-1, m.ncells * nrt)
    nrhs = x.shape[0]
    g = np.ascontiguousarray(flux_dg, dtype=np.float64).reshape(nrhs, -1)
    f = np.ascontiguousarray(rhs_dg, dtype=np.float64).reshape(nrhs, -1)
    if g.shape[1] != m.ncells * nd * 2 or f.shape[1] != m.ncells * nd:
        raise RuntimeError("Equilibration: Input sizes does not match")
    div2 = np.zeros((nrhs, m.ncells))
    sig2 = np.zeros((nrhs, m.ncells))
    jump = np.zeros((nrhs, m.nfacets))
    fn = lib().eqlb_ev_estimate if conforming_flux else lib().eqlb_se_estimate
    _check(fn(dmesh._h, C.c_int32(k), C.c_int32(nrhs), _hp(x), _hp(g), _hp(f),
              _hp(div2), _hp(sig2), _hp(jump), C.c_int32(MEM_HOST), None))
    return div2, sig2, jump


def estimate_stress(dmesh: DeviceMesh, k: int, flux_hdiv, korn=None, pi_1: float = 1.0):
    """eqlb_se_estimate_stress on host arrays: flux_hdiv [2, ncells*k(k+2)] (rows of the equilibrated
    stress), korn [ncells] cell-wise Korn constants or None.  Returns (cell_energy [ncells],
    cell_wsym [ncells], node_asym [nnodes])."""
    m = dmesh.mesh
    x = np.ascontiguousarray(flux_hdiv, dtype=np.float64)
    if x.size != 2 * m.ncells * k * (k + 2):
        raise RuntimeError("Equilibration: Input sizes does not match")
    kc = None if korn is None else np.ascontiguousarray(korn, dtype=np.float64)
    if kc is not None and kc.size != m.ncells:
        raise RuntimeError("Equilibration: Input sizes does not match")
    energy, wsym, asym = np.zeros(m.ncells), np.zeros(m.ncells), np.zeros(m.nnodes)
    _check(lib().eqlb_se_estimate_stress(dmesh._h, C.c_int32(k), _hp(x), _hp(kc) if kc is not None else None,
                                         C.c_double(pi_1), _hp(energy), _hp(wsym), _hp(asym),
                                         C.c_int32(MEM_HOST), None))
    return energy, wsym, asym


def oscillation(dmesh: DeviceMesh, k: int, flux, flux_dg, qpoints, qweights, fvalues, korn=None):
    """eqlb_oscillation on host arrays: flux [nrhs, ncells*k(k+2)], flux_dg [nrhs, ncells*k(k+1)] or None
    (conforming flux in the broken layout), fvalues [nrhs, ncells, nq].  Returns [nrhs, ncells]."""
    m = dmesh.mesh
    x = np.ascontiguousarray(flux, dtype=np.float64).reshape(-1, m.ncells * k * (k + 2))
    nrhs = x.shape[0]
    g = None if flux_dg is None else np.ascontiguousarray(flux_dg, dtype=np.float64).reshape(nrhs, -1)
    qp = np.ascontiguousarray(qpoints, dtype=np.float64)
    qw = np.ascontiguousarray(qweights, dtype=np.float64)
    nq = qw.size
    fv = np.ascontiguousarray(fvalues, dtype=np.float64)
    if fv.size != nrhs * m.ncells * nq or (g is not None and g.shape[1] != m.ncells * k * (k + 1)):
        raise RuntimeError("Equilibration: Input sizes does not match")
    kc = None if korn is None else np.ascontiguousarray(korn, dtype=np.float64)
    out = np.zeros((nrhs, m.ncells))
    _check(lib().eqlb_oscillation(dmesh._h, C.c_int32(k), C.c_int32(nrhs), _hp(x),
                                  _hp(g) if g is not None else None, C.c_int32(nq), _hp(qp), _hp(qw), _hp(fv),
                                  _hp(kc) if kc is not None else None, _hp(out), C.c_int32(MEM_HOST), None))
    return out


def halo_pack(x_ptr, cells_ptr, buf_ptr, nrhs, nlist, nrt, ncells, clear=True, stream=0):
    """eqlb_halo_pack on raw device pointers (asynchronous on `stream`)."""
    _check(lib().eqlb_halo_pack(C.c_int32(nrhs), C.c_int32(nlist), C.c_int32(nrt), C.c_int64(ncells),
                                C.c_void_p(cells_ptr), C.c_void_p(x_ptr), C.c_void_p(buf_ptr),
                                C.c_int32(int(clear)), C.c_void_p(stream)))


def halo_unpack_add(x_ptr, cells_ptr, buf_ptr, nrhs, nlist, nrt, ncells, stream=0):
    _check(lib().eqlb_halo_unpack_add(C.c_int32(nrhs), C.c_int32(nlist), C.c_int32(nrt),
                                      C.c_int64(ncells), C.c_void_p(cells_ptr), C.c_void_p(x_ptr),
                                      C.c_void_p(buf_ptr), C.c_void_p(stream)))


class RcclComm:
    """An RCCL communicator made through the library (eqlb_rccl_get_unique_id / eqlb_rccl_comm_create): what
    a host without an RCCL binding of its own uses for eqlb_halo_exchange / eqlb_halo_reduce.  The 128-byte
    unique id is made on one rank (`RcclComm.unique_id()`) and distributed by the caller."""

    def __init__(self, unique_id: bytes, nranks: int, rank: int):
        if len(unique_id) != 128:
            raise RuntimeError("RcclComm: the unique id has 128 bytes")
        self._h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), 128)
        _check(lib().eqlb_rccl_comm_create(buf, C.c_int32(nranks), C.c_int32(rank), C.byref(self._h)))
        self.nranks, self.rank = nranks, rank

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _check(lib().eqlb_rccl_get_unique_id(buf))
        return buf.raw

    @property
    def handle(self):
        return self._h.value

    def destroy(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().eqlb_rccl_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class HaloPlan:
    """Host-side argument block of eqlb_halo_reduce / eqlb_halo_exchange: peers and, per peer, device index
    lists and device staging buffers (raw pointers; the caller keeps the memory alive)."""

    def __init__(self, peers, send_idx_ptrs, nsend, send_buf_ptrs, recv_idx_ptrs, nrecv, recv_buf_ptrs):
        n = len(peers)
        self.n = n
        self.peers = (C.c_int32 * n)(*[int(q) for q in peers])
        self.send_idx = (C.c_void_p * n)(*[C.c_void_p(int(p) or None) for p in send_idx_ptrs])
        self.recv_idx = (C.c_void_p * n)(*[C.c_void_p(int(p) or None) for p in recv_idx_ptrs])
        self.send_buf = (C.c_void_p * n)(*[C.c_void_p(int(p) or None) for p in send_buf_ptrs])
        self.recv_buf = (C.c_void_p * n)(*[C.c_void_p(int(p) or None) for p in recv_buf_ptrs])
        self.nsend = (C.c_int64 * n)(*[int(v) for v in nsend])
        self.nrecv = (C.c_int64 * n)(*[int(v) for v in nrecv])


def halo_reduce(comm, plan: HaloPlan, x_ptr, nrhs, nrt, nentries, stream=0):
    """eqlb_halo_reduce: pack (+ clear), grouped RCCL send / recv, unpack-add - one call, asynchronous."""
    h = comm.handle if isinstance(comm, RcclComm) else comm
    _check(lib().eqlb_halo_reduce(C.c_void_p(h), C.c_int32(nrhs), C.c_int32(nrt), C.c_int64(nentries),
                                  C.c_void_p(x_ptr), C.c_int32(plan.n), plan.peers, plan.send_idx, plan.nsend,
                                  plan.send_buf, plan.recv_idx, plan.nrecv, plan.recv_buf, C.c_void_p(stream)))


def halo_exchange(comm, plan: HaloPlan, nrhs, nrt, stream=0):
    """eqlb_halo_exchange: the grouped send / recv alone (between halo_pack and halo_unpack_add)."""
    h = comm.handle if isinstance(comm, RcclComm) else comm
    sc = (C.c_int64 * plan.n)(*[int(v) * nrhs * nrt for v in plan.nsend])
    rc = (C.c_int64 * plan.n)(*[int(v) * nrhs * nrt for v in plan.nrecv])
    _check(lib().eqlb_halo_exchange(C.c_void_p(h), C.c_int32(plan.n), plan.peers, plan.send_buf, sc,
                                    plan.recv_buf, rc, C.c_void_p(stream)))


def get_reference_table(k, degree_dg, name):
    nrt, nd, nq = k * (k + 2), (degree_dg + 1) * (degree_dg + 2) // 2, k * (k + 1) // 2
    shapes = {"S": (3, nrt, nrt), "F": (3, 3, nd, k), "H": (3, nd, nq), "D": (3, nd, 2, nq)}
    out = np.zeros(shapes[name])
    n = lib().eqlb_get_reference_table(C.c_int32(k), C.c_int32(degree_dg), name.encode(),
                                       _hp(out), C.c_int32(out.size))
    if n != out.size:
        raise RuntimeError(lib().eqlb_last_error().decode())
    return out


def reconstruct_fluxes_semiexplt(flux_hdiv, flux_dg, rhs_dg, boundary_data, reconstruct_stress):
    """Same name and argument order as the reference binding (wrappers.cpp:97-115); the
    arguments are flat arrays [nrhs, ...] and `boundary_data` is a configured
    SemiExplicitEquilibrator (it carries the facet types like base::BoundaryData does)."""
    if bool(reconstruct_stress) != bool(getattr(boundary_data, "reconstruct_stress", False)):
        raise RuntimeError("reconstruct_stress does not match the equilibrator handle")
    return boundary_data.equilibrate_host(flux_dg, rhs_dg, flux_hdiv)
