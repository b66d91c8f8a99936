"""Unstructured Delaunay meshes (scipy): vertex valences 3 ... 10 mixed in one mesh, boundary patches
of every size - all lanes-per-patch bins, tiles with ragged rims, full and partial patch groups in
the same wave-block.  HIP path (tiled and slot scatter, SE and EV, stress) against the oracle."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def delaunay_mesh(npts, seed):
    from scipy.spatial import Delaunay
    from dolfinx_eqlb_amd.mesh import create_mesh
    for attempt in range(20):
        rng = np.random.default_rng(seed + 1000 * attempt)
        nb = int(3.5 * np.sqrt(npts))
        th = np.sort(rng.uniform(0.0, 2.0 * np.pi, nb))
        ring = np.stack([np.cos(th), 0.7 * np.sin(th)], axis=1)
        r = np.sqrt(rng.uniform(0.0, 0.93, npts))
        ph = rng.uniform(0.0, 2.0 * np.pi, npts)
        inner = np.stack([r * np.cos(ph), 0.7 * r * np.sin(ph)], axis=1)
        pts = np.concatenate([ring, inner])
        tri = Delaunay(pts).simplices.astype(np.int32)
        # counter-clockwise cells, no slivers of (numerically) zero area
        a, b, c = pts[tri[:, 0]], pts[tri[:, 1]], pts[tri[:, 2]]
        area = 0.5 * ((b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0]))
        tri = tri[np.abs(area) > 1e-9]
        used = np.unique(tri)
        if used.size != pts.shape[0]:
            continue
        mesh = create_mesh(pts, tri)
        val = np.diff(mesh.node_cells_offsets)
        if val.min() >= 2 and val.max() <= 63:
            return mesh
    raise RuntimeError("no admissible Delaunay mesh")


@pytest.mark.parametrize("k", [1, 2, 3])
def test_se_on_delaunay_mesh(oracle_mod, k):
    from dolfinx_eqlb_amd import cpp
    from synthetic import facet_types, make_compatible_data
    mesh = delaunay_mesh(900, seed=k)
    val = np.diff(mesh.node_cells_offsets)
    assert val.max() >= 8 and np.unique(val).size >= 5  # a genuine mix of valences
    ft = facet_types(mesh, lambda x: x[:, 0] < -0.2)
    G, f = make_compatible_data(mesh, k, ft, seed=4)
    ref = oracle_mod.se_reconstruct(mesh, k, ft, G[None], f[None])
    for scatter in (0, 2):
        eq = cpp.SemiExplicitEquilibrator(cpp.DeviceMesh(mesh), k, 1)
        eq.set_option("scatter", scatter)
        eq.set_boundary(ft)
        x = eq.equilibrate_host(G[None], f[None])
        assert np.abs(x - ref).max() <= 1e-9 * np.abs(ref).max(), scatter


def test_ev_and_stress_on_delaunay_mesh(oracle_mod):
    from dolfinx_eqlb_amd import cpp
    from dolfinx_eqlb_amd.eqlb.conforming import conforming_dofmap
    from synthetic import facet_types, make_compatible_data, make_compatible_stress_data
    k = 2
    mesh = delaunay_mesh(700, seed=11)
    ft = facet_types(mesh)
    G, f = make_compatible_data(mesh, k, ft, seed=5)
    cd, nd = conforming_dofmap(mesh, k)
    ev = cpp.ConstrainedMinEquilibrator(cpp.DeviceMesh(mesh), k, 1)
    ev.set_boundary(ft)
    x = ev.equilibrate_host(G[None], f[None])
    ref = oracle_mod.ev_reconstruct(mesh, k, ft, G[None], f[None], cd, nd)
    assert np.abs(x - ref).max() <= 1e-9 * np.abs(ref).max()
    ft2 = np.repeat(ft, 2, axis=0)
    Gs, fs = make_compatible_stress_data(mesh, k, ft2)
    eq = cpp.SemiExplicitEquilibrator(cpp.DeviceMesh(mesh), k, 2, reconstruct_stress=True)
    eq.set_boundary(ft2)
    xs = eq.equilibrate_host(Gs, fs)
    refs = oracle_mod.se_reconstruct(mesh, k, ft2, Gs, fs, stress=True)
    assert np.abs(xs - refs).max() <= 1e-9 * np.abs(refs).max()


@pytest.mark.parametrize("kind", ["square", "delaunay", "strip"])
def test_device_bisection_of_the_tiles(oracle_mod, kind, monkeypatch):
    """Tiles by recursive coordinate bisection on the device (one radix sort per level, eqlb_tiling_device.hip)
    against the host bisection (EQLB_TILING=host): same number of tiles, a comparable number of patch instances
    (both cut across the longer side of a segment's box), every cell in exactly one tile, and the same results
    (to rounding: which wave-blocks run the specialised full-patch instance depends on the tiles)."""
    from dolfinx_eqlb_amd import cpp
    from dolfinx_eqlb_amd.mesh import create_rectangle, create_unit_square
    from synthetic import facet_types, make_compatible_data
    k = 2
    if kind == "square":
        mesh = create_unit_square(40, shuffle_seed=3, perturb=0.2)
    elif kind == "strip":
        mesh = create_rectangle(96, 12, 0.0, 8.0, 0.0, 1.0)
    else:
        mesh = delaunay_mesh(2600, seed=5)
    assert mesh.ncells >= 4096
    ft = facet_types(mesh, None)
    G, f = make_compatible_data(mesh, k, ft, seed=2)
    out, info = {}, {}
    for mode in ("host", "device"):
        monkeypatch.setenv("EQLB_TILING", mode)
        eq = cpp.SemiExplicitEquilibrator(cpp.DeviceMesh(mesh), k, 1)  # a new device mesh: the tile order is cached per mesh
        eq.set_option("tile_cells", 64)
        eq.set_boundary(ft)
        out[mode] = eq.equilibrate_host(G[None], f[None])
        info[mode] = eq.tiling_info()
    ref = oracle_mod.se_reconstruct(mesh, k, ft, G[None], f[None])
    for mode in out:
        assert np.abs(out[mode] - ref).max() <= 1e-11 * np.abs(ref).max(), mode
    assert np.abs(out["host"] - out["device"]).max() <= 1e-13 * np.abs(ref).max()
    assert info["host"]["ntiles"] == info["device"]["ntiles"]
    assert info["device"]["patch_instances"] <= 1.15 * info["host"]["patch_instances"]
