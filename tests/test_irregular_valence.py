"""Meshes with irregular / high vertex valence (gmsh and adaptively refined meshes
of the reference's demos): the lanes-per-patch bins P = 16, 32, 64 of the HIP kernels.  A polar
disk mesh has a centre node of valence `nsectors`."""

import numpy as np
import pytest

import kkt_reference as kr
from dolfinx_eqlb_amd.eqlb import check_eqlb_conditions as chk
from dolfinx_eqlb_amd.eqlb.conforming import conforming_dofmap
from dolfinx_eqlb_amd.mesh import create_disk
from synthetic import facet_types, make_compatible_data


def disk_case(ns, nr, k, neumann=False, seed=7):
    mesh = create_disk(ns, nr, shuffle_seed=seed)
    sel = (lambda x: x[:, 1] > 0.0) if neumann else None  # flux BC on the upper half of the rim
    ft = facet_types(mesh, sel)
    G, f = make_compatible_data(mesh, k, ft)
    return mesh, ft, G[None], f[None]


@pytest.mark.parametrize("k", [1, 2, 3])
def test_oracle_high_valence_patch_is_the_minimiser(oracle_mod, k):
    mesh, ft, G, f = disk_case(12, 2, k, neumann=True)
    x = oracle_mod.se_reconstruct(mesh, k, ft, G, f)[0]
    res, nrm = chk.divergence_residual(mesh, k, x, G[0], f[0])
    assert res < 1e-10 * nrm and chk.check_jump_condition(mesh, k, x, G[0], atol=1e-11)
    # the valence-12 centre patch and a rim patch against the independent minimiser
    for node in (0, mesh.nnodes - 1):
        cells, st, sol, u = oracle_mod.se_patch(mesh, k, ft, G, f, node)
        kc, kcoef, resid, nn = kr.solve_patch(mesh, k, node, ft, G[0], f[0])
        order = [list(kc).index(c) for c in cells]
        assert resid < 1e-11 and np.abs(sol[0] - kcoef[order]).max() < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 2, 3])
@pytest.mark.parametrize("ns,nr", [(12, 5), (24, 4), (40, 3), (63, 2)])
@pytest.mark.parametrize("scatter", [0, 2])
def test_gpu_high_valence(oracle_mod, k, ns, nr, scatter):
    from dolfinx_eqlb_amd import cpp
    mesh, ft, G, f = disk_case(ns, nr, k, neumann=True)
    ref = oracle_mod.se_reconstruct(mesh, k, ft, G, f)
    eq = cpp.SemiExplicitEquilibrator(cpp.DeviceMesh(mesh), k, 1)
    eq.set_option("scatter", scatter)
    eq.set_boundary(ft)
    x = eq.equilibrate_host(G, f)
    assert np.abs(x - ref).max() <= 1e-11 * np.abs(ref).max()


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 2, 3])
@pytest.mark.parametrize("ns", [12, 40])
def test_gpu_high_valence_ev(oracle_mod, k, ns):
    from dolfinx_eqlb_amd import cpp
    mesh, ft, G, f = disk_case(ns, 3, k, neumann=True)
    cd, nd = conforming_dofmap(mesh, k)
    ref = oracle_mod.ev_reconstruct(mesh, k, ft, G, f, cd, nd)
    eq = cpp.ConstrainedMinEquilibrator(cpp.DeviceMesh(mesh), k, 1)
    eq.set_boundary(ft)
    x = eq.equilibrate_host(G, f)
    assert np.abs(x - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())


@pytest.mark.gpu
def test_gpu_patch_too_large_is_refused():
    from dolfinx_eqlb_amd import cpp
    mesh = create_disk(70, 1)
    ft = facet_types(mesh, None)
    eq = cpp.SemiExplicitEquilibrator(cpp.DeviceMesh(mesh), 1, 1)
    with pytest.raises(RuntimeError, match="limit 63"):
        eq.set_boundary(ft)
