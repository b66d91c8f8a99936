"""The quadrature-free formulation the HIP kernels implement (tests/proto_gpu_math.py,
exact reference tensors + own-frame flux moments) reproduces the oracle."""

import numpy as np
import pytest

import proto_gpu_math as pg
from cases import make_case


@pytest.mark.parametrize("k", [1, 2, 3])
@pytest.mark.parametrize("bc", ["dirichlet", "neumann_lt", "neumann_bottom"])
def test_prototype_equals_oracle(oracle_mod, k, bc):
    mesh, ft, G, f = make_case(3, k, bc)
    ref = oracle_mod.se_reconstruct(mesh, k, ft, G, f)[0]
    fans = oracle_mod.build_patches(mesh, ft)
    x = pg.reconstruct(mesh, k, k - 1, ft, G[0], f[0], fans)
    assert np.abs(x - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("k", [1, 2, 3])
@pytest.mark.parametrize("bc", ["dirichlet", "neumann_lt", "neumann_bottom"])
def test_prototype_ev_equals_oracle(oracle_mod, k, bc):
    """EV patch problem in the reduced unknowns of the SE kernel == saddle-point LU oracle."""
    from dolfinx_eqlb_amd.eqlb.conforming import conforming_dofmap, conforming_to_broken
    mesh, ft, G, f = make_case(3, k, bc)
    cd, nd = conforming_dofmap(mesh, k)
    ref = conforming_to_broken(mesh, k, oracle_mod.ev_reconstruct(mesh, k, ft, G, f, cd, nd)[0])
    fans = oracle_mod.build_patches(mesh, ft)
    x = pg.reconstruct(mesh, k, k - 1, ft, G[0], f[0], fans, ev=True)
    assert np.abs(x - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("k", [1, 2, 3])
def test_prototype_ev_inhomogeneous_bc(oracle_mod, k):
    from cases import BCS
    from dolfinx_eqlb_amd.eqlb.conforming import (broken_to_conforming, conforming_dofmap,
                                                  conforming_to_broken)
    from dolfinx_eqlb_amd.mesh import create_unit_square
    from synthetic import (boundary_dofs_from_field, facet_types,
                                            make_compatible_data)

    def w(x, y):
        return (0 * x + 0.8, 0 * x - 0.6) if k == 1 else (1.0 + 0.5 * x - 0.3 * y,
                                                           -0.7 + 0.2 * x + 0.4 * y)
    mesh = create_unit_square(3, shuffle_seed=5, perturb=0.3)
    ft = facet_types(mesh, BCS["neumann_lt"])
    G, f = make_compatible_data(mesh, k, ft, neumann_flux=w)
    cd, nd = conforming_dofmap(mesh, k)
    bvb = boundary_dofs_from_field(mesh, k, ft[0], w)
    bv = broken_to_conforming(mesh, k, bvb)
    ref = conforming_to_broken(mesh, k, oracle_mod.ev_reconstruct(
        mesh, k, ft, G[None], f[None], cd, nd, boundary_values=bv[None])[0])
    fans = oracle_mod.build_patches(mesh, ft)
    x = pg.reconstruct(mesh, k, k - 1, ft, G, f, fans, ev=True, bvals=bvb)
    assert np.abs(x - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())
