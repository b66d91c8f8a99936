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
