#!/usr/bin/env python3
"""Generates tests/golden/config0_crossed32_k1_galerkin.npz: BASELINE.json configs[0] at its size.

The set-up of the reference's demo/poisson/demo_reconstruction.py (:44-54 exact solution
u = sin(2 pi x) cos(2 pi y), :89-95 crossed 32 x 32 unit square, :396-406 bc_type neumann_hom:
primal Dirichlet data on x = 0, 1, homogeneous flux BCs on y = 0, 1, :504 primal problem solved
with Pi_0 f for RT_1) with the stand-ins of this repository: the P1 Galerkin solve of
tests/galerkin.py supplies sigma_h = -grad u_h, the expected corrector comes from the CPU oracle
(after its checks in tests/test_oracle.py).  The reference itself cannot run here,
so the vector pins the oracle and the HIP path against regressions - parity with dolfinx_eqlb stays
unpinned by execution.  Run from the repository root:
    python tests/golden/make_golden_config0.py
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]

import galerkin as gk  # noqa: E402
from golden_util import save_case  # noqa: E402
from dolfinx_eqlb_amd.eqlb import check_eqlb_conditions as chk  # noqa: E402
from dolfinx_eqlb_amd.mesh import create_unit_square  # noqa: E402
from synthetic import facet_types  # noqa: E402
from oracle import oracle  # noqa: E402


def f_ext(x, y):
    return 8 * np.pi ** 2 * np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y)


def config0(n=32, k=1):
    mesh = create_unit_square(n)
    ft = facet_types(mesh, lambda mp: (np.abs(mp[:, 1]) < 1e-12) | (np.abs(mp[:, 1] - 1) < 1e-12))
    fh, _, _ = gk.project_rhs(mesh, k, f_ext)
    u, cd = gk.solve_poisson(mesh, k, f_ext, f_dg=fh, dirichlet_facets=np.nonzero(ft[0] == 1)[0])
    G = gk.discrete_flux(mesh, k, u, cd)
    return mesh, ft, G[None], fh[None]


if __name__ == "__main__":
    mesh, ft, G, f = config0()
    x = oracle.se_reconstruct(mesh, 1, ft, G, f)
    res, nrm = chk.divergence_residual(mesh, 1, x[0], G[0], f[0])
    assert res < 1e-10 * nrm and chk.check_jump_condition(mesh, 1, x[0], G[0], atol=1e-10)
    assert chk.boundary_flux_residual(mesh, 1, x[0], G[0], np.nonzero(ft[0] == 2)[0]) < 1e-10
    save_case(os.path.join(HERE, "config0_crossed32_k1_galerkin.npz"), mesh, 1, ft, G, f, x)
    print("config0", mesh.ncells, "cells", mesh.nnodes, "patches", x.shape, "div residual", res / nrm)
