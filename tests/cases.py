"""Shared small test problems (meshes, BC layouts, compatible data)."""

import numpy as np

from dolfinx_eqlb_amd.mesh import create_unit_square
from dolfinx_eqlb_amd.synthetic import facet_types, make_compatible_data


def neumann_left_top(mp):
    return (np.abs(mp[:, 0]) < 1e-12) | (np.abs(mp[:, 1] - 1) < 1e-12)


def neumann_all(mp):
    return np.ones(mp.shape[0], dtype=bool)


def neumann_bottom(mp):
    return np.abs(mp[:, 1]) < 1e-12


BCS = {"dirichlet": None, "neumann_lt": neumann_left_top, "neumann_bottom": neumann_bottom}


def make_case(n, k, bc="dirichlet", shuffle=77, perturb=0.3, diagonal="crossed", seed=20241003,
              nrhs=1):
    mesh = create_unit_square(n, diagonal=diagonal, shuffle_seed=shuffle, perturb=perturb)
    ft = facet_types(mesh, BCS[bc], nrhs=nrhs)
    G, f = [], []
    for r in range(nrhs):
        g_, f_ = make_compatible_data(mesh, k, ft, seed=seed + r)
        G.append(g_)
        f.append(f_)
    return mesh, ft, np.stack(G), np.stack(f)
