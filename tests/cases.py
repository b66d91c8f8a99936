"""Shared small test problems (meshes, BC layouts, compatible data)."""

import numpy as np

from dolfinx_eqlb_amd.mesh import create_unit_square
from synthetic import facet_types, make_compatible_data


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


# the 12 boundary layouts of python/test/unit/test_stressqlb_bcond.py:147-166: per side of the unit square
# (x = 0, y = 0; the sides x = 1, y = 1 are primal Dirichlet for both rows) whether stress row 0 / row 1
# carries a flux (traction) condition
BCOND_LAYOUTS = {
    1: [[True, False], [False, False]], 2: [[False, True], [False, False]], 3: [[False, False], [False, True]],
    4: [[False, False], [True, False]], 5: [[True, False], [False, True]], 6: [[True, False], [True, False]],
    7: [[False, True], [False, True]], 8: [[False, True], [True, False]], 9: [[True, False], [True, True]],
    10: [[False, True], [True, True]], 11: [[True, True], [False, True]], 12: [[True, True], [True, False]],
}
# "Expected fails for degree 2: BCs 8, 10 and 12 / TODO - Extend patch grouping to handle these cases"
# (test_stressqlb_bcond.py:164-165): the two-cell corner patch between the sides x = 0 and y = 0 cannot be
# made weakly symmetric on its own at RT_2
BCOND_EXPECTED_FAILS = {(2, 8), (2, 10), (2, 12)}
BCOND_MESHES = {"crossed2": dict(n=2, shuffle_seed=None, perturb=0.0),
                "crossed4p": dict(n=4, shuffle_seed=4, perturb=0.2)}


def bcond_case(mesh_name, k, id_bc):
    """(mesh, facet_type [2, nf], G, f, boundary_values): Galerkin elasticity data (tests/galerkin.py)."""
    import galerkin as gk
    mesh = create_unit_square(**BCOND_MESHES[mesh_name])
    ft = gk.elasticity_facet_types(mesh, BCOND_LAYOUTS[id_bc])
    G, f, bv = gk.solve_elasticity(mesh, k, ft, seed=1000 * k + id_bc)
    return mesh, ft, G, f, bv
