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


def double_fan_mesh(order=0):
    """Two interior nodes I1, I2 joined by an edge, eight boundary nodes around them; six of the boundary
    nodes have two cells.  With flux BCs on the whole boundary for both stress rows the RT_2 stress
    equilibration forms TWO groups of boundary patches (se/reconstruction.hpp:170-234) whose internal patches
    (around I1 and I2) overlap: the reference's result depends on the node order, `order` = 1 swaps it."""
    from dolfinx_eqlb_amd.mesh import create_mesh
    x = np.array([[-0.5, 0.0], [0.5, 0.05], [1.5, 0.0], [1.0, 1.0], [0.05, 1.0], [-1.0, 1.1], [-1.5, 0.0],
                  [-1.0, -1.0], [0.0, -1.1], [1.1, -1.0]])
    i1, i2, b = 0, 1, [2, 3, 4, 5, 6, 7, 8, 9]
    cells = [[i2, b[0], b[1]], [i2, b[1], b[2]], [i2, b[6], b[7]], [i2, b[7], b[0]],
             [i1, b[2], b[3]], [i1, b[3], b[4]], [i1, b[4], b[5]], [i1, b[5], b[6]],
             [i1, i2, b[2]], [i1, b[6], i2]]
    cells = np.array(cells, dtype=np.int32)
    if order:
        perm = np.arange(x.shape[0])[::-1].copy()   # new id of old node i
        xn = np.empty_like(x)
        xn[perm] = x
        x, cells = xn, perm[cells].astype(np.int32)
    return create_mesh(x, cells)


def fan_chain_mesh():
    """Three interior nodes in a row I1 - I3 - I2, ten boundary nodes, six of them with two cells: with tractions
    on the whole boundary two groups (around I1 and I2) whose internal patches share NO cell."""
    from dolfinx_eqlb_amd.mesh import create_mesh
    x = np.array([[-1.0, 0.0], [1.0, 0.05], [0.0, 0.0], [2.0, 0.0], [1.5, 1.0], [0.5, 1.0], [-0.5, 1.1], [-1.5, 1.0],
                  [-2.0, 0.0], [-1.5, -1.0], [-0.5, -1.0], [0.5, -1.1], [1.5, -1.0]])
    i1, i2, i3 = 0, 1, 2
    cells = [[i2, 3, 4], [i2, 4, 5], [i2, 11, 12], [i2, 12, 3], [i1, 6, 7], [i1, 7, 8], [i1, 8, 9], [i1, 9, 10],
             [i3, 5, 6], [i3, 10, 11], [i3, i2, 5], [i3, 11, i2], [i1, i3, 6], [i1, 10, i3]]
    return create_mesh(x, np.array(cells, dtype=np.int32))
