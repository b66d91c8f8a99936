"""Golden vectors: small complete problems (mesh + data + expected corrector) as .npz."""

import numpy as np

from dolfinx_eqlb_amd.mesh import create_mesh


def save_case(path, mesh, k, ft, G, f, expected):
    np.savez_compressed(path, x=mesh.x[:, :2], cell_nodes=mesh.cell_nodes, k=np.int32(k),
                        facet_type=ft, flux_dg=G, rhs_dg=f, flux_hdiv=expected)


def load_case(path):
    d = np.load(path, allow_pickle=False)
    mesh = create_mesh(d["x"], d["cell_nodes"])
    return mesh, int(d["k"]), d["facet_type"], d["flux_dg"], d["rhs_dg"], d["flux_hdiv"]


def load_bcond(mesh_name, k):
    """tests/golden/stress_bcond_<mesh>_k<k>.npz (make_golden_stress_bcond.py): mesh and, per layout id,
    (facet_type, flux_dg, rhs_dg, boundary_values, flux_hdiv)."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                             f"stress_bcond_{mesh_name}_k{k}.npz"), allow_pickle=False)
    mesh = create_mesh(d["x"], d["cell_nodes"])
    cases = {int(i): tuple(d[key][j] for key in ("facet_type", "flux_dg", "rhs_dg", "boundary_values", "flux_hdiv"))
             for j, i in enumerate(d["ids"])}
    return mesh, cases
