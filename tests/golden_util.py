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
