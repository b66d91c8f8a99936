"""Conforming (H(div)) version of the hierarchic RT_k space of `create_hierarchic_rt`
(python/dolfinx_eqlb/elmtlib/e_raviart_thomas.py:14-196) - the output space of the
constrained-minimisation equilibrator (`FluxEqlbEV`, eqlb/FluxEqlbEV.py:94-101 uses the Basix
RT_k space there; Basix is not available to this build).

Global DOFs: k moments per facet E in the GLOBAL facet frame - parameter s in [0,1] from the lower
to the higher node id, normal n_E = (t_y, -t_x) for the tangent t = x_hi - x_lo,
g_{E,j} = int_E (sigma.n_E) s^j / |E| ds-normalised as in the element definition - followed by the
k^2-k interior DOFs of every cell.  A cell sees the DOFs of its local facet f through
c_local = T_f g with T_f = -I if the cell traverses the facet low->high (facet_perm = 0) and
T_f = B (B_ji = C(j,i)(-1)^i) otherwise; both maps are involutions.

Host-side layout helpers only (index tables and a change of representation); the equilibration
itself runs in libeqlb_amd.so.
"""

from math import comb

import numpy as np


def reversal_matrix(k):
    return np.array([[comb(j, i) * (-1.0) ** i for i in range(k)] for j in range(k)])


def conforming_dofmap(mesh, k):
    """(cell_dofs [ncells, k(k+2)] int32 in the local hierarchic order, ndofs)."""
    ni = k * k - k
    nrt = k * (k + 2)
    cd = np.empty((mesh.ncells, nrt), dtype=np.int32)
    j = np.arange(k, dtype=np.int32)
    for f in range(3):
        cd[:, f * k:(f + 1) * k] = mesh.cell_facets[:, f, None] * k + j[None, :]
    cd[:, 3 * k:] = mesh.nfacets * k + np.arange(mesh.ncells, dtype=np.int32)[:, None] * ni \
        + np.arange(ni, dtype=np.int32)[None, :]
    return cd, mesh.nfacets * k + mesh.ncells * ni


def conforming_to_broken(mesh, k, x_conf, cell_dofs=None):
    """Coefficients in the discontinuous hierarchic RT_k (cell*k(k+2) + local) of a conforming
    function: c_{T,f} = T_f g_E."""
    if cell_dofs is None:
        cell_dofs, _ = conforming_dofmap(mesh, k)
    x_conf = np.asarray(x_conf, dtype=np.float64)
    c = x_conf[..., cell_dofs]  # [..., ncells, nrt]
    B = reversal_matrix(k)
    out = c.copy()
    for f in range(3):
        g = c[..., f * k:(f + 1) * k]
        rev = mesh.facet_perm[:, f].astype(bool)
        out[..., f * k:(f + 1) * k] = np.where(rev[:, None], g @ B.T, -g)
    return out.reshape(x_conf.shape[:-1] + (-1,))


def broken_to_conforming(mesh, k, x_broken, cell_dofs=None, ndofs=None):
    """Inverse of `conforming_to_broken` for a broken vector with continuous normal traces: the
    facet DOFs are read from the first cell of each facet."""
    if cell_dofs is None:
        cell_dofs, ndofs = conforming_dofmap(mesh, k)
    nrt = k * (k + 2)
    xb = np.asarray(x_broken, dtype=np.float64).reshape(-1, mesh.ncells, nrt)
    B = reversal_matrix(k)
    out = np.zeros((xb.shape[0], ndofs))
    g = xb.copy()
    for f in range(3):
        c = xb[:, :, f * k:(f + 1) * k]
        rev = mesh.facet_perm[:, f].astype(bool)
        g[:, :, f * k:(f + 1) * k] = np.where(rev[None, :, None], c @ B.T, -c)
    # later cells overwrite earlier ones; for conforming input all candidates agree
    for r in range(xb.shape[0]):
        out[r, cell_dofs.reshape(-1)] = g[r].reshape(-1)
    return out.reshape(np.asarray(x_broken).shape[:-1] + (ndofs,))
