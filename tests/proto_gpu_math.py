"""numpy prototype of the formulation the HIP kernels implement (test infrastructure).

Same patch problem as the reference, restructured for the GPU ("Kernel math"):
every patch cell ("lane") computes cell-local quantities from exact reference tensors
(tools/gen_tables.py) - no quadrature -, neighbours exchange k facet moments, a prefix sum
fixes the zero-order moments, and the reduced SPD system in the unknowns
[d | (k-1) higher moments per patch facet | interior DOFs per cell] is solved.

All facet quantities are OUTWARD flux moments mu_j = int_E (w . n_out) s^j in the cell's own
facet parameter s; the RT coefficients are c_{f,j} = pf_f * mu_j, pf_f = +-sign(detJ).
The prototype loops over patches in Python (small meshes only) but is written lane-wise so
that it transliterates to the kernel.
"""

import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
from gen_tables import tables_float  # noqa: E402


def patch_solve(mesh, tab, fan, node_idx, facet_type_r, G, f, ev=False, bvals=None):
    """One patch, one RHS. fan: dict from oracle.build_patches; returns (cells, coeffs[n, nrt]).

    ev=True: the constrained-minimisation (EV) patch problem in the same reduced unknowns: no jump
    data (conforming particular solution), divergence data hat f + grad hat . G and the extra load
    (phi_h, hat G); bvals [ncells, nrt]: broken-layout boundary DOFs of the flux BC (EV only)."""
    k, nrt, nd, nq = tab["k"], tab["nrt"], tab["nd"], tab["nq"]
    kb = k - 1
    nadd = (k - 1) * (k - 2) // 2
    ndiv = k * (k + 1) // 2 - 1
    S, F, H, D, B = tab["S"], tab["F"], tab["H"], tab["D"], tab["B"]
    NREF, NOUT = tab["NREF"], tab["NOUT"]
    n = int(fan["ncells"][node_idx])
    cells = fan["cells"][node_idx]
    fl = fan["fcts_local"][node_idx]
    il = fan["inodes_local"][node_idx]
    fcts = fan["fcts"][node_idx]
    interior = cells[0] >= 0
    nf = n if interior else n + 1
    bc0 = (not interior) and facet_type_r[fcts[0]] == 2
    bcn = (not interior) and facet_type_r[fcts[n]] == 2

    lanes = []
    for i in range(n):  # lane i = cell T_{i+1}, minus facet E_i, plus facet E_{i+1}
        a = i + 1
        c = cells[a]
        fm, fp, ln = fl[2 * a - 1], fl[2 * a], il[a]
        x = mesh.x[mesh.cell_nodes[c], :2]
        J = np.stack([x[1] - x[0], x[2] - x[0]], axis=1)
        detJ = J[0, 0] * J[1, 1] - J[0, 1] * J[1, 0]
        adj = np.array([[J[1, 1], -J[0, 1]], [-J[1, 0], J[0, 0]]])  # detJ * K
        sgn = 1.0 if detJ > 0 else -1.0
        pf_m = sgn if NOUT[fm] else -sgn
        pf_p = sgn if NOUT[fp] else -sgn
        # reversal flags w.r.t. the neighbours (se/solve_patch_semiexplt.hpp:324-389)
        def perm(cell, fct):
            return mesh.facet_perm[cell, np.nonzero(mesh.cell_facets[cell] == fct)[0][0]]
        rev_m = rev_p = False
        if interior or a > 1:
            rev_m = perm(cells[a - 1], fcts[a - 1]) != perm(c, fcts[a - 1])
        if interior or a < n:
            rev_p = perm(cells[a + 1], fcts[a]) != perm(c, fcts[a])
        Gc = G[c]  # [nd, 2]
        fc = f[c]
        nu_m = adj.T @ NREF[fm]
        nu_p = adj.T @ NREF[fp]
        gm = pf_m * np.einsum("ij,i->j", F[fm, ln], Gc @ nu_m)
        gp = pf_p * np.einsum("ij,i->j", F[fp, ln], Gc @ nu_p)
        Ghat = Gc @ adj.T  # [nd, X] = (adj G_i)_X
        R = detJ * (fc @ H[ln]) - np.einsum("iX,iXq->q", Ghat, D[ln])
        blin = None
        if ev:
            gm = np.zeros(k)
            gp = np.zeros(k)
            ghat_ref = np.array([[-1.0, -1.0], [1.0, 0.0], [0.0, 1.0]])[ln]
            R = detJ * (fc @ H[ln]) + (Ghat @ ghat_ref) @ tab["HG"]
            JtG = Gc @ J  # [nd, c] = (J^T G_d)_c
            blin = sgn * np.einsum("idc,dc->i", tab["WGF"][ln], JtG)
        R0 = sgn * R[0]
        # active DOF indices: [minus facet k | plus facet k | add | div]
        idx = [fm * k + j for j in range(k)] + [fp * k + j for j in range(k)] \
            + [3 * k + ndiv + q for q in range(nadd)] + [3 * k + q for q in range(ndiv)]
        g = J.T @ J
        Mact = (g[0, 0] * S[0] + g[0, 1] * S[1] + g[1, 1] * S[2])[np.ix_(idx, idx)] / abs(detJ)
        sg = np.array([pf_m] * k + [pf_p] * k + [1.0] * (nadd + ndiv))
        My = Mact * sg[:, None] * sg[None, :]
        bm = bp = np.zeros(k)
        if ev and bvals is not None:
            # prescribed outward moments pf * HB b of hat_a * g on the end facets
            bm = pf_m * tab["HB"][fm, ln] @ bvals[c, fm * k:(fm + 1) * k]
            bp = pf_p * tab["HB"][fp, ln] @ bvals[c, fp * k:(fp + 1) * k]
        lanes.append(dict(c=c, fm=fm, fp=fp, ln=ln, pf_m=pf_m, pf_p=pf_p, rev_m=rev_m,
                          rev_p=rev_p, gm=gm - bm, gp=gp - bp, R=R, R0=R0, My=My, idx=idx,
                          blin=blin, sg=sg))

    # jumps on the plus facets (owner frame), zero-order chain
    for i, L in enumerate(lanes):
        has_nb = interior or i < n - 1
        if has_nb:
            nb = lanes[(i + 1) % n]
            Bp = B if L["rev_p"] else np.eye(k)
            L["J"] = L["gp"] + Bp @ nb["gm"]
        else:
            L["J"] = np.zeros(k)
    t = 0.0
    for i, L in enumerate(lanes):
        Jprev0 = lanes[i - 1]["J"][0] if (interior or i > 0) else 0.0
        t += L["R0"] + Jprev0
        L["t"] = t
    delta = 0.0
    if bc0:
        delta = lanes[0]["gm"][0]
    elif bcn:
        delta = -lanes[n - 1]["gp"][0] - lanes[n - 1]["t"]
    d_fixed = bc0 or bcn

    # particular solution in own-frame outward moments
    for i, L in enumerate(lanes):
        mu_p = np.zeros(k)
        mu_p[0] = L["t"] + delta
        if (not interior) and i == n - 1 and bcn:
            mu_p[1:] = -L["gp"][1:]
        L["mu_p"] = mu_p
    for i, L in enumerate(lanes):
        if interior or i > 0:
            prev = lanes[i - 1]
            v = prev["mu_p"] + prev["J"]
            Bm = B if L["rev_m"] else np.eye(k)
            L["Bm"] = Bm
            L["mu_m"] = -Bm @ v
        else:
            L["Bm"] = np.eye(k)
            mu_m = np.zeros(k)
            mu_m[0] = -delta
            if bc0:
                mu_m[1:] = -L["gm"][1:]
            L["mu_m"] = mu_m

    # reduced system
    dim = 1 + kb * nf + nadd * n
    A = np.zeros((dim, dim))
    rhs = np.zeros(dim)
    for i, L in enumerate(lanes):
        ny = 2 * k + nadd
        ytil = np.concatenate([L["mu_m"], L["mu_p"], np.zeros(nadd)])
        full = np.concatenate([ytil, L["R"][1:]])
        w = L["My"] @ full
        nh = 1 + 2 * kb + nadd
        Q = np.zeros((ny, nh))  # local unknowns [d | um (kb) | up (kb) | ua]
        Q[:k, 0] = -L["Bm"][:, 0]
        Q[k, 0] = 1.0
        for j in range(1, k):
            Q[:k, j] = -L["Bm"][:, j]
            Q[k + j, kb + j] = 1.0
        for q in range(nadd):
            Q[2 * k + q, 1 + 2 * kb + q] = 1.0
        Te = Q.T @ L["My"][:ny, :ny] @ Q
        Le = -Q.T @ w[:ny]
        if ev:
            Le += Q.T @ (L["sg"] * L["blin"][L["idx"]])[:ny]
        fi_m = i
        fi_p = (i + 1) % nf if interior else i + 1
        gidx = [0] + [1 + fi_m * kb + j for j in range(kb)] + [1 + fi_p * kb + j for j in range(kb)] \
            + [1 + nf * kb + i * nadd + q for q in range(nadd)]
        L["gidx"], L["Q"], L["ytil"] = gidx, Q, ytil
        A[np.ix_(gidx, gidx)] += Te
        rhs[gidx] += Le
    fixed = []
    if d_fixed:
        fixed.append(0)
    if bc0:
        fixed += [1 + j for j in range(kb)]
    if bcn:
        fixed += [1 + n * kb + j for j in range(kb)]
    for q in fixed:
        A[q, :] = 0
        A[:, q] = 0
        A[q, q] = 1
        rhs[q] = 0
    u = np.linalg.solve(A, rhs)

    out = np.zeros((n, nrt))
    for i, L in enumerate(lanes):
        y = L["ytil"] + L["Q"] @ u[L["gidx"]]
        cfull = np.zeros(nrt)
        for j in range(k):
            cfull[L["fm"] * k + j] = L["pf_m"] * y[j]
            cfull[L["fp"] * k + j] = L["pf_p"] * y[k + j]
        for q in range(nadd):
            cfull[3 * k + ndiv + q] = y[2 * k + q]
        for q in range(ndiv):
            cfull[3 * k + q] = L["R"][1 + q]
        out[i] = cfull
    return cells[1:n + 1], out


def reconstruct(mesh, k, deg, facet_type, flux_dg, rhs_dg, fans, ev=False, bvals=None):
    tab = tables_float(k, deg)
    nd, nrt = tab["nd"], tab["nrt"]
    G = flux_dg.reshape(mesh.ncells, nd, 2)
    f = rhs_dg.reshape(mesh.ncells, nd)
    ft = np.asarray(facet_type).reshape(-1, mesh.nfacets)[0]
    x = np.zeros((mesh.ncells, nrt))
    for node in range(mesh.nnodes):
        cells, coef = patch_solve(mesh, tab, fans, node, ft, G, f, ev=ev,
                                  bvals=None if bvals is None else bvals.reshape(mesh.ncells, nrt))
        np.add.at(x, cells, coef)
    return x.reshape(-1)
