"""numpy statement of the weak-symmetry step as the fused stress kernel computes it (test
infrastructure; the companion of proto_gpu_math.py for se/solve_patch_weaksym.hpp:59-233).

Per patch the reference solves  [A 0 B0; 0 A B1; B0^T B1^T 0(+c)] [u0; u1; gamma] = [0; 0; Lc]
by the Schur complement on gamma (se/PatchData.hpp:598-663).  Here the same solution is computed the
way the lanes of a patch group do it - lane i <-> cell T_{i+1} <-> facet row E_i <-> ring point i:

  * A = [Z C^T; C A_c]: border (d, x_0) + tridiagonal chain x_1 .. x_{nf-1} (the layout of the
    semi-explicit solver); every solve with A is a parallel cyclic reduction on the chain over the
    lanes + a replicated 2 x 2 border system;
  * the columns of B_k are sparse (row E_i meets the ring points i-1, i, i+1 and the patch node; the
    d row is dense): column c of Y_k = A^-1 B_k is one more right-hand side of the chain reduction,
    its border part z^(c) = Zs^-1 q^(c) is computed by lane c from its neighbours' data alone;
  * S = sum_k B_k^T Y_k = T + sum_k Q_k^T Zs^-1 Q_k with T[r][c] = sum_{chain i} B[E_i][r] t_i^(c) (three
    terms per entry) and the rank-2 border part;
  * S gamma = -R is eliminated without pivoting (S is SPD on boundary patches; on interior patches
    S 1 = 0 and the mean-value row fixes the constant: lambda = sum R / sum M, gamma_node := 0);
  * u_k = -A^-1 (B_k gamma): one more solve per stress row.
"""

import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
from gen_tables import combo, tables_float  # noqa: E402


def element_data(mesh, tab, fan, node, c0, c1):
    """Per-lane element quantities of the patch of `node` (what the kernel has in registers):
    Te [3, 3], Be [2, 3, 3], Lce [3], Ce; c0, c1 [ncells, 8]: patch-local stress rows."""
    k, nrt = 2, 8
    n = int(fan["ncells"][node])
    cells = fan["cells"][node]
    fl, il, fcts = fan["fcts_local"][node], fan["inodes_local"][node], fan["fcts"][node]
    interior = cells[0] >= 0
    TE, VQ, V = tab["TE"], tab["VQ"], tab["V"]

    def perm(cell, fct):
        return mesh.facet_perm[cell, np.nonzero(mesh.cell_facets[cell] == fct)[0][0]]
    lanes = []
    for i in range(n):
        a = i + 1
        c = cells[a]
        fm, fp, ln = int(fl[2 * a - 1]), int(fl[2 * a]), int(il[a])
        x = mesh.x[mesh.cell_nodes[c], :2]
        J = np.stack([x[1] - x[0], x[2] - x[0]], axis=1)
        detJ = np.linalg.det(J)
        sgn = 1.0 if detJ > 0 else -1.0
        rev_m = False
        if interior or a > 1:
            rev_m = bool(perm(cells[a - 1], fcts[a - 1]) != perm(c, fcts[a - 1]))
        ci = combo(fm, fp, int(rev_m))
        g = J.T @ J / abs(detJ)
        te = g[0, 0] * TE[ci][0] + g[0, 1] * TE[ci][1] + g[1, 1] * TE[ci][2]
        Te = np.zeros((3, 3))
        for h in range(3):
            for gg in range(h + 1):
                Te[h, gg] = Te[gg, h] = te[h * (h + 1) // 2 + gg]
        v0, v1 = VQ[ci][0], VQ[ci][1]  # [NH, 3]
        Be = np.stack([J[1, 0] * v0 + J[1, 1] * v1, -(J[0, 0] * v0 + J[0, 1] * v1)])
        w0 = c0[c] * J[1, 0] - c1[c] * J[0, 0]
        w1 = c0[c] * J[1, 1] - c1[c] * J[0, 1]
        Lce = -sgn * (V[:, :, 0] @ w0 + V[:, :, 1] @ w1)
        pf_m = sgn if fm == 1 else -sgn
        pf_p = sgn if fp == 1 else -sgn
        lanes.append(dict(c=c, fm=fm, fp=fp, ln=ln, Te=Te, Be=Be, Lce=Lce, Ce=abs(detJ) / 6.0, rev_m=rev_m,
                          pf_m=pf_m, pf_p=pf_p))
    return n, interior, lanes


def pcr_solve(P, Dp, OffC, cols):
    """Parallel cyclic reduction of the chain: Dp diagonal, OffC[i] coupling of row i to row i + 1;
    rows outside the chain are identity rows with zero couplings.  cols [ncol, P] right-hand sides
    (zero outside the chain).  The level structure of EQLB_PCR_LEVEL in eqlb_se_kernels.hip."""
    b = Dp.copy()
    am = np.concatenate([[0.0], OffC[:-1]])  # coupling to row i - 1
    r = cols.copy()
    s = 1
    while s < P:
        ib = 1.0 / b

        def lo(v):
            return np.concatenate([np.zeros(v.shape[:-1] + (s,)), v[..., :-s]], axis=-1)

        def hi(v):
            return np.concatenate([v[..., s:], np.zeros(v.shape[:-1] + (s,))], axis=-1)
        cp = hi(am)  # coupling to row i + s
        al, ga = am * lo(ib), cp * hi(ib)
        b = b - al * am - ga * cp
        r = r - al * lo(r) - ga * hi(r)
        am = -al * lo(am)
        am[:2 * s] = 0.0
        s *= 2
    return r / b


def weaksym_lanes(P, n, interior, lanes):
    """Corrections ul [2, n, 3] (local unknowns d, um, up of every lane, per stress row)."""
    nf = n if interior else n + 1
    act = np.arange(P) < n
    rowv = np.arange(P) < nf

    def lane_arr(f):
        return np.array([f(lanes[i]) if i < n else 0.0 for i in range(P)])

    def prev(v):
        """value of the cell before facet E_i: lane i - 1, for interior patches lane n - 1 for i = 0"""
        out = np.zeros(P)
        for i in range(P):
            if rowv[i] and (i > 0 or interior):
                out[i] = v[i - 1] if i > 0 else v[n - 1]
        return out

    Te = lambda a, b_: lane_arr(lambda L: L["Te"][a, b_])  # noqa: E731
    alpha = Te(0, 0).sum()
    bt = Te(0, 1) + prev(Te(0, 2))
    Dg = Te(1, 1) + prev(Te(2, 2))
    Off = Te(1, 2)
    Dg[~rowv] = 1.0
    in_chain = (np.arange(P) >= 1) & rowv
    wraps = np.array([interior and i == n - 1 for i in range(P)])
    Dp = np.where(in_chain, Dg, 1.0)
    OffC = np.where(in_chain & ~wraps, Off, 0.0)
    B1 = np.where(in_chain, bt, 0.0)
    B2 = np.where(in_chain, np.where(np.arange(P) == 1, Off[0], 0.0) + np.where(wraps, Off, 0.0), 0.0)
    Z = np.array([[alpha, bt[0]], [bt[0], Dg[0]]])
    s1, s2 = pcr_solve(P, Dp, OffC, np.stack([B1, B2]))
    Sred = np.array([[B1 @ s1, B2 @ s1], [B2 @ s1, B2 @ s2]])
    Zi = np.linalg.inv(Z - Sred)

    def solve_A(b_d, b_x0, b_chain):
        """A y = b: (y_d, y_x0, y_chain [P])"""
        t = pcr_solve(P, Dp, OffC, b_chain[None])[0]
        z = Zi @ np.array([b_d - B1 @ t, b_x0 - B2 @ t])
        return z[0], z[1], t - s1 * z[0] - s2 * z[1]

    # ---- B_k, R, M per lane ----
    Brow = np.zeros((P, 2, 3))  # [lane i][k][m + 1]: row E_i x ring point i + m
    bc = np.zeros((P, 2))       # row E_i x patch node
    Bd = np.zeros((P, 2))       # d row x ring point i
    Bdc = np.zeros(2)           # d row x patch node
    for k in range(2):
        be = lambda h, which: lane_arr(lambda L: L["Be"][k, h, L[which]])  # noqa: E731
        Brow[:, k, 0] = prev(be(2, "fp"))
        Brow[:, k, 1] = be(1, "fp") + prev(be(2, "fm"))
        Brow[:, k, 2] = be(1, "fm")
        bc[:, k] = be(1, "ln") + prev(be(2, "ln"))
        Bd[:, k] = np.where(rowv, be(0, "fp") + prev(be(0, "fm")), 0.0)
        Bdc[k] = be(0, "ln").sum()
    Rring = np.where(rowv, lane_arr(lambda L: L["Lce"][L["fp"]]) + prev(lane_arr(lambda L: L["Lce"][L["fm"]])), 0.0)
    Mring = np.where(rowv, lane_arr(lambda L: L["Ce"]) + prev(lane_arr(lambda L: L["Ce"])), 0.0)
    Rc = lane_arr(lambda L: L["Lce"][L["ln"]]).sum()
    Mc = lane_arr(lambda L: L["Ce"]).sum()

    def ring(i):  # ring point of facet index i (cyclic for interior patches)
        return i % nf if interior else i

    # ---- columns of the chain right-hand sides, absolute layout: col[c][k][lane] ----
    cols = np.zeros((nf, 2, P))
    for i in range(P):
        if not in_chain[i]:
            continue
        for m in (-1, 0, 1):
            c = i + m
            if interior:
                c %= nf
            if 0 <= c < nf:
                cols[c, :, i] += Brow[i, :, m + 1]
    t = np.zeros((nf, 2, P))
    for k in range(2):
        t[:, k, :] = pcr_solve(P, Dp, OffC, cols[:, k, :])
    # border part of every column, computed by lane c: q^(c), z^(c)
    q = np.zeros((nf, 2, 2))
    for c in range(nf):
        for k in range(2):
            tred1 = tred2 = 0.0
            for m in (-1, 0, 1):  # rows E_{c+m} meet ring point c through their entry -m
                i = c + m
                if interior:
                    i %= nf
                if 0 <= i < P and in_chain[i]:
                    tred1 += s1[i] * Brow[i, k, 1 - m]
                    tred2 += s2[i] * Brow[i, k, 1 - m]
            bx0 = 0.0
            for m in (-1, 0, 1):
                if ring(0 + m) == c and (interior or m >= 0):
                    bx0 += Brow[0, k, m + 1]
            q[c, k] = [Bd[c, k] - tred1, bx0 - tred2]
    z = np.einsum("ab,ckb->cka", Zi, q)
    # S on the ring points
    S = np.zeros((nf, nf))
    for r in range(nf):
        for c in range(nf):
            v = 0.0
            for k in range(2):
                for m in (-1, 0, 1):
                    i = r + m
                    if interior:
                        i %= nf
                    if 0 <= i < P and in_chain[i]:
                        v += Brow[i, k, 1 - m] * t[c, k, i]
                v += q[r, k] @ z[c, k]
            S[r, c] = v
    if interior:
        lam = (Rc + Rring.sum()) / (Mc + Mring.sum())
        rhs = -(Rring[:nf] - lam * Mring[:nf])
        gam_ring = np.linalg.solve(S, rhs) if np.linalg.cond(S) < 1e13 else np.linalg.lstsq(S, rhs, rcond=None)[0]
        gam_c = 0.0
        shift = (Mring[:nf] @ gam_ring) / (Mc + Mring.sum())
        gam_ring = gam_ring - shift
        gam_c -= shift
    else:
        # the patch node joins: dense column (all rows E_i carry bc_i)
        tcen = np.zeros((2, P))
        qcen = np.zeros((2, 2))
        for k in range(2):
            tcen[k] = pcr_solve(P, Dp, OffC, np.where(in_chain, bc[:, k], 0.0)[None])[0]
            qcen[k] = [Bdc[k] - s1 @ np.where(in_chain, bc[:, k], 0.0), bc[0, k] - s2 @ np.where(in_chain, bc[:, k], 0.0)]
        zcen = np.einsum("ab,kb->ka", Zi, qcen)
        Sf = np.zeros((nf + 1, nf + 1))
        Sf[1:, 1:] = S
        for c in range(nf):
            v = sum(np.where(in_chain, bc[:, k], 0.0) @ t[c, k] + qcen[k] @ z[c, k] for k in range(2))
            Sf[0, 1 + c] = Sf[1 + c, 0] = v
        Sf[0, 0] = sum(np.where(in_chain, bc[:, k], 0.0) @ tcen[k] + qcen[k] @ zcen[k] for k in range(2))
        g = np.linalg.solve(Sf, -np.concatenate([[Rc], Rring[:nf]]))
        gam_c, gam_ring = g[0], g[1:]
    # ---- u_k = -A^-1 B_k gamma ----
    gr = np.zeros(P + 2)
    gr[:nf] = gam_ring

    def gring(i):
        i = ring(i)
        return gam_ring[i] if 0 <= i < nf else 0.0
    ul = np.zeros((2, n, 3))
    for k in range(2):
        vrow = np.array([Brow[i, k, 0] * gring(i - 1) + Brow[i, k, 1] * gring(i) + Brow[i, k, 2] * gring(i + 1)
                         + bc[i, k] * gam_c if rowv[i] else 0.0 for i in range(P)])
        if not interior:  # no wrap: lane 0 has no ring point -1, lane nf - 1 no ring point nf
            pass
        vd = Bd[:nf, k] @ gam_ring + Bdc[k] * gam_c
        yd, yx0, ych = solve_A(-vd, -vrow[0], np.where(in_chain, -vrow, 0.0))
        xs = ych.copy()
        xs[0] = yx0
        for i in range(n):
            fi_p = (i + 1) % n if interior else i + 1
            ul[k, i] = [yd, xs[i], xs[fi_p]]
    return ul


def stress_correction(mesh, fan, node, c0, c1):
    """Rows 0 / 1 of the weak-symmetry correction of the patch of `node` in RT coefficients
    [2, ncells, 8] (se/solve_patch_weaksym.hpp:189-232), from the patch-local rows c0, c1."""
    tab = tables_float(2, 1)
    n, interior, lanes = element_data(mesh, tab, fan, node, c0, c1)
    nf = n if interior else n + 1
    P = 4
    while P < nf:
        P *= 2
    ul = weaksym_lanes(P, n, interior, lanes)
    out = np.zeros((2, mesh.ncells, 8))
    B = np.array([[1.0, 0.0], [1.0, -1.0]])
    for k in range(2):
        for i, L in enumerate(lanes):
            u = ul[k, i]
            Bm = B if L["rev_m"] else np.eye(2)
            s = -Bm @ u[:2]
            yp = np.array([u[0], u[2]])
            out[k, L["c"], L["fm"] * 2:L["fm"] * 2 + 2] += L["pf_m"] * s
            out[k, L["c"], L["fp"] * 2:L["fp"] * 2 + 2] += L["pf_p"] * yp
    return out
