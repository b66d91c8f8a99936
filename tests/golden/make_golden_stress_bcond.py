#!/usr/bin/env python3
"""Generates tests/golden/stress_bcond_<mesh>_k<k>.npz: the 12 mixed boundary layouts of the reference's
python/test/unit/test_stressqlb_bcond.py:147-166 (per side and per stress row either a traction or a
displacement condition) for k = 2, 3, 4 on the reference's 2 x 2 crossed unit square and on a perturbed,
orientation-shuffled 4 x 4 one.

Inputs are REAL discrete stresses: the P_k^2 Galerkin solve of tests/galerkin.py::solve_elasticity stands in
for solve_primal_problem_general_usquare (:27-144: random DG_{k-1} body force and tractions, u_r = 0 on the
other sides), sigma_h = -2 eps(u_h) - div u_h I row by row.  The expected output is the CPU oracle's
(oracle/eqlb_oracle.c); before saving, the generator asserts what the reference's test asserts - flux BCs,
divergence, jumps and weak symmetry - and that weak symmetry FAILS exactly for the reference's documented
expected fails (k = 2, layouts 8, 10, 12, :164-165) on the reference's mesh.  The reference itself cannot run
here: parity with dolfinx_eqlb stays unpinned by execution.
    python tests/golden/make_golden_stress_bcond.py
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]

from cases import BCOND_EXPECTED_FAILS, BCOND_LAYOUTS, BCOND_MESHES, bcond_case  # noqa: E402
from test_oracle_stress import asym_moments  # noqa: E402
from dolfinx_eqlb_amd.eqlb import check_eqlb_conditions as chk  # noqa: E402
from oracle import oracle  # noqa: E402


if __name__ == "__main__":
    for mname in BCOND_MESHES:
        for k in (2, 3, 4):
            out = {key: [] for key in ("facet_type", "flux_dg", "rhs_dg", "boundary_values", "flux_hdiv")}
            for id_bc in sorted(BCOND_LAYOUTS):
                mesh, ft, G, f, bv = bcond_case(mname, k, id_bc)
                x = oracle.se_reconstruct(mesh, k, ft, G, f, boundary_values=bv, stress=True)
                for r in range(2):
                    res, nrm = chk.divergence_residual(mesh, k, x[r], G[r], f[r])
                    assert res < 1e-10 * nrm
                    assert chk.check_jump_condition(mesh, k, x[r], G[r], atol=1e-10)
                    fb = np.nonzero(ft[r] == 2)[0]
                    assert chk.boundary_flux_residual(mesh, k, x[r], G[r], fb, boundary_values=bv[r]) < 1e-10
                asym = np.abs(asym_moments(mesh, k, x)[1]).max()
                expected_fail = (k, id_bc) in BCOND_EXPECTED_FAILS
                if mname == "crossed2":
                    assert (asym > 1e-5) == expected_fail, (mname, k, id_bc, asym)
                elif not expected_fail:
                    assert asym < 1e-11, (mname, k, id_bc, asym)
                for key, val in zip(out, (ft, G, f, bv, x)):
                    out[key].append(val)
                print(mname, k, id_bc, f"asym {asym:.2e}", "(expected fail)" if expected_fail else "")
            np.savez_compressed(os.path.join(HERE, f"stress_bcond_{mname}_k{k}.npz"), x=mesh.x[:, :2],
                                cell_nodes=mesh.cell_nodes, k=np.int32(k), ids=np.array(sorted(BCOND_LAYOUTS)),
                                **{key: np.stack(val) for key, val in out.items()})
