#!/usr/bin/env python3
"""Generates tests/golden/ev_*.npz (constrained-minimisation equilibrator) with the CPU oracle,
AFTER it passed the independent-minimiser checks of tests/test_oracle_ev.py.  `flux_hdiv` holds the
conforming hierarchic RT_k DOFs in the default numbering of
dolfinx_eqlb_amd/eqlb/conforming.py.  Run from the repository root:
    python tests/golden/make_golden_ev.py
"""

import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]

from cases import make_case  # noqa: E402
from golden_util import save_case  # noqa: E402
from oracle import oracle  # noqa: E402

from dolfinx_eqlb_amd.eqlb.conforming import conforming_dofmap  # noqa: E402

CASES = [  # (name, n, k, bc, shuffle, perturb, nrhs)
    ("ev_crossed2_k2_dirichlet", 2, 2, "dirichlet", None, 0.0, 1),
    ("ev_crossed4_k1_shuffled_neumann", 4, 1, "neumann_lt", 1234, 0.3, 1),
    ("ev_crossed4_k2_shuffled_neumann", 4, 2, "neumann_lt", 1234, 0.3, 2),
    ("ev_crossed4_k3_shuffled_neumann", 4, 3, "neumann_bottom", 1234, 0.3, 1),
]

if __name__ == "__main__":
    for name, n, k, bc, shuffle, perturb, nrhs in CASES:
        mesh, ft, G, f = make_case(n, k, bc, shuffle=shuffle, perturb=perturb, nrhs=nrhs)
        cd, nd = conforming_dofmap(mesh, k)
        x = oracle.ev_reconstruct(mesh, k, ft, G, f, cd, nd)
        save_case(os.path.join(HERE, name + ".npz"), mesh, k, ft, G, f, x)
        print(name, x.shape)
