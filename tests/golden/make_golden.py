#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the CPU oracle.

The reference ships no golden vectors and cannot run here, so these fixtures
are produced by the oracle AFTER it passed the KKT-minimiser and predicate checks of
tests/test_oracle.py; they pin the oracle against regressions and give the HIP path fixed
vectors that travel to the GPU box.  Run from the repository root:
    python tests/golden/make_golden.py
"""

import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]

from cases import make_case  # noqa: E402
from golden_util import save_case  # noqa: E402
from oracle import oracle  # noqa: E402

CASES = [  # (name, n, k, bc, shuffle, perturb, nrhs)
    ("crossed2_k1_dirichlet", 2, 1, "dirichlet", None, 0.0, 1),
    ("crossed2_k2_dirichlet", 2, 2, "dirichlet", None, 0.0, 1),
    ("crossed2_k3_dirichlet", 2, 3, "dirichlet", None, 0.0, 1),
    ("crossed4_k1_shuffled_neumann", 4, 1, "neumann_lt", 1234, 0.3, 1),
    ("crossed4_k2_shuffled_neumann", 4, 2, "neumann_lt", 1234, 0.3, 2),
    ("crossed4_k3_shuffled_neumann", 4, 3, "neumann_lt", 1234, 0.3, 1),
    ("crossed4_k2_shuffled_dirichlet", 4, 2, "dirichlet", 99, 0.3, 1),
]

if __name__ == "__main__":
    for name, n, k, bc, shuffle, perturb, nrhs in CASES:
        mesh, ft, G, f = make_case(n, k, bc, shuffle=shuffle, perturb=perturb, nrhs=nrhs)
        x = oracle.se_reconstruct(mesh, k, ft, G, f)
        save_case(os.path.join(HERE, name + ".npz"), mesh, k, ft, G, f, x)
        print(name, x.shape)
