#!/usr/bin/env python3
"""Extended randomised parity run (not part of the test suite): the cases of tests/test_fuzz.py for many more
seeds, on the GPU box.  python tools/fuzz_extended.py [first_seed] [count]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


class _Skip(Exception):
    pass


def main():
    import pytest
    import test_fuzz as tf
    from oracle import oracle
    oracle.build()
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    nrun = nskip = 0
    worst = 0.0
    for seed in range(first, first + count):
        for k in (1, 2, 3):
            try:
                tf.test_gpu_random_meshes(oracle, seed, k)
                nrun += 1
            except pytest.skip.Exception:
                nskip += 1
        for k in (2, 3):
            orig = tf.random_case

            def shifted(s, kk, nrhs, _o=orig, _seed=seed):
                return _o(_seed + 50000, kk, nrhs)
            tf.random_case = shifted
            try:
                tf.test_gpu_random_meshes_stress(oracle, 0, k)
                nrun += 1
            except pytest.skip.Exception:
                nskip += 1
            finally:
                tf.random_case = orig
        if (seed - first) % 10 == 9:
            print(f"seed {seed}: {nrun} cases ok, {nskip} skipped", flush=True)
    print(f"fuzz OK: {nrun} cases, {nskip} skipped (refused meshes)")


if __name__ == "__main__":
    main()
