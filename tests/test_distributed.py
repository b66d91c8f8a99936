"""N > 1 path on CPU: world_size-2/3 gloo runs of the strip partition + halo reduction
(the per-rank patch solver is the oracle here; on the GPU box it is the HIP library)."""

import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n,k", [(2, 4, 2), (3, 3, 1), (2, 3, 3)])
def test_strip_partition_halo_reduction(oracle_mod, world, n, k):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(n), str(k)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("OK") == world


@pytest.mark.parametrize("world,mode,k", [(3, "se", 2), (3, "se", 3), (2, "ev", 2), (3, "ev", 1), (3, "se_local", 2)])
def test_general_partition_on_a_delaunay_mesh(oracle_mod, world, mode, k):
    """Partition (any mesh, any node ownership; several neighbours per rank) + halo reduction of the
    broken rows (SE) / the conforming DOFs (EV)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_dist_worker_general.py"), mode, str(k)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("OK") == world


def test_single_rank_partition_is_the_plain_mesh():
    from dolfinx_eqlb_amd.distributed import StripPartition
    p = StripPartition(4)
    assert p.node_mask is None and p.ncells_owned == p.mesh.ncells == 64
    assert sum(p.patch_cells_per_bin()) == 3 * p.mesh.ncells
