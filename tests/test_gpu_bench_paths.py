"""bench.py end to end on the device: the default one-GPU line and the host side of the N > 1 path
(torch.distributed replaced by a single-process stand-in, tools/smoke_bench_world2.py - no RCCL)."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable] + cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout.strip().splitlines()


def test_bench_line_has_the_contract_fields():
    out = _run(["bench.py", "--n", "120", "--steps", "5", "--warmup", "1"])
    d = json.loads(out[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["roofline"]["bound"] == "hbm" and 0.0 < d["roofline"]["frac"] < 1.0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1
    assert d["div_residual_rel"] < 1e-10


@pytest.mark.parametrize("rank,world,extra", [(0, 2, []), (1, 2, []), (3, 8, []), (0, 2, ["--stress"]),
                                              (0, 2, ["--k", "3"]), (1, 3, ["--ev"]), (0, 2, ["--ev", "--k", "3"])])
def test_multi_gpu_host_path(rank, world, extra):
    out = _run(["tools/smoke_bench_world2.py", "--n", "100", "--steps", "3", "--warmup", "1"] + extra,
               env={"SMOKE_RANK": str(rank), "SMOKE_WORLD": str(world)})
    if rank == 0:
        d = json.loads(out[-1])
        assert d["n_gpus"] == world and d["scaling"] == "weak"
        if not extra:
            assert "behind the interior tiles" in d["config"]["partition"]
        assert d["div_residual_rel"] < 1e-10  # rank 0 receives nothing: its owned rows (EV: held DOFs) are complete
        assert d["halo"]["rank0_bytes_sent"] > 0 and d["halo"]["halo_only_ms"] > 0.0
