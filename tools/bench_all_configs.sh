#!/bin/bash
# bench_all_configs.sh: one bench.py line per configuration quoted in DESIGN.md section 7 (on the GPU box)
set -e
mkdir -p gpurun_out
run() { n=$1; shift; python bench.py --no-cpu-baseline "$@" > gpurun_out/cfg_$n.log 2>&1; python - "$n" <<'PY'
import json, sys
n = sys.argv[1]
d = json.loads(open(f"gpurun_out/cfg_{n}.log").read().strip().splitlines()[-1])
w = d.get("ms_per_step_windows", {})
print(f"{n:10s} ms/step {d['ms_per_step']:.4f} [min {w.get('min', 0):.4f} med {w.get('median', 0):.4f} max {w.get('max', 0):.4f}] value {d['value']:.3e} kernels {d['roofline'].get('all_kernels_ms')} res {d.get('div_residual_rel')}")
PY
}
run k2 --steps 50
run k1 --steps 50 --k 1
run k3 --steps 30 --k 3
run k2shuf --steps 50 --shuffle 1234
run stress --steps 20 --stress
run ev2 --steps 50 --ev
run ev3 --steps 20 --ev --k 3
run ev1 --steps 50 --ev --k 1
run k3big --steps 10 --k 3 --n 1414
run k2big --steps 10 --n 1414
run k2r4 --steps 20 --nrhs 4
run stress3 --steps 10 --stress --k 3
run k4 --steps 5 --k 4 --n 250
run stress4 --steps 3 --stress --k 4 --n 250
