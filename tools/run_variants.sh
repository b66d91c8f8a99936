#!/bin/bash
# run_variants.sh "bench args" NAME...: bench.py once per timing-experiment library (on the GPU box)
ARGS="$1"; shift
mkdir -p gpurun_out
for v in "$@"; do
  if [ "$v" = "base" ]; then L=dolfinx_eqlb_amd/libeqlb_amd.so; else L=build_exp/lib_$v.so; fi
  EQLB_AMD_LIB=$L python bench.py --no-cpu-baseline $ARGS > gpurun_out/var_$v.log 2>&1 || { echo "$v FAILED"; tail -3 gpurun_out/var_$v.log; exit 1; }
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads(open(f"gpurun_out/var_{v}.log").read().strip().splitlines()[-1])
print(f"{v:16s} ms/step {d['ms_per_step']:.4f}  kernels {d['roofline'].get('all_kernels_ms')}")
PY
done
