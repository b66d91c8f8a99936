#!/bin/bash
# Round-3 evidence in one gpurun call: kernel traces (rocprofv3 --kernel-trace --stats) and PMC passes of the
# configurations quoted in DESIGN.md section 7.0.  Output: gpurun_out/r3p/*.csv (copy what is quoted into profiles/).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p
mkdir -p $O
WHAT=${1:-traces}   # traces | pmc | pmc2  (a gpurun call is limited to 20 minutes)
trace() { n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$n -- python3 bench.py --no-cpu-baseline --windows 2 "$@" > $O/t_$n.log 2>&1 || echo "trace $n failed"
  f=$(find $O/t_$n -name "*_kernel_stats.csv" | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --windows 2 $*  (MI355X, gfx950, ROCm 7.2); bench line: $(grep -a '^{"metric"' $O/t_$n.log | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms_per_step", round(d["ms_per_step"],4), "windows", [round(v,4) for v in d["ms_per_step_windows"]["all"]], "kernel_ms", d["roofline"]["all_kernels_ms"])' 2>/dev/null); the trace covers warm-up, settle probes and all windows: its average lies above the settled windows"; head -8 "$f"; } > $O/${n}_trace.csv
  rm -rf $O/t_$n
  cat $O/${n}_trace.csv | head -4
}
if [ "$WHAT" = traces ]; then
trace headline --steps 200
trace k2_8M --steps 40 --n 1414
trace stress --steps 100 --stress
trace k3 --steps 100 --k 3
trace k3_8M --steps 20 --k 3 --n 1414
trace ev2 --steps 200 --ev
trace ev3 --steps 100 --ev --k 3
trace k2_r4 --steps 100 --nrhs 4
trace k4 --steps 50 --k 4 --n 250
fi
if [ "$WHAT" = stress ]; then
trace stress --steps 100 --stress
fi
if [ "$WHAT" = pmc ]; then
bash tools/pmc_kernel.sh k_se_patch_tiled r3p/pmc_headline > $O/pmc_headline.log 2>&1; cp $O/pmc_headline/summary.csv $O/headline_pmc.csv
bash tools/pmc_kernel.sh k_se_stress_tiled r3p/pmc_stress --stress > $O/pmc_stress.log 2>&1; cp $O/pmc_stress/summary.csv $O/stress_pmc.csv
fi
if [ "$WHAT" = pmc2 ]; then
bash tools/pmc_kernel.sh k_se_patch_tiled r3p/pmc_k3 --k 3 > $O/pmc_k3.log 2>&1; cp $O/pmc_k3/summary.csv $O/k3_pmc.csv
bash tools/pmc_kernel.sh k_se_patch_tiled r3p/pmc_k2_8M --n 1414 > $O/pmc_k2_8M.log 2>&1; cp $O/pmc_k2_8M/summary.csv $O/k2_8M_pmc.csv
fi
rm -rf $O/pmc_headline $O/pmc_stress $O/pmc_k3 $O/pmc_k2_8M
ls $O
