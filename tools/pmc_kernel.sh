#!/bin/bash
# PMC passes of one kernel of a bench.py configuration (rocprofv3 --pmc in separate runs, no tracing):
#   tools/pmc_kernel.sh <kernel substring> <out dir under gpurun_out> [bench.py flags ...]
# writes <out dir>/summary.csv (mean per launch of every counter).
K="$1"; OUT="gpurun_out/$2"; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf "$OUT" && mkdir -p "$OUT"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --settle 0 --windows 1 "$@" > "$OUT/log$i.txt" 2>&1 || echo "pass $i failed"
done
python3 - "$K" "$OUT" "$*" <<'PY'
import collections, csv, glob, sys
kern, out, flags = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if kern in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/summary.csv", "w") as fh:
    fh.write(f"# PMC counters of kernels matching '{kern}' (mean per launch), rocprofv3 --pmc passes, bench.py --steps 3 {flags}\ncounter,value,launches\n")
    for k in sorted(acc):
        fh.write(f"{k},{sum(acc[k]) / len(acc[k]):.1f},{len(acc[k])}\n")
print(open(out + "/summary.csv").read())
PY
