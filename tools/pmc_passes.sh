cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc9 && mkdir -p gpurun_out/pmc9
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_REQ_sum SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc9/p$i -- python3 bench.py --no-cpu-baseline --steps 3 > gpurun_out/pmc9/log$i.txt 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import collections, csv, glob
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc9/p*/**/*_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_se_patch_tiled" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
with open("gpurun_out/pmc9/summary.csv", "w") as fh:
    fh.write("# PMC counters of k_se_patch_tiled<2,1,0> (mean per launch), rocprofv3 --pmc passes, bench.py --steps 3\ncounter,value\n")
    for k in sorted(acc):
        fh.write(f"{k},{sum(acc[k]) / len(acc[k]):.1f}\n")
print(open("gpurun_out/pmc9/summary.csv").read())
PY
