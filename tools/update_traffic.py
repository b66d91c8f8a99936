#!/usr/bin/env python3
"""update_traffic.py: profiles/traffic.json entries from the PMC summaries of tools/pmc_kernel.sh
(profiles/rNN_*_pmc.csv: counter,value,launches).  traffic = FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, KiB ->
bytes; kernel_sha = sha256 of the kernel's source file NOW - run it right after the passes, on the tree they measured."""
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dolfinx_eqlb_amd", "csrc")
ENTRIES = {  # traffic.json key: (pmc summary, source file, note)
    "k_se_patch_tiled<K=2>": ("r03_headline_pmc.csv", "eqlb_se_kernels.hip", ""),
    "k_se_patch_tiled<K=3>": ("r03_k3_pmc.csv", "eqlb_se_kernels.hip", ""),
    "k_se_stress_tiled": ("r03_stress_pmc.csv", "eqlb_stress_tiled.hip", "fused kernel only - full patches; "),
}


def main():
    path = os.path.join(ROOT, "profiles", "traffic.json")
    data = json.load(open(path))
    for key, (pmc, src, note) in ENTRIES.items():
        f = os.path.join(ROOT, "profiles", pmc)
        if not os.path.exists(f):
            print("missing", f, file=sys.stderr)
            continue
        vals = {}
        for row in csv.reader(line for line in open(f) if not line.startswith("#")):
            if len(row) >= 2 and row[0] != "counter":
                vals[row[0]] = float(row[1])
        fetch, write = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
        sha = hashlib.sha256(open(os.path.join(CSRC, src), "rb").read()).hexdigest()[:16]
        data[key] = {"traffic": int(round((2.0 * fetch + write) * 1024.0)), "insts_valu": int(round(vals["SQ_INSTS_VALU"])),
                     "source": f"profiles/{pmc} ({note}FETCH {fetch:,.1f} KiB x 2 + WRITE {write:,.1f} KiB)".replace(",", " "),
                     "kernel_sha": sha}
        print(key, data[key])
    json.dump(data, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
