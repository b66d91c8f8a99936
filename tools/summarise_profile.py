#!/usr/bin/env python3
"""Condense a rocprofv3 run directory (kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE
PMC passes) into one small CSV under profiles/.

usage: tools/summarise_profile.py <prof_dir> <out_csv> "<title>"
<prof_dir> holds trace/, pmc_fetch/, pmc_write/ as written by the gpurun recipe in DESIGN.md.
"""
import collections
import csv
import glob
import sys


def main(prof, out_path, title):
    out = [f"# {title}", "# rocprofv3 --kernel-trace --stats  (MI355X, gfx950, ROCm 7.2)", ""]
    stats = glob.glob(f"{prof}/trace/**/*_kernel_stats.csv", recursive=True)
    if stats:
        out += open(stats[0]).read().splitlines()[:9]
    out += ["", "# PMC passes (separate runs: rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE), mean per launch.",
            "# Raw counter unit = KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports",
            "# 1/2 of the bytes of wide coalesced loads -> x2 (calibrated on k_reduce_slots, whose read",
            "# volume is known by construction).",
            "kernel,FETCH_SIZE_raw_KiB,FETCH_MB_corrected,WRITE_SIZE_KiB,WRITE_MB"]
    vals = collections.defaultdict(dict)
    for name, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        fs = glob.glob(f"{prof}/{name}/**/*_counter_collection.csv", recursive=True)
        if not fs:
            continue
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(fs[0])):
            acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            vals[k][key] = sum(v) / len(v)
    for k, v in vals.items():
        if "eqlb" in k and "FETCH_SIZE" in v:
            w = v.get("WRITE_SIZE", 0.0)
            out.append(f"\"{k}\",{v['FETCH_SIZE']:.1f},{2 * v['FETCH_SIZE'] * 1024 / 1e6:.1f},"
                       f"{w:.1f},{w * 1024 / 1e6:.1f}")
    open(out_path, "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
