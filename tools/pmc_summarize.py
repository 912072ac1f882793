#!/usr/bin/env python3
"""Summarises the rocprofv3 outputs of tools/gpu.sh check into small files for profiles/:
  <round>_kernel_stats.csv   copy of the --stats kernel summary
  <round>_pmc_traffic.json   per kernel: launches, average FETCH_SIZE / WRITE_SIZE per launch
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  gfx950 correction
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-reports wide
coalesced streaming reads by 2x; the raw and the doubled figure are both stored, WRITE_SIZE is
stored raw (uncalibrated)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def counter_table(root, counter):
    files = glob.glob(os.path.join(root, "**", "*counter_collection*.csv"), recursive=True)
    acc = defaultdict(lambda: [0.0, 0])
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            k = row.get("Kernel_Name", "?").split("(")[0]
            acc[k][0] += float(row["Counter_Value"])
            acc[k][1] += 1
    return acc


def main():
    out_root, rnd = sys.argv[1], sys.argv[2]
    dst = os.path.join(out_root, "summary")
    os.makedirs(dst, exist_ok=True)
    for f in glob.glob(os.path.join(out_root, "prof", "**", "*kernel_stats*.csv"), recursive=True)[:1]:
        shutil.copy(f, os.path.join(dst, f"{rnd}_kernel_stats.csv"))
    fetch = counter_table(os.path.join(out_root, "pmc_fetch"), "FETCH_SIZE")
    write = counter_table(os.path.join(out_root, "pmc_write"), "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        short = k.split("<")[0].split("::")[-1]
        e = res.setdefault(short, {"kernel": k[:120]})
        if k in fetch and fetch[k][1]:
            e["launches_fetch_pass"] = fetch[k][1]
            e["fetch_bytes_per_launch_raw"] = 1024.0 * fetch[k][0] / fetch[k][1]
            e["fetch_bytes_per_launch_x2"] = 2048.0 * fetch[k][0] / fetch[k][1]
        if k in write and write[k][1]:
            e["launches_write_pass"] = write[k][1]
            e["write_bytes_per_launch_raw"] = 1024.0 * write[k][0] / write[k][1]
    json.dump(res, open(os.path.join(dst, f"{rnd}_pmc_traffic.json"), "w"), indent=1)
    for k, e in res.items():
        print(k, {a: round(b) for a, b in e.items() if isinstance(b, float)})
    for f in glob.glob(os.path.join(out_root, "prof_3launch", "**", "*kernel_stats*.csv"), recursive=True)[:1]:
        shutil.copy(f, os.path.join(dst, f"{rnd}_kernel_stats_assembly_in_its_own_launch.csv"))
    for f in glob.glob(os.path.join(out_root, "prof_full", "**", "*kernel_stats*.csv"), recursive=True)[:1]:
        shutil.copy(f, os.path.join(dst, f"{rnd}_full_iteration_kernel_stats.csv"))
    # SQ counters per kernel and launch (pmc_sq* passes)
    lines = []
    for d in ("pmc_sq1", "pmc_sq3"):
        acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
        for f in glob.glob(os.path.join(out_root, d, "**", "*counter_collection*.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1]
                a = acc[k][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
        for k, cs in acc.items():
            if "rocclr" not in k:
                lines.append(f"{d} {k} " + str({c: round(v[0] / v[1]) for c, v in sorted(cs.items())}))
    open(os.path.join(dst, f"{rnd}_pmc_sq_summary.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
