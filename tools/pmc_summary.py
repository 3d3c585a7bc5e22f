#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel from the per-dispatch CSVs of tools/prof_pmc.sh / prof_final.sh.

The profiled command runs whole GOPs of 31 frame indices, so dispatch i of a kernel works on frame index i % 31:
index 0 is the I picture, the other 30 are P pictures.  Both means are printed."""
import csv
import glob
import os
import sys
from collections import defaultdict

GOP = 31
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for path in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
    with open(path) as f:
        rows = [r for r in csv.DictReader(f)]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    for row in rows:
        name = row.get("Kernel_Name", "").split("(")[0]
        if "k_recon" not in name and "k_post" not in name and "k_frame" not in name:
            continue
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name in sorted(acc):
    print(name)
    for c in sorted(acc[name]):
        v = acc[name][c]
        p = [x for i, x in enumerate(v) if i % GOP] or v
        ip = [x for i, x in enumerate(v) if i % GOP == 0] or v
        print("  %-28s n=%4d mean_P=%16.1f mean_I=%16.1f max=%16.1f" % (c, len(v), sum(p) / len(p), sum(ip) / len(ip), max(v)))
