#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel from the per-dispatch CSVs of tools/prof_pmc.sh."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, "g*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "").split("(")[0]
            if "k_recon" not in name and "k_post" not in name:
                continue
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name in sorted(acc):
    print(name)
    for c in sorted(acc[name]):
        v = acc[name][c]
        # the first 31 dispatches are warm-up (frame 0 = I picture); report the mean of the rest and of all
        tail = v[32:] if len(v) > 34 else v
        print("  %-28s n=%3d mean_all=%16.1f mean_timedP=%16.1f max=%16.1f" % (c, len(v), sum(v) / len(v), sum(tail) / len(tail), max(v)))
