#!/usr/bin/env python3
"""Per-phase dynamic instruction counts of k_frame from the counting builds of tools/phase_insts.sh (see there)."""
import csv
import glob
import os
import sys
from collections import defaultdict

GOP, N_PICTURES, PIXELS = 31, 64, 1920 * 1080
root = sys.argv[1]
COUNTERS = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"]


def means(variant):
    acc = defaultdict(list)
    for path in sorted(glob.glob(os.path.join(root, variant, "**", "*counter_collection.csv"), recursive=True)):
        rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r.get("Dispatch_Id", 0)))
        for row in rows:
            if "k_frame" in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for c, v in acc.items():
        p = [x for i, x in enumerate(v) if i % GOP] or v            # dispatch i works on frame index i % 31: 0 = the I picture
        k = [x for i, x in enumerate(v) if i % GOP == 0] or v
        out[c] = (sum(p) / len(p), sum(k) / len(k), len(v))
    return out


VARIANTS = ("full", "recon9", "recon1", "recon2", "recon3", "recon4", "recon5", "post9", "post1", "post2", "post3")
m = {v: means(v) for v in VARIANTS}
if not m["full"]:
    sys.exit("no counters for the full build under " + root)
steps = [
    ("recon: records -> LDS, mark, descriptors, compaction (static waves: their copy too)", "recon9", "recon1"),
    ("recon: addresses + every load of the wave (reference rows, coefficient words)", "recon1", "recon2"),
    ("recon: first IDCT round, row pass (events -> rows, classes, dequantiser, f32 products)", "recon2", "recon3"),
    ("recon: prediction into the strip (alignment, half-pel filter, border taps)", "recon3", "recon4"),
    ("recon: column pass of round 0 + the remaining rounds (row + column pass each)", "recon4", "recon5"),
    ("recon: strip -> frame", "recon5", "full"),
    ("post: strip addresses, loads, commit to LDS", "post9", "post1"),
    ("post: deblock, horizontal edges (packed quartets, 3 planes)", "post1", "post2"),
    ("post: deblock, vertical edges", "post2", "post3"),
    ("post: BT.601 (per pixel 1 multiply + 3 adds + 2 shift-saturate-packs) + RGBA stores", "post3", "full"),
]


def val(variant, counter, which):
    return m[variant][counter][which] if m.get(variant) and counter in m[variant] else float("nan")


for which, label in ((0, "P pictures (mean of the launches of frame indices 1..30)"), (1, "the GOP's I picture (frame index 0)")):
    full = val("full", "SQ_INSTS_VALU", which)
    print("# k_frame, %s: %d pictures per launch; full build: %.2f M vector instructions = %.2f lane operations per output pixel"
          % (label, N_PICTURES, full / 1e6, full * 64 / (N_PICTURES * PIXELS)))
    print("%-90s %8s %7s %10s %8s %7s %8s %8s" % ("phase (difference of two counting builds)", "VALU M", "share", "lane-op/px",
                                                 "SALU M", "LDS M", "VMEM rd", "VMEM wr"))
    total = 0.0
    for name, lo, hi in steps:
        d = [val(hi, c, which) - val(lo, c, which) for c in COUNTERS]
        total += d[0]
        print("%-90s %8.2f %6.1f%% %10.2f %8.2f %7.2f %8.3f %8.3f" % (name, d[0] / 1e6, 100 * d[0] / full, d[0] * 64 / (N_PICTURES * PIXELS),
                                                                     d[1] / 1e6, d[2] / 1e6, d[3] / 1e6, d[4] / 1e6))
    # what neither series holds: the dispatch prologue of every wave (which kind am I, where) = recon9 + post9 - full
    pro = val("recon9", "SQ_INSTS_VALU", which) + val("post9", "SQ_INSTS_VALU", which) - full
    print("%-90s %8.2f %6.1f%% %10.2f" % ("k_frame's dispatch prologue of all waves (recon9 + post9 - full)", pro / 1e6, 100 * pro / full,
                                        pro * 64 / (N_PICTURES * PIXELS)))
    print("# rows + prologue = %.2f M of %.2f M" % ((total + pro) / 1e6, full / 1e6))
    print()
