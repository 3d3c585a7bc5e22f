#!/bin/bash
# FETCH_SIZE per true HBM byte for k_frame's own access shapes (tools/probes/fetch_calib.hip); counters only.
# usage (GPU box, repo root): bash tools/fetch_calib.sh <tag>   -> gpurun_out/<tag>/fetch_calibration.json
set -u
TAG=${1:-fetch_calib}
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/probes/fetch_calib.hip -o /tmp/fetch_calib 2> $OUT/build.log || exit 1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -- /tmp/fetch_calib 2048 > $OUT/run.log 2>&1
# the request counters FETCH_SIZE is derived from, by request size (a pass of their own)
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/pmc2 -- /tmp/fetch_calib 2048 > $OUT/run2.log 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum --output-format csv -d $OUT/pmc3 -- /tmp/fetch_calib 2048 > $OUT/run3.log 2>&1
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
nbytes = int(open(os.path.join(out, "run.log")).read().split("bytes")[1].split()[0])
acc = collections.defaultdict(list)
for path in glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == "FETCH_SIZE":
            acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
res = {}
for k, v in sorted(acc.items()):
    v = v[1:] or v                                   # the first launch of a shape may find part of the memset in the caches
    kib = sum(v) / len(v)
    # what the strip shapes really need from HBM: every line their loads touch; the skew makes them touch the whole region
    res[k] = {"fetch_size_kib": kib, "true_bytes": nbytes, "bytes_per_counter_unit": nbytes / kib if kib else None,
              "factor_vs_kib": (nbytes / 1024.0) / kib if kib else None, "launches": len(v)}
json.dump({"buffer_bytes": nbytes, "what": "rocprofv3 --pmc FETCH_SIZE over tools/probes/fetch_calib.hip: every kernel reads each 64-byte "
           "line of a 2 GiB buffer exactly once in one access shape; factor_vs_kib = true KiB / FETCH_SIZE (2.0 = the guide's "
           "calibration for wide streaming reads)", "shapes": res}, open(os.path.join(out, "fetch_calibration.json"), "w"), indent=1)
req = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("pmc2", "pmc3"):
    for path in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            req[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in sorted(req.items()):
    if k not in res:
        continue
    m = {name: sum(v[1:] or v) / len(v[1:] or v) for name, v in c.items()}
    res[k]["requests"] = m
    n32, n64, n128, tot = (m.get("TCC_EA0_RDREQ_%s_sum" % s, 0.0) for s in ("32B", "64B", "128B")) if False else (
        m.get("TCC_EA0_RDREQ_32B_sum", 0.0), m.get("TCC_EA0_RDREQ_64B_sum", 0.0), m.get("TCC_EA0_RDREQ_128B_sum", 0.0), m.get("TCC_EA0_RDREQ_sum", 0.0))
    res[k]["bytes_by_request_size"] = 32 * n32 + 64 * n64 + 128 * n128
    res[k]["bytes_by_request_size_over_true"] = res[k]["bytes_by_request_size"] / nbytes
    res[k]["dram_32b_units_times_32_over_true"] = 32 * m.get("TCC_EA0_RDREQ_DRAM_32B_sum", 0.0) / nbytes
    print("%-22s RDREQ %.0f = 32B %.0f + 64B %.0f + 128B %.0f; sized bytes / true = %.3f; DRAM_32B x 32 / true = %.3f" % (
        k, tot, n32, n64, n128, res[k]["bytes_by_request_size_over_true"], res[k]["dram_32b_units_times_32_over_true"]))
json.dump({"buffer_bytes": nbytes, "shapes": res}, open(os.path.join(out, "fetch_calibration.json"), "w"), indent=1)
for k, r in res.items():
    print("%-22s FETCH_SIZE %12.0f KiB for %d bytes -> factor %.3f" % (k, r["fetch_size_kib"], nbytes, r["factor_vs_kib"] or 0))
PY
