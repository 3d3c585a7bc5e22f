#!/bin/bash
# Round-end evidence for profiles/: rocprofv3 kernel stats of the default bench command, then the two PMC
# passes (FETCH_SIZE, WRITE_SIZE; counters only) that feed roofline.traffic.  usage: bash tools/prof_final.sh r01
set -u
TAG=${1:-rXX}
R=$PWD; OUT=$R/gpurun_out/final_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 62 --warmup 31 --no-cpu-baseline --no-extra > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py --steps 62 --warmup 31 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/pmc_$C.log
done
cd $R
timeout 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for path in glob.glob(os.path.join(out, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].split("(")[0].split("::")[-1]
            if name in ("k_recon", "k_post"):
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
traffic = {}
for k, v in acc.items():
    # timed region = the dispatches after the 31 warm-up ones; counters are in KiB.
    # gfx950: FETCH_SIZE reports exactly half of the bytes read (profiles/r01_copy_bw.txt) -> doubled.
    f = v["FETCH_SIZE"][31:] or v["FETCH_SIZE"]; w = v["WRITE_SIZE"][31:] or v["WRITE_SIZE"]
    fetch = 2.0 * 1024 * sum(f) / len(f); write = 1024.0 * sum(w) / len(w)
    traffic[k] = {"fetch_bytes_per_launch": int(fetch), "write_bytes_per_launch": int(write),
                  "hbm_bytes_per_launch": int(fetch + write), "launches_sampled": len(f)}
json.dump({"tag": tag, "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 62 --warmup 31`; "
           "FETCH_SIZE x2 (gfx950 calibration), KiB -> bytes", "kernels": traffic}, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(traffic))
PY
cat $OUT/bench.json | tail -1 | cut -c1-600
ls $OUT/trace/*/ | head
