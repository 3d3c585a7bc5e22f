#!/bin/bash
# Evidence for profiles/: rocprofv3 kernel stats of the bench workload, then the two PMC passes (FETCH_SIZE,
# WRITE_SIZE; counters only, separate runs) that feed roofline.traffic, then the default bench line (which reads them).
# usage (GPU box, repo root): bash tools/prof_final.sh r02_a      -> gpurun_out/final_r02_a/
set -u
TAG=${1:-rXX}
R=$PWD; OUT=$R/gpurun_out/final_$TAG; mkdir -p $OUT
PROF_ARGS="--gops-per-step 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-parity-gate"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-extra > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py $PROF_ARGS > /dev/null 2> $OUT/pmc_$C.log
done
cd $R
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.getcwd())
import bench
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for path in glob.glob(os.path.join(out, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
        rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r.get("Dispatch_Id", 0)))
        for row in rows:
            name = row["Kernel_Name"].split("(")[0].split("::")[-1]
            if name in ("k_recon", "k_post", "k_frame"):
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
traffic = {}
for k, v in acc.items():
    # every dispatch of the profiled command (3 GOPs of 31 frame indices: 1 I + 30 P each) -- the same mix as the
    # bench's timed region; counters are in KiB.
    # gfx950: FETCH_SIZE reports exactly half of the bytes read (profiles/r01_copy_bw.txt) -> doubled.
    f, w = v["FETCH_SIZE"], v["WRITE_SIZE"]
    fetch = 2.0 * 1024 * sum(f) / len(f); write = 1024.0 * sum(w) / len(w)
    traffic[k] = {"fetch_bytes_per_launch": int(fetch), "write_bytes_per_launch": int(write),
                  "hbm_bytes_per_launch": int(fetch + write), "launches_sampled": len(f)}
json.dump({"tag": tag, "kernel_source_hash": bench.kernel_source_hash(),
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --gops-per-step 1 --steps 2 "
                   "--warmup 1` (93 launches per kernel: 3 I + 90 P pictures x 64 streams); FETCH_SIZE x2 (gfx950 "
                   "calibration), KiB -> bytes",
           "kernels": traffic}, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(traffic))
PY
# the bench line below reports this traffic: same kernel sources, same box (the file is also copied to profiles/ by hand)
cp $OUT/traffic.json profiles/traffic_latest.json
timeout 1200 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-900
ls $OUT/trace/*/ | head
