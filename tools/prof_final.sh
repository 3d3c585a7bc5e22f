#!/bin/bash
# Evidence for profiles/ from ONE lease of one GPU box (profiles and bench lines from different
# boxes cannot be compared; every file is stamped with this box's measured copy ceiling):
#   box.txt                      the box's copy / read / write ceilings (tools/probes/ceiling.hip, the tuned shapes)
#   bench.json                   the default `python bench.py` line (unprofiled), which reads traffic.json
#   bench_under_rocprof.json + trace/   rocprofv3 --kernel-trace --stats of the same command
#   traffic.json                 FETCH_SIZE / WRITE_SIZE PMC passes (counters only, separate runs)
#   pmc_summary.txt              SQ / TCC / TCP counters of k_frame (tools/prof_pmc.sh groups)
#   e2e_kernel_stats.csv         rocprofv3 kernel stats of the end-to-end path alone (tools/probes/e2e_trace.py)
# usage (GPU box, repo root): bash tools/prof_final.sh r03_x      -> gpurun_out/final_r03_x/
set -u
TAG=${1:-rXX}
R=$PWD; OUT=$R/gpurun_out/final_$TAG; mkdir -p $OUT
PROF_ARGS="--gops-per-step 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-parity-gate"
cd /tmp && export TMPDIR=/tmp
# ---- the box
( [ -x $R/build/ceiling ] && timeout 200 $R/build/ceiling 1024 | grep "BEST\|hipMemcpy" ) > $OUT/box.txt 2>&1
STAMP="box: $(grep 'BEST copy' $OUT/box.txt | awk '{print $3" GB/s copy ceiling"}'), $(hostname), $(date -u +%Y-%m-%dT%H:%MZ)"
echo "$STAMP" >> $OUT/box.txt
echo "== $STAMP"
# ---- kernel trace of the default bench command
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-extra > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
# ---- HBM traffic: two counter-only passes
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py $PROF_ARGS > /dev/null 2> $OUT/pmc_$C.log
done
cd $R
python3 - "$OUT" "$TAG" "$STAMP" <<'PY'
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.getcwd())
import bench
out, tag, stamp = sys.argv[1], sys.argv[2], sys.argv[3]
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
    if not f or not w:
        continue
    fetch = 2.0 * 1024 * sum(f) / len(f); write = 1024.0 * sum(w) / len(w)
    traffic[k] = {"fetch_bytes_per_launch": int(fetch), "write_bytes_per_launch": int(write),
                  "hbm_bytes_per_launch": int(fetch + write), "launches_sampled": len(f)}
json.dump({"tag": tag, "box": stamp, "kernel_source_hash": bench.kernel_source_hash(),
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --gops-per-step 1 --steps 2 "
                   "--warmup 1` (93 launches per kernel: 3 I + 90 P pictures x 64 streams); FETCH_SIZE x2 (gfx950 "
                   "calibration), KiB -> bytes",
           "kernels": traffic}, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(traffic))
PY
# the bench line below reports this traffic: same kernel sources, same box
cp $OUT/traffic.json profiles/traffic_latest.json
# ---- the default bench line, unprofiled (what the driver runs)
timeout 1200 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
# ---- SQ / TCC / TCP counters of the launch
bash tools/prof_pmc.sh final_${TAG}_pmc > /dev/null 2>&1
cp gpurun_out/final_${TAG}_pmc/summary_table.txt $OUT/pmc_summary.txt 2>/dev/null
# ---- the end-to-end path alone: kernel stats (GPU time per 64-picture call)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e2e_trace -- python3 $R/tools/probes/e2e_trace.py > $OUT/e2e.json 2> $OUT/e2e.log
cd $R
for f in $OUT/trace/*/*kernel_stats.csv; do cp $f $OUT/bench_kernel_stats.csv; done 2>/dev/null
for f in $OUT/e2e_trace/*/*kernel_stats.csv; do cp $f $OUT/e2e_kernel_stats.csv; done 2>/dev/null
for f in $OUT/bench_kernel_stats.csv $OUT/e2e_kernel_stats.csv $OUT/pmc_summary.txt; do [ -f $f ] && sed -i "1i # $STAMP" $f; done
tail -1 $OUT/bench.json | cut -c1-600
head -6 $OUT/bench_kernel_stats.csv
head -6 $OUT/e2e_kernel_stats.csv
