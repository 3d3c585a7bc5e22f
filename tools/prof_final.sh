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
# (the probe is built here, on the box: build/ does not travel with the snapshot any more)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/probes/ceiling.hip -o /tmp/ceiling 2> /dev/null
( [ -x /tmp/ceiling ] && timeout 200 /tmp/ceiling 1024 | grep "BEST\|hipMemcpy" ) > $OUT/box.txt 2>&1
STAMP="box: $(grep 'BEST copy' $OUT/box.txt | awk '{print $3" GB/s copy ceiling"}'), $(hostname), $(date -u +%Y-%m-%dT%H:%MZ)"
echo "$STAMP" >> $OUT/box.txt
echo "== $STAMP"
# ---- kernel trace of the default bench command
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-extra > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
# ---- HBM traffic: two counter-only passes
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py $PROF_ARGS > /dev/null 2> $OUT/pmc_$C.log
done
# ... and the request counters behind them, by size (round 4: tools/fetch_calib.sh shows that EVERY read request of the L2s
# is 128 bytes on gfx950, for k_frame's gathers and strip loads as for streaming reads -- FETCH_SIZE, which prices a request
# at 64 bytes, times 2 is exact; TCC_EA0_RDREQ_DRAM_32B x 32 bytes agrees to 0.2 %)
timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_32B_sum --output-format csv -d $OUT/pmc_RDREQ -- python3 $R/bench.py $PROF_ARGS > /dev/null 2> $OUT/pmc_RDREQ.log
timeout 600 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum --output-format csv -d $OUT/pmc_WRREQ -- python3 $R/bench.py $PROF_ARGS > /dev/null 2> $OUT/pmc_WRREQ.log
# ... and what the vector ALUs did (round 5: the launch is bound by their issue slots, bench.py reports it as roofline.valu)
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_VALU -- python3 $R/bench.py $PROF_ARGS > /dev/null 2> $OUT/pmc_VALU.log
timeout 600 rocprofv3 --pmc VALUBusy --output-format csv -d $OUT/pmc_VALUBUSY -- python3 $R/bench.py $PROF_ARGS > /dev/null 2> $OUT/pmc_VALUBUSY.log
cd $R
python3 - "$OUT" "$TAG" "$STAMP" <<'PY'
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.getcwd())
import bench
out, tag, stamp = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE", "RDREQ", "WRREQ", "VALU", "VALUBUSY"):
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
    mean = lambda name: (sum(v[name]) / len(v[name])) if v.get(name) else None
    rd, rd128, rd_dram = mean("TCC_EA0_RDREQ_sum"), mean("TCC_EA0_RDREQ_128B_sum"), mean("TCC_EA0_RDREQ_DRAM_32B_sum")
    wr, wr64, wr_dram = mean("TCC_EA0_WRREQ_sum"), mean("TCC_EA0_WRREQ_64B_sum"), mean("TCC_EA0_WRREQ_WRITE_DRAM_32B_sum")
    if rd and rd_dram and wr_dram:
        # the exact figure: the DRAM request counters in 32-byte units (k_frame: 8.5 % of its read requests are 64 bytes,
        # so FETCH_SIZE x 2 overstates its reads by 4 %)
        traffic[k]["fetch_bytes_per_launch_fetch_size_x2"] = traffic[k]["fetch_bytes_per_launch"]
        traffic[k]["write_bytes_per_launch_write_size"] = traffic[k]["write_bytes_per_launch"]
        traffic[k]["fetch_bytes_per_launch"] = int(32 * rd_dram)
        traffic[k]["write_bytes_per_launch"] = int(32 * wr_dram)
        traffic[k]["hbm_bytes_per_launch"] = int(32 * rd_dram + 32 * wr_dram)
    if rd:
        traffic[k]["cross_check"] = {
            "read_requests": int(rd), "of_which_128_bytes": int(rd128 or 0), "read_bytes_dram_32B_units": int(32 * (rd_dram or 0)),
            "write_requests": int(wr or 0), "of_which_64_bytes": int(wr64 or 0), "write_bytes_dram_32B_units": int(32 * (wr_dram or 0)),
            "what": "TCC_EA0_* request counters of the same command: reads are 128-byte requests (FETCH_SIZE prices them at 64: "
                    "the x2), the DRAM counters are in 32-byte units"}
valu = {}
for k, v in acc.items():
    if v.get("SQ_INSTS_VALU") and v.get("VALUBusy"):
        m = lambda name: sum(v[name]) / len(v[name])
        valu[k] = {"insts_per_launch": int(m("SQ_INSTS_VALU")), "busy_frac": round(m("VALUBusy") / 100.0, 4),
                   "gui_active_cycles": int(m("GRBM_GUI_ACTIVE")) if v.get("GRBM_GUI_ACTIVE") else None,
                   "launches_sampled": len(v["SQ_INSTS_VALU"]),
                   "what": "means over every launch of the profiled GOPs (I and P pictures in the bench's mix): SQ_INSTS_VALU = "
                           "vector instructions issued (wave64: x 64 lane operations), VALUBusy = share of the launch's cycles the "
                           "vector ALUs were busy"}
json.dump({"tag": tag, "box": stamp, "kernel_source_hash": bench.kernel_source_hash(), "valu": valu,
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --gops-per-step 1 --steps 2 "
                   "--warmup 1` (93 launches per kernel: 3 I + 90 P pictures x 64 streams); FETCH_SIZE x2 (gfx950: every L2 read "
                   "request is 128 bytes and FETCH_SIZE prices it at 64 -- checked on k_frame's own access shapes, 12-byte gathers "
                   "and skewed strip loads over a known byte count, profiles/r04_e_fetch_calibration.json), KiB -> bytes",
           "fetch_size_factor": 2.0,
           "hbm_bytes_source": "TCC_EA0_RDREQ_DRAM_32B_sum / TCC_EA0_WRREQ_WRITE_DRAM_32B_sum x 32 bytes where collected (exact), "
                               "else FETCH_SIZE x 2 + WRITE_SIZE",
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
