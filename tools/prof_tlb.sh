#!/bin/bash
# The L1 address-translation unit under k_frame: requests, misses, and the cycles it stalls the L1 because its in-flight
# limit is reached (counters only, one group per rocprofv3 run).  usage (GPU box, repo root): bash tools/prof_tlb.sh <tag>
set -u
TAG=${1:-tlb}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--gops-per-step 1 --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-parity-gate"
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $GROUP --output-format csv -d $OUT/g$i -- python3 $R/bench.py $ARGS > $OUT/g$i.log 2>&1
  echo "group $i ($GROUP): rc=$?" >> $OUT/summary.txt
done <<'GROUPS'
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum
TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum GRBM_GUI_ACTIVE
GROUPS
cd $R
python3 tools/pmc_summary.py $OUT > $OUT/summary_table.txt 2>&1
grep -A10 "k_frame" $OUT/summary_table.txt | cut -c1-130
