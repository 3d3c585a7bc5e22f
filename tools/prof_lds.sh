#!/bin/bash
# LDS bank conflicts of k_frame (counters only).  usage (GPU box, repo root): bash tools/prof_lds.sh <tag>
set -u
TAG=${1:-lds}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/g1 -- python3 $R/bench.py --gops-per-step 1 --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-parity-gate > $OUT/g1.log 2>&1
cd $R
python3 tools/pmc_summary.py $OUT > $OUT/summary_table.txt 2>&1
grep -A6 "k_frame" $OUT/summary_table.txt | cut -c1-120
