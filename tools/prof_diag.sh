#!/bin/bash
# Memory-pipeline diagnosis of k_frame: where vector-memory instructions queue (SQ -> TA -> TCP -> TCC -> EA), address
# translation, request latencies.  Counters only, one group per rocprofv3 run.
# usage (GPU box, repo root): bash tools/prof_diag.sh <tag> [bench args...]
set -u
TAG=${1:-diag}; shift || true
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="${@:---gops-per-step 1 --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-parity-gate}"
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $GROUP --output-format csv -d $OUT/g$i -- python3 $R/bench.py $ARGS > $OUT/g$i.log 2>&1
  echo "group $i ($GROUP): rc=$?" >> $OUT/summary.txt
done <<'GROUPS'
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_TA_BUSY_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum
TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
TCC_TAG_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum TCC_BUSY_sum
TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_STREAMING_REQ_sum TCC_BYPASS_REQ_sum
TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum
TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum
TCC_CYCLE_sum TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TCP_GATE_EN1_sum
GROUPS
cd $R
python3 tools/pmc_summary.py $OUT > $OUT/summary_table.txt 2>&1
cat $OUT/summary.txt
grep -A 80 "k_frame" $OUT/summary_table.txt | grep -B1000 -m1 "k_post" | cut -c1-110
