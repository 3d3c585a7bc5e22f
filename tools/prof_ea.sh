#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r04_k_pmc_ea; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--gops-per-step 1 --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-parity-gate"
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $GROUP --output-format csv -d $OUT/g$i -- python3 $R/bench.py $ARGS > $OUT/g$i.log 2>&1
  echo "group $i ($GROUP): rc=$?" >> $OUT/summary.txt
done <<'GROUPS'
TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum GRBM_GUI_ACTIVE
TCC_BUSY_sum TCC_CYCLE_sum TCC_TAG_STALL_sum
TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum
TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum
TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum
GROUPS
cd $R
python3 tools/pmc_summary.py $OUT > $OUT/summary_table.txt 2>&1
cat $OUT/summary.txt; grep -A40 "k_frame" $OUT/summary_table.txt | head -40
