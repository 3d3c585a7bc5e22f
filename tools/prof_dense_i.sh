#!/bin/bash
# counters of k_frame on dense I pictures alone (tools/probes/dense_i_only.py); usage (GPU box): bash tools/prof_dense_i.sh <tag>
set -u
TAG=${1:-dense_i}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $GROUP --output-format csv -d $OUT/g$i -- python3 $R/tools/probes/dense_i_only.py 12 > $OUT/g$i.log 2>&1
  echo "group $i ($GROUP): rc=$?" >> $OUT/summary.txt
done <<'GROUPS'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU
VALUBusy SALUBusy
TA_BUSY_avr TA_TA_BUSY_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
GROUPS
cd $R
python3 tools/pmc_summary.py $OUT > $OUT/summary_table.txt 2>&1
cat $OUT/summary.txt $OUT/summary_table.txt
