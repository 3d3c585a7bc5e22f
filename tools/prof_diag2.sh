#!/bin/bash
# Second diagnosis set of k_frame (counters only, one group per rocprofv3 run): who keeps a compute unit from being
# full (SPI resource-allocation stalls), how full it is (SQ_LEVEL_WAVES / SQ_BUSY_CU_CYCLES), the instruction cache
# and the scalar data cache.  usage (GPU box, repo root): bash tools/prof_diag2.sh <tag>
set -u
TAG=${1:-diag2}; shift || true
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
  rc=$?
  echo "group $i ($GROUP): rc=$rc" >> $OUT/summary.txt
  [ $rc -ne 0 ] && [ $rc -ne 1 ] && { echo "stopping after a failed group" >> $OUT/summary.txt; break; }
done <<'GROUPS'
SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN SPI_CSN_BUSY SPI_CSN_WINDOW_VALID
SPI_RA_LDS_CU_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_SGPR_SIMD_FULL_CSN
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ SQC_TC_STALL
SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_BUSY_CYCLES
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM
GROUPS
cd $R
python3 tools/pmc_summary.py $OUT > $OUT/summary_table.txt 2>&1
cat $OUT/summary.txt $OUT/summary_table.txt
