#!/bin/bash
# SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_WAVE_CYCLES of the two kernels for two builds
R=$PWD; cd /tmp; export TMPDIR=/tmp
for L in "$@"; do
  rm -rf $R/gpurun_out/pmc_ab; mkdir -p $R/gpurun_out/pmc_ab
  H263MI_LIB=$R/$L timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_ab/g1 -- python3 $R/bench.py --steps 8 --warmup 31 --no-cpu-baseline --no-extra > /dev/null 2>&1
  echo "== $L"; (cd $R && python3 tools/pmc_summary.py gpurun_out/pmc_ab | grep -E "k_post|k_recon|INSTS_VALU|INSTS_SALU")
done
