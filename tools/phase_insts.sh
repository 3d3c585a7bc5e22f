#!/bin/bash
# What every phase of k_frame EXECUTES (dynamic instruction counts), by differencing counting builds.
#
# tools/isa_mix.py is a static table: a loop body counts once, both arms of a branch count.  Here the launch's own
# counters say what ran: a series of builds in which the reconstruction waves (H263MI_STOP_RECON=1..5) or the
# post-processing waves (H263MI_STOP_POST=1..3) stop behind a phase (kernels.hip; results wrong by construction), each
# profiled with `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR` over the bench's
# GOPs (counters only, one pass per build).  The difference between two consecutive builds is the phase between their stops.
# usage (GPU box, repo root; the variants built beforehand with tools/build_variant.sh stop_<name> -DH263MI_STOP_...=n):
#   bash tools/phase_insts.sh r05_x      -> gpurun_out/phase_insts_r05_x/table.txt
set -u
TAG=${1:-rXX}
R=$PWD; OUT=$R/gpurun_out/phase_insts_$TAG; mkdir -p $OUT
PROF_ARGS="--gops-per-step 1 --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-parity-gate"
cd /tmp && export TMPDIR=/tmp
for V in full recon9 recon1 recon2 recon3 recon4 recon5 post9 post1 post2 post3; do
  if [ $V = full ]; then LIB=$R/h263-rs_amd/libh263mi.so; else LIB=$R/ab_libs/lib_stop_$V.so; fi
  [ -f $LIB ] || { echo "missing $LIB"; continue; }
  H263MI_LIB=$LIB timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
      --output-format csv -d $OUT/$V -- python3 $R/bench.py $PROF_ARGS > $OUT/$V.log 2>&1
  echo "$V rc=$?"
done
cd $R
python3 tools/phase_insts.py $OUT > $OUT/table.txt 2>&1
cat $OUT/table.txt
