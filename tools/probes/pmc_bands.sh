#!/bin/bash
# HBM traffic of k_frame for 8 / 4 / 2 bands per picture (profiles/r02_k_traffic_by_bands.txt): separate rocprofv3 --pmc
# FETCH_SIZE / WRITE_SIZE passes over a short bench run per build.  The builds are variants of the library,
#   hipcc <flags of h263-rs_amd/Makefile> -DH263MI_FRAME_BANDS=<8|4|2> ... -o h263-rs_amd/variants/lib_b<8|4|2>.so
# usage (GPU box, repo root): bash tools/probes/pmc_bands.sh
R=$PWD; OUT=$R/gpurun_out/s2/pmc_bands; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for L in b8 b4 b2; do
  for C in FETCH_SIZE WRITE_SIZE; do
    H263MI_LIB=$R/h263-rs_amd/variants/lib_$L.so timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/${L}_$C -- python3 $R/bench.py --gops-per-step 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-parity-gate > /dev/null 2> $OUT/${L}_$C.log
  done
done
cd $R
python3 - <<'PY'
import csv, glob, os
out='gpurun_out/s2/pmc_bands'
for L in ('b8','b4','b2'):
    res={}
    for C in ('FETCH_SIZE','WRITE_SIZE'):
        vals=[]
        for path in glob.glob(os.path.join(out, L+'_'+C, '**', '*counter_collection.csv'), recursive=True):
            for row in csv.DictReader(open(path)):
                if 'k_frame' in row['Kernel_Name'] and row['Counter_Name']==C:
                    vals.append(float(row['Counter_Value']))
        res[C]=(sum(vals)/max(len(vals),1), len(vals))
    print(L, 'fetch MB %.1f (n=%d)' % (2*1024*res['FETCH_SIZE'][0]/1e6, res['FETCH_SIZE'][1]), 'write MB %.1f' % (1024*res['WRITE_SIZE'][0]/1e6))
PY
