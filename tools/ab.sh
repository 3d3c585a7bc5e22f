#!/bin/bash
# A/B two builds of libh263mi.so in ONE process sequence on the same GPU box: interleaved runs, kernel averages.
# usage: bash tools/ab.sh <libA.so> <libB.so> [rounds]
A=$1; B=$2; N=${3:-3}
for i in $(seq 1 $N); do
  for L in $A $B; do
    H263MI_LIB=$PWD/$L timeout 300 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('$L', d['ms_per_step'], 'recon', k['k_recon']['avg_ms'], 'post', k['k_post']['avg_ms'])"
  done
done
