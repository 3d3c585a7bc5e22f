#!/bin/bash
# A/B builds of libh263mi.so on the same GPU box: interleaved bench runs, per-kernel averages.
# usage: ROUNDS=3 bash tools/ab.sh <libA.so> <libB.so> [<libC.so> ...]
N=${ROUNDS:-3}
for i in $(seq 1 $N); do
  for L in "$@"; do
    H263MI_LIB=$PWD/$L timeout 300 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('$L', d['ms_per_step'], 'recon', k['k_recon']['avg_ms'], 'post', k['k_post']['avg_ms'])"
  done
done
