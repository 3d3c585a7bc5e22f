#!/bin/bash
# A/B builds of libh263mi.so on the same GPU box: interleaved bench runs, per-kernel averages.
# usage: ROUNDS=3 bash tools/ab.sh <libA.so> <libB.so> [<libC.so> ...]
# (ab_libs/ is in .gpurunignore: stale variants do not travel with every call.  Build the variants ON the box -- tools/build_variant.sh,
#  hipcc is there -- or take the line out of .gpurunignore for the one call that needs libraries built here, e.g. those of earlier rounds:
#  `git archive <round commit> h263-rs_amd include | tar -x -C /tmp/src_rN`, build, copy to ab_libs/; tools/ab_inproc.py runs them beside HEAD.)
N=${ROUNDS:-3}
for i in $(seq 1 $N); do
  for L in "$@"; do
    H263MI_LIB=$PWD/$L timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-parity-gate --no-e2e 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; e=d.get('extra',{}).get('config2_dense_iframe',{})
print('$L', 'ms/frame', d['ms_per_frame_index'], 'recon', k['k_recon']['avg_ms'], 'post', k['k_post']['avg_ms'], 'denseI recon', e.get('k_recon_avg_ms'), 'post', e.get('k_post_avg_ms'), 'frame', e.get('k_frame_avg_ms'))"
  done
done
