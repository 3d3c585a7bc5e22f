#!/usr/bin/env python3
"""How the host of the GPU box scales the CPU baseline: threads vs MP/s, plus the cgroup CPU quota."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "h263-rs_amd")]
import h263mi
from oracle import native_bench

for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
          "/sys/fs/cgroup/cpuset.cpus.effective", "/proc/loadavg"):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, "n/a", e)
print("affinity", len(os.sched_getaffinity(0)))
nb = native_bench.NativeOracle()
W, H = 1920, 1080
streams = [[h263mi.synth_picture_host(h263mi.SYNTH_I_MIXED if f == 0 else h263mi.SYNTH_P, W, H, 100 + s, f) for f in range(31)]
           for s in range(4)]
for t in (1, 2, 4, 8, 16, 32, 64, 128):
    secs = nb.run(W, H, streams, t, 1, 5)
    print("threads %3d  %.2f s  %.1f MP/s  per thread %.1f" % (t, secs, t * 31 * 2.0736 / secs, 31 * 2.0736 / secs), flush=True)
