"""Is the +-3 % between bench processes on one box a property of the process (where its buffers landed) or of the moment
(clocks)?  The bench workload, step by step (one step = 31 frame indices here), in one process; run it several times.
usage (GPU box): python tools/probes/step_times.py"""
import os
import sys
import time

import torch

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "h263-rs_amd"))
import bench  # noqa: E402
import h263mi  # noqa: E402

torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
N, GOP = 64, bench.GOP
wl = bench.Workload(h263mi, N, GOP, 0, 0, stream)
batch = h263mi.Batch(N, bench.W, bench.H, 0, stream, pipeline_post=True)
rgba = h263mi.DeviceBuffer(N * bench.RGBA_BYTES, 0)
bench.run_frames(batch, wl, rgba, GOP, True)
batch.sync()
out = []
for step in range(60):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.run_frames(batch, wl, rgba, GOP, True)
    batch.sync()
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / GOP * 1e3)
print("ms per frame index, 60 steps of one GOP: min %.4f median %.4f max %.4f" % (min(out), sorted(out)[30], max(out)))
print(" ".join("%.3f" % v for v in out))
