"""What do the HIP events around every launch (roofline.avg_launch_ms is measured with them, inside the timed region)
cost?  The bench workload with and without h263mi_batch_timing_begin, same process, interleaved.
usage (GPU box): python tools/probes/timing_overhead.py"""
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
N, GOP, GOPS = 64, bench.GOP, 16
wl = bench.Workload(h263mi, N, GOP, 0, 0, stream)
batch = h263mi.Batch(N, bench.W, bench.H, 0, stream, pipeline_post=True)
rgba = h263mi.DeviceBuffer(N * bench.RGBA_BYTES, 0)
batch.timing_reserve(2 * GOP * GOPS)
bench.run_frames(batch, wl, rgba, GOP, True)
batch.sync()
for rnd in range(3):
    for timed in (False, True):
        if timed:
            batch.timing_begin()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.run_frames(batch, wl, rgba, GOP * GOPS, True)
        batch.sync()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        extra = ""
        if timed:
            kt = batch.timing_end()
            extra = ", k_frame by events %.4f ms" % (kt.frame_ms / max(kt.frame_launches, 1))
        print("events %-5s: %.4f ms per frame index%s" % (timed, dt / (GOP * GOPS) * 1e3, extra), flush=True)
