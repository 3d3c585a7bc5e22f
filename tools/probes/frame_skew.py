"""Speed of the frame-pipelined launch against the distance between the two frame sets of the batch (H263MI_FRAME_SKEW:
the second set starts n * frame_bytes + skew behind the first one, in one allocation).  Everything else stays where it
is inside the process, so only the skew changes between the rows.
usage (GPU box): python tools/probes/frame_skew.py"""
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
rgba = h263mi.DeviceBuffer(N * bench.RGBA_BYTES, 0)
os.environ["H263MI_TRACE_ALLOC"] = "1"
print("rgba at %#x; frame 1: records %#x coefficients %#x" % (rgba.ptr.value, wl.frames[1]["mbs"].ptr.value, wl.frames[1]["co"].ptr.value), flush=True)
skews = [0, 0, 0, 1 << 21, 1 << 21, 1 << 22, 0, 3 << 21, 0, 0]
keep = []
for rep in range(2):
    for skew in skews:
        os.environ["H263MI_FRAME_SKEW"] = str(skew)
        batch = h263mi.Batch(N, bench.W, bench.H, 0, stream, pipeline_post=True)
        bench.run_frames(batch, wl, rgba, GOP, True)
        batch.sync()
        out = []
        for step in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            bench.run_frames(batch, wl, rgba, GOP, True)
            batch.sync()
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / GOP * 1e3)
        print("skew %8d: ms per frame index min %.4f median %.4f" % (skew, min(out), sorted(out)[2]), flush=True)
        batch.close()
        if rep == 1 and skew == 0:
            keep.append(h263mi.DeviceBuffer(64 << 20, 0))       # shifts where the next frame store lands
