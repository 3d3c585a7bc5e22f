"""Does the speed of a batch depend on where its frame store lies relative to the RGBA surface in the address space?
batch_order.py saw frame stores above a 4 GB-aligned address (the RGBA surface below it) run 8-10 % faster than those
below it.  Here: 5 batches, the RGBA surface, then 14 more batches (401 MB each: they walk down through more than one
4 GB window), each timed on the same workload and surface.
usage (GPU box): python tools/probes/batch_va_window.py"""
import os
import sys
import time

os.environ["H263MI_TRACE_ALLOC"] = "1"      # the library prints each frame store's address to stderr: read beside the timings

import torch

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "h263-rs_amd"))
import bench  # noqa: E402
import h263mi  # noqa: E402

L = h263mi.lib()
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
N, GOP = 64, bench.GOP
wl = bench.Workload(h263mi, N, GOP, 0, 0, stream, events=True)


def mk():
    return h263mi.Batch(N, bench.W, bench.H, 0, stream, pipeline_post=True), 0


batches = [mk() for _ in range(5)]
rgba = h263mi.DeviceBuffer(N * bench.RGBA_BYTES, 0)
print("rgba surface at %#x" % rgba.ptr.value, flush=True)
batches += [mk() for _ in range(14)]


def run(b):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.run_frames(b, wl, rgba, GOP * 2, True)
    b.sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (GOP * 2) * 1e3


for b, _ in batches:
    run(b)
res = [[] for _ in batches]
for rnd in range(3):                                   # interleaved: a drift of the clock hits every batch alike
    for k, (b, _) in enumerate(batches):
        res[k].append(run(b))
for k, ts in enumerate(res):
    print("batch %2d (%s the surface)  %.4f ms (spread %.4f)" % (k, "before" if k < 5 else "after", sum(ts) / 3, max(ts) - min(ts)), flush=True)
