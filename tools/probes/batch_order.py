"""Does it matter WHEN a batch (= its frame store, 2 x 200 MB for 64 x 1080p) is allocated relative to the records,
events and RGBA buffers?  tools/ab_inproc.py saw the batch created last run the P workload 5.7-5.9 % slower than its
siblings on two boxes, whichever library it belonged to.  One library, one workload, batches created at different points
of the allocation sequence, timed in turn.
usage (GPU box): python tools/probes/batch_order.py"""
import os
import sys
import time

os.environ["H263MI_TRACE_ALLOC"] = "1"

import torch

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "h263-rs_amd"))
import bench  # noqa: E402
import h263mi  # noqa: E402

torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
N, GOP = 64, bench.GOP


def mk():
    return h263mi.Batch(N, bench.W, bench.H, 0, stream, pipeline_post=True)


batches = [("before everything", mk()), ("second, before everything", mk())]
wl = bench.Workload(h263mi, N, GOP, 0, 0, stream, events=True)
print("workload: records of frame 1 at %#x" % wl.frames[1]["mbs"].ptr.value, flush=True)
batches.append(("after the workload", mk()))
rgba = h263mi.DeviceBuffer(N * bench.RGBA_BYTES, 0)
print("rgba surface: %#x .. +%d" % (rgba.ptr.value, N * bench.RGBA_BYTES), flush=True)
batches.append(("after the RGBA surface", mk()))
spacer = h263mi.DeviceBuffer(777 << 20, 0)
batches.append(("after a 777 MB spacer", mk()))
batches.append(("last", mk()))


def run(b):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.run_frames(b, wl, rgba, GOP * 3, True)
    b.sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (GOP * 3) * 1e3


for name, b in batches:
    run(b)
res = {name: [] for name, _ in batches}
for rnd in range(4):
    for name, b in batches:
        res[name].append(run(b))
base = min(sum(v) / len(v) for v in res.values())
for name, _ in batches:
    v = res[name]
    print("batch created %-28s %.4f ms per frame index (%+.1f %% against the best, spread %.4f)" % (
        name, sum(v) / len(v), 100 * (sum(v) / len(v) / base - 1), max(v) - min(v)), flush=True)
