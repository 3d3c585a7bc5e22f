"""Which buffer's placement moves the time per frame index (tools/probes/realloc_modes.py: 0.25 ... 0.29 ms between two
allocations of the whole workload)?  One process; between the rows only ONE kind of buffer is freed and allocated again
(behind a spacer of a new size, so that the driver hands out other memory): the RGBA output, the batch's frame store, or
the records + coefficients of the 31 frame indices.
usage (GPU box): python tools/probes/realloc_which.py"""
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
os.environ["H263MI_TRACE_ALLOC"] = "1"


def free_workload(wl):
    for fr in wl.frames:
        for k in ("mbs", "co", "base"):
            fr[k].free()


def measure(batch, wl, rgba):
    bench.run_frames(batch, wl, rgba, GOP, True)
    batch.sync()
    out = []
    for step in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.run_frames(batch, wl, rgba, GOP, True)
        batch.sync()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / GOP * 1e3)
    return sorted(out)[1]


wl = bench.Workload(h263mi, N, GOP, 0, 0, stream)
rgba = h263mi.DeviceBuffer(N * bench.RGBA_BYTES, 0)
batch = h263mi.Batch(N, bench.W, bench.H, 0, stream, pipeline_post=True)
print("start: %.4f ms   rgba %#x  records[1] %#x  coefficients[1] %#x" % (
    measure(batch, wl, rgba), rgba.ptr.value, wl.frames[1]["mbs"].ptr.value, wl.frames[1]["co"].ptr.value), flush=True)
spacers = []
for what in ("rgba", "frames", "workload"):
    for trial in range(5):
        spacers.append(h263mi.DeviceBuffer(((len(spacers) * 29) % 97 + 3) << 20, 0))
        if what == "rgba":
            rgba.free()
            rgba = h263mi.DeviceBuffer(N * bench.RGBA_BYTES, 0)
        elif what == "frames":
            batch.close()
            batch = h263mi.Batch(N, bench.W, bench.H, 0, stream, pipeline_post=True)
        else:
            free_workload(wl)
            wl = bench.Workload(h263mi, N, GOP, 0, 0, stream)
        print("new %-8s: %.4f ms   rgba %#x  records[1] %#x  coefficients[1] %#x" % (
            what, measure(batch, wl, rgba), rgba.ptr.value, wl.frames[1]["mbs"].ptr.value, wl.frames[1]["co"].ptr.value), flush=True)
