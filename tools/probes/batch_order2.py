"""Follow-up of batch_order.py: is a batch slow because of WHERE its frame store lies, or because of how it lies
relative to the RGBA surface it writes?  RGBA surface 1, then batch A, then RGBA surface 2, then batch B; each batch
timed against each surface.
usage (GPU box): python tools/probes/batch_order2.py"""
import os
import sys
import time

os.environ["H263MI_TRACE_ALLOC"] = "1"
import torch  # noqa: E402

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


wl = bench.Workload(h263mi, N, GOP, 0, 0, stream, events=True)
b0 = mk()
s1 = h263mi.DeviceBuffer(N * bench.RGBA_BYTES, 0)
print("rgba surface 1: %#x" % s1.ptr.value, flush=True)
bA = mk()
s2 = h263mi.DeviceBuffer(N * bench.RGBA_BYTES, 0)
print("rgba surface 2: %#x" % s2.ptr.value, flush=True)
bB = mk()


def run(b, rgba):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.run_frames(b, wl, rgba, GOP * 3, True)
    b.sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (GOP * 3) * 1e3


cases = [("batch 0 (before both surfaces)", b0), ("batch A (between the surfaces)", bA), ("batch B (after both)", bB)]
for name, b in cases:
    for sname, s in (("surface 1", s1), ("surface 2", s2)):
        run(b, s)
        ts = [run(b, s) for _ in range(3)]
        print("%-34s -> %s: %.4f ms per frame index (spread %.4f)" % (name, sname, sum(ts) / 3, max(ts) - min(ts)), flush=True)
