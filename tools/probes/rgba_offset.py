"""Does the speed depend on where the RGBA output sits relative to the frame sets?  One process (one physical placement),
the RGBA pointer shifted by multiples of 4 KB inside a larger buffer.
usage (GPU box): python tools/probes/rgba_offset.py"""
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
rgba = h263mi.DeviceBuffer(N * bench.RGBA_BYTES + (64 << 20), 0)


class Shifted:
    def __init__(self, ptr):
        self.ptr = ptr


for off in ([int(a) for a in sys.argv[1:]] or [0, 4096, 8192, 16384, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, 3 << 20, 7 << 20, 33 << 20, 0]):
    buf = Shifted(rgba.ptr.value + off)
    bench.run_frames(batch, wl, buf, GOP, True)
    batch.sync()
    ts = []
    for step in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.run_frames(batch, wl, buf, GOP, True)
        batch.sync()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / GOP * 1e3)
    print("RGBA offset %9d: %.4f ms per frame index (min of 6; max %.4f)" % (off, min(ts), max(ts)), flush=True)
