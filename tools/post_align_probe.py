"""Does the placement of the RGBA output relative to the frame store change the speed of the pipeline?  Between
processes k_post is bimodal (0.152 vs 0.169 ms inside the bench loop).  One process, several RGBA offsets and a few
re-allocations, full recon + post steps."""
import os
import sys

import torch

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "h263-rs_amd"))
import bench  # noqa: E402
import h263mi  # noqa: E402

torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
n = 64
wl = bench.Workload(n, 31, 0, 0, stream)


def measure(batch, ptr, steps=31):
    class P:       # run_steps wants an object with .ptr
        pass
    d = P()
    d.ptr = ptr
    bench.run_steps(batch, wl, d, 0, 31)
    batch.sync()
    batch.timing_begin()
    bench.run_steps(batch, wl, d, 31, steps)
    t = batch.timing_end()
    return t.recon_ms / t.recon_launches, t.post_ms / t.post_launches


keep = []
for trial in range(4):
    batch = h263mi.Batch(n, bench.W, bench.H, 0, stream)
    buf = h263mi.DeviceBuffer(n * bench.RGBA_BYTES + (64 << 20), 0)
    for off in (0, 4096, 1 << 20, 3 << 20, 16 << 20, 0):
        r, p = measure(batch, buf.at(off))
        print("alloc %d rgba %#x + %9d: k_recon %.4f  k_post %.4f ms" % (trial, buf.ptr.value, off, r, p))
    keep.append((batch, buf))          # keep them alive so that the next trial gets other addresses
    keep.append(h263mi.DeviceBuffer((37 + 11 * trial) << 20, 0))
