"""Two speeds per build, per process (profiles/r02_l_process_bimodality.txt): does the mode belong to the PROCESS (code
placement, clocks) or to WHERE ITS BUFFERS LANDED?  One process allocates the whole bench workload several times over
(freeing it in between, and holding a spacer allocation of a different size each time so that the driver hands out
different memory) and times a few GOPs each time.
usage (GPU box): python tools/probes/realloc_modes.py [trials]"""
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
GOP = bench.GOP
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 6
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
for trial in range(trials):
    spacer = h263mi.DeviceBuffer((trial * 37 + 1) << 20, 0) if trial else None
    wl = bench.Workload(h263mi, N, GOP, 0, 0, stream)
    batch = h263mi.Batch(N, bench.W, bench.H, 0, stream, pipeline_post=True)
    rgba = h263mi.DeviceBuffer(N * bench.RGBA_BYTES, 0)
    bench.run_frames(batch, wl, rgba, GOP, True)
    batch.sync()
    batch_ptr = wl.frames[1]['co'].ptr.value or 0
    out = []
    for step in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.run_frames(batch, wl, rgba, GOP, True)
        batch.sync()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / GOP * 1e3)
    print("%d streams, trial %d: ms per 64 pictures min %.4f median %.4f  (coefficients of frame 1 at %#x, rgba at %#x)" % (
        N, trial, min(out) * 64 / N, sorted(out)[3] * 64 / N, batch_ptr, rgba.ptr.value or 0), flush=True)
    batch.close()
    rgba.free()
    for fr in wl.frames:
        for k in ("mbs", "co", "base"):
            fr[k].free()
    del wl
    if spacer is not None:
        spacer.free()
