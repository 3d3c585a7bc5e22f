"""Does the order of the work matter?  The bench walks a 64-stream batch frame index by frame index (one launch = picture
f of all 64 streams).  Here the same 64 streams are cut into G groups and each group runs its whole GOP before the next
one starts: the planes a launch writes (n/G x 3.1 MB) are read again by the very next launch, out of the infinity cache.
usage (GPU box): python tools/probes/group_major.py"""
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
N, GOP, REPS = 64, bench.GOP, 8
for groups in (1, 2, 4, 8):
    n = N // groups
    wls = [bench.Workload(h263mi, n, GOP, g * n, 0, stream) for g in range(groups)]
    batches = [h263mi.Batch(n, bench.W, bench.H, 0, stream, pipeline_post=True) for _ in range(groups)]
    rgba = h263mi.DeviceBuffer(n * bench.RGBA_BYTES, 0)

    def run(reps):
        for _ in range(reps):
            for b, wl in zip(batches, wls):
                bench.run_frames(b, wl, rgba, GOP, True)
        for b in batches:
            b.sync()
    run(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(REPS)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%d group(s) of %2d streams: %.1f GP/s, %.4f ms per 64 pictures" %
          (groups, n, N * GOP * REPS * bench.MP_PER_PICTURE / dt / 1e3, dt / (GOP * REPS) * 1e3), flush=True)
    for b in batches:
        b.close()
    del wls, batches, rgba
