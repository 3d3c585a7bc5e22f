"""Do the ends of a launch cost anything?  The bench walks a 64-stream batch frame index by frame index on ONE HIP
stream: launch f + 1 starts when the last wave of launch f has retired, so every launch pays its ramp-up and its tail.
Here the same 64 streams are cut into Q batches of 64 / Q streams, each on a HIP stream of its own, and the frame
indices of the Q batches are submitted interleaved: the chains are independent (streams never exchange data), so the
tail of one batch's launch overlaps the body of another's.
usage (GPU box): python tools/probes/two_queues.py [Q ...]"""
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
N, GOP, REPS = 64, bench.GOP, 8
main_stream = torch.cuda.current_stream().cuda_stream
for queues in ([int(a) for a in sys.argv[1:]] or [1, 2, 4, 1, 2, 4]):
    n = N // queues
    streams = [torch.cuda.Stream() for _ in range(queues)]
    wls = [bench.Workload(h263mi, n, GOP, q * n, 0, main_stream) for q in range(queues)]
    torch.cuda.synchronize()
    batches = [h263mi.Batch(n, bench.W, bench.H, 0, s.cuda_stream, pipeline_post=True) for s in streams]
    rgbas = [h263mi.DeviceBuffer(n * bench.RGBA_BYTES, 0) for _ in range(queues)]

    def run(reps):
        for _ in range(reps):
            for f in range(GOP):
                for b, wl, rgba in zip(batches, wls, rgbas):
                    fr = wl.frames[f]
                    b.decode(fr["ptype"], fr["mbs"].ptr, fr["co"].ptr, fr["base"].ptr, 0, bench.STRENGTH, rgba.ptr, None)
        for b in batches:
            b.sync()
    run(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(REPS)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%d queue(s) of %2d streams: %.1f GP/s, %.4f ms per 64 pictures" %
          (queues, n, N * GOP * REPS * bench.MP_PER_PICTURE / dt / 1e3, dt / (GOP * REPS) * 1e3), flush=True)
    for b in batches:
        b.close()
    del wls, batches, rgbas, streams
