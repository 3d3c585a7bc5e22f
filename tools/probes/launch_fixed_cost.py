"""What does a k_frame launch cost beyond its pictures?  Time per frame index against the number of streams.

Every launch starts with all waves waiting for their first records and ends with the last waves draining while the
compute units empty; consecutive launches cannot overlap (launch f + 1 reads the planes launch f writes).  If
T(n) = a + b * n, `a` is what a multi-picture launch (or a dataflow kernel) could win back.
usage (GPU box): python tools/probes/launch_fixed_cost.py"""
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
GOP = 9
res = []
for n in (8, 16, 32, 64, 96, 128, 192):
    b = h263mi.Batch(n, bench.W, bench.H, 0, stream, pipeline_post=True)
    wl = bench.Workload(h263mi, n, GOP, 0, 0, stream, events=True)
    rgba = h263mi.DeviceBuffer(n * bench.RGBA_BYTES, 0)
    ts = []
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.run_frames(b, wl, rgba, GOP * 6, True)
        b.sync()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / (GOP * 6) * 1e3)
    t = min(ts[1:])
    res.append((n, t))
    print("%4d streams: %.4f ms per frame index = %.3f us per picture" % (n, t, t / n * 1e3), flush=True)
    b.close()
    rgba.free()
    for fr in wl.frames:
        for key in ("mbs", "co", "base", "first", "ev"):
            if fr.get(key) is not None:
                fr[key].free()
# least squares T = a + b n over n >= 32
xs = [(n, t) for n, t in res if n >= 32]
mx = sum(n for n, _ in xs) / len(xs)
my = sum(t for _, t in xs) / len(xs)
bb = sum((n - mx) * (t - my) for n, t in xs) / sum((n - mx) ** 2 for n, _ in xs)
aa = my - bb * mx
print("fit over n >= 32: T(n) = %.4f ms + n * %.3f us  (fixed part = %.1f %% of T(64))" % (aa, bb * 1e3, 100 * aa / (aa + 64 * bb)))
