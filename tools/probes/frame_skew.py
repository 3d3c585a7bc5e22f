"""Does it matter whether the rows of the frame store's planes start on a 64-byte line?  (H263MI_EXP_FRAME_SKEW: the
whole frame store is moved 0 / 16 / 32 / 48 bytes; RGBA runs that start 16 bytes into a line are 10 % faster than
line-aligned ones inside k_frame, profiles/README.md r03_zz.)  One library, one workload; a batch per skew, created and
closed per measurement (placement-neutral), rounds interleaved.
usage (GPU box): python tools/probes/frame_skew.py"""
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
wl = bench.Workload(h263mi, N, GOP, 0, 0, stream, events=True)
dense = bench.Workload(h263mi, N, 1, 0, 0, stream, i_kind=h263mi.SYNTH_I_DENSE, p_frames=False)
rgba = h263mi.DeviceBuffer(N * bench.RGBA_BYTES, 0)


def run(b, w, n, strength=None):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if strength is None:
        bench.run_frames(b, w, rgba, n, True)
    else:
        fr = w.frames[0]
        for i in range(n):
            b.decode(fr["ptype"], fr["mbs"].ptr, fr["co"].ptr, fr["base"].ptr, 0, strength, rgba.ptr, None)
    b.sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


skews = [0, 16, 32, 48, 0]
res = [[] for _ in skews]
resd = [[] for _ in skews]
for rnd in range(5):
    for k, sk in enumerate(skews):
        os.environ["H263MI_EXP_FRAME_SKEW"] = str(sk)
        b = h263mi.Batch(N, bench.W, bench.H, 0, stream, pipeline_post=True)
        run(b, wl, GOP)
        t, td = run(b, wl, GOP * 3), run(b, dense, 60, 0)
        b.close()
        if rnd:
            res[k].append(t)
            resd[k].append(td)
for k, sk in enumerate(skews):
    m, md = sum(res[k]) / len(res[k]), sum(resd[k]) / len(resd[k])
    print("frame store %2d bytes past a line: P workload %.4f ms (%+.1f %%, spread %.4f)   dense I %.4f (%+.1f %%)" % (
        sk, m, 100 * (m / (sum(res[0]) / len(res[0])) - 1), max(res[k]) - min(res[k]), md, 100 * (md / (sum(resd[0]) / len(resd[0])) - 1)), flush=True)
