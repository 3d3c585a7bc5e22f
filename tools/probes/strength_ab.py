"""What a post-filter strength PER STREAM costs the headline launch (<= 1 %).  One batch, one workload, the
bench's GOPs: (a) h263mi_batch_decode_events with the uniform strength 5 -- the kernels get one pointer pair and one strength;
(b) the _ps form with strengths = [5] * 63 + [6] -- every wave takes its picture's strength (and frame set) from the stream's
word, one scalar load; (c) strengths drawn from QUANT_TO_STRENGTH[4..20], what 64 real streams look like.  Interleaved rounds,
wall clock around synced GOPs.  usage (GPU box): python tools/probes/strength_ab.py [rounds]"""
import os
import sys
import time

import torch

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "h263-rs_amd"))
import bench  # noqa: E402
import h263mi  # noqa: E402
import numpy as np  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
N, GOP = 64, bench.GOP
wl = bench.Workload(h263mi, N, GOP, 0, 0, stream, events=True)
rgba = h263mi.DeviceBuffer(N * bench.RGBA_BYTES, 0)
b = h263mi.Batch(N, bench.W, bench.H, 0, stream, pipeline_post=True)
q2s = h263mi.quant_to_strength()
variants = {
    "uniform 5": None,
    "per stream, 63 x 5 + 1 x 6": [5] * 63 + [6],
    "per stream, QUANT_TO_STRENGTH[4..20]": [int(q2s[4 + (7 * s) % 17]) for s in range(N)],
}


def gops(strengths, n_frames):
    g = len(wl.frames)
    for i in range(n_frames):
        fr = wl.frames[i % g]
        if fr.get("first") is not None:
            b.decode_events(fr["ptype"], fr["mbs"].ptr, fr["first"].ptr, fr["ev"].ptr, fr["base"].ptr, fr["blocks"], bench.STRENGTH,
                            rgba.ptr, None, n_events=fr["n_events"], strengths=strengths)
        else:
            b.decode(fr["ptype"], fr["mbs"].ptr, fr["co"].ptr, fr["base"].ptr, fr["blocks"], bench.STRENGTH, rgba.ptr, None,
                     strengths=strengths)
    b.sync()
    torch.cuda.synchronize()


res = {k: [] for k in variants}
for rnd in range(rounds + 1):
    for name, st in variants.items():
        gops(st, GOP)
        t0 = time.perf_counter()
        gops(st, 4 * GOP)
        dt = (time.perf_counter() - t0) / (4 * GOP) * 1e3
        if rnd:
            res[name].append(dt)
base = float(np.mean(res["uniform 5"]))
for name, v in res.items():
    print("%-40s %.4f ms per frame index (%+.2f %%, median %.4f %+.2f %%, spread %.4f)  rounds: %s" % (
        name, np.mean(v), 100 * (np.mean(v) / base - 1), np.median(v), 100 * (np.median(v) / float(np.median(res["uniform 5"])) - 1),
        max(v) - min(v), " ".join("%.4f" % x for x in v)), flush=True)
