"""A/B of several builds of libh263mi.so INSIDE ONE PROCESS, on one set of input / output buffers.

Between processes -- and between two allocations of the workload in one process -- the same build runs 5-14 % apart
(tools/probes/realloc_modes.py): where the driver puts the buffers decides.  Here the records, the coefficients and
the RGBA output are allocated once and every build decodes the same GOPs from them, round after round, interleaved; only
each build's own frame store differs (its placement did not matter in tools/probes/frame_skew.py).
usage (GPU box): python tools/ab_inproc.py [--rounds N] [--dense] libA.so libB.so ..."""
import argparse
import os
import sys
import time

import torch

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "h263-rs_amd"))
import bench  # noqa: E402
import h263mi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--gops", type=int, default=3)
ap.add_argument("--dense-coeffs", action="store_true", help="dense coefficient blocks instead of events")
ap.add_argument("--placements", type=int, default=1, help="allocate the input / output buffers this many times over and run the A/B on each")
ap.add_argument("--strength", type=int, default=None, help="deblocking strength of the P workload (default: the bench's; 0 = no deblocking)")
ap.add_argument("libs", nargs="+")
args = ap.parse_args()

if args.strength is not None:
    bench.STRENGTH = args.strength
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
N, GOP = 64, bench.GOP
handles = []
for path in args.libs:
    h263mi._lib = None
    h263mi.LIB_PATH = os.path.abspath(path)
    handles.append(h263mi.lib())
h263mi._lib = handles[0]
# Every measurement creates its batch and closes it again: the libraries then take turns on ONE frame store's worth of
# device memory (the allocator hands the block just freed to the next batch).  With a batch per library kept for the whole
# run, the batch created last ran the P workload 6-9 % slower than its siblings on some boxes, whichever library it
# belonged to (profiles/README.md r03_v, r03_y): a position in the argument list is not a property of a build.
batches = [None] * len(handles)


class fresh_batch:
    def __init__(self, L):
        self.L = L

    def __enter__(self):
        h263mi._lib = self.L
        self.b = h263mi.Batch(N, bench.W, bench.H, 0, stream, pipeline_post=True)
        return self.b

    def __exit__(self, *exc):
        h263mi._lib = self.L
        self.b.close()


def time_gops(b, w, n_frames, rgba, strength=None):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g = len(w.frames)
    if strength is None:
        bench.run_frames(b, w, rgba, n_frames, True)
    else:
        for i in range(n_frames):
            fr = w.frames[i % g]
            b.decode(fr["ptype"], fr["mbs"].ptr, fr["co"].ptr, fr["base"].ptr, 0, strength, rgba.ptr, None)
    b.sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n_frames * 1e3


spacers = []
for placement in range(args.placements):
    h263mi._lib = handles[0]
    wl = bench.Workload(h263mi, N, GOP, 0, 0, stream, events=not args.dense_coeffs)
    dense = bench.Workload(h263mi, N, 1, 0, 0, stream, i_kind=h263mi.SYNTH_I_DENSE, p_frames=False)
    rgba = h263mi.DeviceBuffer(N * bench.RGBA_BYTES, 0)
    res = [[] for _ in handles]
    resd = [[] for _ in handles]
    for rnd in range(args.rounds + 1):
        for k, L in enumerate(handles):
            with fresh_batch(L) as b:
                time_gops(b, wl, GOP, rgba)                      # (the new frame store's first touch)
                t = time_gops(b, wl, GOP * args.gops, rgba)
                td = time_gops(b, dense, 60, rgba, 0)
            if rnd:                                  # round 0 warms up
                res[k].append(t)
                resd[k].append(td)
    print("placement %d (rgba at %#x, records of frame 1 at %#x)" % (placement, rgba.ptr.value, wl.frames[1]["mbs"].ptr.value))
    base = sum(res[0]) / len(res[0])
    based = sum(resd[0]) / len(resd[0])
    for k, path in enumerate(args.libs):
        m, md = sum(res[k]) / len(res[k]), sum(resd[k]) / len(resd[k])
        print("  %-24s P workload %.4f ms/frame index (%+.1f %%, spread %.4f)   dense I %.4f (%+.1f %%)" % (
            os.path.basename(path), m, 100 * (m / base - 1), max(res[k]) - min(res[k]), md, 100 * (md / based - 1)), flush=True)
    h263mi._lib = handles[0]
    rgba.free()
    for w in (wl, dense):
        for fr in w.frames:
            for key in ("mbs", "co", "base", "first", "ev"):
                if fr.get(key) is not None:
                    fr[key].free()
    spacers.append(h263mi.DeviceBuffer(((placement * 41) % 89 + 5) << 20, 0))

