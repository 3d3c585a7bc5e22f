"""What a key frame costs when its coefficients travel as events (the end-to-end path: the host parser emits nothing else):
the bench workload's mixed-class I picture (20 events per coded block) and the dense one (64 per block), 64 streams, through
h263mi_batch_decode_events on a pipelined batch, library by library.
usage (GPU box): python tools/probes/events_i_picture.py libA.so libB.so ..."""
import os
import sys
import time

import torch

os.environ["H263MI_BENCH_EVENTS_ALWAYS"] = "1"
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "h263-rs_amd"))
import bench  # noqa: E402
import h263mi  # noqa: E402

torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
N = 64
handles = []
for path in sys.argv[1:]:
    h263mi._lib = None
    h263mi.LIB_PATH = os.path.abspath(path)
    handles.append(h263mi.lib())
h263mi._lib = handles[0]
rgba = h263mi.DeviceBuffer(N * bench.RGBA_BYTES, 0)
for name, kind in (("mixed-class I", h263mi.SYNTH_I_MIXED), ("dense I", h263mi.SYNTH_I_DENSE)):
    wl = bench.Workload(h263mi, N, 1, 0, 0, stream, i_kind=kind, p_frames=False, events=True)
    fr = wl.frames[0]
    assert fr["first"] is not None
    for rnd in range(3):
        for k, L in enumerate(handles):
            h263mi._lib = L
            b = h263mi.Batch(N, bench.W, bench.H, 0, stream, pipeline_post=True)
            for timed in (False, True):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(40):
                    b.decode_events(fr["ptype"], fr["mbs"].ptr, fr["first"].ptr, fr["ev"].ptr, fr["base"].ptr, 0, bench.STRENGTH,
                                    rgba.ptr, None)
                b.sync()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 40 * 1e3
            b.close()
            if rnd:
                print("%-14s %-20s %.4f ms per 64 key frames (%d events per picture)" % (
                    name, os.path.basename(sys.argv[1 + k]), dt, fr["n_events"] // N), flush=True)
    h263mi._lib = handles[0]
    # the same picture with its coefficients as dense 128-byte blocks (h263mi_batch_decode), first library
    wd = bench.Workload(h263mi, N, 1, 0, 0, stream, i_kind=kind, p_frames=False, events=False)
    fd = wd.frames[0]
    b = h263mi.Batch(N, bench.W, bench.H, 0, stream, pipeline_post=True)
    for timed in (False, True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            b.decode(fd["ptype"], fd["mbs"].ptr, fd["co"].ptr, fd["base"].ptr, 0, bench.STRENGTH, rgba.ptr, None)
        b.sync()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 40 * 1e3
    b.close()
    print("%-14s %-20s %.4f ms per 64 key frames as dense blocks (%d blocks per picture)" % (
        name, os.path.basename(sys.argv[1]), dt, fd["blocks"] // N), flush=True)
