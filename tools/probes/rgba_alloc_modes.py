"""Does the KIND of device memory behind the RGBA output matter to k_frame?

The RGBA surface is written once per frame index with non-temporal 16-byte stores and never read on the device; it is
half of the launch's bytes.  hipExtMallocWithFlags offers fine-grained (0x1) and uncached (0x3) device memory next to
the default coarse-grained kind: with either, the stores may skip the L2 / the infinity cache and leave them to the
planes that the next launch reads.  One process, one workload, the RGBA surface allocated each way in turn.
usage (GPU box): python tools/probes/rgba_alloc_modes.py [--rounds N]"""
import argparse
import ctypes as C
import os
import sys
import time

import torch

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "h263-rs_amd"))
import bench  # noqa: E402
import h263mi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--gops", type=int, default=3)
args = ap.parse_args()

torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipFree.argtypes = [C.c_void_p]
N, GOP = 64, bench.GOP


class Surface:
    def __init__(self, nbytes, flags):
        self.ptr = C.c_void_p()
        rc = hip.hipExtMallocWithFlags(C.byref(self.ptr), nbytes, flags)
        if rc:
            raise RuntimeError("hipExtMallocWithFlags(%#x) -> %d" % (flags, rc))

    def free(self):
        hip.hipFree(self.ptr)


b = h263mi.Batch(N, bench.W, bench.H, 0, stream, pipeline_post=True)
wl = bench.Workload(h263mi, N, GOP, 0, 0, stream, events=True)


def run(rgba):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.run_frames(b, wl, rgba, GOP * args.gops, True)
    b.sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (GOP * args.gops) * 1e3


modes = [("default (coarse-grained)", 0x0), ("fine-grained", 0x1), ("uncached", 0x3), ("default again", 0x0)]
for name, flags in modes:
    try:
        s = Surface(N * bench.RGBA_BYTES, flags)
    except RuntimeError as e:
        print("%-28s %s" % (name, e), flush=True)
        continue
    run(s)
    ts = [run(s) for _ in range(args.rounds)]
    print("%-28s %.4f ms per frame index (min %.4f max %.4f)  rgba at %#x" % (name, sum(ts) / len(ts), min(ts), max(ts), s.ptr.value), flush=True)
    s.free()
b.close()
