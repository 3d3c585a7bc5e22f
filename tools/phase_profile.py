"""Wall-clock ticks per phase of k_recon (diagnosis build -DH263MI_PROFILE_PHASES, H263MI_LIB=<that .so>).
usage (GPU box): H263MI_LIB=$PWD/gpurun_ab_PH.so python tools/phase_profile.py"""
import ctypes as C
import os
import sys

import torch

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "h263-rs_amd"))
import bench  # noqa: E402
import h263mi  # noqa: E402

L = h263mi.lib()
L.h263mi_debug_read_phases.argtypes = [C.c_void_p, C.c_int]
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
n = 64
wl = bench.Workload(h263mi, n, 9, 0, 0, stream)
batch = h263mi.Batch(n, bench.W, bench.H, 0, stream)
buf = (C.c_ulonglong * 8)()
names = ["records -> LDS", "mark + compact", "issue loads", "row pass, round 0", "predict (waits for rows)", "column passes + rounds", "store"]
waves_per_launch = n * 15 * 68          # one wave per 8 macroblocks of a macroblock row
NP = len(names)


def report(label, launches):
    L.h263mi_debug_read_phases(buf, 1)
    tot = sum(buf[:NP])
    print("%s: %.0f ticks per wave (s_memtime)" % (label, tot / (waves_per_launch * launches)))
    for i in range(NP):
        print("  %-22s %8.1f ticks  %5.1f %%" % (names[i], buf[i] / (waves_per_launch * launches), 100.0 * buf[i] / max(tot, 1)))


fr = wl.frames[0]
batch.submit(fr["ptype"], fr["mbs"].ptr, fr["co"].ptr, fr["base"].ptr)
batch.sync()
report("I picture (mixed classes)", 1)
for fr in wl.frames[1:]:
    batch.submit(fr["ptype"], fr["mbs"].ptr, fr["co"].ptr, fr["base"].ptr)
batch.sync()
report("P pictures", len(wl.frames) - 1)

# the same inside the frame-pipelined launch (k_frame: reconstruction waves beside post-processing waves)
batch.close()
batch = h263mi.Batch(n, bench.W, bench.H, 0, stream, pipeline_post=True)
d_rgba = h263mi.DeviceBuffer(n * bench.RGBA_BYTES, 0)
fr = wl.frames[0]
batch.decode(fr["ptype"], fr["mbs"].ptr, fr["co"].ptr, fr["base"].ptr, 0, bench.STRENGTH, d_rgba.ptr, None)
batch.sync()
L.h263mi_debug_read_phases(buf, 1)
for rep in range(3):
    for fr in wl.frames[1:]:
        batch.decode(fr["ptype"], fr["mbs"].ptr, fr["co"].ptr, fr["base"].ptr, 0, bench.STRENGTH, d_rgba.ptr, None)
batch.sync()
report("P pictures inside k_frame", 3 * (len(wl.frames) - 1))
