"""Dense I pictures alone (BASELINE configs[1]: every block Full, 64 x 1080p) through the frame-pipelined launch, for
counter runs: rocprofv3 --pmc ... -- python3 tools/probes/dense_i_only.py [launches]"""
import os
import sys
import time

import torch

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "h263-rs_amd"))
import bench  # noqa: E402
import h263mi  # noqa: E402

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
N = 64
b = h263mi.Batch(N, bench.W, bench.H, 0, stream, pipeline_post=True)
wl = bench.Workload(h263mi, N, 1, 0, 0, stream, i_kind=h263mi.SYNTH_I_DENSE, p_frames=False)
rgba = h263mi.DeviceBuffer(N * bench.RGBA_BYTES, 0)
fr = wl.frames[0]
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(launches):
        b.decode(fr["ptype"], fr["mbs"].ptr, fr["co"].ptr, fr["base"].ptr, 0, 0, rgba.ptr, None)
    b.sync()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / launches * 1e3
print("dense I pictures, 64 x 1080p, strength 0: %.4f ms per launch" % t)
b.close()
