import json, os, sys
import torch
R = os.getcwd()
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "h263-rs_amd"))
import bench, h263mi
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
d_rgba = h263mi.DeviceBuffer(64 * bench.RGBA_BYTES, 0)
def stat():
    try:
        return dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().strip().splitlines())
    except Exception as e:
        return {}
for thr in (8, 12, 14, 15, 16, 20):
    s0 = stat()
    d = bench.e2e_bitstream(h263mi, 64, 0, stream, d_rgba, parser_threads=thr, corpus="kinds")
    s1 = stat()
    print(thr, "threads:", d["pictures_per_s"], "pictures/s; one thread", d["one_parser_thread_pictures_per_s"],
          "throttled periods +%d, +%.1f ms" % (int(s1.get("nr_throttled", 0)) - int(s0.get("nr_throttled", 0)),
                                               (int(s1.get("throttled_usec", 0)) - int(s0.get("throttled_usec", 0))) / 1e3), flush=True)
