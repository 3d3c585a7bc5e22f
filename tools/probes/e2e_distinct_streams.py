"""Does the end-to-end figure depend on how many DISTINCT streams the 64 are made of?  (A parser thread that meets the same
15 KB picture again and again has its branches predicted from history; 64 real streams are all different.)
bench.e2e_bitstream(realistic=True) with n_distinct = 1, 2, 4, 8, 16.  usage (GPU box): python tools/probes/e2e_distinct_streams.py"""
import os, sys
import torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "h263-rs_amd"))
import bench, h263mi
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
d_rgba = h263mi.DeviceBuffer(64 * bench.RGBA_BYTES, 0)
threads, quota = h263mi.default_parser_threads(64)
for k in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16]:
    e = bench.e2e_bitstream(h263mi, 64, 0, stream, d_rgba, n_distinct=k, parser_threads=threads, corpus="kinds")
    print("n_distinct %2d: %8.0f pictures/s on %d threads, one parser thread %6.0f, parity %s" % (
        k, e["pictures_per_s"], threads, e["one_parser_thread_pictures_per_s"], e["parity_vs_oracle"]), flush=True)
