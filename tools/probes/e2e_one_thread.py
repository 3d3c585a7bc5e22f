"""The end-to-end path with ONE parser thread (64 realistic 1080p streams, GOPs of 31): best GOP of four.  bench.py's
`one_parser_thread_pictures_per_s` is the same measurement behind its many-thread run.  usage (GPU box): python tools/probes/e2e_one_thread.py"""
import os, sys, time
import torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "h263-rs_amd")); sys.path.insert(0, os.path.join(R, "tests"))
import bench, h263mi, recgen
import sorenson_enc as enc
from test_bitstream_e2e import make_codable
W, H, n = bench.W, bench.H, 64
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
d_rgba = h263mi.DeviceBuffer(n * bench.RGBA_BYTES, 0)
n_distinct = int(os.environ.get("E2E_DISTINCT", "8"))    # (one stream repeated flatters the branch predictor: e2e_distinct_streams.py)
pics = []
for s in range(n_distinct):
    row = []
    for f in range(8):
        mbs, co = (recgen.realistic_intra_picture(W, H, 300 + s) if f == 0 else recgen.realistic_inter_picture(W, H, 7000 + 100 * s + f))
        row.append(enc.encode_picture(W, H, 0 if f == 0 else 1, 10, make_codable(mbs, 10, f, 0 if f == 0 else 1), co, temporal_reference=f))
    pics.append(row)
batch = h263mi.Batch(n, W, H, 0, stream, pipeline_post=True)
import numpy as np
variant = [int(v) for v in np.random.default_rng(20261004).integers(0, n_distinct, n)]     # (not s % n_distinct: see bench.e2e_bitstream)
prepared = [batch.prepare_pictures([pics[variant[s]][f] for s in range(n)]) for f in range(8)]
order = [0] + [1 + k % 7 for k in range(30)]
if len(sys.argv) > 1:                                    # a many-thread run in front, as in bench.py
    for rep in range(10):
        for f in order:
            batch.decode_next_pictures_ex(None, n_threads=int(sys.argv[1]), prepared=prepared[f], strength=5, d_rgba=d_rgba.ptr)
    batch.sync()
best = 1e9
for rep in range(4):
    t0 = time.perf_counter()
    for f in order:
        batch.decode_next_pictures_ex(None, n_threads=1, prepared=prepared[f], strength=5, d_rgba=d_rgba.ptr)
    batch.sync()
    dt = time.perf_counter() - t0
    best = min(best, dt)
print("one parser thread: best GOP %.1f ms = %d pictures/s" % (best * 1e3, n * 31 / best))
batch.close()
