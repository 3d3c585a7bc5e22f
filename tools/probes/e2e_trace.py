"""Where does the host time of the end-to-end path go?  Runs bench.e2e_bitstream (64 Sorenson Spark 1080p streams
through h263mi_batch_decode_next_pictures) with H263MI_TRACE_E2E=1: the library prints, per call, the time in the parser
threads, waiting for the staging slot, packing into pinned memory and enqueueing copies + launches.
usage (GPU box): python tools/probes/e2e_trace.py"""
import json
import os
import sys

os.environ["H263MI_TRACE_E2E"] = "1"
import torch  # noqa: E402

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "h263-rs_amd"))
import bench  # noqa: E402
import h263mi  # noqa: E402

torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
d_rgba = h263mi.DeviceBuffer(64 * bench.RGBA_BYTES, 0)
threads = int(sys.argv[1]) if len(sys.argv) > 1 else None           # parser threads (default: the container's CPU quota)
print(json.dumps(bench.e2e_bitstream(h263mi, 64, 0, stream, d_rgba, parser_threads=threads, corpus="dense", n_distinct=4)), flush=True)
print(json.dumps(bench.e2e_bitstream(h263mi, 64, 0, stream, d_rgba, parser_threads=threads, corpus="kinds")), flush=True)
