"""NUMA placement A/B on ONE GPU: the end-to-end legs with the host side of the batch (parser threads,
pinned staging memory) on the socket of the GPU ("local", the default), on another node on purpose ("far": H263MI_NUMA_NODE)
and left alone ("unbound": H263MI_NUMA=0).  Interleaved rounds inside one process: every leg makes its own batch, and a batch
reads the switches when it is made.  usage (GPU box): python tools/probes/numa_ab.py [rounds]"""
import glob
import json
import os
import sys

import torch

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "h263-rs_amd"))
import bench  # noqa: E402
import h263mi  # noqa: E402
import numpy as np  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
N = 64
d_rgba = h263mi.DeviceBuffer(N * bench.RGBA_BYTES, 0)
nodes = sorted(int(p.rsplit("node", 1)[1]) for p in glob.glob("/sys/devices/system/node/node[0-9]*"))
b = h263mi.Batch(1, 176, 144)
dev_node = b.host_placement()[0]
b.close()
print("NUMA nodes of the host: %s; node of device 0: %d; CPUs this process may run on: %d" % (nodes, dev_node, len(os.sched_getaffinity(0))), flush=True)
for k in nodes:
    try:
        print("  node%d cpus %s" % (k, open("/sys/devices/system/node/node%d/cpulist" % k).read().strip()))
    except Exception:
        pass
threads, quota = h263mi.default_parser_threads(N)
modes = {"local": {}, "unbound": {"H263MI_NUMA": "0"}}
others = [k for k in nodes if k != dev_node]
if dev_node >= 0 and others:
    modes["far (node %d)" % others[-1]] = {"H263MI_NUMA_NODE": str(others[-1])}
res = {m: {"kinds": [], "distinct": [], "single_P_ms": [], "placement": None} for m in modes}
for rnd in range(rounds):
    for m, env in modes.items():
        for k in ("H263MI_NUMA", "H263MI_NUMA_NODE"):
            os.environ.pop(k, None)
        os.environ.update(env)
        for corpus in ("kinds", "distinct"):
            e = bench.e2e_bitstream(h263mi, N, 0, stream, d_rgba, corpus=corpus, n_distinct=8, parser_threads=threads, min_seconds=1.0)
            assert e["parity_vs_oracle"] == "ok"
            res[m][corpus].append(e["pictures_per_s"])
            res[m]["placement"] = e["host_placement"]
        s1 = bench.single_stream_latency(h263mi, 0, stream, n_p=8, reps=2)
        res[m]["single_P_ms"].append(s1["pinned_direct"]["P_picture_ms"])
        print("round %d %-16s kinds %.0f distinct %.0f pictures/s, single stream P %.3f ms, placement %s" % (
            rnd, m, res[m]["kinds"][-1], res[m]["distinct"][-1], res[m]["single_P_ms"][-1], res[m]["placement"]), flush=True)
print(json.dumps({m: {"kinds_pictures_per_s": round(float(np.mean(v["kinds"])), 1), "distinct_pictures_per_s": round(float(np.mean(v["distinct"])), 1),
                      "single_stream_P_ms": round(float(np.mean(v["single_P_ms"])), 4), "placement": v["placement"]} for m, v in res.items()}))
