"""Per-call wall time of the end-to-end path over one GOP (64 realistic 1080p streams, 16 parser threads): which calls are
slow -- the key frame, the first P pictures behind it, all of them?  usage (GPU box): python tools/probes/e2e_per_call.py [threads]
E2E_GOPS=22 in the environment: that many GOPs, and the median / best GOP behind the first two at the end (A/B runs of two builds or
two switches -- H263MI_DIRECT_WORDS=0, H263MI_SPARSE_RECORDS=0 -- alternate this on one lease: profiles/r05_r_ab_direct_words.txt)."""
import os, sys, time
import torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "h263-rs_amd")); sys.path.insert(0, os.path.join(R, "tests"))
import bench, h263mi, recgen
import sorenson_enc as enc
from test_bitstream_e2e import make_codable
W, H, n = bench.W, bench.H, 64
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
d_rgba = h263mi.DeviceBuffer(n * bench.RGBA_BYTES, 0)
# E2E_DISTINCT different streams (default 8): a parser thread that meets the same picture again and again has its branches
# predicted from history (tools/probes/e2e_distinct_streams.py: 120 k pictures/s with one stream repeated, 113 k with 16)
n_distinct = int(os.environ.get("E2E_DISTINCT", "8"))
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
gops = []


def cpu_stat():
    """usage_usec, nr_throttled, throttled_usec of this process's cgroup (cgroup v2), or None"""
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(d["usage_usec"]), int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
    except Exception:
        return None


stat0, t_stat0 = None, 0.0
for rep in range(int(os.environ.get("E2E_GOPS", "3"))):
    times = []
    if rep == 2:
        stat0, t_stat0 = cpu_stat(), time.perf_counter()
    t_gop = time.perf_counter()
    for f in order:
        t0 = time.perf_counter()
        batch.decode_next_pictures_ex(None, n_threads=threads, prepared=prepared[f], strength=5, d_rgba=d_rgba.ptr)
        times.append(time.perf_counter() - t0)
    t_sync = time.perf_counter()
    batch.sync()
    t_end = time.perf_counter()
    print("GOP %d: %.2f ms (%d pictures/s); I call %.3f ms; P calls: first %.3f, median %.3f, max %.3f ms; final sync %.3f ms" % (
        rep, (t_end - t_gop) * 1e3, n * 31 / (t_end - t_gop), times[0] * 1e3, times[1] * 1e3, sorted(times[1:])[15] * 1e3,
        max(times[1:]) * 1e3, (t_end - t_sync) * 1e3), flush=True)
    gops.append(t_end - t_gop)
if len(gops) > 4:
    g = sorted(gops[2:])
    # (the MEAN is what a server gets: a process that uses more CPU time than its cgroup's quota is frozen for the rest of the
    # scheduler period, which a median does not show)
    print("GOPs %d..%d: median %.2f ms (%d pictures/s), best %.2f ms (%d pictures/s), mean %.2f ms (%d pictures/s), worst %.2f ms" % (
        2, len(gops) - 1, g[len(g) // 2] * 1e3, n * 31 / g[len(g) // 2], g[0] * 1e3, n * 31 / g[0],
        sum(g) / len(g) * 1e3, n * 31 * len(g) / sum(g), g[-1] * 1e3), flush=True)
    stat1, t_stat1 = cpu_stat(), time.perf_counter()
    if stat0 and stat1:
        print("cgroup over those GOPs: %.1f CPUs busy on average, throttled in %d scheduler periods for %.1f ms of thread time" % (
            (stat1[0] - stat0[0]) * 1e-6 / (t_stat1 - t_stat0), stat1[1] - stat0[1], (stat1[2] - stat0[2]) * 1e-3), flush=True)
print("P calls of the last GOP (ms):", " ".join("%.2f" % (t * 1e3) for t in times[1:]))
batch.close()
