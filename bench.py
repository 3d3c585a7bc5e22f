#!/usr/bin/env python3
"""bench.py -- decoded megapixels/s of the MI355X macroblock back-end (BASELINE.json metric).

Workload (config.workload): BASELINE configs[3] -- a batch of 64 independent 1080p streams on
each GPU.  A step is one frame index over the whole batch: k_recon (dequant + IDCT + half-pel
MC + residual add/clip) and k_post (deblock strength 5 + BT.601 -> RGBA) for 64 pictures.
Frames cycle through a GOP of 31: one I picture (mixed block classes) then 30 P pictures
(half-pel vectors in [-32, 31], 25 % coded blocks, quant 10).  Records are generated on the
device beforehand (counter-based splitmix64), so inputs are resident in HBM when
the timed region starts.  With --gpus N every rank decodes its own 64 streams (weak scaling,
no data-path collective); RCCL only carries the barrier and the max-over-ranks reduction.

One JSON line is printed by rank 0.  `roofline` is for the kernel with the larger share of the
timed region; `cpu_baseline` is the C oracle (a port of the reference CPU path, not the Rust
binary) timed on this box's host cores.
"""
import argparse
import json
import os
import sys
import threading
import time

import torch  # first: torch brings its own HIP runtime; the C-ABI library must share it

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "h263-rs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

import h263mi  # noqa: E402
import shard  # noqa: E402

W, H = 1920, 1080
MBS_PP = 120 * 68
MP_PER_PICTURE = W * H / 1e6
Y_BYTES, C_BYTES = W * H, 960 * 540
YUV_BYTES = Y_BYTES + 2 * C_BYTES            # 3 110 400
RGBA_BYTES = W * H * 4                       # 8 294 400
HDR_BYTES = MBS_PP * 32                      # 261 120
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
STRENGTH = 5                                 # QUANT_TO_STRENGTH[10] (deblock.rs:5-8)


class Workload:
    """Device-resident records of `gop` frame indices for `n` streams."""

    def __init__(self, n, gop, first_stream, device_id, stream, i_kind=h263mi.SYNTH_I_MIXED, p_frames=True):
        self.n, self.frames = n, []
        for f in range(gop):
            kind = i_kind if (f == 0 or not p_frames) else h263mi.SYNTH_P
            cap = n * MBS_PP * (6 if kind != h263mi.SYNTH_P else 2)
            d_mbs = h263mi.DeviceBuffer(n * MBS_PP * 32, device_id)
            d_co = h263mi.DeviceBuffer(cap * 128, device_id)
            d_base = h263mi.DeviceBuffer(n * 8, device_id)
            blocks = h263mi.synth_batch_device(kind, W, H, n, first_stream, f, d_mbs.ptr, d_co.ptr, cap, d_base.ptr,
                                               device_id, stream)
            ptype = h263mi.PICTURE_I if kind != h263mi.SYNTH_P else h263mi.PICTURE_P
            self.frames.append(dict(kind=kind, ptype=ptype, mbs=d_mbs, co=d_co, base=d_base, blocks=blocks))

    def recon_bytes(self, f):
        """algorithmic bytes of one k_recon launch: headers + coefficients + reference read
        (P only) + reconstructed planes written."""
        fr = self.frames[f]
        b = self.n * HDR_BYTES + fr["blocks"] * 128 + self.n * YUV_BYTES
        if fr["ptype"] == h263mi.PICTURE_P:
            b += self.n * YUV_BYTES
        return b

    def post_bytes(self):
        """algorithmic bytes of one k_post launch: the RGBA frames written.  Re-reading the
        reconstructed planes is the price of running deblock + convert as a second kernel and is
        NOT counted (a fully fused pipeline would keep them on chip)."""
        return self.n * RGBA_BYTES


def run_steps(batch, wl, d_rgba, first, count):
    g = len(wl.frames)
    for i in range(first, first + count):
        fr = wl.frames[i % g]
        batch.submit(fr["ptype"], fr["mbs"].ptr, fr["co"].ptr, fr["base"].ptr)
        batch.render_rgba(STRENGTH, d_rgba.ptr, None)


def cpu_baseline(budget_s=20.0):
    """The oracle (C restatement of the reference CPU path) on the host cores of this box: the same
    synthetic 1080p stream (I + P pictures, deblock strength 5 on the three planes, BT.601), one
    independent stream per thread like the reference's single-threaded-per-stream design."""
    from oracle import oracle as orc   # checker/baseline only: never on the product path

    def decode_stream(stream_id, n_frames, out):
        ref, px = None, 0
        for f in range(n_frames):
            kind = h263mi.SYNTH_I_MIXED if f == 0 else h263mi.SYNTH_P
            mbs, co = h263mi.synth_picture_host(kind, W, H, stream_id, f)
            t0 = time.perf_counter()
            rc, ref = orc.decode_picture(W, H, mbs, co, ref)
            assert rc == 0
            filt = tuple(orc.deblock(p, pw, STRENGTH) for p, pw in zip(ref, (W, 960, 960)))
            orc.yuv420_to_rgba(*filt, W)
            out[0] += time.perf_counter() - t0
            px += 1
        out[1] = px

    orc.lib()
    probe = [0.0, 0]
    decode_stream(0, 3, probe)                       # 1 I + 2 P on one core
    per_frame = probe[0] / probe[1]
    one_thread = MP_PER_PICTURE / per_frame
    cores = os.cpu_count() or 1
    n_frames = int(max(3, min(31, budget_s / per_frame)))
    outs = [[0.0, 0] for _ in range(cores)]
    threads = [threading.Thread(target=decode_stream, args=(100 + i, n_frames, outs[i])) for i in range(cores)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    wall = time.perf_counter() - t0
    busy = max(o[0] for o in outs)                   # excludes the record generation
    value = cores * n_frames * MP_PER_PICTURE / busy
    return {"value": round(value, 2), "unit": "MP/s", "cores": cores, "kind": "port",
            "sample": "%d threads x 1 stream x %d pictures (1 I + %d P) of the bench workload at 1920x1080, "
                      "recon + deblock(5) x3 planes + BT.601; C oracle (port of the h263-rs CPU path, not the "
                      "Rust binary); %.1f s wall" % (cores, n_frames, n_frames - 1, wall),
            "one_thread_mp_s": round(one_thread, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=62)
    ap.add_argument("--warmup", type=int, default=31)
    ap.add_argument("--streams", type=int, default=64, help="streams per GPU")
    ap.add_argument("--gop", type=int, default=31)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--overlap", action="store_true",
                    help="k_post on a second stream (post of picture i beside recon of picture i+1); measured: no gain, "
                         "both kernels fill the chip")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MI355X back-end has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("H263MI_FORCE_DIST"):      # (the env switch exercises the RCCL path on one GPU)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # "nccl" is RCCL on ROCm

    stream = torch.cuda.current_stream().cuda_stream
    n = args.streams
    my_streams = shard.streams_of_rank(rank, world, n)           # weak scaling: 64 streams per GPU
    wl = Workload(n, args.gop, my_streams[0], local_rank, stream)
    batch = h263mi.Batch(n, W, H, local_rank, stream, overlap_post=args.overlap)
    d_rgba = h263mi.DeviceBuffer(n * RGBA_BYTES, local_rank)

    run_steps(batch, wl, d_rgba, 0, args.warmup)
    batch.sync()
    batch.timing_begin()
    # barrier + synchronize | exactly K steps | synchronize + barrier; MAX over ranks
    elapsed = shard.timed_region(dist, lambda: run_steps(batch, wl, d_rgba, args.warmup, args.steps),
                                 torch.cuda.synchronize)
    kt = batch.timing_end()
    batch.sync()

    pictures = shard.aggregate_pictures(dist, n * args.steps)
    value = pictures * MP_PER_PICTURE / elapsed

    # ---- roofline of the dominant kernel (HIP events on the launch stream, timed region only)
    g = len(wl.frames)
    recon_alg = sum(wl.recon_bytes(i % g) for i in range(args.warmup, args.warmup + args.steps)) / max(args.steps, 1)
    post_alg = wl.post_bytes()
    recon_avg_ms = kt.recon_ms / max(kt.recon_launches, 1)
    post_avg_ms = kt.post_ms / max(kt.post_launches, 1)
    kernels = {
        "k_recon": {"avg_ms": recon_avg_ms, "launches": kt.recon_launches, "alg_bytes_per_launch": recon_alg},
        "k_post": {"avg_ms": post_avg_ms, "launches": kt.post_launches, "alg_bytes_per_launch": post_alg},
    }
    for k in kernels.values():
        k["achieved_gbs"] = k["alg_bytes_per_launch"] / (k["avg_ms"] * 1e-3) / 1e9 if k["avg_ms"] > 0 else 0.0
    dom = "k_post" if kt.post_ms >= kt.recon_ms else "k_recon"
    # HBM traffic of that kernel from the PMC passes committed under profiles/ (FETCH_SIZE + WRITE_SIZE,
    # collected by tools/prof_final.sh in separate rocprofv3 --pmc runs of this same workload; null if absent)
    traffic = None
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
        if n == 64 and world == 1:
            traffic = tr["kernels"][dom]["hbm_bytes_per_launch"]
    except Exception:
        traffic = None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(kernels[dom]["achieved_gbs"], 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(kernels[dom]["achieved_gbs"] / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "avg_launch_ms": round(kernels[dom]["avg_ms"], 4),
                "alg_bytes_per_launch": int(kernels[dom]["alg_bytes_per_launch"]),
                "pipeline_achieved": round((recon_alg + post_alg) * args.steps / elapsed / 1e9, 1),
                "kernels": {k: {"avg_ms": round(v["avg_ms"], 4), "launches": v["launches"],
                                "alg_bytes_per_launch": int(v["alg_bytes_per_launch"]),
                                "achieved_gbs": round(v["achieved_gbs"], 1)} for k, v in kernels.items()}}

    extra = {}
    if rank == 0 and world == 1 and not args.no_extra:
        # BASELINE configs[1]: dense 1080p I pictures (every block Full): dequant + IDCT + YUV->RGBA, no deblock
        del wl
        dense = Workload(n, 1, 0, local_rank, stream, i_kind=h263mi.SYNTH_I_DENSE, p_frames=False)
        fr = dense.frames[0]
        for it in range(3):
            batch.submit(fr["ptype"], fr["mbs"].ptr, fr["co"].ptr, fr["base"].ptr)
            batch.render_rgba(0, d_rgba.ptr, None)
        batch.sync()
        reps = 20
        batch.timing_begin()
        t1 = time.perf_counter()
        for it in range(reps):
            batch.submit(fr["ptype"], fr["mbs"].ptr, fr["co"].ptr, fr["base"].ptr)
            batch.render_rgba(0, d_rgba.ptr, None)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        kd = batch.timing_end()
        alg = dense.recon_bytes(0) + dense.post_bytes()
        extra["config2_dense_iframe"] = {
            "mp_per_s": round(n * reps * MP_PER_PICTURE / dt, 1),
            "k_recon_avg_ms": round(kd.recon_ms / max(kd.recon_launches, 1), 4),
            "k_post_avg_ms": round(kd.post_ms / max(kd.post_launches, 1), 4),
            "alg_bytes_per_picture": int(alg / n), "pipeline_gbs": round(alg * reps / dt / 1e9, 1)}

    out = {
        "metric": "decoded megapixels/sec (IDCT+MC+YUV->RGB)",
        "value": round(value, 1), "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]: batch of %d independent 1920x1080 streams per GPU; step = one "
                               "frame index over the batch; GOP %d = 1 I (mixed block classes) + %d P (half-pel MVs "
                               "in [-32,31], 25%% coded blocks, quant 10); dequant+IDCT+MC+add/clip, deblock "
                               "strength %d, BT.601 RGBA; records pre-generated in HBM" % (n, args.gop, args.gop - 1, STRENGTH),
                   "streams_per_gpu": n, "width": W, "height": H, "gop": args.gop,
                   "parallelism": "streams sharded per GPU, no data-path collective"},
        "realtime_1080p30_streams": round(value / (MP_PER_PICTURE * 30), 1),
        "roofline": roofline,
    }
    if extra:
        out["extra"] = extra
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    batch.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
