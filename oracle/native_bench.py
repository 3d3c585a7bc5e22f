"""Native CPU baseline of bench.py: the C oracle rebuilt for THIS machine and timed with one stream per thread.

TEST / BENCH INFRASTRUCTURE ONLY (like everything under oracle/): used by bench.py's `cpu_baseline` leg and by
tests/.  BASELINE.md section 3 / SURVEY 8(d): `-O3 -march=native -ffp-contract=off`, (i) one thread, (ii) T threads
with one stream each, T = physical cores.  The build happens where the timing happens (`-march=native` code must
not travel between machines), into oracle/_native/, keyed by the CPU's flag set.
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

from . import oracle as orc

_HERE = os.path.dirname(os.path.abspath(__file__))
FLAGS = ["-O3", "-march=native", "-std=c11", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fwrapv", "-pthread"]


class BenchPicture(C.Structure):
    _fields_ = [("mbs", C.c_void_p), ("n_mbs", C.c_size_t), ("coeffs", C.c_void_p), ("n_blocks", C.c_size_t)]


def _cpu_key():
    try:
        txt = open("/proc/cpuinfo").read()
        flags = [ln for ln in txt.splitlines() if ln.startswith("flags")][0]
        model = [ln for ln in txt.splitlines() if ln.startswith("model name")][0]
    except Exception:
        flags, model = "unknown", "unknown"
    return hashlib.sha256((flags + model).encode()).hexdigest()[:12]


class NativeOracle:
    def __init__(self):
        out_dir = os.path.join(_HERE, "_native")
        os.makedirs(out_dir, exist_ok=True)
        srcs = [os.path.join(_HERE, "h263_oracle.c"), os.path.join(_HERE, "bench_streams.c")]
        self.path = os.path.join(out_dir, "libh263oracle_native_%s.so" % _cpu_key())
        if not os.path.exists(self.path) or any(os.path.getmtime(self.path) < os.path.getmtime(s) for s in srcs):
            subprocess.check_call(["gcc"] + FLAGS + ["-shared", "-o", self.path] + srcs)
        self.flags = "gcc " + " ".join(FLAGS)
        L = C.CDLL(self.path)
        orc.bind(L)
        L.orc_bench_streams.restype = C.c_double
        L.orc_bench_streams.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint16, C.c_uint16, C.POINTER(BenchPicture), C.c_int,
                                        C.c_uint8, C.POINTER(C.c_uint64)]
        self.L = L

    def check_against_portable(self, w, h, pictures, strength):
        """the native build must produce the portable (-O2) oracle's bytes before it is worth timing"""
        ref_n = ref_p = None
        cw = (w + 1) // 2
        for mbs, co in pictures:
            rc_n, ref_n = orc.decode_picture(w, h, mbs, co, ref_n, L=self.L)
            rc_p, ref_p = orc.decode_picture(w, h, mbs, co, ref_p)
            assert rc_n == rc_p == 0
            for a, b in zip(ref_n, ref_p):
                assert np.array_equal(a, b), "native oracle build differs from the portable one (planes)"
            fn = tuple(orc.deblock(p, pw, strength, L=self.L) for p, pw in zip(ref_n, (w, cw, cw)))
            fp = tuple(orc.deblock(p, pw, strength) for p, pw in zip(ref_p, (w, cw, cw)))
            for a, b in zip(fn, fp):
                assert np.array_equal(a, b), "native oracle build differs from the portable one (deblock)"
            assert np.array_equal(orc.yuv420_to_rgba(*fn, w, L=self.L), orc.yuv420_to_rgba(*fp, w)), \
                "native oracle build differs from the portable one (RGBA)"

    def run(self, w, h, streams, n_threads, n_gops, strength, checksums=None):
        """streams: list (distinct streams) of lists (frames) of (mbs, coeffs); returns wall seconds"""
        n_distinct, n_frames = len(streams), len(streams[0])
        keep, pics = [], (BenchPicture * (n_distinct * n_frames))()
        for s, frames in enumerate(streams):
            assert len(frames) == n_frames
            for f, (mbs, co) in enumerate(frames):
                mbs = np.ascontiguousarray(mbs, dtype=orc.MB_RECORD_DTYPE)
                co = np.ascontiguousarray(co, dtype=np.int16).reshape(-1, 64)
                keep.append((mbs, co))
                pics[s * n_frames + f] = BenchPicture(mbs.ctypes.data, mbs.size, co.ctypes.data if co.size else None,
                                                      co.shape[0])
        sums = (C.c_uint64 * n_threads)()
        secs = self.L.orc_bench_streams(n_threads, n_gops, n_frames, w, h, pics, n_distinct, strength, sums)
        if secs < 0:
            raise RuntimeError("orc_bench_streams failed: %r" % secs)
        if checksums is not None:
            checksums[:] = list(sums)
        return secs
