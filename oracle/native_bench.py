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
        srcs = [os.path.join(_HERE, "h263_oracle.c"), os.path.join(_HERE, "bench_streams.c"), os.path.join(_HERE, "simd_stages.c")]
        self.srcs = srcs
        self.path = os.path.join(out_dir, "libh263oracle_native_%s.so" % _cpu_key())
        if not os.path.exists(self.path) or any(os.path.getmtime(self.path) < os.path.getmtime(s) for s in srcs):
            # (built under a private name and moved into place: several test workers may get here at the same time)
            tmp = "%s.%d.tmp" % (self.path, os.getpid())
            subprocess.check_call(["gcc"] + FLAGS + ["-shared", "-o", tmp] + srcs)
            os.replace(tmp, self.path)
        self.flags = "gcc " + " ".join(FLAGS)
        L = C.CDLL(self.path)
        orc.bind(L)
        L.orc_bench_streams.restype = C.c_double
        L.orc_bench_streams.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint16, C.c_uint16, C.POINTER(BenchPicture), C.c_int,
                                        C.c_uint8, C.POINTER(C.c_uint64)]
        L.orc_bench_stages.restype = C.c_double
        L.orc_bench_stages.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint16, C.c_uint16, C.POINTER(BenchPicture), C.c_int,
                                       C.c_uint8, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        u8p = C.POINTER(C.c_uint8)
        L.orc_deblock_simd.restype = C.c_int
        L.orc_deblock_simd.argtypes = [u8p, C.c_size_t, C.c_size_t, C.c_uint8, u8p]
        L.orc_yuv420_to_rgba_simd.restype = C.c_int
        L.orc_yuv420_to_rgba_simd.argtypes = [u8p, C.c_size_t, u8p, u8p, C.c_size_t, C.c_size_t, u8p]
        self.L = L

    # ---- the explicit 128-bit forms of deblock / BT.601 (simd_stages.c): baseline material, checked against the oracle
    def deblock_simd(self, plane, width, strength):
        plane = np.ascontiguousarray(plane, dtype=np.uint8).ravel()
        out = np.empty_like(plane)
        u8p = C.POINTER(C.c_uint8)
        rc = self.L.orc_deblock_simd(plane.ctypes.data_as(u8p), plane.size, width, strength, out.ctypes.data_as(u8p))
        assert rc == 0, rc
        return out

    def yuv420_to_rgba_simd(self, y, cb, cr, width):
        y, cb, cr = (np.ascontiguousarray(p, dtype=np.uint8).ravel() for p in (y, cb, cr))
        out = np.empty(y.size * 4, np.uint8)
        u8p = C.POINTER(C.c_uint8)
        rc = self.L.orc_yuv420_to_rgba_simd(y.ctypes.data_as(u8p), y.size, cb.ctypes.data_as(u8p), cr.ctypes.data_as(u8p),
                                            cb.size, width, out.ctypes.data_as(u8p))
        assert rc == 0, rc
        return out

    def vectorisation_report(self):
        """What gcc's auto-vectoriser made of the oracle's loops at the baseline's flags (-fopt-info-vec-optimized), per
        function: BASELINE.md section 3 wants to know whether deblock and BT.601 were vectorised -- the reference's are
        explicit SIMD (deblock.rs:99-127, bt601.rs:12-59)."""
        src = self.srcs[0]
        try:
            out = subprocess.run(["gcc"] + FLAGS + ["-fopt-info-vec-optimized", "-c", src, "-o", os.devnull],
                                 capture_output=True, text=True, timeout=120).stderr
        except Exception as e:                       # pragma: no cover
            return {"error": str(e)}
        # function extents of the oracle source: a line that starts an identifier at column 0 and ends in ')' or '{'
        import re
        starts = []
        for i, ln in enumerate(open(src).read().splitlines(), 1):
            m = re.match(r"^(?:static\s+)?(?:inline\s+)?[a-z_0-9]+\s+\**([a-z_0-9]+)\(", ln)
            if m and not ln.rstrip().endswith(";"):
                starts.append((i, m.group(1)))
        hits = {}
        for m in re.finditer(r"h263_oracle\.c:(\d+):\d+: optimized: (loop|basic block part) vectorized using (\d+) byte vectors", out):
            line = int(m.group(1))
            fn = [name for (at, name) in starts if at <= line]
            hits.setdefault(fn[-1] if fn else "?", set()).add("%s, %s-byte vectors" % (m.group(2), m.group(3)))
        want = ("idct_1d", "orc_idct_channel", "gather_block", "deblock_horiz", "deblock_vert", "orc_deblock_process_simd_lane",
                "orc_yuv420_to_rgba", "yuv_to_rgba_1")
        return {fn: sorted(hits[fn]) if fn in hits else "not vectorised" for fn in want}

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

    def check_simd_stages(self, w, h, pictures, strength):
        """the explicit-SIMD deblock / BT.601 of simd_stages.c must produce the oracle's bytes before they are timed"""
        ref = None
        cw = (w + 1) // 2
        for mbs, co in pictures:
            rc, ref = orc.decode_picture(w, h, mbs, co, ref, L=self.L)
            assert rc == 0
        want = tuple(orc.deblock(p, pw, strength, L=self.L) for p, pw in zip(ref, (w, cw, cw)))
        got = tuple(self.deblock_simd(p, pw, strength) for p, pw in zip(ref, (w, cw, cw)))
        for a, b in zip(got, want):
            assert np.array_equal(a, np.asarray(b).ravel()), "simd_stages.c: deblock differs from the oracle"
        assert np.array_equal(self.yuv420_to_rgba_simd(*want, w), np.asarray(orc.yuv420_to_rgba(*want, w, L=self.L)).ravel()), \
            "simd_stages.c: BT.601 differs from the oracle"

    RECON, DEBLOCK, RGBA = 1, 2, 4

    def run(self, w, h, streams, n_threads, n_gops, strength, checksums=None, stages=7, simd=False):
        """streams: list (distinct streams) of lists (frames) of (mbs, coeffs); returns wall seconds.  stages: RECON |
        DEBLOCK | RGBA bits (a stage without the reconstruction runs on one fixed picture); simd: deblock / BT.601 in
        the reference's explicit 128-bit shape (simd_stages.c)"""
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
        secs = self.L.orc_bench_stages(n_threads, n_gops, n_frames, w, h, pics, n_distinct, strength, stages, 1 if simd else 0, sums)
        if secs < 0:
            raise RuntimeError("orc_bench_streams failed: %r" % secs)
        if checksums is not None:
            checksums[:] = list(sums)
        return secs
