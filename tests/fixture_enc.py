"""ctypes access to tests/fixture_enc (the C++ Sorenson Spark writer and picture generator for fixtures).  Test and bench
infrastructure: what it writes is held byte for byte to tests/sorenson_enc.py (tests/test_fixture_enc.py)."""
import ctypes as C
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

# the record layout of include/h263mi.h (h263mi_mb_record, 32 bytes) -- kept here so that the workload generator imports
# nothing of the oracle (bench.py may use the oracle as its checker only)
MB_RECORD_DTYPE = np.dtype([
    ("mb_type", "u1"), ("quant", "u1"), ("cbp", "u1"), ("kill", "u1"),
    ("mv", "<i2", (4, 2)), ("intradc", "u1", (6,)), ("reserved", "u1", (2,)),
    ("coeff_index", "<u4"),
])

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        d = os.path.join(HERE, "fixture_enc")
        subprocess.check_call(["make", "-C", d, "-s"])
        L = C.CDLL(os.path.join(d, "libfixture_enc.so"))
        vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
        L.fx_encode.argtypes = [i32, i32, i32, i32, i32, i32, vp, sz, vp, vp, sz, C.POINTER(sz)]
        L.fx_picture.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, i32, i32, i32, i32, i32, vp, vp, sz, C.POINTER(sz), vp, sz,
                                 C.POINTER(sz)]
        _lib = L
    return _lib


def encode_picture(width, height, picture_type, pquant, mbs, coeffs, temporal_reference=0, deblock_flag=0):
    """records + coefficient blocks -> bytes: tests/sorenson_enc.py's encode_picture (Sorenson flavour, no stuffing / overflow
    options), in C++"""
    mbs = np.ascontiguousarray(mbs, MB_RECORD_DTYPE)
    co = np.ascontiguousarray(coeffs, np.int16).reshape(-1, 64)
    cap = 64 + len(mbs) * 64 + co.size * 4
    out = np.empty(cap, np.uint8)
    n = C.c_size_t(0)
    rc = lib().fx_encode(width, height, picture_type, pquant, temporal_reference, deblock_flag, mbs.ctypes.data, len(mbs),
                         co.ctypes.data if co.size else None, out.ctypes.data, cap, C.byref(n))
    if rc != 0:
        raise ValueError("fx_encode: records the syntax cannot express (rc %d)" % rc)
    return out[:n.value].tobytes()


def picture(seed, stream, frame, width, height, intra, pquant, deblock_flag=1, with_records=False):
    """picture `frame` of stream `stream` of the generated corpus: bytes, or (bytes, records, coefficient blocks)"""
    n = ((width + 15) // 16) * ((height + 15) // 16)
    mbs = np.zeros(n, MB_RECORD_DTYPE)
    co = np.zeros((6 * n, 64), np.int16)
    cap = 64 + n * 64 + 6 * n * 24
    out = np.empty(cap, np.uint8)
    nb, ny = C.c_size_t(0), C.c_size_t(0)
    rc = lib().fx_picture(seed, stream, frame, width, height, int(bool(intra)), pquant, deblock_flag, mbs.ctypes.data, co.ctypes.data,
                          6 * n, C.byref(nb), out.ctypes.data, cap, C.byref(ny))
    if rc != 0:
        raise ValueError("fx_picture failed (rc %d)" % rc)
    data = out[:ny.value].tobytes()
    return (data, mbs, co[:nb.value].copy()) if with_records else data


def corpus(seed, n_streams, n_frames, width, height, quants, deblock_flag=1, threads=8):
    """[stream][frame] -> bytes: every picture distinct (frame 0 of a stream is its key frame); ctypes releases the GIL, so
    the pictures are generated on `threads` threads"""
    lib()
    jobs = [(s, f) for s in range(n_streams) for f in range(n_frames)]
    with ThreadPoolExecutor(max_workers=threads) as ex:
        pics = list(ex.map(lambda sf: picture(seed, sf[0], sf[1], width, height, sf[1] == 0, quants[sf[0]], deblock_flag), jobs))
    return [[pics[s * n_frames + f] for f in range(n_frames)] for s in range(n_streams)]
