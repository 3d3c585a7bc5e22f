"""ctypes binding of the CPU oracle (oracle/h263_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by bench.py's `cpu_baseline` leg and by
__graft_entry__.smoke() as the checker.  The product (h263-rs_amd/) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libh263oracle.so")

MB_RECORD_DTYPE = np.dtype([
    ("mb_type", "u1"), ("quant", "u1"), ("cbp", "u1"), ("kill", "u1"),
    ("mv", "<i2", (4, 2)), ("intradc", "u1", (6,)), ("reserved", "u1", (2,)),
    ("coeff_index", "<u4"),
])
assert MB_RECORD_DTYPE.itemsize == 32

ORC_ZERO, ORC_DC, ORC_HORIZ, ORC_VERT, ORC_FULL = range(5)
ERR_UNCODED_IFRAME_BLOCKS = -15
ERR_INVALID_ARGUMENT = -100


class DctBlock(C.Structure):
    _fields_ = [("tag", C.c_int32), ("v", C.c_float * 64)]


class Block(C.Structure):
    _fields_ = [("has_intradc", C.c_int32), ("intradc", C.c_uint8), ("n_tcoef", C.c_int32),
                ("run", C.c_uint8 * 80), ("level", C.c_int16 * 80)]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "h263_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def bind(L):
    """declare the orc_* signatures on a loaded library handle (the portable build below, or the native one of
    oracle/native_bench.py)"""
    u8p = C.POINTER(C.c_uint8)
    L.orc_intradc_into_level.restype = C.c_int16
    L.orc_intradc_into_level.argtypes = [C.c_uint8]
    L.orc_average_sum_of_mvs.restype = C.c_int16
    L.orc_average_sum_of_mvs.argtypes = [C.c_int16]
    L.orc_lerp_parameters.restype = None
    L.orc_lerp_parameters.argtypes = [C.c_int16, C.POINTER(C.c_int16), C.POINTER(C.c_int)]
    L.orc_inverse_rle.restype = None
    L.orc_inverse_rle.argtypes = [C.POINTER(Block), C.POINTER(DctBlock), C.c_size_t, C.c_size_t,
                                  C.c_size_t, C.c_uint8]
    L.orc_idct_channel.restype = None
    L.orc_idct_channel.argtypes = [C.POINTER(DctBlock), C.c_size_t, C.c_void_p, C.c_size_t,
                                   C.c_size_t, C.c_size_t]
    L.orc_gather.restype = C.c_int
    L.orc_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_decode_picture.restype = C.c_int
    L.orc_decode_picture.argtypes = [C.c_uint16, C.c_uint16, C.c_void_p, C.c_size_t, C.c_void_p,
                                     C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_deblock.restype = C.c_int
    L.orc_deblock.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint8, C.c_void_p]
    for name in ("orc_deblock_process_scalar", "orc_deblock_process_simd_lane"):
        f = getattr(L, name)
        f.restype = None
        f.argtypes = [u8p, u8p, u8p, u8p, C.c_uint8]
    L.orc_yuv420_to_rgba.restype = C.c_int
    L.orc_yuv420_to_rgba.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_size_t, C.c_void_p]
    L.orc_basis_table.restype = C.POINTER(C.c_float)
    L.orc_basis_table.argtypes = []
    return L


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = bind(C.CDLL(_LIB_PATH))
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def chroma_dims(w, h):
    return (w + 1) // 2, (h + 1) // 2


def mb_dims(w, h):
    return (w + 15) // 16, (h + 15) // 16


def basis_table():
    p = lib().orc_basis_table()
    return np.ctypeslib.as_array(p, shape=(8, 8)).copy()


def quant_to_strength():
    return np.array((C.c_uint8 * 32).in_dll(lib(), "orc_quant_to_strength"), dtype=np.uint8)


def process_scalar(a, b, c, d, strength):
    v = [C.c_uint8(x) for x in (a, b, c, d)]
    lib().orc_deblock_process_scalar(*[C.byref(x) for x in v], strength)
    return tuple(x.value for x in v)


def process_simd_lane(a, b, c, d, strength):
    v = [C.c_uint8(x) for x in (a, b, c, d)]
    lib().orc_deblock_process_simd_lane(*[C.byref(x) for x in v], strength)
    return tuple(x.value for x in v)


def deblock(data, width, strength, L=None):
    """deblock::deblock(data, width, strength) -> new buffer (deblock.rs:305)."""
    data = np.ascontiguousarray(data, dtype=np.uint8).ravel()
    out = np.empty_like(data)
    rc = (L or lib()).orc_deblock(_ptr(data), data.size, width, strength, _ptr(out))
    if rc != 0:
        raise ValueError("orc_deblock rc=%d" % rc)
    return out


def yuv420_to_rgba(y, cb, cr, y_width, L=None):
    """yuv::bt601::yuv420_to_rgba(y, chroma_b, chroma_r, y_width) (bt601.rs:105)."""
    y = np.ascontiguousarray(y, dtype=np.uint8).ravel()
    cb = np.ascontiguousarray(cb, dtype=np.uint8).ravel()
    cr = np.ascontiguousarray(cr, dtype=np.uint8).ravel()
    out = np.empty(y.size * 4, dtype=np.uint8)
    if cb.size != cr.size:
        raise ValueError("chroma size mismatch")
    rc = (L or lib()).orc_yuv420_to_rgba(_ptr(y), y.size, _ptr(cb), _ptr(cr), cb.size, y_width, _ptr(out))
    if rc != 0:
        raise ValueError("orc_yuv420_to_rgba rc=%d" % rc)
    return out


def inverse_rle(has_intradc, intradc, runs, levels, quant):
    """Returns (tag, float32[64]) for a single block placed at (0,0)."""
    b = Block()
    b.has_intradc = 1 if has_intradc else 0
    b.intradc = intradc
    b.n_tcoef = len(runs)
    for i, (r, l) in enumerate(zip(runs, levels)):
        b.run[i] = r
        b.level[i] = l
    d = DctBlock()
    d.tag = ORC_ZERO
    lib().orc_inverse_rle(C.byref(b), C.byref(d), 0, 0, 1, quant)
    return d.tag, np.array(d.v[:], dtype=np.float32)


def idct_blocks(tags, values, pred, blk_per_line, width):
    """idct_channel over a plane.  tags: int[n]; values: float32[n,64]; pred: uint8 plane (flat)."""
    n = len(tags)
    arr = (DctBlock * n)()
    for i in range(n):
        arr[i].tag = int(tags[i])
        for k in range(64):
            arr[i].v[k] = float(values[i][k])
    out = np.ascontiguousarray(pred, dtype=np.uint8).ravel().copy()
    lib().orc_idct_channel(arr, n, _ptr(out), out.size, blk_per_line, width)
    return out


def gather(mb_types, mvs, ref, width, height):
    """gather(); ref = (y, cb, cr) or None.  Returns (rc, (y, cb, cr))."""
    mbw, mbh = mb_dims(width, height)
    cw, ch = chroma_dims(width, height)
    mb_types = np.ascontiguousarray(mb_types, dtype=np.uint8)
    mvs = np.ascontiguousarray(mvs, dtype=np.int16).reshape(-1, 4, 2)
    ny = np.zeros(width * height, np.uint8)
    ncb = np.zeros(cw * ch, np.uint8)
    ncr = np.zeros(cw * ch, np.uint8)
    r = [None, None, None] if ref is None else [np.ascontiguousarray(p, dtype=np.uint8).ravel() for p in ref]
    rc = lib().orc_gather(_ptr(mb_types), _ptr(mvs), mb_types.size, _ptr(r[0]), _ptr(r[1]), _ptr(r[2]),
                          width, height, mbw, _ptr(ny), _ptr(ncb), _ptr(ncr))
    return rc, (ny, ncb, ncr)


def decode_picture(width, height, mbs, coeffs, ref=None, L=None):
    """Record-level reconstruction (tail of decode_next_picture, state.rs:421-458).

    mbs: structured array MB_RECORD_DTYPE (len <= mbw*mbh); coeffs: int16[n_blocks, 64];
    ref: (y, cb, cr) flat uint8 arrays or None.  Returns (rc, (y, cb, cr))."""
    cw, ch = chroma_dims(width, height)
    mbs = np.ascontiguousarray(mbs, dtype=MB_RECORD_DTYPE)
    coeffs = np.ascontiguousarray(coeffs, dtype=np.int16).reshape(-1, 64)
    oy = np.empty(width * height, np.uint8)
    ocb = np.empty(cw * ch, np.uint8)
    ocr = np.empty(cw * ch, np.uint8)
    r = [None, None, None] if ref is None else [np.ascontiguousarray(p, dtype=np.uint8).ravel() for p in ref]
    rc = (L or lib()).orc_decode_picture(width, height, _ptr(mbs), mbs.size, _ptr(coeffs), coeffs.shape[0],
                                  _ptr(r[0]), _ptr(r[1]), _ptr(r[2]), _ptr(oy), _ptr(ocb), _ptr(ocr))
    return rc, (oy, ocb, ocr)
