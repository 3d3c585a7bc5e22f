"""ctypes access to the CPU-only build of the host bitstream parser (tests/parser)."""
import ctypes as C
import os
import subprocess

import numpy as np

from oracle.oracle import MB_RECORD_DTYPE

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
TCOEF, MCBPC_I, MCBPC_P, CBPY, MVD = range(5)
EOF_ERR = -16


class PictureDesc(C.Structure):
    _fields_ = [("width", C.c_uint16), ("height", C.c_uint16), ("picture_type", C.c_uint8), ("pquant", C.c_uint8),
                ("use_deblocker", C.c_uint8), ("reserved0", C.c_uint8), ("temporal_reference", C.c_uint16),
                ("reserved1", C.c_uint16)]


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", os.path.join(HERE, "parser"), "-s"])
        L = C.CDLL(os.path.join(HERE, "parser", "libh263parse_test.so"))
        L.pt_read_vlc.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.pt_decode_block.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.POINTER(C.c_size_t)]
        L.pt_reader_script.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.pt_reader_script.restype = None
        L.pt_parse_picture.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(PictureDesc), C.c_void_p,
                                       C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                       C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_int]
        L.pt_parse_header.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_void_p]
        L.pt_read_umv.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.pt_read_umv.restype = None
        L.pt_context_reset.restype = None
        L.pt_compare_parser_paths.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_int)]
        L.pt_compare_record_destinations.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_size_t, C.POINTER(C.c_int),
                                                     C.POINTER(C.c_int)]
        _lib = L
    return _lib


def _bytes(data):
    a = np.frombuffer(bytes(bytearray(data)), dtype=np.uint8).copy()
    return a if a.size else np.zeros(1, np.uint8), len(data)


def read_vlc(table, data, n):
    a, ln = _bytes(data)
    out = np.zeros((n, 4), np.int32)
    k = lib().pt_read_vlc(table, a.ctypes.data, ln, n, out.ctypes.data)
    return out[:k]


def decode_block(data, sorenson, version, intra, tcoef_present):
    a, ln = _bytes(data)
    out = np.zeros(4 + 3 * 80, np.int32)
    used = C.c_size_t()
    rc = lib().pt_decode_block(a.ctypes.data, ln, int(sorenson), -1 if version is None else version, int(intra),
                               int(tcoef_present), out.ctypes.data, C.byref(used))
    n = int(out[3])
    return rc, bool(out[1]), int(out[2]), [tuple(int(v) for v in out[4 + 3 * i:7 + 3 * i]) for i in range(n)], used.value


def reader_script(data, ops):
    a, ln = _bytes(data)
    o = np.array(ops, np.int32).reshape(-1, 2)
    res = np.zeros((len(o), 2), np.int64)
    lib().pt_reader_script(a.ctypes.data, ln, len(o), o.ctypes.data, res.ctypes.data)
    return [(int(r), int(v)) for r, v in res]


def compare_parser_paths(data, options=1):
    """(difference code, return code): the windowed fast paths of parse_picture against its field-by-field form on the
    same bytes; difference code 0 = every output agrees."""
    a, ln = _bytes(data)
    rc = C.c_int()
    diff = lib().pt_compare_parser_paths(a.ctypes.data, ln, options, C.byref(rc))
    return diff, rc.value


def compare_record_destinations(data, cap, options=1):
    """(difference code, return code, used the caller's array): parse_picture writing its records into a caller's
    array of `cap` records (ParsedPicture::mbs_ext) against the same parse into its own vector."""
    a, ln = _bytes(data)
    rc, used = C.c_int(), C.c_int()
    diff = lib().pt_compare_record_destinations(a.ctypes.data, ln, options, cap, C.byref(rc), C.byref(used))
    return diff, rc.value, bool(used.value)


def context_reset():
    """forget the last decoded picture (a fresh H263State)"""
    lib().pt_context_reset()


HEADER_FIELDS = ("rc is_picture picture_type width height format_kind options has_plusptype has_opptype mv_range "
                 "quantizer temporal_reference n_extra bits_used").split()


def parse_header(data, options=0, use_context=False):
    a, ln = _bytes(data)
    out = np.zeros(len(HEADER_FIELDS), np.int32)
    lib().pt_parse_header(a.ctypes.data, ln, options, int(use_context), out.ctypes.data)
    return dict(zip(HEADER_FIELDS, (int(v) for v in out)))


def read_umv(data):
    a, ln = _bytes(data)
    out = np.zeros(3, np.int32)
    lib().pt_read_umv(a.ctypes.data, ln, out.ctypes.data)
    return tuple(int(v) for v in out)


def parse_picture(data, options=1, cap_mbs=20000, cap_blocks=120000, use_context=False):
    a, ln = _bytes(data)
    d = PictureDesc()
    mbs = np.zeros(cap_mbs, MB_RECORD_DTYPE)
    co = np.zeros((cap_blocks, 64), np.int16)
    n_mbs, n_blocks, bits = C.c_size_t(), C.c_size_t(), C.c_size_t()
    rc = lib().pt_parse_picture(a.ctypes.data, ln, options, C.byref(d), mbs.ctypes.data, cap_mbs, co.ctypes.data,
                                cap_blocks, C.byref(n_mbs), C.byref(n_blocks), C.byref(bits), int(use_context))
    return rc, d, mbs[:n_mbs.value].copy(), co[:n_blocks.value].copy(), bits.value


def parse_picture_limited(data, max_w, max_h, options=1):
    """(rc, 32-bit words the parser's arrays were sized to) with ParsedPicture::size_fits = (w <= max_w and h <= max_h);
    max_w = 0: no limit"""
    a, ln = _bytes(data)
    words = C.c_size_t()
    f = lib().pt_parse_picture_limited
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_size_t)]
    rc = f(a.ctypes.data, ln, options, max_w, max_h, C.byref(words))
    return rc, words.value


def parse_picture_events(data, options=1, cap_blocks=120000, cap_events=2000000):
    """(rc, block_first_event, events) of the product's form of the parse (events only): coded block k of the picture owns
    events[first[k]:first[k + 1]], each LEVEL << 16 | raster position"""
    a, ln = _bytes(data)
    first = np.zeros(cap_blocks + 1, np.uint32)
    ev = np.zeros(cap_events, np.uint32)
    nb, ne = C.c_size_t(), C.c_size_t()
    f = lib().pt_parse_picture_events
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                  C.POINTER(C.c_size_t)]
    rc = f(a.ctypes.data, ln, options, first.ctypes.data, cap_blocks + 1, ev.ctypes.data, cap_events, C.byref(nb), C.byref(ne))
    return rc, first[:nb.value + 1].copy(), ev[:ne.value].copy()
