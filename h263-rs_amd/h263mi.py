"""ctypes binding of libh263mi.so (the C ABI of include/h263mi.h) for the tests and bench.py.

Python here is plumbing only: every compute call goes through the C ABI into the gfx950
kernels.  There is no fallback -- if the library or a GPU is missing, calls raise.

The class and function names mirror the reference API (ruffle-rs/h263-rs):
  H263State.decode_next_picture / get_last_picture / as_yuv   h263/src/decoder/state.rs
  deblock(data, width, strength), QUANT_TO_STRENGTH           deblock/src/deblock.rs:5-8,305
  yuv420_to_rgba(y, chroma_b, chroma_r, y_width)              yuv/src/bt601.rs:105
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("H263MI_LIB", os.path.join(_HERE, "libh263mi.so"))   # override: A/B runs of two builds

OK = 0
# h263/src/error.rs:6-58, in order (include/h263mi.h)
(ERR_INTERNAL_DECODER_ERROR, ERR_MIDDLE_OF_BITSTREAM, ERR_INVALID_MACROBLOCK_HEADER, ERR_INVALID_MACROBLOCK_CODED_BITS,
 ERR_INVALID_INTRA_DC, ERR_INVALID_SHORT_COEFFICIENT, ERR_INVALID_LONG_COEFFICIENT, ERR_INVALID_MVD, ERR_INVALID_PTYPE,
 ERR_INVALID_PLUS_PTYPE, ERR_INVALID_GOB_HEADER, ERR_INVALID_BITSTREAM, ERR_PICTURE_FORMAT_MISSING,
 ERR_PICTURE_FORMAT_INVALID, ERR_UNCODED_IFRAME_BLOCKS, ERR_UNHANDLED_IO_ERROR, ERR_UNIMPLEMENTED_DECODING) = range(-1, -18, -1)
ERR_INVALID_ARGUMENT = -100
ERR_NO_DEVICE = -101
ERR_HIP = -102
ERR_OUT_OF_MEMORY = -103
ERR_NO_PICTURE = -104

def events_from_dense(coeffs, intra_blocks=None):
    """dense (n, 64) LEVEL blocks -> (block_first_event, events) of h263mi_submit_picture_events.  intra_blocks: mask
    of blocks whose element 0 is not a TCOEF (the DC of an intra block travels in the record)."""
    c = np.ascontiguousarray(coeffs, np.int16).reshape(-1, 64)
    nz = c != 0
    if intra_blocks is not None:
        nz[np.asarray(intra_blocks, bool), 0] = False
    counts = nz.sum(axis=1)
    first = np.zeros(len(c) + 1, np.uint32)
    np.cumsum(counts, out=first[1:])
    blk, pos = np.nonzero(nz)
    ev = (c[blk, pos].astype(np.uint16).astype(np.uint32) << 16) | pos.astype(np.uint32)
    return first, ev


SORENSON_SPARK_BITSTREAM = 1
USE_SCALABILITY_MODE = 2
PICTURE_I, PICTURE_P, PICTURE_DISPOSABLE_P = 0, 1, 2
(PICTURE_RESERVED_SORENSON, PICTURE_PB, PICTURE_IMPROVED_PB, PICTURE_B, PICTURE_EI, PICTURE_EP,
 PICTURE_RESERVED) = range(3, 10)
SYNTH_I_DENSE, SYNTH_I_MIXED, SYNTH_P = 0, 1, 2

MB_RECORD_DTYPE = np.dtype([
    ("mb_type", "u1"), ("quant", "u1"), ("cbp", "u1"), ("kill", "u1"),
    ("mv", "<i2", (4, 2)), ("intradc", "u1", (6,)), ("reserved", "u1", (2,)),
    ("coeff_index", "<u4"),
])

EXPORTS = [
    "h263mi_strerror", "h263mi_abi_version",
    "h263mi_state_new", "h263mi_state_free", "h263mi_state_is_sorenson", "h263mi_state_reset",
    "h263mi_state_cleanup_buffers", "h263mi_submit_picture", "h263mi_decode_next_picture",
    "h263mi_parse_picture_header",
    "h263mi_get_last_picture", "h263mi_get_reference_picture", "h263mi_copy_yuv", "h263mi_render_rgba",
    "h263mi_quant_to_strength", "h263mi_deblock", "h263mi_bt601_yuv420_to_rgba", "h263mi_deblock_on",
    "h263mi_bt601_yuv420_to_rgba_on",
    "h263mi_batch_create", "h263mi_batch_destroy", "h263mi_batch_mbs_per_picture", "h263mi_batch_submit",
    "h263mi_batch_decode", "h263mi_batch_decode_events", "h263mi_batch_decode_next_pictures", "h263mi_batch_decode_next_pictures_ex",
    "h263mi_batch_sync_streams", "h263mi_batch_reset_stream", "h263mi_batch_set_active", "h263mi_batch_stream_has_picture",
    "h263mi_batch_render_rgba", "h263mi_batch_sync", "h263mi_batch_reset", "h263mi_batch_copy_yuv",
    "h263mi_batch_timing_begin", "h263mi_batch_timing_end", "h263mi_batch_timing_reserve", "h263mi_probe_bandwidth",
    "h263mi_probe_bandwidth_shape",
    "h263mi_batch_submit_host", "h263mi_submit_picture_events", "h263mi_batch_submit_host_events",
    "h263mi_device_count", "h263mi_device_malloc", "h263mi_device_free", "h263mi_device_memcpy_h2d",
    "h263mi_device_memcpy_d2h", "h263mi_device_synchronize",
    "h263mi_synth_picture_host", "h263mi_synth_batch_device", "h263mi_synth_batch_device_strided",
    "h263mi_default_parser_threads",
    "h263mi_render_rgba_pinned", "h263mi_host_alloc", "h263mi_host_free", "h263mi_host_register", "h263mi_host_unregister",
    "h263mi_debug_fail_nth_hip_call",
    "h263mi_mixed_create", "h263mi_mixed_destroy", "h263mi_mixed_decode_next_pictures", "h263mi_mixed_sync",
    "h263mi_mixed_stream_size", "h263mi_mixed_size_classes", "h263mi_mixed_set_memory_limit",
    "h263mi_mixed_frame_store_bytes", "h263mi_mixed_copy_yuv", "h263mi_mixed_reset_stream",
    # ABI 7: one post-filter strength per stream, explicit host share, NUMA placement report
    "h263mi_batch_decode_ps", "h263mi_batch_decode_events_ps", "h263mi_batch_render_rgba_ps",
    "h263mi_batch_decode_next_pictures_ps", "h263mi_mixed_decode_next_pictures_ps",
    "h263mi_set_ranks_per_node", "h263mi_batch_host_placement", "h263mi_debug_host_placement",
]
STRENGTH_FROM_HEADER = 0xFF
CFG_OVERLAP_POST, CFG_PIPELINE_POST, CFG_TRUSTED_ARRAYS = 1, 2, 4


class H263Error(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        msg = lib().h263mi_strerror(code).decode() if _lib is not None else str(code)
        super().__init__("%s: %s (%d)" % (what, msg, code))


class PictureDesc(C.Structure):
    _fields_ = [("width", C.c_uint16), ("height", C.c_uint16), ("picture_type", C.c_uint8), ("pquant", C.c_uint8),
                ("use_deblocker", C.c_uint8), ("reserved0", C.c_uint8), ("temporal_reference", C.c_uint16),
                ("reserved1", C.c_uint16)]


class BackendCfg(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("flags", C.c_uint32), ("stream", C.c_void_p)]


class FrameView(C.Structure):
    _fields_ = [("width", C.c_uint16), ("height", C.c_uint16), ("chroma_width", C.c_uint16),
                ("chroma_height", C.c_uint16), ("temporal_reference", C.c_uint16), ("picture_type", C.c_uint8),
                ("pquant", C.c_uint8), ("use_deblocker", C.c_uint8), ("reserved", C.c_uint8 * 3),
                ("dev_y", C.c_void_p), ("dev_cb", C.c_void_p), ("dev_cr", C.c_void_p),
                ("dev_pitch_y", C.c_uint32), ("dev_pitch_c", C.c_uint32)]


class KernelTimes(C.Structure):
    _fields_ = [("recon_ms", C.c_double), ("recon_launches", C.c_uint32), ("pad0", C.c_uint32),
                ("post_ms", C.c_double), ("post_launches", C.c_uint32), ("pad1", C.c_uint32),
                ("frame_ms", C.c_double), ("frame_launches", C.c_uint32), ("pad2", C.c_uint32)]


def build(force=False):
    """hipcc --offload-arch=gfx950 build of the shared library (h263-rs_amd/Makefile)."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "-s", "clean"])
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return LIB_PATH


_lib = None


class _MissingExport:
    argtypes = restype = None

    def __init__(self, name):
        self.name = name

    def __call__(self, *a):
        raise RuntimeError("%s: this build of libh263mi.so does not export %s" % (LIB_PATH, self.name))


def _has(name):
    return not isinstance(getattr(lib(), name, None), _MissingExport)


def lib():
    """Load the C-ABI library; fails loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libh263mi.so is missing: run `make -C h263-rs_amd` (hipcc, gfx950). "
                               "There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        # A library of an EARLIER round (tools/ab_inproc.py runs them beside HEAD) lacks the newer exports: a stand-in takes
        # the signature assignments below and raises when called.  (tests/test_abi.py holds HEAD's library to every export.)
        for name in EXPORTS:
            try:
                getattr(L, name)
            except AttributeError:
                setattr(L, name, _MissingExport(name))
        vp, sz, u8, u16, u32, i32 = C.c_void_p, C.c_size_t, C.c_uint8, C.c_uint16, C.c_uint32, C.c_int
        L.h263mi_strerror.restype = C.c_char_p
        L.h263mi_strerror.argtypes = [i32]
        L.h263mi_state_new.argtypes = [u32, C.POINTER(BackendCfg), C.POINTER(vp)]
        L.h263mi_state_free.argtypes = [vp]
        L.h263mi_state_free.restype = None
        L.h263mi_state_is_sorenson.argtypes = [vp]
        L.h263mi_state_reset.argtypes = [vp]
        L.h263mi_state_cleanup_buffers.argtypes = [vp]
        L.h263mi_submit_picture.argtypes = [vp, C.POINTER(PictureDesc), vp, sz, vp, sz]
        L.h263mi_submit_picture_events.argtypes = [vp, C.POINTER(PictureDesc), vp, sz, vp, sz, vp, sz]
        L.h263mi_decode_next_picture.argtypes = [vp, vp, sz, C.POINTER(sz)]
        L.h263mi_parse_picture_header.argtypes = [vp, vp, sz, C.POINTER(PictureDesc)]
        L.h263mi_get_last_picture.argtypes = [vp, C.POINTER(FrameView)]
        L.h263mi_get_reference_picture.argtypes = [vp, C.POINTER(FrameView)]
        L.h263mi_copy_yuv.argtypes = [vp, vp, vp, vp]
        L.h263mi_render_rgba.argtypes = [vp, u8, vp]
        L.h263mi_deblock.argtypes = [vp, sz, sz, u8, vp]
        L.h263mi_bt601_yuv420_to_rgba.argtypes = [vp, sz, vp, vp, sz, sz, vp]
        L.h263mi_batch_create.argtypes = [u32, u16, u16, C.POINTER(BackendCfg), C.POINTER(vp)]
        L.h263mi_batch_destroy.argtypes = [vp]
        L.h263mi_batch_destroy.restype = None
        L.h263mi_batch_mbs_per_picture.argtypes = [vp]
        L.h263mi_batch_mbs_per_picture.restype = u32
        L.h263mi_batch_submit.argtypes = [vp, u8, vp, vp, vp]
        L.h263mi_batch_decode.argtypes = [vp, u8, vp, vp, vp, C.c_uint64, u8, vp, vp]
        L.h263mi_batch_decode_events.argtypes = [vp, u8, vp, vp, vp, vp, C.c_uint64, C.c_uint64, u8, vp, vp]
        L.h263mi_render_rgba_pinned.argtypes = [vp, u8, vp]
        L.h263mi_host_alloc.argtypes = [sz, C.POINTER(vp)]
        L.h263mi_host_free.argtypes = [vp]
        L.h263mi_host_register.argtypes = [vp, sz]
        L.h263mi_host_unregister.argtypes = [vp]
        L.h263mi_debug_fail_nth_hip_call.argtypes = [i32]
        L.h263mi_mixed_create.argtypes = [u32, C.POINTER(BackendCfg), C.POINTER(vp)]
        L.h263mi_mixed_destroy.argtypes = [vp]
        L.h263mi_mixed_destroy.restype = None
        L.h263mi_mixed_decode_next_pictures.argtypes = [vp, u32, vp, vp, vp, u32, vp, u8, vp, vp, vp]
        L.h263mi_mixed_sync.argtypes = [vp, vp]
        L.h263mi_mixed_stream_size.argtypes = [vp, u32, C.POINTER(u16), C.POINTER(u16)]
        L.h263mi_mixed_size_classes.argtypes = [vp]
        L.h263mi_mixed_size_classes.restype = u32
        L.h263mi_mixed_set_memory_limit.argtypes = [vp, C.c_uint64]
        L.h263mi_mixed_frame_store_bytes.argtypes = [vp]
        L.h263mi_mixed_frame_store_bytes.restype = C.c_uint64
        L.h263mi_mixed_copy_yuv.argtypes = [vp, u32, vp, vp, vp]
        L.h263mi_mixed_reset_stream.argtypes = [vp, u32]
        L.h263mi_batch_decode_next_pictures.argtypes = [vp, u32, vp, vp, vp, u32]
        L.h263mi_batch_decode_next_pictures_ex.argtypes = [vp, u32, vp, vp, vp, u32, vp, u8, vp, vp]
        L.h263mi_batch_sync_streams.argtypes = [vp, vp]
        L.h263mi_batch_reset_stream.argtypes = [vp, u32]
        L.h263mi_batch_set_active.argtypes = [vp, vp]
        L.h263mi_batch_stream_has_picture.argtypes = [vp, u32]
        L.h263mi_batch_submit_host.argtypes = [vp, u8, vp, vp, vp, vp]
        L.h263mi_batch_submit_host_events.argtypes = [vp, u8, vp, vp, vp, vp, vp, vp]
        L.h263mi_batch_render_rgba.argtypes = [vp, u8, vp, vp]
        L.h263mi_batch_sync.argtypes = [vp]
        L.h263mi_batch_reset.argtypes = [vp]
        L.h263mi_batch_copy_yuv.argtypes = [vp, u32, vp, vp, vp]
        L.h263mi_batch_timing_begin.argtypes = [vp]
        L.h263mi_batch_timing_end.argtypes = [vp, C.POINTER(KernelTimes)]
        L.h263mi_batch_timing_reserve.argtypes = [vp, u32]
        L.h263mi_probe_bandwidth.argtypes = [C.POINTER(BackendCfg), i32, sz, i32, C.POINTER(C.c_double)]
        L.h263mi_probe_bandwidth_shape.argtypes = [C.POINTER(BackendCfg), i32, sz, i32, C.POINTER(C.c_double), C.POINTER(C.c_char_p)]
        L.h263mi_device_count.argtypes = [C.POINTER(i32)]
        L.h263mi_device_malloc.argtypes = [i32, sz, C.POINTER(vp)]
        L.h263mi_device_free.argtypes = [i32, vp]
        L.h263mi_device_memcpy_h2d.argtypes = [i32, vp, vp, sz]
        L.h263mi_device_memcpy_d2h.argtypes = [i32, vp, vp, sz]
        L.h263mi_device_synchronize.argtypes = [i32]
        L.h263mi_synth_picture_host.argtypes = [i32, u16, u16, u32, u32, vp, vp, sz, C.POINTER(sz)]
        L.h263mi_synth_batch_device.argtypes = [C.POINTER(BackendCfg), i32, u16, u16, u32, u32, u32, vp, vp, sz, vp,
                                                C.POINTER(sz)]
        L.h263mi_synth_batch_device_strided.argtypes = [C.POINTER(BackendCfg), i32, u16, u16, u32, u32, u32, u32, vp, vp, sz, vp,
                                                        C.POINTER(sz)]
        L.h263mi_batch_decode_ps.argtypes = [vp, u8, vp, vp, vp, C.c_uint64, u8, vp, vp, vp]
        L.h263mi_batch_decode_events_ps.argtypes = [vp, u8, vp, vp, vp, vp, C.c_uint64, C.c_uint64, u8, vp, vp, vp]
        L.h263mi_batch_render_rgba_ps.argtypes = [vp, u8, vp, vp, vp]
        L.h263mi_batch_decode_next_pictures_ps.argtypes = [vp, u32, vp, vp, vp, u32, vp, u8, vp, vp, vp]
        L.h263mi_mixed_decode_next_pictures_ps.argtypes = [vp, u32, vp, vp, vp, u32, vp, u8, vp, vp, vp, vp]
        L.h263mi_set_ranks_per_node.argtypes = [u32]
        L.h263mi_set_ranks_per_node.restype = None
        L.h263mi_batch_host_placement.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(u32), vp, u32]
        L.h263mi_debug_host_placement.argtypes = [vp, u32, i32, u32, C.c_char_p, C.POINTER(i32), vp, u32, C.POINTER(u32)]
        L.h263mi_default_parser_threads.restype = u32
        L.h263mi_default_parser_threads.argtypes = [u32, C.POINTER(u32)]
        _lib = L
    return _lib


def _check(rc, what):
    if rc != OK:
        raise H263Error(rc, what)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def device_count():
    n = C.c_int(0)
    rc = lib().h263mi_device_count(C.byref(n))
    return n.value if rc == OK else 0


QUANT_TO_STRENGTH = None


def quant_to_strength():
    """deblock::QUANT_TO_STRENGTH (deblock.rs:5-8)."""
    return np.array((C.c_uint8 * 32).in_dll(lib(), "h263mi_quant_to_strength"), dtype=np.uint8)


def deblock(data, width, strength):
    """deblock::deblock(data, width, strength) -> Vec<u8> (deblock.rs:305), on the GPU."""
    data = np.ascontiguousarray(data, dtype=np.uint8).ravel()
    out = np.empty_like(data)
    _check(lib().h263mi_deblock(_p(data), data.size, width, strength, _p(out)), "deblock")
    return out


def yuv420_to_rgba(y, chroma_b, chroma_r, y_width):
    """yuv::bt601::yuv420_to_rgba(y, chroma_b, chroma_r, y_width) -> Vec<u8> (bt601.rs:105), on the GPU."""
    y = np.ascontiguousarray(y, dtype=np.uint8).ravel()
    cb = np.ascontiguousarray(chroma_b, dtype=np.uint8).ravel()
    cr = np.ascontiguousarray(chroma_r, dtype=np.uint8).ravel()
    if cb.size != cr.size:
        raise H263Error(ERR_INVALID_ARGUMENT, "yuv420_to_rgba")
    out = np.empty(y.size * 4, dtype=np.uint8)
    _check(lib().h263mi_bt601_yuv420_to_rgba(_p(y), y.size, _p(cb), _p(cr), cb.size, y_width, _p(out)),
           "yuv420_to_rgba")
    return out


class DecodedPicture:
    """Host-side copy of DecodedPicture (h263/src/decoder/picture.rs:8-58)."""

    def __init__(self, view, y, cb, cr):
        self.width, self.height = view.width, view.height
        self.chroma_width, self.chroma_height = view.chroma_width, view.chroma_height
        self.temporal_reference = view.temporal_reference
        self.picture_type = view.picture_type
        self.pquant = view.pquant
        self.use_deblocker = view.use_deblocker
        self._yuv = (y, cb, cr)

    def luma_samples_per_row(self):
        return self.width

    def chroma_samples_per_row(self):
        return self.chroma_width

    def as_luma(self):
        return self._yuv[0]

    def as_chroma_b(self):
        return self._yuv[1]

    def as_chroma_r(self):
        return self._yuv[2]

    def as_yuv(self):
        return self._yuv


class H263State:
    """H263State (h263/src/decoder/state.rs:16-490) over the C ABI."""

    def __init__(self, decoder_options=SORENSON_SPARK_BITSTREAM, device_id=0, stream=None, cfg_flags=0):
        self._h = C.c_void_p()
        self._cfg = BackendCfg(device_id, cfg_flags, stream)
        _check(lib().h263mi_state_new(decoder_options, C.byref(self._cfg), C.byref(self._h)), "H263State::new")

    def close(self):
        if self._h:
            lib().h263mi_state_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def is_sorenson(self):
        return bool(lib().h263mi_state_is_sorenson(self._h))

    def reset(self):
        _check(lib().h263mi_state_reset(self._h), "reset")

    def cleanup_buffers(self):
        _check(lib().h263mi_state_cleanup_buffers(self._h), "cleanup_buffers")

    def submit_picture(self, width, height, mbs, coeffs, picture_type=PICTURE_I, temporal_reference=0, pquant=1,
                       use_deblocker=0):
        """Record-level decode_next_picture (state.rs:421-483); raises H263Error on failure."""
        mbs = np.ascontiguousarray(mbs, dtype=MB_RECORD_DTYPE)
        coeffs = np.ascontiguousarray(coeffs, dtype=np.int16).reshape(-1, 64)
        d = PictureDesc(width, height, picture_type, pquant, use_deblocker, 0, temporal_reference, 0)
        _check(lib().h263mi_submit_picture(self._h, C.byref(d), _p(mbs) if mbs.size else None, mbs.size,
                                           _p(coeffs) if coeffs.size else None, coeffs.shape[0]), "submit_picture")

    def submit_picture_events(self, width, height, mbs, block_first_event, events, picture_type=PICTURE_I,
                              temporal_reference=0, pquant=1, use_deblocker=0):
        """submit_picture with sparse coefficient transport (h263mi_submit_picture_events)."""
        mbs = np.ascontiguousarray(mbs, dtype=MB_RECORD_DTYPE)
        first = np.ascontiguousarray(block_first_event, dtype=np.uint32)
        ev = np.ascontiguousarray(events, dtype=np.uint32)
        d = PictureDesc(width, height, picture_type, pquant, use_deblocker, 0, temporal_reference, 0)
        _check(lib().h263mi_submit_picture_events(self._h, C.byref(d), _p(mbs) if mbs.size else None, mbs.size, _p(first),
                                                  first.size - 1, _p(ev) if ev.size else None, ev.size),
               "submit_picture_events")

    def decode_next_picture(self, data):
        data = np.frombuffer(bytes(data), dtype=np.uint8)
        used = C.c_size_t(0)
        _check(lib().h263mi_decode_next_picture(self._h, _p(data), data.size, C.byref(used)), "decode_next_picture")
        return used.value

    def parse_picture(self, data):
        """H263State::parse_picture (state.rs:102-111): header fields of the picture that starts `data`."""
        data = np.frombuffer(bytes(data), dtype=np.uint8)
        d = PictureDesc()
        _check(lib().h263mi_parse_picture_header(self._h, _p(data), data.size, C.byref(d)), "parse_picture")
        return d

    def _view(self, fn, what):
        v = FrameView()
        rc = fn(self._h, C.byref(v))
        if rc == ERR_NO_PICTURE:
            return None
        _check(rc, what)
        return v

    def get_last_picture(self):
        """Option<&DecodedPicture> (state.rs:61-67): None before the first picture."""
        v = self._view(lib().h263mi_get_last_picture, "get_last_picture")
        if v is None:
            return None
        y = np.empty(v.width * v.height, np.uint8)
        cb = np.empty(v.chroma_width * v.chroma_height, np.uint8)
        cr = np.empty_like(cb)
        _check(lib().h263mi_copy_yuv(self._h, _p(y), _p(cb), _p(cr)), "as_yuv")
        return DecodedPicture(v, y, cb, cr)

    def has_reference_picture(self):
        return self._view(lib().h263mi_get_reference_picture, "get_reference_picture") is not None

    def render_rgba(self, strength=0):
        v = self._view(lib().h263mi_get_last_picture, "get_last_picture")
        if v is None:
            raise H263Error(ERR_NO_PICTURE, "render_rgba")
        out = np.empty(v.width * v.height * 4, np.uint8)
        _check(lib().h263mi_render_rgba(self._h, strength, _p(out)), "render_rgba")
        return out

    def render_rgba_pinned(self, strength, pinned):
        """h263mi_render_rgba_pinned: RGBA of the last picture straight into the caller's page-locked buffer (a PinnedBuffer
        or a registered numpy array); returns a view of the w*h*4 bytes"""
        v = self._view(lib().h263mi_get_last_picture, "get_last_picture")
        if v is None:
            raise H263Error(ERR_NO_PICTURE, "render_rgba_pinned")
        arr = pinned.array if isinstance(pinned, PinnedBuffer) else pinned
        n = v.width * v.height * 4
        assert arr.nbytes >= n
        _check(lib().h263mi_render_rgba_pinned(self._h, strength, _p(arr)), "render_rgba_pinned")
        return arr[:n]


class PinnedBuffer:
    """page-locked, device-visible host memory (h263mi_host_alloc) as a numpy uint8 array"""

    def __init__(self, nbytes):
        p = C.c_void_p()
        _check(lib().h263mi_host_alloc(nbytes, C.byref(p)), "host_alloc")
        self.ptr, self.nbytes = p.value, nbytes
        self.array = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,))

    def free(self):
        if self.ptr:
            self.array = None
            lib().h263mi_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def host_register(arr):
    _check(lib().h263mi_host_register(_p(arr), arr.nbytes), "host_register")


def host_unregister(arr):
    _check(lib().h263mi_host_unregister(_p(arr)), "host_unregister")


def debug_fail_nth_hip_call(n):
    """TEST HOOK: the n-th HIP call of the host entry points from now on fails (0 = off); returns calls still to go"""
    return lib().h263mi_debug_fail_nth_hip_call(n)


def synchronize(device_id=0):
    _check(lib().h263mi_device_synchronize(device_id), "device_synchronize")


class DeviceBuffer:
    def __init__(self, nbytes, device_id=0):
        self.device_id, self.nbytes = device_id, nbytes
        self.ptr = C.c_void_p()
        _check(lib().h263mi_device_malloc(device_id, nbytes, C.byref(self.ptr)), "device_malloc")

    def upload(self, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        _check(lib().h263mi_device_memcpy_h2d(self.device_id, C.c_void_p(self.ptr.value + offset), _p(arr),
                                              arr.nbytes), "h2d")

    def download(self, nbytes=None, offset=0, dtype=np.uint8):
        nbytes = self.nbytes - offset if nbytes is None else nbytes
        out = np.empty(nbytes, np.uint8)
        _check(lib().h263mi_device_memcpy_d2h(self.device_id, _p(out), C.c_void_p(self.ptr.value + offset), nbytes),
               "d2h")
        return out.view(dtype)

    def at(self, offset):
        return C.c_void_p(self.ptr.value + offset)

    def free(self):
        if self.ptr:
            lib().h263mi_device_free(self.device_id, self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Batch:
    """N independent streams advancing in lock step on one GPU (h263mi_batch_*)."""

    def __init__(self, n_streams, width, height, device_id=0, stream=None, overlap_post=False, pipeline_post=False,
                 trusted_arrays=False):
        """trusted_arrays: H263MI_CFG_TRUSTED_ARRAYS -- the caller vouches for the device arrays of submit / decode /
        decode_events; the default (ABI 7) bounds every one of them"""
        self.n, self.width, self.height, self.device_id = n_streams, width, height, device_id
        self._cfg = BackendCfg(device_id, (CFG_OVERLAP_POST if overlap_post else 0) | (CFG_PIPELINE_POST if pipeline_post else 0) |
                               (CFG_TRUSTED_ARRAYS if trusted_arrays else 0), stream)
        self._h = C.c_void_p()
        _check(lib().h263mi_batch_create(n_streams, width, height, C.byref(self._cfg), C.byref(self._h)),
               "batch_create")
        self.mbs_per_picture = lib().h263mi_batch_mbs_per_picture(self._h)

    def close(self):
        if self._h:
            lib().h263mi_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def submit(self, picture_type, d_mbs, d_coeffs, d_coeff_base=None):
        _check(lib().h263mi_batch_submit(self._h, picture_type, d_mbs, d_coeffs, d_coeff_base), "batch_submit")

    def _strengths(self, strengths):
        """None, or one post-filter strength per stream -> (keep-alive array, pointer) for the *_ps entry points"""
        if strengths is None:
            return None, None
        arr = np.ascontiguousarray(strengths, np.uint8)
        assert arr.size == self.n
        return arr, _p(arr)

    def decode(self, picture_type, d_mbs, d_coeffs, d_coeff_base=None, coeff_pool_blocks=0, strength=0, d_rgba=None,
               d_deblocked=None, strengths=None):
        """submit + render_rgba in one call (h263mi_batch_decode[_ps]).  strengths: one value per stream (ABI 7)"""
        keep, ps = self._strengths(strengths)
        if ps is None and not _has("h263mi_batch_decode_ps"):          # (a library of an earlier round: A/B runs)
            _check(lib().h263mi_batch_decode(self._h, picture_type, d_mbs, d_coeffs, d_coeff_base, coeff_pool_blocks, strength,
                                             d_rgba, d_deblocked), "batch_decode")
            return
        _check(lib().h263mi_batch_decode_ps(self._h, picture_type, d_mbs, d_coeffs, d_coeff_base, coeff_pool_blocks, strength, ps,
                                            d_rgba, d_deblocked), "batch_decode")

    def decode_events(self, picture_type, d_mbs, d_block_first_event, d_events, d_coeff_base=None, coeff_pool_blocks=0,
                      strength=0, d_rgba=None, d_deblocked=None, n_events=0, strengths=None):
        """h263mi_batch_decode_events[_ps]: decode with the coefficients as sparse events in device memory.  n_events: words
        in d_events, coeff_pool_blocks: blocks in the pool; a block whose bounds do not ascend or reach beyond them is not
        read and its picture is rejected.  0 = not told: the library bounds the arrays by the allocations they lie in (ABI 7)
        -- unless the batch was made with trusted_arrays, where 0 means the caller vouches and nothing is checked."""
        keep, ps = self._strengths(strengths)
        if ps is None and not _has("h263mi_batch_decode_events_ps"):
            _check(lib().h263mi_batch_decode_events(self._h, picture_type, d_mbs, d_block_first_event, d_events, d_coeff_base,
                                                    coeff_pool_blocks, n_events, strength, d_rgba, d_deblocked), "batch_decode_events")
            return
        _check(lib().h263mi_batch_decode_events_ps(self._h, picture_type, d_mbs, d_block_first_event, d_events, d_coeff_base,
                                                   coeff_pool_blocks, n_events, strength, ps, d_rgba, d_deblocked), "batch_decode_events")

    def decode_next_pictures(self, data_list, decoder_options=SORENSON_SPARK_BITSTREAM, n_threads=0, prepared=None):
        """one coded picture per stream (bytes-like objects) through the host parser threads and the GPU; returns the
        bytes consumed per stream.  `prepared` (from prepare_pictures) skips the per-call ctypes marshalling."""
        pd, ln, keep = prepared if prepared is not None else self.prepare_pictures(data_list)
        used = (C.c_size_t * self.n)()
        _check(lib().h263mi_batch_decode_next_pictures(self._h, decoder_options, pd, ln, used, n_threads),
               "batch_decode_next_pictures")
        return list(used)

    def prepare_pictures(self, data_list):
        """(None entries: the stream has no picture in the call -- decode_next_pictures_ex only)"""
        assert len(data_list) == self.n
        # an EMPTY picture (b"") is not "no picture": it goes to the parser and gets its end-of-stream error, so it
        # travels as a non-NULL pointer with length 0; only None becomes NULL
        empty = np.zeros(1, np.uint8)
        keep = [(np.frombuffer(bytes(d), dtype=np.uint8) if len(d) else empty) if d is not None else None for d in data_list]
        pd = (C.c_void_p * self.n)(*[k.ctypes.data if k is not None else None for k in keep])
        ln = (C.c_size_t * self.n)(*[(len(d) if d is not None else 0) for d in data_list])
        return pd, ln, keep

    def decode_next_pictures_ex(self, data_list, decoder_options=SORENSON_SPARK_BITSTREAM, n_threads=0, prepared=None,
                                strength=0, d_rgba=None, d_deblocked=None, strengths=None):
        """h263mi_batch_decode_next_pictures_ex / _ps: every stream its own H263State.  strength may be STRENGTH_FROM_HEADER
        (each picture with QUANT_TO_STRENGTH[its pquant] when its header sets USE_DEBLOCKER, else 0); strengths: one value
        per stream.  Returns (bytes consumed per stream, error code per stream: 0 = decoded or no data)."""
        pd, ln, keep = prepared if prepared is not None else self.prepare_pictures(data_list)
        used = (C.c_size_t * self.n)()
        rcs = (C.c_int * self.n)()
        keep_s, ps = self._strengths(strengths)
        _check(lib().h263mi_batch_decode_next_pictures_ps(self._h, decoder_options, pd, ln, used, n_threads, rcs, strength, ps,
                                                          d_rgba, d_deblocked), "batch_decode_next_pictures_ex")
        return list(used), list(rcs)

    def host_placement(self):
        """h263mi_batch_host_placement: (NUMA node of the device, node the pinned staging memory lies on, CPUs the host
        threads are confined to) -- -1 / -1 / [] where unknown or not made yet"""
        dn, sn, k = C.c_int(-1), C.c_int(-1), C.c_uint32(0)
        cpus = (C.c_uint16 * 1024)()
        _check(lib().h263mi_batch_host_placement(self._h, C.byref(dn), C.byref(sn), C.byref(k), cpus, 1024), "host_placement")
        return dn.value, sn.value, [int(cpus[i]) for i in range(min(k.value, 1024))]

    def sync_streams(self):
        """h263mi_batch_sync_streams: the device's verdict per stream (0 or an error code); never raises for those"""
        rcs = (C.c_int * self.n)()
        rc = lib().h263mi_batch_sync_streams(self._h, rcs)
        if rc != OK and not any(rcs):
            raise H263Error(rc, "batch_sync_streams")
        return list(rcs)

    def reset_stream(self, stream):
        _check(lib().h263mi_batch_reset_stream(self._h, stream), "batch_reset_stream")

    def set_active(self, active=None):
        arr = None if active is None else (C.c_uint8 * self.n)(*[1 if a else 0 for a in active])
        _check(lib().h263mi_batch_set_active(self._h, arr), "batch_set_active")

    def stream_has_picture(self, stream):
        return bool(lib().h263mi_batch_stream_has_picture(self._h, stream))

    def submit_host(self, picture_type, mbs_list, coeffs_list):
        """one picture per stream from host records: lists of MB_RECORD_DTYPE arrays and (n, 64) int16 arrays"""
        n = len(mbs_list)
        mbs = [np.ascontiguousarray(m, MB_RECORD_DTYPE) for m in mbs_list]
        cos = [np.ascontiguousarray(c, np.int16).reshape(-1, 64) for c in coeffs_list]
        pm = (C.c_void_p * n)(*[m.ctypes.data if m.size else None for m in mbs])
        pc = (C.c_void_p * n)(*[c.ctypes.data if c.size else None for c in cos])
        nm = (C.c_uint32 * n)(*[len(m) for m in mbs])
        nc = (C.c_uint32 * n)(*[len(c) for c in cos])
        _check(lib().h263mi_batch_submit_host(self._h, picture_type, pm, nm, pc, nc), "batch_submit_host")

    def submit_host_events(self, picture_type, mbs_list, first_list, events_list):
        """submit_host with sparse coefficient transport: per stream (block_first_event, events) arrays"""
        n = len(mbs_list)
        mbs = [np.ascontiguousarray(m, MB_RECORD_DTYPE) for m in mbs_list]
        fs = [np.ascontiguousarray(f, np.uint32) for f in first_list]
        evs = [np.ascontiguousarray(e, np.uint32) for e in events_list]
        pm = (C.c_void_p * n)(*[m.ctypes.data if m.size else None for m in mbs])
        pf = (C.c_void_p * n)(*[f.ctypes.data for f in fs])
        pe = (C.c_void_p * n)(*[e.ctypes.data if e.size else None for e in evs])
        nm = (C.c_uint32 * n)(*[len(m) for m in mbs])
        nb = (C.c_uint32 * n)(*[len(f) - 1 for f in fs])
        ne = (C.c_uint32 * n)(*[len(e) for e in evs])
        _check(lib().h263mi_batch_submit_host_events(self._h, picture_type, pm, nm, pf, nb, pe, ne), "batch_submit_host_events")

    def render_rgba(self, strength, d_rgba, d_deblocked=None, strengths=None):
        keep, ps = self._strengths(strengths)
        _check(lib().h263mi_batch_render_rgba_ps(self._h, strength, ps, d_rgba, d_deblocked), "batch_render_rgba")

    def sync(self):
        _check(lib().h263mi_batch_sync(self._h), "batch_sync")

    def reset(self):
        _check(lib().h263mi_batch_reset(self._h), "batch_reset")

    def copy_yuv(self, stream):
        w, h = self.width, self.height
        cw, ch = (w + 1) // 2, (h + 1) // 2
        y, cb, cr = np.empty(w * h, np.uint8), np.empty(cw * ch, np.uint8), np.empty(cw * ch, np.uint8)
        _check(lib().h263mi_batch_copy_yuv(self._h, stream, _p(y), _p(cb), _p(cr)), "batch_copy_yuv")
        return y, cb, cr

    def timing_begin(self):
        _check(lib().h263mi_batch_timing_begin(self._h), "timing_begin")

    def timing_reserve(self, n_launches):
        _check(lib().h263mi_batch_timing_reserve(self._h, n_launches), "timing_reserve")

    def timing_end(self):
        t = KernelTimes()
        _check(lib().h263mi_batch_timing_end(self._h, C.byref(t)), "timing_end")
        return t


PROBE_COPY, PROBE_READ, PROBE_WRITE = 0, 1, 2


class MixedBatch:
    """h263mi_mixed: n streams of different (and changing) picture sizes, one launch per size class and call"""

    def __init__(self, n_streams, device_id=0, stream=None, pipeline_post=False):
        cfg = BackendCfg(device_id, 2 if pipeline_post else 0, stream)
        h = C.c_void_p()
        _check(lib().h263mi_mixed_create(n_streams, C.byref(cfg), C.byref(h)), "mixed_create")
        self._h, self.n = h, n_streams

    def close(self):
        if self._h:
            lib().h263mi_mixed_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def decode_next_pictures(self, data_list, decoder_options=SORENSON_SPARK_BITSTREAM, n_threads=0, strength=0, rgba=None,
                             raise_on_error=True, strengths=None):
        """data_list: bytes or None per stream; rgba: DeviceBuffer or None per stream (or None: no rendering).
        Returns (bytes consumed, error codes, picture headers) per stream; with raise_on_error=False a call-level error
        does not raise and comes back as a fourth element (the per-stream results are valid either way)."""
        assert len(data_list) == self.n
        empty = np.zeros(1, np.uint8)
        keep = [(np.frombuffer(bytes(d), dtype=np.uint8) if len(d) else empty) if d is not None else None for d in data_list]
        pd = (C.c_void_p * self.n)(*[k.ctypes.data if k is not None else None for k in keep])
        ln = (C.c_size_t * self.n)(*[(len(d) if d is not None else 0) for d in data_list])
        used = (C.c_size_t * self.n)()
        rcs = (C.c_int * self.n)()
        descs = (PictureDesc * self.n)()
        ptrs = caps = None
        if rgba is not None:
            ptrs = (C.c_void_p * self.n)(*[(r.ptr.value if hasattr(r.ptr, "value") else r.ptr) if r is not None else None for r in rgba])
            caps = (C.c_size_t * self.n)(*[r.nbytes if r is not None else 0 for r in rgba])
        ps = None
        if strengths is not None:
            keep_s = np.ascontiguousarray(strengths, np.uint8)
            assert keep_s.size == self.n
            ps = _p(keep_s)
        rc = lib().h263mi_mixed_decode_next_pictures_ps(self._h, decoder_options, pd, ln, used, n_threads, rcs, strength, ps, ptrs, caps,
                                                        descs)
        if not raise_on_error:
            return list(used), list(rcs), list(descs), rc
        _check(rc, "mixed_decode_next_pictures")
        return list(used), list(rcs), list(descs)

    def sync(self):
        rcs = (C.c_int * self.n)()
        rc = lib().h263mi_mixed_sync(self._h, rcs)
        if rc != OK and not any(rcs):
            raise H263Error(rc, "mixed_sync")
        return list(rcs)

    def stream_size(self, stream):
        w, h = C.c_uint16(0), C.c_uint16(0)
        lib().h263mi_mixed_stream_size(self._h, stream, C.byref(w), C.byref(h))
        return w.value, h.value

    def size_classes(self):
        return lib().h263mi_mixed_size_classes(self._h)

    def set_memory_limit(self, n_bytes):
        """what the frame stores of all size classes together may take (0 = no limit; default: half of the device's memory)"""
        _check(lib().h263mi_mixed_set_memory_limit(self._h, int(n_bytes)), "mixed_set_memory_limit")

    def frame_store_bytes(self):
        return int(lib().h263mi_mixed_frame_store_bytes(self._h))

    def copy_yuv(self, stream):
        w, h = self.stream_size(stream)
        if not w:
            raise H263Error(ERR_NO_PICTURE, "mixed_copy_yuv")
        cw, ch = (w + 1) // 2, (h + 1) // 2
        y, cb, cr = np.empty(w * h, np.uint8), np.empty(cw * ch, np.uint8), np.empty(cw * ch, np.uint8)
        _check(lib().h263mi_mixed_copy_yuv(self._h, stream, _p(y), _p(cb), _p(cr)), "mixed_copy_yuv")
        return y, cb, cr

    def reset_stream(self, stream):
        _check(lib().h263mi_mixed_reset_stream(self._h, stream), "mixed_reset_stream")


def probe_bandwidth(mode, nbytes=1 << 30, reps=10, device_id=0, stream=None, with_shape=False):
    """GB/s the device sustains for a streaming copy / read / write kernel: the fastest of the library's launch shapes
    (h263mi_probe_bandwidth); with_shape: (GB/s, description of the shape that won)."""
    cfg = BackendCfg(device_id, 0, stream)
    out = C.c_double(0.0)
    if with_shape:
        name = C.c_char_p()
        _check(lib().h263mi_probe_bandwidth_shape(C.byref(cfg), mode, nbytes, reps, C.byref(out), C.byref(name)), "probe_bandwidth")
        return out.value, (name.value or b"").decode()
    _check(lib().h263mi_probe_bandwidth(C.byref(cfg), mode, nbytes, reps, C.byref(out)), "probe_bandwidth")
    return out.value


def synth_picture_host(kind, width, height, stream_id, frame_idx):
    total = ((width + 15) // 16) * ((height + 15) // 16)
    mbs = np.zeros(total, MB_RECORD_DTYPE)
    coeffs = np.zeros((total * 6, 64), np.int16)
    n = C.c_size_t(0)
    _check(lib().h263mi_synth_picture_host(kind, width, height, stream_id, frame_idx, _p(mbs), _p(coeffs), total * 6,
                                           C.byref(n)), "synth_picture_host")
    return mbs, coeffs[:n.value].copy()


def debug_host_placement(pci_ids, device, ranks, sysfs_root=None):
    """h263mi_debug_host_placement (no GPU needed): (NUMA node, CPUs) a batch on `device` would be placed on"""
    arr = (C.c_char_p * len(pci_ids))(*[p.encode() for p in pci_ids])
    node, k = C.c_int(-1), C.c_uint32(0)
    cpus = (C.c_uint16 * 1024)()
    _check(lib().h263mi_debug_host_placement(arr, len(pci_ids), device, ranks, sysfs_root.encode() if sysfs_root else None,
                                             C.byref(node), cpus, 1024, C.byref(k)), "debug_host_placement")
    return node.value, [int(cpus[i]) for i in range(min(k.value, 1024))]


def set_ranks_per_node(ranks):
    """h263mi_set_ranks_per_node: how many processes share this node's CPUs with this one (0 = back to the environment)"""
    lib().h263mi_set_ranks_per_node(int(ranks))


def default_parser_threads(n_streams):
    """(threads, cpu_quota): the parser threads a batch call with n_threads = 0 uses for n_streams streams, and the CPU-time
    quota (in CPUs, 0 = none) that choice was made under (h263mi_default_parser_threads)."""
    q = C.c_uint32(0)
    t = lib().h263mi_default_parser_threads(n_streams, C.byref(q))
    return int(t), int(q.value)


def synth_batch_device(kind, width, height, n_streams, first_stream_id, frame_idx, d_mbs, d_coeffs, capacity_blocks,
                       d_coeff_base, device_id=0, stream=None, stream_stride=1):
    """picture p of the batch is stream first_stream_id + p * stream_stride"""
    cfg = BackendCfg(device_id, 0, stream)
    total = C.c_size_t(0)
    if stream_stride == 1 and not _has("h263mi_synth_batch_device_strided"):
        _check(lib().h263mi_synth_batch_device(C.byref(cfg), kind, width, height, n_streams, first_stream_id, frame_idx, d_mbs,
                                               d_coeffs, capacity_blocks, d_coeff_base, C.byref(total)), "synth_batch_device")
        return total.value
    _check(lib().h263mi_synth_batch_device_strided(C.byref(cfg), kind, width, height, n_streams, first_stream_id, stream_stride,
                                                   frame_idx, d_mbs, d_coeffs, capacity_blocks, d_coeff_base, C.byref(total)),
           "synth_batch_device")
    return total.value
