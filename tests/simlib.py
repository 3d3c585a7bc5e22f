"""ctypes access to the CPU logic checker of the kernel phases (tests/sim/sim.cpp).

Test infrastructure only.  The same helpers pack tightly stored planes into the pitched
device frame layout and back, so GPU tests reuse them.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SIM_DIR = os.path.join(HERE, "sim")


class FrameLayout(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "width", "height", "cwidth", "cheight", "mbw", "mbh", "pitch_y", "pitch_c", "rows_y", "rows_c",
        "off_cb", "off_cr", "frame_bytes", "pad")]


_libs = {}


def lib(asan=False):
    name = "libh263mi_sim_asan.so" if asan else "libh263mi_sim.so"
    if name not in _libs:
        subprocess.check_call(["make", "-C", SIM_DIR, "-s", name])
        L = C.CDLL(os.path.join(SIM_DIR, name))
        L.sim_layout.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(FrameLayout)]
        L.sim_recon.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.sim_recon_ex.argtypes = L.sim_recon.argtypes + [C.c_void_p, C.c_void_p]
        L.sim_post.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                               C.c_int]
        L.sim_synth_picture.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                        C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        _libs[name] = L
    return _libs[name]


def layout(w, h):
    L = FrameLayout()
    lib().sim_layout(w, h, C.byref(L))
    return L


def pack_frame(L, planes, fill=0x5A):
    """(y, cb, cr) tightly packed -> one pitched frame (padding filled with `fill`)."""
    f = np.full(L.frame_bytes, fill, np.uint8)
    y, cb, cr = (np.asarray(p, np.uint8) for p in planes)
    f[:L.pitch_y * L.rows_y].reshape(L.rows_y, L.pitch_y)[:L.height, :L.width] = y.reshape(L.height, L.width)
    for off, p in ((L.off_cb, cb), (L.off_cr, cr)):
        f[off:off + L.pitch_c * L.rows_c].reshape(L.rows_c, L.pitch_c)[:L.cheight, :L.cwidth] = \
            p.reshape(L.cheight, L.cwidth)
    return f


def unpack_frame(L, f):
    y = f[:L.pitch_y * L.rows_y].reshape(L.rows_y, L.pitch_y)[:L.height, :L.width].copy().ravel()
    out = [y]
    for off in (L.off_cb, L.off_cr):
        out.append(f[off:off + L.pitch_c * L.rows_c].reshape(L.rows_c, L.pitch_c)[:L.cheight, :L.cwidth].copy().ravel())
    return tuple(out)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def pad_records(mbs, w, h):
    """state.rs:421-427: missing macroblocks become Inter / mv 0 / nothing coded."""
    from oracle.oracle import MB_RECORD_DTYPE
    total = ((w + 15) // 16) * ((h + 15) // 16)
    out = np.zeros(total, MB_RECORD_DTYPE)
    out["quant"] = 1
    out[:len(mbs)] = mbs
    return out


def to_sparse_records(mbs, w, h):
    """dense raster records -> (records of the macroblocks that are coded, group index) of ReconArgs::mb_group_index: a
    macroblock that is not coded (INTER, no vector, nothing coded) has no record; one word per group of 8 macroblocks of a row"""
    mbw, mbh = (w + 15) // 16, (h + 15) // 16
    gpl = (mbw + 7) // 8
    index = np.zeros(gpl * mbh, np.uint32)
    keep = []
    for k in range(len(mbs)):
        m = mbs[k]
        if int(m["mb_type"]) == 0 and int(m["cbp"]) == 0 and not np.asarray(m["mv"]).any():
            continue
        line, col = divmod(k, mbw)
        g = line * gpl + col // 8
        if not index[g] & 0xff:
            index[g] = len(keep) << 8
        index[g] |= 1 << (col % 8)
        keep.append(k)
    return np.ascontiguousarray(mbs[keep]), index


def recon(w, h, mbs, coeffs, ref=None, asan=False, events=False, sparse_records=False):
    """One picture through the kernel phases on the CPU.  Returns (status, (y, cb, cr)).  events: the coefficients reach
    the reconstruction wave as sparse events (block_first_event + events), not as dense blocks."""
    L = layout(w, h)
    mbs = pad_records(mbs, w, h)
    keep_alive = None
    if sparse_records:
        mbs, gi = to_sparse_records(mbs, w, h)
        if len(mbs) == 0:
            mbs = np.zeros(1, mbs.dtype)
        base = np.zeros(1, np.uint64)
        keep_alive = (gi, base)
        lib(asan).sim_set_sparse_records.argtypes = [C.c_void_p, C.c_void_p]
        lib(asan).sim_set_sparse_records(_p(gi), _p(base))
    coeffs = np.ascontiguousarray(coeffs, np.int16).reshape(-1, 64)
    if events:
        # (an intra block's DC is not an event: it travels in the record)
        intra = np.zeros(coeffs.shape[0], bool)
        for m in mbs:
            if int(m["mb_type"]) in (3, 4):
                k = int(m["coeff_index"])
                intra[k:k + bin(int(m["cbp"])).count("1")] = True
        nz = coeffs != 0
        nz[intra, 0] = False
        first = np.zeros(coeffs.shape[0] + 1, np.uint32)
        np.cumsum(nz.sum(axis=1), out=first[1:])
        blk, pos = np.nonzero(nz)
        ev = ((coeffs[blk, pos].astype(np.uint16).astype(np.uint32) << 16) | pos.astype(np.uint32))
        ev = np.concatenate([ev, np.zeros(4, np.uint32)])
        reff = pack_frame(L, ref) if ref is not None else None
        cur = np.full(L.frame_bytes, 0xC3, np.uint8)
        status = np.zeros(1, np.uint32)
        dummy = np.zeros((1, 64), np.int16)
        rc = lib(asan).sim_recon_ex(w, h, 1, _p(mbs), _p(dummy), coeffs.shape[0], None, _p(reff), 1 if ref is not None else 0,
                                    _p(cur), _p(status), _p(first), _p(ev))
        assert rc == 0, "sim_recon_ex: %d (-2: desc_pix_origin != task_pix_origin)" % rc
        return int(status[0]), unpack_frame(L, cur)
    cpad = np.zeros((coeffs.shape[0] + 1, 64), np.int16)
    cpad[:coeffs.shape[0]] = coeffs
    reff = pack_frame(L, ref) if ref is not None else None
    cur = np.full(L.frame_bytes, 0xC3, np.uint8)
    status = np.zeros(1, np.uint32)
    rc = lib(asan).sim_recon(w, h, 1, _p(mbs), _p(cpad), coeffs.shape[0], None, _p(reff), 1 if ref is not None else 0,
                             _p(cur), _p(status))
    assert rc == 0, "sim_recon: %d (-2: desc_pix_origin != task_pix_origin)" % rc
    return int(status[0]), unpack_frame(L, cur)


def post(w, h, planes, strength, want_rgba=True, want_planes=True, luma_only=False, asan=False):
    L = layout(w, h)
    if luma_only:
        planes = (planes[0], np.zeros(L.cwidth * L.cheight, np.uint8), np.zeros(L.cwidth * L.cheight, np.uint8))
    f = pack_frame(L, planes)
    rgba = np.full(w * h * 4, 0x11, np.uint8) if want_rgba else None
    po = np.full(w * h + 2 * L.cwidth * L.cheight, 0x22, np.uint8) if want_planes else None
    lib(asan).sim_post(w, h, 1, _p(f), strength, _p(rgba), _p(po), 1 if luma_only else 0)
    out_planes = None
    if po is not None:
        n, c = w * h, L.cwidth * L.cheight
        out_planes = (po[:n], po[n:n + c], po[n + c:n + 2 * c])
    return rgba, out_planes


def synth_picture(kind, w, h, stream_id, frame_idx):
    from oracle.oracle import MB_RECORD_DTYPE
    total = ((w + 15) // 16) * ((h + 15) // 16)
    mbs = np.zeros(total, MB_RECORD_DTYPE)
    coeffs = np.zeros((total * 6, 64), np.int16)
    n = C.c_uint64()
    rc = lib().sim_synth_picture(kind, w, h, stream_id, frame_idx, _p(mbs), _p(coeffs), total * 6, C.byref(n))
    assert rc == 0
    return mbs, coeffs[:n.value].copy()
