"""Independent numpy restatement of the reference's reconstruction path.

TEST INFRASTRUCTURE ONLY (cross-check of oracle/h263_oracle.c; never imported by the
product).  Written from the reference text separately from the C oracle and with a
different structure (vectorised over the 8 outputs of idct_1d, whole-plane clamped
gathers) so that a transcription slip in either shows up as a disagreement:

  dequant / classify : h263/src/decoder/cpu/rle.rs:82-172
  idct               : h263/src/decoder/cpu/idct.rs:39-65, 82-201
  motion compensation: h263/src/decoder/cpu/gather.rs:16-204, h263/src/types.rs:721-768
  picture assembly   : h263/src/decoder/state.rs:173-191, 421-458

Parity status: UNPINNED by the reference's own tests (it has none for these functions).
"""
import numpy as np

F = np.float32

# idct.rs:39-48 -- decimal literals of the reference, parsed to binary32 by numpy.
BASIS = np.array([
    "0.70710677 0.70710677 0.70710677 0.70710677 0.70710677 0.70710677 0.70710677 0.70710677".split(),
    "0.98078525 0.8314696 0.5555702 0.19509023 -0.19509032 -0.55557036 -0.83146966 -0.9807853".split(),
    "0.9238795 0.38268343 -0.38268352 -0.9238796 -0.9238795 -0.38268313 0.3826836 0.92387956".split(),
    "0.8314696 -0.19509032 -0.9807853 -0.55557 0.55557007 0.98078525 0.19509007 -0.8314698".split(),
    "0.70710677 -0.70710677 -0.70710665 0.707107 0.70710677 -0.70710725 -0.70710653 0.7071068".split(),
    "0.5555702 -0.9807853 0.19509041 0.83146936 -0.8314698 -0.19508928 0.9807853 -0.55557007".split(),
    "0.38268343 -0.9238795 0.92387974 -0.3826839 -0.38268384 0.9238793 -0.92387974 0.3826839".split(),
    "0.19509023 -0.55557 0.83146936 -0.9807852 0.98078525 -0.83147013 0.55557114 -0.19508967".split(),
], dtype=np.float32)

# expected binary32 bit patterns of the table above.
BASIS_HEX = """
3F3504F3 3F3504F3 3F3504F3 3F3504F3 3F3504F3 3F3504F3 3F3504F3 3F3504F3
3F7B14BE 3F54DB31 3F0E39D9 3E47C5BC BE47C5C2 BF0E39DC BF54DB32 BF7B14BF
3F6C835E 3EC3EF15 BEC3EF18 BF6C8360 BF6C835E BEC3EF0B 3EC3EF1B 3F6C835F
3F54DB31 BE47C5C2 BF7B14BF BF0E39D6 3F0E39D7 3F7B14BE 3E47C5B1 BF54DB34
3F3504F3 BF3504F3 BF3504F1 3F3504F7 3F3504F3 BF3504FB BF3504EF 3F3504F4
3F0E39D9 BF7B14BF 3E47C5C8 3F54DB2D BF54DB34 BE47C57C 3F7B14BF BF0E39D7
3EC3EF15 BF6C835E 3F6C8362 BEC3EF25 BEC3EF23 3F6C835B BF6C8362 3EC3EF25
3E47C5BC BF0E39D6 3F54DB2D BF7B14BD 3F7B14BE BF54DB3A 3F0E39E9 BE47C596
"""

# rle.rs:6-71 as raster index x + 8*y per zigzag position.
ZIGZAG_RASTER = np.array([
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63], dtype=np.int64)

ZERO, DC, HORIZ, VERT, FULL = range(5)


def intradc_level(code):
    return 1024 if code == 0xFF else (int(code) << 3)


def _wrap_i16(v):
    """i16 arithmetic of a Rust release build: the value modulo 2^16, as a signed number."""
    return ((np.asarray(v, dtype=np.int64) + 32768) % 65536) - 32768


def dequant(level, quant):
    """rle.rs:130-133 for an array of non-zero LEVELs (zeros stay zero), as a RELEASE build of the reference executes
    it: every intermediate is an i16 that wraps (a dev build panics on the overflow instead).  quant * (2|L| + 1)
    exceeds 32767 for Sorenson's 11-bit escape LEVELs at quantisers from 16 up."""
    level = np.asarray(level, dtype=np.int64)
    mag = _wrap_i16(quant * _wrap_i16(_wrap_i16(2 * _wrap_i16(np.abs(level))) + 1))
    mag = _wrap_i16(mag - (1 if quant % 2 == 0 else 0))
    return np.clip(_wrap_i16(np.sign(level) * mag), -2048, 2047) * (level != 0)


def classify_dense(coeff, is_intra, intradc_code, coded, kill, quant):
    """Dense-record equivalent of inverse_rle: returns (tag, float32[8,8] block[y][x])."""
    blk = np.zeros((8, 8), np.float32)
    if coded and kill:
        return ZERO, blk                        # rle.rs:125-127
    vals = dequant(np.asarray(coeff, np.int64).reshape(8, 8), quant) if coded else np.zeros((8, 8), np.int64)
    if is_intra:
        vals = vals.copy()
        vals[0, 0] = 0                           # TCOEFs of an intra block start at zigzag 1
    any_tcoef = bool(np.any(vals != 0))
    if not any_tcoef:                            # rle.rs:94-109
        if is_intra and intradc_level(intradc_code) != 0:
            blk[0, 0] = intradc_level(intradc_code)
            return DC, blk
        return ZERO, blk
    blk[:] = vals
    if is_intra:
        blk[0, 0] = intradc_level(intradc_code)
    is_horiz = not np.any(vals[1:, :] != 0)      # no non-zero TCOEF with y > 0
    is_vert = not np.any(vals[:, 1:] != 0)       # no non-zero TCOEF with x > 0
    if is_horiz and is_vert:
        return (DC if blk[0, 0] != 0 else ZERO), blk
    if is_horiz:
        return HORIZ, blk
    if is_vert:
        return VERT, blk
    return FULL, blk


def idct_1d(v):
    """idct.rs:52-65: out[i] = ((0 + v0*B0i) + v1*B1i) + ... sequential, separately rounded."""
    v = np.asarray(v, np.float32)
    out = np.zeros(8, np.float32)
    for f in range(8):
        out = (out + (v[f] * BASIS[f]).astype(np.float32)).astype(np.float32)
    return out


def _clip_idct(v):
    """((v + signum(v)*0.5) as i16).clamp(-256, 255) with v already divided by 4."""
    v = np.asarray(v, np.float32)
    sg = np.where(np.signbit(v), F(-1), F(1)).astype(np.float32)
    t = (v + sg * F(0.5)).astype(np.float32)
    t = np.clip(np.trunc(t.astype(np.float64)), -32768, 32767).astype(np.int64)
    return np.clip(t, -256, 255)


def idct_residual(tag, blk):
    """Residual (int64[8,8], indexed [y][x]) that idct_channel adds to the plane (idct.rs:109-197)."""
    if tag == ZERO:
        return np.zeros((8, 8), np.int64)
    if tag == DC:
        dc = F(blk[0, 0])
        r = _clip_idct(np.array([(dc * F(0.5)) / F(4.0)], np.float32))[0]
        # signum is taken from dc; dc*0.5/4 has the same sign
        return np.full((8, 8), r, np.int64)
    if tag == HORIZ:
        t = idct_1d(blk[0, :])
        r = _clip_idct_with_sign(((t * BASIS[0, 0]).astype(np.float32) / F(4.0)).astype(np.float32), t)
        return np.tile(r[None, :], (8, 1))
    if tag == VERT:
        t = idct_1d(blk[:, 0])
        r = _clip_idct_with_sign(((t * BASIS[0, 0]).astype(np.float32) / F(4.0)).astype(np.float32), t)
        return np.tile(r[:, None], (1, 8))
    inter = np.zeros((8, 8), np.float32)
    for row in range(8):
        inter[:, row] = idct_1d(blk[row, :])     # transposition (idct.rs:171-177)
    out = np.zeros((8, 8), np.float32)
    for row in range(8):
        out[row, :] = idct_1d(inter[row, :])     # out[x][y]
    res = _clip_idct((out / F(4.0)).astype(np.float32))
    return res.T.copy()                          # -> [y][x]


def _clip_idct_with_sign(v, sign_src):
    sg = np.where(np.signbit(sign_src), F(-1), F(1)).astype(np.float32)
    t = (v + sg * F(0.5)).astype(np.float32)
    t = np.clip(np.trunc(t.astype(np.float64)), -32768, 32767).astype(np.int64)
    return np.clip(t, -256, 255)


def lerp_params(hp):
    hp = int(hp)
    return hp >> 1, hp & 1      # floor(hp/2), odd  == types.rs:721-729 for every i16


def chroma_mv(s):
    """types.rs:759-768 average_sum_of_mvs on an i16 sum."""
    s = int(np.int16(s))
    whole = (s >> 4) << 1
    frac = s & 15
    if frac <= 2:
        return whole
    if frac >= 14:
        return whole + 2
    return whole + 1


def gather_block(plane, x, y, mvx, mvy):
    """8x8 prediction for destination (x, y) from 2-D `plane` with clamped taps (gather.rs:47-126)."""
    h, w = plane.shape
    dx, ix = lerp_params(mvx)
    dy, iy = lerp_params(mvy)
    us = x + dx + np.arange(8)
    vs = y + dy + np.arange(8)

    def tap(du, dv):
        uu = np.clip(us + du, 0, w - 1)
        vv = np.clip(vs + dv, 0, h - 1)
        return plane[np.ix_(vv, uu)].astype(np.int64)

    a = tap(0, 0)
    if not ix and not iy:
        return a
    b, c, d = tap(1, 0), tap(0, 1), tap(1, 1)
    if ix and iy:
        return (a + b + c + d + 2) // 4
    if ix:
        return (a + b + 1) // 2
    return (a + c + 1) // 2


def decode_picture(width, height, mbs, coeffs, ref=None):
    """Returns (rc, (y, cb, cr)) with flat uint8 planes; rc = -15 on inter MB without reference."""
    w, h = width, height
    cw, ch = (w + 1) // 2, (h + 1) // 2
    mbw, mbh = (w + 15) // 16, (h + 15) // 16
    n_total = mbw * mbh
    coeffs = np.asarray(coeffs, np.int16).reshape(-1, 64)
    Y = np.zeros((mbh * 16, mbw * 16), np.int64)      # padded work planes, cropped at the end
    CB = np.zeros((mbh * 8, mbw * 8), np.int64)
    CR = np.zeros((mbh * 8, mbw * 8), np.int64)
    if ref is not None:
        ry = np.asarray(ref[0], np.uint8).reshape(h, w)
        rcb = np.asarray(ref[1], np.uint8).reshape(ch, cw)
        rcr = np.asarray(ref[2], np.uint8).reshape(ch, cw)
    for i in range(n_total):
        if i < len(mbs):
            m = mbs[i]
            mb_type, quant, cbp, kill = int(m["mb_type"]), int(m["quant"]), int(m["cbp"]), int(m["kill"])
            mv = np.asarray(m["mv"], np.int64)
            dcs = m["intradc"]
            ci = int(m["coeff_index"])
        else:                                          # state.rs:421-427
            mb_type, quant, cbp, kill = 0, 1, 0, 0
            mv = np.zeros((4, 2), np.int64)
            dcs = [0] * 6
            ci = 0
        inter = mb_type in (0, 1, 2, 5)
        intra = mb_type in (3, 4)
        px, py = (i % mbw) * 16, (i // mbw) * 16
        if inter:
            if ref is None:
                return -15, None                       # gather.rs:149
            for b, (ox, oy) in enumerate(((0, 0), (8, 0), (0, 8), (8, 8))):
                Y[py + oy:py + oy + 8, px + ox:px + ox + 8] = gather_block(ry, px + ox, py + oy, mv[b, 0], mv[b, 1])
            cx, cy = chroma_mv(mv[:, 0].sum()), chroma_mv(mv[:, 1].sum())
            CB[py // 2:py // 2 + 8, px // 2:px // 2 + 8] = gather_block(rcb, px // 2, py // 2, cx, cy)
            CR[py // 2:py // 2 + 8, px // 2:px // 2 + 8] = gather_block(rcr, px // 2, py // 2, cx, cy)
        for b in range(6):
            coded = (cbp >> b) & 1
            c = None
            if coded:
                c = coeffs[ci]
                ci += 1
            tag, blk = classify_dense(c, intra, int(dcs[b]), coded, coded and ((kill >> b) & 1), quant)
            res = idct_residual(tag, blk)
            if b < 4:
                ox, oy = (b & 1) * 8, (b >> 1) * 8
                sl = (slice(py + oy, py + oy + 8), slice(px + ox, px + ox + 8))
                Y[sl] = np.clip(Y[sl] + res, 0, 255)
            else:
                P = CB if b == 4 else CR
                sl = (slice(py // 2, py // 2 + 8), slice(px // 2, px // 2 + 8))
                P[sl] = np.clip(P[sl] + res, 0, 255)
    return 0, (Y[:h, :w].astype(np.uint8).ravel(), CB[:ch, :cw].astype(np.uint8).ravel(),
               CR[:ch, :cw].astype(np.uint8).ravel())
