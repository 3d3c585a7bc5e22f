"""The reconstruction half of the oracle against the STANDARD, not against the reference's text.

The reference holds no test vectors for dequantisation / IDCT / motion compensation, so that half of the oracle
is pinned by reading idct.rs / gather.rs -- one reading, however often it is cross-checked against itself.  These tests
anchor it from the other side: ITU-T H.263 says what a decoder must compute, independently of how h263-rs computes it.

  * Annex A (the IEEE 1180-1990 procedure): the inverse transform's output may differ from the mathematically exact
    8x8 IDCT (double precision, rounded to nearest) by at most 1 in any pixel, with bounded mean and mean-square errors
    per pixel and overall -- checked over 10 000 random blocks per range, both signs.  A transposed basis, a wrong
    scale or a wrong sign anywhere in the restatement fails this by a wide margin.
  * 6.1.2: half-pel luminance prediction is a, (a + b + 1) / 2, (a + c + 1) / 2, (a + b + c + d + 2) / 4 (integer division,
    no rounding control in the baseline), the chrominance vector of a macroblock with one vector is the luminance vector
    halved with quarter positions moved to the half position (Table 9 of 6.1.1 -- equal to Annex F's sum / 8 rule for four
    equal vectors).

What these tests cannot see is the last bit of the reference's float32 arithmetic (Annex A allows many IDCTs); that is what
the soft-float model, the numpy restatement and the mutation tests are for."""
import numpy as np
import pytest

from oracle import oracle as orc

K = np.arange(8)
C = np.where(K == 0, 1 / np.sqrt(2.0), 1.0)
# basis[u][x] = C(u) cos((2x + 1) u pi / 16) / 2  (H.263 6.2.4 / Annex A)
BASIS64 = (C[:, None] * np.cos((2 * K[None, :] + 1) * K[:, None] * np.pi / 16.0)) / 2.0


def exact_idct(F):
    """F[v][u] (vertical, horizontal frequency) -> f[y][x], float64"""
    return BASIS64.T @ F @ BASIS64


def dequant(level, q):
    """6.2.1: |REC| = q (2 |LEVEL| + 1) - (q even), sign of LEVEL; 0 stays 0; clipped to [-2048, 2047]"""
    a = np.abs(level).astype(np.int64)
    rec = np.where(a == 0, 0, q * (2 * a + 1) - (1 - q % 2))
    return np.clip(np.sign(level) * rec, -2048, 2047)


def oracle_residuals(levels, q):
    """n inter blocks of LEVELs (n, 64; position x + 8y) through orc.decode_picture on a flat prediction of 128, zero vectors:
    the residual the oracle adds, as (n, 8, 8) -- exact wherever 128 + residual stays inside 0..255"""
    n = len(levels)
    per = 4                                                       # luma blocks of a macroblock
    n_mb = (n + per - 1) // per
    w, h = 16 * n_mb, 16
    mbs = np.zeros(n_mb, orc.MB_RECORD_DTYPE)
    mbs["mb_type"] = 0
    mbs["quant"] = q
    co = np.zeros((n_mb * per, 64), np.int16)
    co[:n] = levels
    mbs["cbp"] = 0x0f                                             # bits 0..3: the four luma blocks
    mbs["coeff_index"] = np.arange(n_mb) * per
    ref = (np.full(w * h, 128, np.uint8), np.full(w * h // 4, 128, np.uint8), np.full(w * h // 4, 128, np.uint8))
    rc, (y, cb, cr) = orc.decode_picture(w, h, mbs, co, ref)
    assert rc == 0
    y = y.reshape(h, w).astype(np.int64) - 128
    out = np.zeros((n_mb * per, 8, 8), np.int64)
    for m in range(n_mb):
        for b in range(per):
            out[m * per + b] = y[8 * (b >> 1):8 * (b >> 1) + 8, 16 * m + 8 * (b & 1):16 * m + 8 * (b & 1) + 8]
    return out[:n]


def test_block_order_and_position_convention_of_the_helper():
    # one coefficient at (u = 1, v = 0) of luma block 1 (top right) gives a horizontal cosine there and nothing elsewhere
    lv = np.zeros((4, 64), np.int16)
    lv[1, 1] = 20
    res = oracle_residuals(lv, 4)
    assert not res[0].any() and not res[2].any() and not res[3].any()
    assert (res[1][0] == res[1][7]).all() and res[1][0, 0] > 0 > res[1][0, 7]


@pytest.mark.parametrize("max_level,q,name", [(127, 1, "[-255, 255]"), (2, 1, "[-5, 5]"), (149, 1, "[-299, 299]"), (31, 4, "[-251, 251], q = 4")])
@pytest.mark.parametrize("sign", [1, -1])
def test_idct_meets_the_accuracy_specification_of_annex_a(max_level, q, name, sign):
    rng = np.random.default_rng(1180 + max_level + q)
    n = 10000
    levels = (sign * rng.integers(-max_level, max_level + 1, (n, 64))).astype(np.int16)
    got = oracle_residuals(levels, q)
    F = dequant(levels.astype(np.int64), q).reshape(n, 8, 8).astype(np.float64)
    want = np.clip(np.rint(np.stack([exact_idct(f) for f in F])), -256, 255).astype(np.int64)
    # the helper sees the residual through a prediction of 128: compare where neither side saturates 0..255
    lo, hi = -128, 127
    ok = (want > lo) & (want < hi) & (got > lo) & (got < hi)
    assert ok.mean() > 0.3, name
    err = np.where(ok, got - want, 0)
    cnt = np.maximum(ok.sum(axis=0), 1)
    assert np.abs(err).max() <= 1, "peak error, %s" % name
    assert (np.square(err).sum(axis=0) / cnt).max() <= 0.06, "mean square error of a pixel, %s" % name
    assert np.square(err).sum() / ok.sum() <= 0.02, "overall mean square error, %s" % name
    assert np.abs(err.sum(axis=0) / cnt).max() <= 0.015, "mean error of a pixel, %s" % name
    assert abs(err.sum() / ok.sum()) <= 0.0015, "overall mean error, %s" % name


def test_idct_of_nothing_is_nothing():
    assert not oracle_residuals(np.zeros((4, 64), np.int16), 7).any()


def chroma_vector(m):
    """6.1.1, one vector per macroblock: the luminance component halved, quarter positions to the half position; half-pel units"""
    a = abs(int(m))
    return int(np.sign(m)) * (2 * (a >> 2) + (1 if a & 3 else 0))


def predict(plane, x0, y0, n, mvx, mvy):
    """6.1.2 for an n x n block at (x0, y0), vector (mvx, mvy) in half-pel units, all taps inside the plane"""
    ix, iy, hx, hy = mvx >> 1, mvy >> 1, mvx & 1, mvy & 1
    p = plane.astype(np.int64)
    a = p[y0 + iy:y0 + iy + n, x0 + ix:x0 + ix + n]
    b = p[y0 + iy:y0 + iy + n, x0 + ix + hx:x0 + ix + hx + n]
    c = p[y0 + iy + hy:y0 + iy + hy + n, x0 + ix:x0 + ix + n]
    d = p[y0 + iy + hy:y0 + iy + hy + n, x0 + ix + hx:x0 + ix + hx + n]
    if hx and hy:
        return (a + b + c + d + 2) >> 2
    if hx:
        return (a + b + 1) >> 1
    if hy:
        return (a + c + 1) >> 1
    return a


def test_half_pel_prediction_follows_6_1_2_for_every_phase_and_the_chroma_vector_rule():
    rng = np.random.default_rng(612)
    w, h = 96, 80
    mbw, mbh = w // 16, h // 16
    for trial in range(12):
        ref = tuple(rng.integers(0, 256, s, dtype=np.uint8) for s in (w * h, w * h // 4, w * h // 4))
        mbs = np.zeros(mbw * mbh, orc.MB_RECORD_DTYPE)
        mbs["mb_type"] = 0
        mbs["quant"] = 5
        mv = np.zeros((mbw * mbh, 2), np.int64)
        for i in range(mbw * mbh):
            mx, my = i % mbw, i // mbw
            # any vector whose taps stay inside the picture (the border rule is the reference's own business)
            lo_x, hi_x = -32 * mx, 32 * (mbw - 1 - mx) - 2
            lo_y, hi_y = -32 * my, 32 * (mbh - 1 - my) - 2
            mv[i] = (rng.integers(max(lo_x, -31), min(hi_x, 31) + 1), rng.integers(max(lo_y, -31), min(hi_y, 31) + 1))
        mbs["mv"] = np.repeat(mv[:, None, :], 4, axis=1)
        rc, (y, cb, cr) = orc.decode_picture(w, h, mbs, np.zeros((0, 64), np.int16), ref)
        assert rc == 0
        Y, CB, CR = y.reshape(h, w), cb.reshape(h // 2, w // 2), cr.reshape(h // 2, w // 2)
        R = (ref[0].reshape(h, w), ref[1].reshape(h // 2, w // 2), ref[2].reshape(h // 2, w // 2))
        for i in range(mbw * mbh):
            mx, my = i % mbw, i // mbw
            vx, vy = int(mv[i][0]), int(mv[i][1])
            assert (Y[16 * my:16 * my + 16, 16 * mx:16 * mx + 16] == predict(R[0], 16 * mx, 16 * my, 16, vx, vy)).all(), (trial, i, vx, vy)
            cx, cy = chroma_vector(vx), chroma_vector(vy)
            for got, plane in ((CB, R[1]), (CR, R[2])):
                assert (got[8 * my:8 * my + 8, 8 * mx:8 * mx + 8] == predict(plane, 8 * mx, 8 * my, 8, cx, cy)).all(), (trial, i, vx, vy, cx, cy)
