"""Pins the reconstruction half of the C oracle (dequantisation, block classes, IDCT: rle.rs:112-171,
idct.rs:39-201 -- functions the reference holds no test for) against an implementation that does not use the host
FPU: oracle/softfloat_idct.py carries out every binary32 multiply and add on integers with an explicit
round-to-nearest-even step.  Required agreement: all 4 096 DC values, the 204 DC values whose Dc
class differs from the Full arithmetic, the Vert columns of SURVEY appendix B.2, 10^5 random Full blocks -- and the
committed mutation fixtures."""
import ctypes as C
import json
import os
from concurrent.futures import ProcessPoolExecutor

import numpy as np

from oracle import oracle as orc
from oracle import softfloat_idct as sf

GOLD = os.path.join(os.path.dirname(__file__), "golden")
DCT = np.dtype([("tag", "<i4"), ("v", "<f4", (64,))])
assert DCT.itemsize == C.sizeof(orc.DctBlock)
TAG = {"zero": orc.ORC_ZERO, "dc": orc.ORC_DC, "horiz": orc.ORC_HORIZ, "vert": orc.ORC_VERT, "full": orc.ORC_FULL}


def c_residual(tags, values):
    """clipped_idct of n blocks through the C oracle's idct_channel (idct.rs:82-201), as [n, y, x] in [-255, 255]:
    over a prediction of 0 the positive residuals show, over 255 the negative ones"""
    n = len(tags)
    arr = np.zeros(n, DCT)
    arr["tag"] = tags
    arr["v"] = values
    outs = []
    for pred in (0, 255):
        plane = np.full(64 * n, pred, np.uint8)
        orc.lib().orc_idct_channel(arr.ctypes.data_as(C.POINTER(orc.DctBlock)), n, plane.ctypes.data_as(C.c_void_p),
                                   plane.size, n, 8 * n)
        outs.append(plane.reshape(8, n, 8).transpose(1, 0, 2).astype(np.int32))
    return np.where(outs[0] > 0, outs[0], outs[1] - 255)


def c_block(coeffs, force_full=False):
    """one block of dequantised raster coefficients through the C oracle, classified like rle.rs:138-171"""
    tag, val = sf.classify(list(coeffs))
    if force_full and tag != "zero":
        tag, val = "full", list(coeffs)
    v = np.zeros(64, np.float32)
    if tag == "dc":
        v[0] = val
    elif tag in ("horiz", "vert"):
        v[:8] = val
    elif tag == "full":
        v[:] = val
    return c_residual([TAG[tag]], v[None])[0]


def clip255(res):
    return np.clip(np.asarray(res), -255, 255)


def test_softfloat_primitives_against_known_bit_patterns():
    # hand-checkable binary32 facts
    assert sf.f32_from_int(1) == 0x3F800000 and sf.f32_from_int(-2048) == 0xC5000000 and sf.f32_from_int(0) == 0
    assert sf.f32_mul(0x3F800000, 0x3F3504F3) == 0x3F3504F3
    assert sf.f32_add(0x3F800000, 0x3F800000) == 0x40000000
    assert sf.f32_add(0x3F800000, 0xBF800000) == 0                      # exact cancellation gives +0
    assert sf.f32_add(0x80000000, 0x80000000) == 0x80000000             # -0 + -0 = -0
    assert sf.f32_add(0x4B800000, 0x3F800000) == 0x4B800000             # 2^24 + 1: tie, rounds to even
    assert sf.f32_add(0x4B800001, 0x3F800000) == 0x4B800002             # 2^24 + 2 + 1: tie, rounds to even (up)
    assert sf.f32_mul(0x00000001, 0x3F000000) == 0                      # smallest subnormal / 2: tie to even = 0
    assert sf.f32_mul(0x00000003, 0x3F000000) == 0x00000002             # 3 ulp / 2 = 1.5 ulp -> 2
    assert sf.f32_fma(0x3F800001, 0x3F800001, 0xBF800000) == 0x34800000 # (1+u)^2 - 1 = 2u + u^2: one rounding
    assert sf.f32_to_i16(0xC3808000) == -257 and sf.f32_to_i16(0x3F7FFFFF) == 0 and sf.f32_to_i16(0x47800000) == 32767
    # the vectorised integer form agrees with the scalar one, incl. cancellation and far-apart exponents
    rng = np.random.default_rng(5)
    a = rng.integers(0, 1 << 32, 4000, dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, 1 << 32, 4000, dtype=np.uint64).astype(np.uint32)
    ea, eb = (a >> 23) & 0xFF, (b >> 23) & 0xFF            # keep products and sums inside the binary32 range
    keep = (ea > 0x58) & (ea < 0x98) & (eb > 0x58) & (eb < 0x98)
    a, b = a[keep], b[keep]
    b[::3] = a[::3] ^ np.uint32(0x80000000)                             # exact cancellation
    b[1::3] = (a[1::3] ^ np.uint32(0x80000000)) + np.uint32(3)          # near cancellation
    vm, va = sf.vf32_mul(a, b), sf.vf32_add(a, b)
    assert vm.dtype == np.uint32 and va.dtype == np.uint32
    for i in range(len(a)):
        assert int(vm[i]) == sf.f32_mul(int(a[i]), int(b[i])), i
        assert int(va[i]) == sf.f32_add(int(a[i]), int(b[i])), i


def test_dequantisation_rule():
    # rle.rs:130-133 against the C oracle's inverse_rle, every quantiser, levels over the 11-bit range
    for q in range(1, 32):
        for level in (-1023, -127, -2, -1, 1, 2, 3, 40, 127, 1023):
            tag, v = orc.inverse_rle(False, 0, [0], [level], q)
            assert tag == orc.ORC_DC and int(v[0]) == sf.dequant(level, q), (q, level)


def test_all_4096_dc_values():
    dcs = np.arange(-2048, 2048)
    vals = np.zeros((4096, 64), np.float32)
    vals[:, 0] = dcs
    got_c = c_residual(np.full(4096, orc.ORC_DC), vals)
    for k, dc in enumerate(dcs):
        co = [0] * 64
        co[0] = int(dc)
        want = sf.block_residual(co)
        assert (clip255(want) == got_c[k]).all(), dc
        assert len(set(v for row in want for v in row)) == 1            # a Dc block is flat


def test_the_204_dc_values_whose_dc_class_differs_from_the_full_arithmetic():
    """Dc(dc) uses the exact factor 0.5 (idct.rs:114-120); running the same block through the Full
    path (B00 * B00 in two rounded steps) gives a different pixel for 204 of the 4 096 values -- the classes must not
    be merged.  Soft-float and C oracle must agree on WHICH values."""
    differs_sf, differs_c = [], []
    dcs = np.arange(-2048, 2048)
    vals = np.zeros((4096, 64), np.float32)
    vals[:, 0] = dcs
    dc_c = c_residual(np.full(4096, orc.ORC_DC), vals)
    full_c = c_residual(np.full(4096, orc.ORC_FULL), vals)
    for k, dc in enumerate(dcs):
        co = [0] * 64
        co[0] = int(dc)
        a, b = sf.block_residual(co), sf.block_residual(co, force_full=True)
        if a != b and dc != 0:
            differs_sf.append(int(dc))
        assert (clip255(b) == full_c[k]).all(), dc
        if (dc_c[k] != full_c[k]).any():
            differs_c.append(int(dc))
    assert differs_sf == differs_c
    assert len(differs_sf) == 204


def test_vert_columns_of_survey_appendix_b2():
    for col in ([19, 0, 0, 0, -7, 0, 0, 0], [-38, 0, 0, 0, -2, 0, 0, 0], [-34, 0, 0, 0, -2, 0, 0, 0]):
        co = [0] * 64
        for y, v in enumerate(col):
            co[8 * y] = v
        assert sf.classify(co)[0] == "vert"
        vert, full = sf.block_residual(co), sf.block_residual(co, force_full=True)
        assert (clip255(vert) == c_block(co)).all()
        assert (clip255(full) == c_block(co, force_full=True)).all()
        assert vert != full                                           # these columns are why Vert keeps its own arithmetic
    # row-only blocks: Horiz equals Full for every input tried, in both implementations
    rng = np.random.default_rng(7)
    for _ in range(300):
        co = [0] * 64
        co[:8] = [int(v) for v in rng.integers(-300, 301, 8)]
        if not any(co[1:8]):
            continue
        h = sf.block_residual(co)
        assert h == sf.block_residual(co, force_full=True)
        assert (clip255(h) == c_block(co)).all()


def _random_full_blocks(seed, n):
    rng = np.random.default_rng(seed)
    co = np.zeros((n, 64), np.int64)
    for b in range(n):
        k = int(rng.integers(2, 20))
        pos = rng.choice(64, k, replace=False)
        co[b, pos] = rng.integers(-2048, 2048, k) if b % 10 == 0 else rng.integers(-600, 601, k)
    return co


def _chunk_mismatches(seed):
    co = _random_full_blocks(seed, 12500)
    res, _ = sf.vfull_residual(co)
    got_c = c_residual(np.full(len(co), orc.ORC_FULL), co.astype(np.float32))
    return int((clip255(res) != got_c).any(axis=(1, 2)).sum())


def test_100000_random_full_blocks():
    # 6.4 M pixels; the vectorised integer soft-float in 4 worker processes (about 20 s)
    with ProcessPoolExecutor(4) as ex:
        bad = sum(ex.map(_chunk_mismatches, range(100, 108)))
    assert bad == 0


def test_scalar_and_vectorised_softfloat_agree_on_whole_blocks():
    co = _random_full_blocks(9, 40)
    res, _ = sf.vfull_residual(co)
    for b in range(len(co)):
        assert sf.block_residual([int(v) for v in co[b]], force_full=True) == res[b].tolist()


def test_mutation_fixture_is_consistent():
    """tests/golden/idct_sensitive_blocks.json (tools/find_sensitive_blocks.py): the C oracle gives `residual`, the
    mutated soft-float arithmetic gives the listed differing pixels -- so an implementation that fuses or
    re-associates cannot match the oracle on these blocks."""
    doc = json.load(open(os.path.join(GOLD, "idct_sensitive_blocks.json")))
    assert len(doc["blocks"]) >= 40
    n_fma = n_pw = 0
    for blk in doc["blocks"]:
        co = [sf.dequant(l, blk["quant"]) if l else 0 for l in blk["levels"]]
        assert sf.classify(co)[0] == "full"
        assert (c_block(co) == clip255(blk["residual"])).all()
        assert sf.block_residual(co) == blk["residual"]
        for mode, pixels in blk["detects"].items():
            mut = sf.block_residual(co, mode)
            for y, x, v in pixels:
                assert mut[y][x] == v != blk["residual"][y][x]
                assert -128 <= v <= 127 and -128 <= blk["residual"][y][x] <= 127       # visible over a prediction of 128
        n_fma += "fma" in blk["detects"]
        n_pw += "pairwise" in blk["detects"]
    assert n_fma >= 20 and n_pw >= 20
