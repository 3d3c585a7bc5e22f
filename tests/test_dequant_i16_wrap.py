"""The dequantiser where the reference's i16 arithmetic overflows (rle.rs:130-133).

`quant as i16 * ((2 * tcoef.level.abs()) + 1)` is an i16 product.  Sorenson's 11-bit escape LEVELs
(parser/block.rs:694-708) push it past 32 767 from quant = 16 up; the dev profile panics there, a release build --
/Cargo.toml:16-17, no overflow-checks: what Ruffle ships -- WRAPS, and the wrapped value is what gets clamped to
[-2048, 2047].  "Identical to the reference" means identical to that.

 1. tests/golden/dequant_i16_wrap_known_answers.json: values worked out by hand, arithmetic beside every one;
 2. the C oracle, the numpy restatement and the soft-float model must reproduce them, and agree with each other over
    every int16 LEVEL at every quantiser;
 3. the kernel's wrapping dequantiser (recon_kernel.inl: dequant_pair_wrap, through tests/sim) must too, its fast form
    (dequant_pair_i16) must be 16 x the same value wherever |LEVEL| <= 511 -- the condition under which a round takes it --
    and the detector of wider LEVELs (wide_bits_of, on coefficient rows and on event words) must fire exactly outside
    [-512, 511];
 4. whole pictures with 11-bit LEVELs at quantisers 16..31 through the sim's wave against the oracle, and: forcing
    EVERY round through the wide form changes nothing (which rounds take it is a matter of speed only).
The MI355X side of this is tests/test_gpu_round5.py and the fourth mutant of tests/test_gpu_mutation.py.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import recgen
import simlib
from oracle import np_restatement as npr
from oracle import oracle as orc
from oracle import softfloat_idct as sf

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "dequant_i16_wrap_known_answers.json")))["cases"]


def wrap16(v):
    return ((np.asarray(v, np.int64) + 32768) % 65536) - 32768


def release_build_value(level, q):
    """rle.rs:130-133 step by step on int64 with an explicit wrap behind every i16 operation"""
    level = np.asarray(level, np.int64)
    a = wrap16(np.abs(level))                       # i16::abs
    m = wrap16(q * wrap16(wrap16(2 * a) + 1))       # quant as i16 * ((2 * abs) + 1)
    m = wrap16(m + (0 if q % 2 == 1 else -1))       # + parity
    return np.clip(wrap16(np.sign(level) * m), -2048, 2047)


def oracle_value(level, q):
    tag, v = orc.inverse_rle(False, 0, [0], [int(level)], q)
    assert tag == orc.ORC_DC
    return int(v[0])


@pytest.mark.parametrize("case", CASES, ids=lambda c: "q%d_L%d" % (c["quant"], c["level"]))
def test_hand_derived_known_answers(case):
    q, lv, want = case["quant"], case["level"], case["value"]
    assert oracle_value(lv, q) == want, case["arithmetic"]
    assert int(npr.dequant(np.array([lv]), q)[0]) == want
    assert sf.dequant(lv, q) == want
    assert int(release_build_value(lv, q)) == want


def test_the_three_restatements_agree_on_every_int16_level():
    levels = np.array([v for v in range(-32768, 32768) if v != 0], np.int64)
    for q in range(1, 32):
        want = release_build_value(levels, q)
        assert (npr.dequant(levels, q) == want).all(), q
        # the scalar ones on the ranges where something happens: all 11-bit LEVELs, the i16 extremes, a stride elsewhere
        probe = list(range(-1024, 1024)) + [-32768, -32767, -16385, -16384, 16383, 16384, 32766, 32767] + list(range(-32768, 32768, 257))
        for lv in probe:
            if lv == 0:
                continue
            w = int(want[lv + 32768 - (1 if lv > 0 else 0)])
            assert sf.dequant(lv, q) == w, (q, lv)
            assert oracle_value(lv, q) == w, (q, lv)


def test_where_the_wrap_first_shows():
    """No |L| <= 511 overflows at any quantiser (what lets the kernel's fast form stand), q = 16 wraps at L = -1024 only,
    q = 31 from |L| = 529 -- and below q = 16 no 11-bit LEVEL does."""
    lv = np.array([v for v in range(-1024, 1024) if v != 0], np.int64)
    for q in range(1, 32):
        math = np.clip(np.sign(lv) * (q * (2 * np.abs(lv) + 1) - (1 - q % 2)), -2048, 2047)
        differs = lv[release_build_value(lv, q) != math]
        if q < 16:
            assert differs.size == 0, q
        elif q == 16:
            assert differs.tolist() == [-1024]
        else:
            assert np.abs(differs).min() == (32767 // q - 1) // 2 + 1, q
        assert not (np.abs(differs) <= 511).any()
    assert release_build_value(np.array([529, -529, 528, -528]), 31).tolist() == [-2048, 2047, 2047, -2048]


def _pairs(levels):
    lv = np.asarray(levels, np.int16)
    if lv.size % 2:
        lv = np.concatenate([lv, np.zeros(1, np.int16)])
    return lv, np.ascontiguousarray(lv).view(np.uint32)


def test_kernel_wrapping_dequantiser_every_int16_level_every_quantiser():
    L = simlib.lib()
    L.sim_dequant_pairs_wrap.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.sim_dequant_pairs.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lv, packed = _pairs(np.arange(-32768, 32768, dtype=np.int32).astype(np.int16))
    out = np.empty_like(packed)
    narrow = np.abs(lv.astype(np.int64)) <= 511
    for q in range(1, 32):
        want = release_build_value(lv, q) * (lv != 0)
        L.sim_dequant_pairs_wrap(packed.ctypes.data, len(packed), q, out.ctypes.data)
        assert (out.view(np.int16).astype(np.int64) == want).all(), q
        # the fast form: 16 x the same value wherever the round is allowed to take it
        L.sim_dequant_pairs(packed.ctypes.data, len(packed), q, out.ctypes.data)
        assert (out.view(np.int16).astype(np.int64)[narrow] == 16 * want[narrow]).all(), q


def test_wide_level_detector_fires_exactly_outside_pm512():
    L = simlib.lib()
    L.sim_wide_bits.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    rng = np.random.default_rng(3)
    # every int16 value in every one of the 8 positions of a row, the other seven narrow
    for pos in range(8):
        rows = rng.integers(-512, 512, (65536, 8)).astype(np.int16)
        rows[:, pos] = np.arange(-32768, 32768, dtype=np.int32).astype(np.int16)
        got = np.empty(65536, np.uint32)
        L.sim_wide_bits(rows.ctypes.data, 65536, got.ctypes.data)
        v = rows[:, pos].astype(np.int64)
        assert ((got != 0) == ((v < -512) | (v > 511))).all(), pos
    # ... and on event words, LEVEL << 16 | position, every LEVEL with every position
    L.sim_wide_bits_events.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lv = np.arange(-32768, 32768, dtype=np.int64)
    for pos in (0, 1, 31, 62, 63):
        ev = ((lv & 0xffff) << 16 | pos).astype(np.uint32)
        got = np.empty(65536, np.uint32)
        L.sim_wide_bits_events(ev.ctypes.data, 65536, got.ctypes.data)
        assert ((got != 0) == ((lv < -512) | (lv > 511))).all(), pos


@pytest.mark.parametrize("w,h,quant", [(176, 144, 31), (64, 48, 16), (100, 60, 17), (48, 32, 24)])
def test_pictures_with_11_bit_levels_at_large_quantisers(w, h, quant):
    """inter and intra macroblocks, Full / Horiz / Vert / Dc blocks, LEVELs over the whole 11-bit range (Sorenson's
    escape) at the quantisers where the i16 product wraps: the sim's wave == the oracle's release-build arithmetic"""
    ref = recgen.random_planes(w, h, 5)
    mbs, coeffs = recgen.inter_picture(w, h, seed=quant + w, mv_range=40, p_4v=0.3, p_intra=0.3, p_coded=0.7,
                                       quant=quant, max_level=1023, sparse_low=False)
    coeffs = coeffs.copy()
    rng = np.random.default_rng(quant)
    flat = coeffs.reshape(-1)
    nz = np.flatnonzero(flat)
    flat[nz[rng.random(nz.size) < 0.05]] = -1024          # the one LEVEL that wraps at q = 16
    rc, want = orc.decode_picture(w, h, mbs, coeffs, ref)
    assert rc == 0
    for events in (False, True):
        st, got = simlib.recon(w, h, mbs, coeffs, ref, events=events)
        assert st == 0
        for g, e, name in zip(got, want, "Y Cb Cr".split()):
            assert (g == e).all(), (events, name, np.flatnonzero(g != e)[:10])
    mbs, coeffs = recgen.intra_picture(w, h, seed=quant, max_level=1023, quant=quant)
    rc, want = orc.decode_picture(w, h, mbs, coeffs, None)
    assert rc == 0
    for events in (False, True):
        st, got = simlib.recon(w, h, mbs, coeffs, None, events=events)
        assert st == 0
        for g, e in zip(got, want):
            assert (g == e).all()


@pytest.mark.parametrize("max_level,quant", [(127, 10), (511, 31), (1023, 31)])
def test_forcing_every_round_through_the_wide_form_changes_nothing(max_level, quant):
    w, h = 176, 144
    L = simlib.lib()
    ref = recgen.random_planes(w, h, 9)
    mbs, coeffs = recgen.inter_picture(w, h, seed=max_level, mv_range=30, p_4v=0.2, p_intra=0.2, p_coded=0.6, quant=quant,
                                       max_level=max_level, sparse_low=False)
    _, normal = simlib.recon(w, h, mbs, coeffs, ref)
    L.sim_force_wide_rounds(1)
    try:
        _, forced = simlib.recon(w, h, mbs, coeffs, ref)
    finally:
        L.sim_force_wide_rounds(0)
    rc, want = orc.decode_picture(w, h, mbs, coeffs, ref)
    assert rc == 0
    for a, b, e in zip(normal, forced, want):
        assert (a == e).all() and (b == e).all()
