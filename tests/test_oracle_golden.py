"""The CPU oracle against every known-answer test the reference holds for the hot path.

Fixtures under tests/golden/ are DATA extracted from the reference's in-source tests by
tools/extract_golden.py (yuv/src/bt601.rs:198-483, deblock/src/deblock.rs:320-558).
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def bt():
    return json.load(open(os.path.join(GOLD, "bt601_reference_tests.json")))


@pytest.fixture(scope="module")
def db():
    return json.load(open(os.path.join(GOLD, "deblock_reference_tests.json")))


def yuv_to_rgb(yuv):
    # the reference's test helper (bt601.rs:62-93): a 4-pixel call with identical pixels
    y, cb, cr = yuv
    out = orc.yuv420_to_rgba([y, y, y, y], [cb, cb], [cr, cr], 4)
    px = out.reshape(4, 4)
    assert (px == px[0]).all() and px[0, 3] == 255
    return tuple(int(v) for v in px[0, :3])


def rgb_to_yuv_f32(rgb):
    # test-only inverse of the reference (bt601.rs:229-240), f32 arithmetic + round-half-away
    r, g, b = (np.float32(v) for v in rgb)
    f = np.float32
    y = f(16.0) + (f(65.481) * r) / f(255.0) + (f(128.553) * g) / f(255.0) + (f(24.966) * b) / f(255.0)
    u = f(128.0) - (f(37.797) * r) / f(255.0) - (f(74.203) * g) / f(255.0) + (f(112.0) * b) / f(255.0)
    v = f(128.0) + (f(112.0) * r) / f(255.0) - (f(93.786) * g) / f(255.0) - (f(18.214) * b) / f(255.0)
    rnd = lambda x: int(np.floor(np.float64(x) + 0.5))
    return rnd(y), rnd(u), rnd(v)


def test_bt601_single_pixel(bt):                       # bt601.rs:199-225 + 415
    for case in bt["single_pixel"]:
        assert yuv_to_rgb(case["yuv"]) == tuple(case["rgb"]), case


def test_bt601_inverse_helper_matches_reference_table(bt):   # bt601.rs:243-271
    for case in bt["rgb_to_yuv"]:
        assert rgb_to_yuv_f32(case["rgb"]) == tuple(case["yuv"]), case


def test_bt601_roundtrip(bt):                          # bt601.rs:274-326
    for case in bt["roundtrip_exact"]:
        assert yuv_to_rgb(rgb_to_yuv_f32(case["rgb_in"])) == tuple(case["rgb_out"]), case
    for rgb in bt["roundtrip_pm1_palette"]:
        rgb2 = yuv_to_rgb(rgb_to_yuv_f32(rgb))
        assert all(abs(a - b) <= 1 for a, b in zip(rgb, rgb2)), (rgb, rgb2)


def test_bt601_pictures(bt):                           # bt601.rs:329-483
    assert len(bt["pictures"]) == 10
    for p in bt["pictures"]:
        out = orc.yuv420_to_rgba(p["y"], p["cb"], p["cr"], p["y_width"])
        assert out.tolist() == p["rgba"], p["y_width"]


def test_quant_to_strength(db):                        # deblock.rs:5-8
    assert orc.quant_to_strength().tolist() == db["quant_to_strength"]


def test_deblock_process_const():                      # deblock.rs:323-334
    for val in range(256):
        for s in range(1, 13):
            assert orc.process_scalar(val, val, val, val, s) == (val,) * 4
            assert orc.process_simd_lane(val, val, val, val, s) == (val,) * 4


def test_deblock_process_symmetric_input():            # deblock.rs:337-349 (sampled: every 5th outer value)
    for outer in range(0, 256, 5):
        for inner in range(256):
            for s in (1, 2, 5, 8, 12):
                assert orc.process_scalar(outer, inner, inner, outer, s) == (outer, inner, inner, outer)
                assert orc.process_simd_lane(outer, inner, inner, outer, s) == (outer, inner, inner, outer)


def test_deblock_process_table(db):                    # deblock.rs:352-439
    assert len(db["process_rows"]) == 37
    for row in db["process_rows"]:
        a, b, c, d = row["in"]
        s, exp = row["strength"], tuple(row["out"])
        assert orc.process_scalar(a, b, c, d, s) == exp, row
        r = orc.process_scalar(d, c, b, a, s)              # direction symmetry
        assert (r[3], r[2], r[1], r[0]) == exp, row
        r = orc.process_scalar(255 - a, 255 - b, 255 - c, 255 - d, s)   # value symmetry
        assert tuple(255 - v for v in r) == exp, row


def test_deblock_image(db):                            # deblock.rs:442-558 (mixed SIMD/scalar semantics)
    img = db["image"]
    for s in ("4", "8", "12"):
        out = orc.deblock(img["data"], img["width"], int(s))
        assert out.tolist() == img["expected"][s], s


def test_deblock_floor_vs_trunc_disagree_somewhere():
    # the two semantics differ on a sizeable share of inputs; the oracle
    # must keep them distinct (floor in SIMD lanes, trunc in the scalar tails).
    rng = np.random.default_rng(7)
    diff = 0
    for _ in range(4000):
        a, b, c, d = (int(v) for v in rng.integers(0, 256, 4))
        s = int(rng.integers(1, 13))
        diff += orc.process_scalar(a, b, c, d, s) != orc.process_simd_lane(a, b, c, d, s)
    assert 100 < diff < 1000, diff
