"""The CPU baseline runner of bench.py (oracle/native_bench.py + oracle/bench_streams.c): the -O3 -march=native
build must reproduce the portable oracle byte for byte, and the threaded runner must decode every stream like a
single-threaded run does (digest per thread)."""
import numpy as np

import recgen
from oracle import native_bench


def _stream(w, h, seed, n_frames=3):
    pics = [recgen.intra_picture(w, h, seed=seed)]
    for f in range(1, n_frames):
        pics.append(recgen.inter_picture(w, h, seed=seed * 100 + f, mv_range=20, p_4v=0.2, p_intra=0.05))
    return pics


def test_native_build_matches_portable_and_threads_agree():
    w, h = 176, 144
    nb = native_bench.NativeOracle()
    assert "-march=native" in nb.flags and "-ffp-contract=off" in nb.flags
    streams = [_stream(w, h, 1), _stream(w, h, 2)]
    nb.check_against_portable(w, h, streams[0], 5)
    nb.check_against_portable(w, h, streams[1], 12)
    one = [0]
    assert nb.run(w, h, streams[:1], 1, 2, 5, one) > 0
    two = [0]
    assert nb.run(w, h, streams[1:], 1, 2, 5, two) > 0
    many = [0] * 5
    assert nb.run(w, h, streams, 5, 2, 5, many) > 0
    assert many == [one[0], two[0], one[0], two[0], one[0]]       # thread t decodes stream t % 2
    assert one[0] != two[0]


def test_explicit_simd_stages_match_the_reference_goldens_and_the_oracle():
    """oracle/simd_stages.c (the explicit 128-bit deblock / BT.601 the CPU baseline times) against the reference's own
    vectors -- the 11x17 image at strengths 4 / 8 / 12 (deblock.rs:442-558), the ten pictures of bt601.rs:329-483 -- and
    against the oracle on sizes that exercise the SIMD region, the scalar tails and the width % 4 remainder."""
    import json
    import os
    from oracle import oracle as orc
    gold = os.path.join(os.path.dirname(__file__), "golden")
    nb = native_bench.NativeOracle()
    img = json.load(open(os.path.join(gold, "deblock_reference_tests.json")))["image"]
    for s in ("4", "8", "12"):
        assert nb.deblock_simd(np.array(img["data"], np.uint8), img["width"], int(s)).tolist() == img["expected"][s]
    for p in json.load(open(os.path.join(gold, "bt601_reference_tests.json")))["pictures"]:
        if not p["y"]:
            continue
        got = nb.yuv420_to_rgba_simd(np.array(p["y"], np.uint8), np.array(p["cb"], np.uint8), np.array(p["cr"], np.uint8), p["y_width"])
        assert got.tolist() == p["rgba"], p["y_width"]
    rng = np.random.default_rng(11)
    for w, h in ((8, 8), (9, 2), (10, 10), (16, 9), (37, 19), (100, 60), (176, 144), (133, 77)):
        plane = rng.integers(0, 256, w * h, dtype=np.uint8)
        for s in (1, 5, 12):
            assert np.array_equal(nb.deblock_simd(plane, w, s), np.asarray(orc.deblock(plane, w, s)).ravel()), (w, h, s)
        cw, ch = (w + 1) // 2, (h + 1) // 2
        cb, cr = rng.integers(0, 256, cw * ch, dtype=np.uint8), rng.integers(0, 256, cw * ch, dtype=np.uint8)
        assert np.array_equal(nb.yuv420_to_rgba_simd(plane, cb, cr, w), np.asarray(orc.yuv420_to_rgba(plane, cb, cr, w)).ravel()), (w, h)


def test_stage_selection_of_the_baseline_runner():
    """orc_bench_stages: every stage alone and all together run, and the all-stages digest does not depend on the form
    (scalar or explicit SIMD) of deblock / BT.601 -- they produce the same bytes"""
    w, h = 176, 144
    nb = native_bench.NativeOracle()
    streams = [_stream(w, h, 3)]
    nb.check_simd_stages(w, h, streams[0], 5)
    a, b = [0], [0]
    assert nb.run(w, h, streams, 1, 2, 5, a, stages=7, simd=False) > 0
    assert nb.run(w, h, streams, 1, 2, 5, b, stages=7, simd=True) > 0
    assert a == b
    for bits in (nb.RECON, nb.DEBLOCK, nb.RGBA, nb.DEBLOCK | nb.RGBA):
        for simd in (False, True):
            c, d = [0], [0]
            assert nb.run(w, h, streams, 1, 1, 5, c, stages=bits, simd=simd) > 0
            assert nb.run(w, h, streams, 2, 1, 5, d, stages=bits, simd=False) > 0 and len(d) == 2
    rep = nb.vectorisation_report()
    assert "deblock_horiz" in rep and "orc_yuv420_to_rgba" in rep
