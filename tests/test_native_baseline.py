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
