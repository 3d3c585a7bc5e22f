"""The C++ fixture writer (tests/fixture_enc) against the Python one (tests/sorenson_enc.py), byte for byte, and the
generated pictures through the host parser: records in == records out.  Both writers are test infrastructure; what their
bits MEAN is pinned independently by the hand-derived known answers (tests/golden/*known_answers.json)."""
import numpy as np
import pytest

import fixture_enc as fx
import parselib as pl
import recgen
import sorenson_enc as enc
from test_bitstream_e2e import assert_records_equal, make_codable


@pytest.mark.parametrize("w,h", [(176, 144), (352, 288), (100, 60), (320, 240), (1920, 1080)])
def test_cpp_writer_equals_python_writer_byte_for_byte(w, h):
    big = w * h > 500000
    for f in range(2 if big else 4):
        intra = f == 0
        if big:
            mbs, co = (recgen.realistic_intra_picture(w, h, 40 + f) if intra else recgen.realistic_inter_picture(w, h, 50 + f))
        elif intra:
            mbs, co = recgen.intra_picture(w, h, seed=10 + f, max_level=1023 if f else 40)
        else:
            mbs, co = recgen.inter_picture(w, h, seed=20 + f, mv_range=31, p_4v=0.3, p_intra=0.15, p_coded=0.5,
                                           max_level=[3, 60, 1023][f % 3])
        q = [10, 1, 31, 17][f]
        mbs = make_codable(mbs, q, 7 * f + w, 0 if intra else 1)
        want = enc.encode_picture(w, h, 0 if intra else 1, q, mbs, co, temporal_reference=f + 3, deblock_flag=f & 1)
        got = fx.encode_picture(w, h, 0 if intra else 1, q, mbs, co, temporal_reference=f + 3, deblock_flag=f & 1)
        assert got == want, "frame %d: %d vs %d bytes" % (f, len(got), len(want))


def test_generated_pictures_parse_back_to_their_records():
    w, h = 352, 288
    for stream in range(3):
        for frame in range(4):
            q = 4 + 5 * stream
            data, mbs, co = fx.picture(99, stream, frame, w, h, frame == 0, q, deblock_flag=stream & 1, with_records=True)
            pl.context_reset()
            rc, d, got, got_co, used = pl.parse_picture(data, options=1)
            assert rc == 0
            assert (d.width, d.height, d.picture_type, d.pquant, d.use_deblocker) == (w, h, 0 if frame == 0 else 1, q, stream & 1)
            assert_records_equal(got, mbs)
            assert np.array_equal(got_co, co)
            if frame:
                skipped = ((mbs["cbp"] == 0) & ~mbs["mv"].reshape(len(mbs), -1).any(axis=1) & (mbs["mb_type"] == 0)).mean()
                assert 0.4 < skipped < 0.9, skipped
    # distinct keys give distinct pictures; the same key the same bytes
    a = fx.picture(1, 0, 1, w, h, False, 10)
    assert a == fx.picture(1, 0, 1, w, h, False, 10) and a != fx.picture(1, 1, 1, w, h, False, 10) != fx.picture(2, 0, 1, w, h, False, 10)


def test_corpus_is_generated_in_parallel_and_is_all_distinct():
    c = fx.corpus(5, 6, 4, 176, 144, [4, 7, 10, 13, 16, 19])
    flat = [p for s in c for p in s]
    assert len(set(flat)) == len(flat) == 24
