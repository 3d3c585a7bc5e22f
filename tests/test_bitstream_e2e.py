"""decode_next_picture(bytes): Sorenson Spark pictures serialised by tests/sorenson_enc.py from seeded
macroblock records, parsed back by the host parser (CPU tests) and decoded on the GPU (gpu tests) --
BASELINE config 1(iii): a QCIF Sorenson I picture through the real bitstream path."""
import numpy as np
import pytest

import parselib as pl
import recgen
import sorenson_enc as enc
from oracle import oracle as orc


def make_codable(mbs, pquant, seed, picture_type):
    """Give the records a quantiser walk DQUANT can express and the Q / non-Q types that go with it."""
    rng = np.random.default_rng(seed)
    mbs = mbs.copy()
    q = pquant
    for i in range(len(mbs)):
        m = mbs[i]
        t = int(m["mb_type"])
        intra, four = t in (3, 4), t in (2, 5)
        uncoded = picture_type != 0 and not intra and int(m["cbp"]) == 0 and not np.asarray(m["mv"]).any()
        step = int(rng.choice([0, 0, 0, -2, -1, 1, 2]))
        nq = int(np.clip(q + step, 1, 31))           # always within DQUANT's reach of the quantiser in force
        if uncoded:
            nq = q                                   # COD = 1 carries no DQUANT
            four = False
        dq = nq - q
        m["quant"] = nq
        m["mb_type"] = (4 if dq else 3) if intra else ((5 if dq else 2) if four else (1 if dq else 0))
        q = nq
    return mbs


def assert_records_equal(got, want):
    assert len(got) == len(want)
    for name in ("mb_type", "quant", "cbp", "kill", "mv", "intradc"):
        assert (got[name] == want[name]).all(), name
    coded = want["cbp"] != 0
    assert (got["coeff_index"][coded] == want["coeff_index"][coded]).all()


@pytest.mark.parametrize("w,h", [(176, 144), (128, 96), (320, 240), (100, 60), (16, 16), (352, 288), (300, 20)])
def test_intra_picture_round_trips_through_the_parser(w, h):
    mbs, coeffs = recgen.intra_picture(w, h, seed=w + h, max_level=40)
    mbs = make_codable(mbs, 7, w, 0)
    data = enc.encode_picture(w, h, 0, 7, mbs, coeffs, temporal_reference=5, deblock_flag=1)
    rc, d, got, gco, bits = pl.parse_picture(data)
    assert rc == 0 and (d.width, d.height, d.picture_type, d.pquant, d.use_deblocker, d.temporal_reference) == \
        (w, h, 0, 7, 1, 5)
    assert_records_equal(got, mbs)
    # intra blocks ignore coefficient 0 (the DC comes from INTRADC)
    want = coeffs.copy()
    want[:, 0] = 0
    assert (gco == want).all()
    assert bits <= len(data) * 8 and len(data) * 8 - bits < 32


@pytest.mark.parametrize("w,h", [(176, 144), (128, 96), (320, 240), (100, 60), (48, 32)])
def test_inter_picture_round_trips_through_the_parser(w, h):
    mbs, coeffs = recgen.inter_picture(w, h, seed=3 * w + h, mv_range=32, p_4v=0.3, p_intra=0.15, p_coded=0.4,
                                       quant=9, max_level=100, sparse_low=False)
    # a few escapes of every width: |level| > 63 needs the 11-bit form, runs > table range the 7-bit one
    mbs = make_codable(mbs, 9, h, 1)
    data = enc.encode_picture(w, h, 1, 9, mbs, coeffs)
    rc, d, got, gco, _ = pl.parse_picture(data)
    assert rc == 0 and d.picture_type == 1
    assert_records_equal(got, mbs)
    intra = np.isin(mbs["mb_type"], (3, 4))
    want = coeffs.copy()
    for i in np.flatnonzero(intra):
        n = bin(int(mbs[i]["cbp"])).count("1")
        want[int(mbs[i]["coeff_index"]):int(mbs[i]["coeff_index"]) + n, 0] = 0
    assert (gco == want).all()


def test_stuffing_short_pictures_and_run_overflow():
    w, h = 64, 48
    mbs, coeffs = recgen.inter_picture(w, h, seed=11, mv_range=20, p_coded=0.5, quant=6, max_level=20)
    mbs = make_codable(mbs, 6, 1, 1)
    # MCBPC stuffing in front of some macroblocks is skipped (state.rs:206)
    rc, _, got, _, _ = pl.parse_picture(enc.encode_picture(w, h, 1, 6, mbs, coeffs, stuffing_every=3))
    assert rc == 0
    assert_records_equal(got, mbs)
    # a picture that simply ends after 5 macroblocks: EOF at a macroblock boundary ends the picture (state.rs:411)
    rc, _, got, _, _ = pl.parse_picture(enc.encode_picture(w, h, 1, 6, mbs[:5], coeffs))
    assert rc == 0 and len(got) == 5
    # a run that walks past zigzag 63 sets the kill bit (rle.rs:125-127)
    first = int(np.flatnonzero(mbs["cbp"] & 1)[0])
    rc, _, got, _, _ = pl.parse_picture(enc.encode_picture(w, h, 1, 6, mbs, coeffs, overflow_blocks={(first, 0)}))
    assert rc == 0 and got[first]["kill"] == 1 and (np.delete(got["kill"], first) == 0).all()


def test_parser_error_classes():
    w, h = 64, 48
    mbs, coeffs = recgen.intra_picture(w, h, seed=2, max_level=30)
    mbs = make_codable(mbs, 5, 2, 0)
    data = enc.encode_picture(w, h, 0, 5, mbs, coeffs)
    # truncated inside a block: the block error fails the whole decode (the `?` of state.rs:287-381) ...
    cut = len(data) // 2
    rc = pl.parse_picture(data[:cut])[0]
    assert rc in (pl.EOF_ERR, 0)
    # ... and somewhere in the stream there is a cut that lands inside a block
    assert any(pl.parse_picture(data[:c])[0] == pl.EOF_ERR for c in range(cut, cut + 12))
    assert pl.parse_picture(bytes([0x12, 0x34, 0x56, 0x78, 0x9A]))[0] == -2          # no start code: MiddleOfBitstream
    # read as standard H.263 the Sorenson version field is a GOB number != 0: not a picture (picture.rs:660-662)
    assert pl.parse_picture(data, options=0)[0] == -2
    bad = bytearray(enc.encode_picture(w, h, 2, 5, mbs[:1], coeffs))                 # disposable P: macroblock.rs:461-465
    assert pl.parse_picture(bytes(bad))[0] == -17
    # reserved source format 7 has no dimensions: PictureFormatInvalid (state.rs:169-171)
    bw = enc.BitWriter()
    bw.put(1, 17); bw.put(1, 5); bw.put(0, 8); bw.put(7, 3); bw.put(0, 2); bw.put(0, 1); bw.put(5, 5); bw.put(0, 1)
    assert pl.parse_picture(bw.tobytes() + b"\x00\x00")[0] == -14


# ---------------------------------------------------------------------------------------------------------------
# GPU: H263State::decode_next_picture over bytes
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(176, 144), (320, 240), (100, 60)])
def test_decode_next_picture_stream_matches_oracle(w, h):
    import h263mi
    st = h263mi.H263State(h263mi.SORENSON_SPARK_BITSTREAM)
    mbs, coeffs = recgen.intra_picture(w, h, seed=w, max_level=40)
    mbs = make_codable(mbs, 8, 1, 0)
    data = enc.encode_picture(w, h, 0, 8, mbs, coeffs, temporal_reference=0, deblock_flag=1)
    hdr = st.parse_picture(data)
    assert (hdr.width, hdr.height, hdr.picture_type, hdr.pquant, hdr.use_deblocker) == (w, h, 0, 8, 1)
    used = st.decode_next_picture(data)
    assert len(data) - 4 <= used <= len(data)
    rc, ref = orc.decode_picture(w, h, mbs, coeffs, None)
    pic = st.get_last_picture()
    assert pic.use_deblocker == 1 and pic.pquant == 8
    for g, e in zip(pic.as_yuv(), ref):
        assert (g == e).all()
    for f in range(1, 4):
        mbs, coeffs = recgen.inter_picture(w, h, seed=10 * f + h, mv_range=32, p_4v=0.3, p_intra=0.1, p_coded=0.3,
                                           quant=8, max_level=60)
        mbs = make_codable(mbs, 8, f, 1)
        data = enc.encode_picture(w, h, 1, 8, mbs, coeffs, temporal_reference=f)
        st.decode_next_picture(data)
        rc, ref = orc.decode_picture(w, h, mbs, coeffs, ref)
        pic = st.get_last_picture()
        assert pic.temporal_reference == f
        for g, e in zip(pic.as_yuv(), ref):
            assert (g == e).all()
    s = int(h263mi.quant_to_strength()[8])
    cw = (w + 1) // 2
    filt = tuple(orc.deblock(p, pw, s) for p, pw in zip(ref, (w, cw, cw)))
    assert (st.render_rgba(s) == orc.yuv420_to_rgba(*filt, w)).all()
    st.close()


@pytest.mark.gpu
def test_decode_next_picture_errors_leave_state_unchanged():
    import h263mi
    w, h = 176, 144
    st = h263mi.H263State(h263mi.SORENSON_SPARK_BITSTREAM)
    pm, pc = recgen.inter_picture(w, h, seed=1, mv_range=16, quant=8)
    p_data = enc.encode_picture(w, h, 1, 8, make_codable(pm, 8, 1, 1), pc)
    with pytest.raises(h263mi.H263Error) as e:                 # P picture first: no reference (gather.rs:149)
        st.decode_next_picture(p_data)
    assert e.value.code == h263mi.ERR_UNCODED_IFRAME_BLOCKS and st.get_last_picture() is None
    im, ic = recgen.intra_picture(w, h, seed=2, max_level=30)
    im = make_codable(im, 8, 2, 0)
    i_data = enc.encode_picture(w, h, 0, 8, im, ic)
    st.decode_next_picture(i_data)
    before = st.get_last_picture().as_yuv()
    for bad in (i_data[:len(i_data) // 2 + 3], bytes([1, 2, 3, 4, 5, 6])):
        with pytest.raises(h263mi.H263Error):
            st.decode_next_picture(bad)
        for g, e in zip(st.get_last_picture().as_yuv(), before):
            assert (g == e).all()
    # a P picture that ends early: the missing macroblocks are zero-motion copies (state.rs:421-427)
    pm2 = make_codable(pm, 8, 1, 1)
    st.decode_next_picture(enc.encode_picture(w, h, 1, 8, pm2[:40], pc))
    rc, want = orc.decode_picture(w, h, pm2[:40], pc, before)
    for g, e in zip(st.get_last_picture().as_yuv(), want):
        assert (g == e).all()
    # a Sorenson picture handed to a standard H.263 state: its version field reads as a GOB number (picture.rs:660-662)
    st2 = h263mi.H263State(0)
    with pytest.raises(h263mi.H263Error) as e:
        st2.decode_next_picture(i_data)
    assert e.value.code == h263mi.ERR_MIDDLE_OF_BITSTREAM
    st.close()
    st2.close()
