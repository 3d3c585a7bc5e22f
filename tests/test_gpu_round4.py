"""GPU tests added in round 4.

* the EXACT path of the headline number: bench.py's own Workload (records resident in HBM, the GOP's I picture as dense
  blocks, the P pictures as sparse events) through `h263mi_batch_decode_events` on a frame-pipelined batch of 64 x 1080p
  -- k_frame with the event transport, the 4-band deal, both walk directions -- with EVERY picture's RGBA and the last
  planes of streams 0 / 31 / 63 checked against the oracle (round 3 checked the dense call at this geometry and the event
  call only at <= 352x288; the bench's own gate looks at the last picture only);

Everything goes through the C ABI and is compared with the oracle bit for bit."""
import os
import sys

import numpy as np
import pytest

import h263mi
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
W, H = 1920, 1080
MBS_PP = 120 * 68


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if h263mi.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu-marked tests must run on the MI355X box")


def assert_planes_equal(got, want, what=""):
    for g, e, name in zip(got, want, ("Y", "Cb", "Cr")):
        bad = np.flatnonzero(np.asarray(g) != np.asarray(e))
        assert bad.size == 0, "%s %s: %d bytes differ, first at %s" % (what, name, bad.size, bad[:8])


def _rgba_want(planes, strength, w=W):
    cw = (w + 1) // 2
    filt = planes if strength == 0 else tuple(orc.deblock(p, pw, strength) for p, pw in zip(planes, (w, cw, cw)))
    return orc.yuv420_to_rgba(*filt, w)


# ---------------------------------------------------------------------------------------------
# the timed path of bench.py, picture by picture
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("odd_start", [False, True])
def test_the_bench_path_events_k_frame_64x1080p_every_picture(odd_start):
    """bench.py:run_frames on bench.py:Workload(events=True), exactly: batch.decode (dense) for the I picture,
    batch.decode_events for the P pictures, Batch(64, 1920, 1080, pipeline_post=True).  1 I + 6 P; odd_start runs one
    more launch pair first, so that every picture index is walked in the other direction (the direction alternates from
    one k_frame launch to the next)."""
    import bench
    n, first_stream, gop = 64, 5, 7
    check = (0, 31, 63)
    strength = bench.STRENGTH
    wl = bench.Workload(h263mi, n, gop, first_stream, 0, None, events=True)
    # the transport the bench line reports: the I picture stayed dense, every P picture travels as events
    assert wl.frames[0].get("first") is None and all(fr.get("first") is not None for fr in wl.frames[1:])
    b = h263mi.Batch(n, W, H, 0, None, pipeline_post=True)
    scratch = h263mi.DeviceBuffer(n * W * H * 4)
    if odd_start:
        # I + one P + I again: three launches (k_recon, k_frame, k_frame) ahead of the checked GOP shift the parity
        for f in (0, 1):
            fr = wl.frames[f]
            if fr.get("first") is not None:
                b.decode_events(fr["ptype"], fr["mbs"].ptr, fr["first"].ptr, fr["ev"].ptr, fr["base"].ptr, 0, strength, scratch.ptr, None)
            else:
                b.decode(fr["ptype"], fr["mbs"].ptr, fr["co"].ptr, fr["base"].ptr, 0, strength, scratch.ptr, None)
    d_rgba = [h263mi.DeviceBuffer(n * W * H * 4) for _ in range(gop)]
    b.timing_reserve(4 * gop)
    b.timing_begin()
    for f in range(gop):                                     # bench.run_frames(pipeline=True), one RGBA buffer per frame index
        fr = wl.frames[f]
        if fr.get("first") is not None:
            b.decode_events(fr["ptype"], fr["mbs"].ptr, fr["first"].ptr, fr["ev"].ptr, fr["base"].ptr, 0, strength,
                            d_rgba[f].ptr, None)
        else:
            b.decode(fr["ptype"], fr["mbs"].ptr, fr["co"].ptr, fr["base"].ptr, 0, strength, d_rgba[f].ptr, None)
    b.sync()
    kt = b.timing_end()
    # every launch of the GOP but (without a picture before it) the first is a k_frame; the last post-processing runs at the sync
    assert kt.frame_launches == (gop if odd_start else gop - 1) and kt.post_launches == 1
    assert kt.recon_launches == (0 if odd_start else 1)
    for s in check:
        ref = None
        for f in range(gop):
            kind = h263mi.SYNTH_I_MIXED if f == 0 else h263mi.SYNTH_P
            mbs, co = h263mi.synth_picture_host(kind, W, H, first_stream + s, f)
            rc, ref = orc.decode_picture(W, H, mbs, co, ref)
            assert rc == 0
            got = d_rgba[f].download(W * H * 4, s * W * H * 4)
            bad = np.flatnonzero(got != _rgba_want(ref, strength))
            assert bad.size == 0, "RGBA stream %d frame %d: %d bytes differ, first at pixel %s" % (
                s, f, bad.size, divmod(int(bad[0]) // 4, W))
        assert_planes_equal(b.copy_yuv(s), ref, "last picture of stream %d" % s)
    b.close()


def _upload(arr):
    d = h263mi.DeviceBuffer(max(arr.nbytes, 16))
    if arr.nbytes:
        d.upload(arr)
    return d


# ---------------------------------------------------------------------------------------------
# "on error the state is unchanged" (state.rs:142, 464-487) with a failure injected at EVERY HIP call a submit makes:
# same size (the ping-pong must not lose the last picture) and a size change (the old frame store must survive until the
# new picture's launch is queued -- round 3 gave it up before the allocations and copies that can fail)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("new_size", [(176, 144), (96, 80)])
def test_a_failure_at_any_hip_call_leaves_the_state_unchanged(new_size):
    import recgen
    w0, h0 = 176, 144
    st = h263mi.H263State()
    m0, c0 = recgen.intra_picture(w0, h0, seed=3)
    st.submit_picture(w0, h0, m0, c0, h263mi.PICTURE_I, temporal_reference=7, pquant=9)
    rc, want0 = orc.decode_picture(w0, h0, m0, c0, None)
    assert rc == 0
    rgba0 = _rgba_want(want0, 5, w0)
    w1, h1 = new_size
    same = (w1, h1) == (w0, h0)
    if same:                       # a P picture on top of the I picture
        m1, c1 = recgen.inter_picture(w1, h1, seed=4, mv_range=30, p_4v=0.2, p_coded=0.5, quant=8)
        ptype, ref = h263mi.PICTURE_P, want0
    else:                          # another size: legal as an I picture (state.rs:157-176)
        m1, c1 = recgen.intra_picture(w1, h1, seed=5)
        ptype, ref = h263mi.PICTURE_I, None
    rc, want1 = orc.decode_picture(w1, h1, m1, c1, ref)
    assert rc == 0
    failures = 0
    for nth in range(1, 200):
        h263mi.debug_fail_nth_hip_call(nth)
        try:
            st.submit_picture(w1, h1, m1, c1, ptype, temporal_reference=8, pquant=8)
            fired = h263mi.debug_fail_nth_hip_call(0) <= 0
            assert not fired, "the %d-th HIP call failed and the submit reported success" % nth
            break                                              # the call needs fewer than nth HIP calls: it went through
        except h263mi.H263Error as e:
            assert e.code in (h263mi.ERR_OUT_OF_MEMORY,), (nth, e.code)
            h263mi.debug_fail_nth_hip_call(0)
            failures += 1
        # the previous picture is still the last picture: header, planes, and it still renders
        pic = st.get_last_picture()
        assert pic is not None, "failure at HIP call %d: the state lost its last picture" % nth
        assert (pic.width, pic.height) == (w0, h0) and pic.temporal_reference == 7
        assert_planes_equal(pic.as_yuv(), want0, "after a failure at HIP call %d" % nth)
        assert (st.render_rgba(5) == rgba0).all()
    else:
        pytest.fail("the submit never went through")
    assert failures >= (6 if not same else 3), failures          # allocations, copies, the launch: all were hit
    pic = st.get_last_picture()
    assert (pic.width, pic.height) == (w1, h1) and pic.temporal_reference == 8
    assert_planes_equal(pic.as_yuv(), want1, "the picture that finally went through")
    st.close()


def test_failure_injection_in_the_batch_parser_entry_keeps_every_stream():
    """h263mi_batch_decode_next_pictures_ex with a failure at each HIP call of the call: either nothing changed for any
    stream (parser state included: the same data decodes afterwards) or the call went through"""
    import recgen
    import sorenson_enc as enc
    from test_bitstream_e2e import make_codable
    w, h, n, q = 176, 144, 4, 6
    b = h263mi.Batch(n, w, h, pipeline_post=True)
    refs = [None] * n

    def pictures(f, intra):
        datas, recs = [], []
        for s in range(n):
            if intra:
                mbs, co = recgen.intra_picture(w, h, seed=77 * f + s, max_level=60)
                mbs = make_codable(mbs, q, s, 0)
            else:
                mbs, co = recgen.inter_picture(w, h, seed=77 * f + s, mv_range=32, p_4v=0.2, p_coded=0.4, quant=q, max_level=60)
                mbs = make_codable(mbs, q, s + f, 1)
            datas.append(enc.encode_picture(w, h, 0 if intra else 1, q, mbs, co, temporal_reference=f))
            recs.append((mbs, co))
        return datas, recs

    d_rgba = h263mi.DeviceBuffer(n * w * h * 4)
    datas, recs = pictures(0, True)
    used, rcs = b.decode_next_pictures_ex(datas, n_threads=2, strength=5, d_rgba=d_rgba.ptr)
    assert not any(rcs)
    for s in range(n):
        rc, refs[s] = orc.decode_picture(w, h, recs[s][0], recs[s][1], None)
    b.sync()
    datas, recs = pictures(1, False)
    failures = 0
    for nth in range(1, 100):
        h263mi.debug_fail_nth_hip_call(nth)
        try:
            used, rcs = b.decode_next_pictures_ex(datas, n_threads=2, strength=5, d_rgba=d_rgba.ptr)
            fired = h263mi.debug_fail_nth_hip_call(0) <= 0
            if fired:                                           # a failure behind the point of no return is not reported
                pass
            assert not any(rcs)
            break
        except h263mi.H263Error:
            h263mi.debug_fail_nth_hip_call(0)
            failures += 1
            b.sync()
            for s in range(n):
                assert_planes_equal(b.copy_yuv(s), refs[s], "stream %d after a failure at HIP call %d" % (s, nth))
    else:
        pytest.fail("the call never went through")
    assert failures >= 3
    b.sync()
    for s in range(n):
        rc, refs[s] = orc.decode_picture(w, h, recs[s][0], recs[s][1], refs[s])
        assert_planes_equal(b.copy_yuv(s), refs[s], "stream %d after the call went through" % s)
        assert (d_rgba.download(w * h * 4, s * w * h * 4) == _rgba_want(refs[s], 5, w)).all()
    b.close()


# ---------------------------------------------------------------------------------------------
# the drop-in caller's surface: decode_next_picture + RGBA into the caller's pinned buffer (no device buffer, no pageable
# copy): bit-exact, for allocated and for registered memory; pageable memory is refused
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h", [(176, 144), (1920, 1080), (100, 60)])
def test_render_rgba_pinned_matches_the_oracle(w, h):
    import recgen
    st = h263mi.H263State()
    pinned = h263mi.PinnedBuffer(w * h * 4 + 64)
    ref = None
    for f, strength in enumerate((0, 5, 12)):
        if f == 0:
            mbs, co = recgen.intra_picture(w, h, seed=f + 1)
        else:
            mbs, co = recgen.inter_picture(w, h, seed=f + 1, mv_range=30, p_4v=0.2, p_coded=0.4, quant=7)
        st.submit_picture(w, h, mbs, co, h263mi.PICTURE_I if f == 0 else h263mi.PICTURE_P, temporal_reference=f)
        rc, ref = orc.decode_picture(w, h, mbs, co, ref)
        assert rc == 0
        pinned.array[:] = 0x5a
        got = st.render_rgba_pinned(strength, pinned)
        assert (got == _rgba_want(ref, strength, w)).all(), (f, strength)
        assert (pinned.array[w * h * 4:] == 0x5a).all()           # nothing beyond the picture was written
        assert (st.render_rgba(strength) == got).all()            # the pageable form agrees
    # memory the caller owns, registered
    mine = np.zeros(w * h * 4 + 4096, np.uint8)
    h263mi.host_register(mine)
    got = st.render_rgba_pinned(12, mine)
    assert (got == _rgba_want(ref, 12, w)).all()
    h263mi.host_unregister(mine)
    with pytest.raises(h263mi.H263Error) as e:
        st.render_rgba_pinned(12, np.zeros(w * h * 4, np.uint8))  # pageable: refused, not copied behind the caller's back
    assert e.value.code == h263mi.ERR_INVALID_ARGUMENT
    pinned.free()
    st.close()


# ---------------------------------------------------------------------------------------------
# a decode call renders only the streams that decoded (non-pipelined batches too); an EMPTY picture is not "no
# picture"; device-resident event bounds are checked against n_events
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pipeline", [False, True])
def test_a_stream_that_sits_a_call_out_keeps_its_rgba(pipeline):
    import recgen
    w, h, n = 96, 64, 4
    b = h263mi.Batch(n, w, h, pipeline_post=pipeline)
    d_rgba = h263mi.DeviceBuffer(n * w * h * 4)
    refs = [None] * n

    def submit(f, intra, skip=None):
        mbs_all, co_all, base, at = [], [], [], 0
        for s in range(n):
            m, c = (recgen.intra_picture(w, h, seed=9 * f + s) if intra else
                    recgen.inter_picture(w, h, seed=9 * f + s, mv_range=20, p_coded=0.5, quant=7))
            if s != skip:
                rc, refs[s] = orc.decode_picture(w, h, m, c, None if intra else refs[s])
            mbs_all.append(m); co_all.append(c); base.append(at); at += len(c)
        d = (_upload(np.concatenate(mbs_all)), _upload(np.concatenate(co_all) if at else np.zeros((1, 64), np.int16)),
             _upload(np.array(base, np.uint64)))
        b.decode(h263mi.PICTURE_I if intra else h263mi.PICTURE_P, d[0].ptr, d[1].ptr, d[2].ptr, max(at, 1), 5, d_rgba.ptr)
        return d
    keep = [submit(0, True)]
    b.sync()
    marker = np.full(w * h * 4, 0xA7, np.uint8)
    d_rgba.upload(marker, 2 * w * h * 4)                          # stream 2's part of the output
    b.set_active([s != 2 for s in range(n)])
    keep.append(submit(1, False, skip=2))
    b.sync()
    got = d_rgba.download()
    for s in range(n):
        part = got[s * w * h * 4:(s + 1) * w * h * 4]
        if s == 2:
            assert (part == 0xA7).all(), "the RGBA of a stream that sat the call out was rewritten"
        else:
            assert (part == _rgba_want(refs[s], 5, w)).all(), s
    b.set_active(None)
    b.close()


def test_empty_picture_is_an_error_and_none_is_no_picture():
    import recgen
    import sorenson_enc as enc
    from test_bitstream_e2e import make_codable
    w, h, n, q = 176, 144, 3, 6
    b = h263mi.Batch(n, w, h)
    datas = []
    for s in range(n):
        mbs, co = recgen.intra_picture(w, h, seed=s + 1, max_level=60)
        datas.append(enc.encode_picture(w, h, 0, q, make_codable(mbs, q, s, 0), co, temporal_reference=0))
    used, rcs = b.decode_next_pictures_ex([datas[0], b"", None], n_threads=1)
    assert rcs[0] == 0 and rcs[1] < 0 and rcs[2] == 0 and used[1] == 0 and used[2] == 0
    assert b.stream_has_picture(0) and not b.stream_has_picture(1) and not b.stream_has_picture(2)
    with pytest.raises(h263mi.H263Error):                       # plain form: every stream decodes, an empty reader fails the call
        b.decode_next_pictures([datas[0], b"", datas[2]], n_threads=1)
    b.close()


def test_event_bounds_from_device_memory_are_checked_against_n_events():
    import recgen
    w, h, n = 96, 64, 2
    b = h263mi.Batch(n, w, h)
    mbs_all, firsts, evs, base, at_b, at_e = [], [], [], [], 0, 0
    want = []
    for s in range(n):
        m, c = recgen.intra_picture(w, h, seed=40 + s)
        rc, planes = orc.decode_picture(w, h, m, c, None)
        want.append(planes)
        intra_blocks = np.ones(len(c), bool)
        first, ev = h263mi.events_from_dense(c, intra_blocks)
        mbs_all.append(m); base.append(at_b)
        firsts.append(first[:-1].astype(np.uint32) + at_e)
        evs.append(ev); at_b += len(c); at_e += len(ev)
    first_all = np.concatenate(firsts + [np.array([at_e], np.uint32)])
    ev_all = np.concatenate(evs + [np.zeros(8, np.uint32)])
    d = (_upload(np.concatenate(mbs_all)), _upload(first_all), _upload(ev_all), _upload(np.array(base, np.uint64)))
    b.decode_events(h263mi.PICTURE_I, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, at_b, n_events=at_e)
    assert b.sync_streams() == [0, 0]
    for s in range(n):
        assert_planes_equal(b.copy_yuv(s), want[s], "stream %d" % s)
    # stream 1's last block claims events beyond the array: rejected, stream 0 unaffected
    bad = first_all.copy()
    bad[-1] = at_e + 1000
    d_bad = _upload(bad)
    b.decode_events(h263mi.PICTURE_I, d[0].ptr, d_bad.ptr, d[2].ptr, d[3].ptr, at_b, n_events=at_e)
    assert b.sync_streams() == [0, h263mi.ERR_INVALID_ARGUMENT]
    # a descending pair in stream 0: rejected as well (a caller that does not say n_events vouches for its arrays: nothing
    # is looked at then, and nothing is promised)
    bad = first_all.copy()
    bad[3] = bad[2] - 1 if bad[2] > 0 else 0
    bad[2] = bad[3] + 5
    d_bad2 = _upload(bad)
    b.decode_events(h263mi.PICTURE_I, d[0].ptr, d_bad2.ptr, d[2].ptr, d[3].ptr, at_b, n_events=at_e)
    rcs = b.sync_streams()
    assert rcs[0] == h263mi.ERR_INVALID_ARGUMENT and rcs[1] == 0
    # and the good arrays decode again afterwards
    b.decode_events(h263mi.PICTURE_I, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, at_b, n_events=at_e)
    assert b.sync_streams() == [0, 0]
    for s in range(n):
        assert_planes_equal(b.copy_yuv(s), want[s], "stream %d, after the rejected pictures" % s)
    b.close()


# ---------------------------------------------------------------------------------------------
# streams of different sizes behind one call (h263mi_mixed_*): QCIF + CIF + 1080p + an odd size in one set, every stream
# against its own oracle chain; a stream changes its size at an I picture (state.rs:157-176) and keeps decoding; a size
# change under inter prediction is that stream's PICTURE_FORMAT_INVALID and leaves it where it was
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pipeline", [True, False])
def test_mixed_sizes_in_one_set(pipeline):
    import recgen
    import sorenson_enc as enc
    from test_bitstream_e2e import make_codable
    q, strength = 6, 5
    sizes = [(176, 144), (352, 288), (1920, 1080), (176, 144), (100, 60), (352, 288), (176, 144), (1920, 1080)]
    n = len(sizes)
    m = h263mi.MixedBatch(n, pipeline_post=pipeline)
    refs = [None] * n
    cur = list(sizes)
    calls = 7
    rgba = [[h263mi.DeviceBuffer(1920 * 1080 * 4) for _ in range(n)] for _ in range(calls)]
    rendered = []                                                # (call, stream, w, h, expected RGBA)

    def picture(s, f, intra, size):
        w, h = size
        if intra:
            mbs, co = recgen.intra_picture(w, h, seed=1000 * f + s, max_level=60)
            mbs = make_codable(mbs, q, s, 0)
        else:
            mbs, co = recgen.inter_picture(w, h, seed=1000 * f + s, mv_range=32, p_4v=0.2, p_intra=0.05, p_coded=0.4, quant=q, max_level=60)
            mbs = make_codable(mbs, q, s + f, 1)
        return enc.encode_picture(w, h, 0 if intra else 1, q, mbs, co, temporal_reference=f), mbs, co

    def step(f, plan, expect_rc):
        """plan[s]: ('I' | 'P', size) or None; default ('P', the stream's current size)"""
        datas = []
        for s in range(n):
            item = plan.get(s, ("P", cur[s]))
            if item is None:
                datas.append(None)
                continue
            kind, size = item
            data, mbs, co = picture(s, f, kind == "I", size)
            datas.append(data)
            if expect_rc.get(s, 0) == 0:
                rc, refs[s] = orc.decode_picture(size[0], size[1], mbs, co, None if kind == "I" else refs[s])
                assert rc == 0
                cur[s] = size
                rendered.append((f, s, size[0], size[1], _rgba_want(refs[s], strength, size[0])))
        used, rcs, descs = m.decode_next_pictures(datas, n_threads=3, strength=strength, rgba=rgba[f])
        for s in range(n):
            assert rcs[s] == expect_rc.get(s, 0), "call %d stream %d: rc %d" % (f, s, rcs[s])
            if datas[s] is not None and rcs[s] == 0:
                assert used[s] > 0 and (descs[s].width, descs[s].height) == cur[s]
        assert not any(m.sync())
        for s in range(n):
            if refs[s] is None:
                assert m.stream_size(s) == (0, 0)
            else:
                assert m.stream_size(s) == cur[s]
                assert_planes_equal(m.copy_yuv(s), refs[s], "after call %d, stream %d" % (f, s))

    step(0, {s: ("I", sizes[s]) for s in range(n)}, {})
    assert m.size_classes() == 4
    step(1, {}, {})
    # stream 3 (QCIF) changes to CIF at an I picture; stream 6 (QCIF) tries to change under inter prediction: refused;
    # stream 4 sits the call out
    step(2, {3: ("I", (352, 288)), 6: ("P", (352, 288)), 4: None}, {6: h263mi.ERR_PICTURE_FORMAT_INVALID})
    assert m.size_classes() == 4
    step(3, {}, {})                                              # 3 predicts at CIF, 6 still at QCIF from its picture of call 1
    # a size no stream has had yet: a new class appears in mid-stream; the 1080p stream 7 shrinks to it
    step(4, {7: ("I", (640, 360)), 1: None}, {})
    assert m.size_classes() == 5
    step(5, {}, {})
    m.reset_stream(0)
    refs[0] = None
    step(6, {0: ("P", (176, 144))}, {0: h263mi.ERR_UNCODED_IFRAME_BLOCKS})
    for f, s, w, h, want in rendered:
        got = rgba[f][s].download(w * h * 4)
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "RGBA of call %d stream %d (%dx%d): %d bytes differ" % (f, s, w, h, bad.size)
    m.close()


@pytest.mark.parametrize("pipeline", [True, False])
def test_a_stream_that_keeps_changing_its_size_does_not_grow_the_set(pipeline):
    """The dimensions come out of the bitstream (a Sorenson custom format carries 16-bit sizes): a stream that brings a key
    frame of a new size every time must not leave a frame store behind for every size it ever had.  A class nobody belongs
    to any more is given up when the next class is made; the pictures -- also the P picture decoded at each size, and the
    RGBA of a picture whose class is given up right after it was rendered -- stay the oracle's."""
    import recgen
    import sorenson_enc as enc
    from test_bitstream_e2e import make_codable
    q, strength = 6, 5
    sizes = [(176, 144), (352, 288), (96, 80), (640, 360), (176, 144), (320, 240), (96, 80), (128, 96)]
    m = h263mi.MixedBatch(2, pipeline_post=pipeline)
    rgba = [[h263mi.DeviceBuffer(640 * 360 * 4) for _ in range(2)] for _ in range(2 * len(sizes))]
    refs, want = [None, None], []
    fixed = (352, 288)                                           # stream 1 stays at CIF
    call = 0
    for k, size in enumerate(sizes):
        for intra in (True, False):
            datas = []
            for s_, (w, h) in enumerate((size, fixed)):
                first = intra and (s_ == 0 or k == 0)
                if first:
                    mbs, co = recgen.intra_picture(w, h, seed=97 * call + s_, max_level=60)
                else:
                    mbs, co = recgen.inter_picture(w, h, seed=97 * call + s_, mv_range=32, p_4v=0.2, p_intra=0.05, p_coded=0.4,
                                                   quant=q, max_level=60)
                mbs = make_codable(mbs, q, call + s_, 0 if first else 1)
                datas.append(enc.encode_picture(w, h, 0 if first else 1, q, mbs, co, temporal_reference=call % 256))
                rc, refs[s_] = orc.decode_picture(w, h, mbs, co, None if first else refs[s_])
                assert rc == 0
                want.append((call, s_, w, h, _rgba_want(refs[s_], strength, w)))
            used, rcs, descs = m.decode_next_pictures(datas, n_threads=2, strength=strength, rgba=rgba[call])
            assert rcs == [0, 0], "call %d: %s" % (call, rcs)
            assert (descs[0].width, descs[0].height) == size
            # stream 0's class, stream 1's class, and at most the one stream 0 has just left
            assert m.size_classes() <= 3, "call %d: %d classes" % (call, m.size_classes())
            call += 1
    assert not any(m.sync())
    assert m.stream_size(0) == sizes[-1] and m.stream_size(1) == fixed
    for s_ in range(2):
        assert_planes_equal(m.copy_yuv(s_), refs[s_], "stream %d at the end" % s_)
    for c, s_, w, h, expect in want:
        got = rgba[c][s_].download(w * h * 4)
        assert np.array_equal(got, expect), "RGBA of call %d stream %d (%dx%d)" % (c, s_, w, h)
    m.close()


def test_a_picture_size_beyond_the_memory_limit_is_refused_for_its_stream_only():
    """A class holds two frames per member (slot); the size comes out of the bitstream.  With a limit that leaves room for
    the QCIF and CIF classes only, the stream that brings a 1080p key frame gets H263MI_ERR_OUT_OF_MEMORY and keeps what it
    had; the others decode; with the limit raised the same key frame goes through."""
    import recgen
    import sorenson_enc as enc
    from test_bitstream_e2e import make_codable
    q = 6
    n = 3
    m = h263mi.MixedBatch(n)

    def key_frame(w, h, seed):
        mbs, co = recgen.intra_picture(w, h, seed=seed, max_level=60)
        mbs = make_codable(mbs, q, seed, 0)
        rc, ref = orc.decode_picture(w, h, mbs, co, None)
        assert rc == 0
        return enc.encode_picture(w, h, 0, q, mbs, co, temporal_reference=0), ref

    small = [key_frame(176, 144, 1), key_frame(352, 288, 2), key_frame(176, 144, 3)]
    used, rcs, _ = m.decode_next_pictures([d for d, _ in small])
    assert rcs == [0, 0, 0] and not any(m.sync())
    have = m.frame_store_bytes()
    # (round 5: a class holds two frames per SLOT -- two QCIF members, one CIF member -- not per stream of the set)
    assert 2 * (2 * 176 * 144 + 352 * 288) * 3 // 2 <= have < 2 * n * (176 * 144 + 352 * 288) * 3 // 2
    m.set_memory_limit(have + 1024 * 1024)                       # nothing like 2 x 3 frames of 1080p (19 MB)
    big, big_ref = key_frame(1920, 1080, 4)
    nxt, nxt_ref = key_frame(352, 288, 5)
    used, rcs, _ = m.decode_next_pictures([big, nxt, None])
    assert rcs == [h263mi.ERR_OUT_OF_MEMORY, 0, 0], rcs
    assert not any(m.sync())
    assert m.stream_size(0) == (176, 144) and m.size_classes() == 2 and m.frame_store_bytes() == have
    assert_planes_equal(m.copy_yuv(0), small[0][1], "the refused stream keeps its picture")
    assert_planes_equal(m.copy_yuv(1), nxt_ref, "the others advance")
    m.set_memory_limit(0)
    used, rcs, _ = m.decode_next_pictures([big, None, None])
    assert rcs == [0, 0, 0] and not any(m.sync())
    assert m.stream_size(0) == (1920, 1080)
    assert_planes_equal(m.copy_yuv(0), big_ref, "the key frame with the limit lifted")
    m.close()


def test_a_hostile_picture_size_is_that_streams_format_error_everywhere():
    """65 535 x 65 535 in a Sorenson header: refused right behind the header (the parser is given the back-end's size limit)
    by the single state, by a batch (per stream) and by a mixed set (per stream); nobody else is disturbed."""
    import recgen
    import sorenson_enc as enc
    from test_bitstream_e2e import make_codable
    hostile = enc.encode_picture(65535, 65535, 0, 10, [], np.zeros((0, 64), np.int16))
    mbs, co = recgen.intra_picture(176, 144, seed=5, max_level=60)
    mbs = make_codable(mbs, 6, 5, 0)
    good = enc.encode_picture(176, 144, 0, 6, mbs, co)
    rc, ref = orc.decode_picture(176, 144, mbs, co, None)
    st = h263mi.H263State()
    st.decode_next_picture(good)
    with pytest.raises(h263mi.H263Error) as e:
        st.decode_next_picture(hostile)
    assert e.value.code == h263mi.ERR_PICTURE_FORMAT_INVALID
    assert_planes_equal(st.get_last_picture().as_yuv(), ref, "state unchanged")
    b = h263mi.Batch(2, 176, 144)
    used, rcs = b.decode_next_pictures_ex([good, hostile], n_threads=2)
    assert rcs == [0, h263mi.ERR_PICTURE_FORMAT_INVALID]
    b.sync()
    assert_planes_equal(b.copy_yuv(0), ref, "batch stream 0")
    b.close()
    m = h263mi.MixedBatch(2)
    used, rcs, _ = m.decode_next_pictures([hostile, good], n_threads=2)
    assert rcs == [h263mi.ERR_PICTURE_FORMAT_INVALID, 0] and not any(m.sync())
    assert m.size_classes() == 1 and m.stream_size(0) == (0, 0)
    assert_planes_equal(m.copy_yuv(1), ref, "mixed stream 1")
    m.close()


def test_overlap_mode_with_streams_that_have_drifted_apart():
    """H263MI_CFG_OVERLAP_POST (k_post on a second HIP stream) together with per-stream state words -- one stream
    sits calls out, so the streams' ping-pong positions differ and every wave reads its stream's word.  The words are now
    copied on the stream of the kernel that reads them, and a reconstruction waits for the post-processing of BOTH frame
    sets.  Many pictures back to back without a sync, every RGBA checked."""
    import recgen
    w, h, n, frames = 176, 144, 4, 8
    cw = (w + 1) // 2
    b = h263mi.Batch(n, w, h, overlap_post=True)
    refs = [None] * n
    d_rgba = [h263mi.DeviceBuffer(n * w * h * 4) for _ in range(frames)]
    want = []
    for f in range(frames):
        sits_out = {2: (1,), 3: (1, 3), 5: (0,)}.get(f, ())
        b.set_active([s not in sits_out for s in range(n)])
        mbs, cos = [], []
        for s in range(n):
            if f == 0:
                m, c = recgen.intra_picture(w, h, seed=11 * s + 3)
            else:
                m, c = recgen.inter_picture(w, h, seed=70 * f + s, mv_range=40, p_4v=0.3, p_coded=0.4, quant=6)
            mbs.append(m)
            cos.append(c)
            if s not in sits_out:
                rc, refs[s] = orc.decode_picture(w, h, m, c, refs[s])
                assert rc == 0
        b.submit_host(h263mi.PICTURE_I if f == 0 else h263mi.PICTURE_P, mbs, cos)
        b.render_rgba(5, d_rgba[f].ptr)                      # (renders every stream's LAST picture)
        want.append([orc.yuv420_to_rgba(*(orc.deblock(p, pw, 5) for p, pw in zip(refs[s], (w, cw, cw))), w) for s in range(n)])
    b.set_active(None)
    b.sync()
    for f in range(frames):
        got = d_rgba[f].download().reshape(n, -1)
        for s in range(n):
            assert (got[s] == want[f][s].reshape(-1)).all(), (f, s)
    for s in range(n):
        assert_planes_equal(b.copy_yuv(s), refs[s], "stream %d" % s)
    b.close()
