"""GPU tests added in round 2: BASELINE configs[3] at its stated size (64 streams), a bounded slice of the
randomised differential run, the reference-store cases the record-level API had no GPU test for, and the error
semantics of the batch API.  Everything goes through the C ABI and is compared with the oracle bit for bit."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import h263mi
import recgen
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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


# ---------------------------------------------------------------------------------------------
# BASELINE configs[3]: 64 independent 1080p streams in one batch, 1 I + 2 P, three streams against the oracle
# ---------------------------------------------------------------------------------------------
def test_64_stream_1080p_batch_matches_the_oracle_on_streams_0_31_63():
    n, first_stream = 64, 7
    b = h263mi.Batch(n, W, H)
    d_rgba = h263mi.DeviceBuffer(n * W * H * 4)
    refs = {0: None, 31: None, 63: None}
    for f in range(3):
        kind = h263mi.SYNTH_I_MIXED if f == 0 else h263mi.SYNTH_P
        cap = n * MBS_PP * (6 if f == 0 else 2)
        d_mbs, d_co, d_base = h263mi.DeviceBuffer(n * MBS_PP * 32), h263mi.DeviceBuffer(cap * 128), h263mi.DeviceBuffer(n * 8)
        total = h263mi.synth_batch_device(kind, W, H, n, first_stream, f, d_mbs.ptr, d_co.ptr, cap, d_base.ptr)
        # the one-call form, with the pool bound in force
        b.decode(h263mi.PICTURE_I if f == 0 else h263mi.PICTURE_P, d_mbs.ptr, d_co.ptr, d_base.ptr, total, 5, d_rgba.ptr)
        b.sync()
        for s in refs:
            mbs, co = h263mi.synth_picture_host(kind, W, H, first_stream + s, f)
            rc, refs[s] = orc.decode_picture(W, H, mbs, co, refs[s])
            assert rc == 0
            assert_planes_equal(b.copy_yuv(s), refs[s], "stream %d frame %d" % (s, f))
            filt = tuple(orc.deblock(p, pw, 5) for p, pw in zip(refs[s], (W, 960, 960)))
            got = d_rgba.download(W * H * 4, s * W * H * 4)
            assert (got == orc.yuv420_to_rgba(*filt, W)).all(), "RGBA stream %d frame %d" % (s, f)
        for d in (d_mbs, d_co, d_base):
            d.free()
    b.close()


# ---------------------------------------------------------------------------------------------
# bounded slice of tools/fuzz_gpu.py (fixed seed)
# ---------------------------------------------------------------------------------------------
def test_fuzz_slice_fixed_seed():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_gpu
    n_pic, n_px = fuzz_gpu.run(budget=20.0, seed=20261003, verbose=False)
    assert n_pic > 50


# ---------------------------------------------------------------------------------------------
# reference store: disposable P pictures and equal temporal references (state.rs:464-483, 72-78)
# ---------------------------------------------------------------------------------------------
def test_disposable_p_picture_at_record_level():
    """A disposable P picture becomes the LAST picture but not the reference (state.rs:474-480); since
    get_reference_picture hands out the last picture whenever a reference exists (state.rs:72-78, mirrored on
    purpose), the next P picture predicts from the disposable one."""
    w, h = 176, 144
    st = h263mi.H263State()
    mbs, co = recgen.inter_picture(w, h, seed=3)
    with pytest.raises(h263mi.H263Error) as e:                   # nothing to predict from yet
        st.submit_picture(w, h, mbs, co, h263mi.PICTURE_DISPOSABLE_P)
    assert e.value.code == h263mi.ERR_UNCODED_IFRAME_BLOCKS and st.get_last_picture() is None
    imbs, ico = recgen.intra_picture(w, h, seed=1)
    st.submit_picture(w, h, imbs, ico, h263mi.PICTURE_I, temporal_reference=0)
    rc, ref = orc.decode_picture(w, h, imbs, ico, None)
    assert st.has_reference_picture()
    for f, pt in enumerate((h263mi.PICTURE_DISPOSABLE_P, h263mi.PICTURE_DISPOSABLE_P, h263mi.PICTURE_P,
                            h263mi.PICTURE_DISPOSABLE_P, h263mi.PICTURE_P), start=1):
        mbs, co = recgen.inter_picture(w, h, seed=10 + f, mv_range=40, p_4v=0.3, p_intra=0.1, p_coded=0.4, quant=6)
        st.submit_picture(w, h, mbs, co, pt, temporal_reference=f)
        rc, ref = orc.decode_picture(w, h, mbs, co, ref)
        assert rc == 0
        pic = st.get_last_picture()
        assert pic.picture_type == pt and pic.temporal_reference == f
        assert_planes_equal(pic.as_yuv(), ref, "picture %d (type %d)" % (f, pt))
        assert st.has_reference_picture()
    st.close()


def test_equal_consecutive_temporal_references():
    """the reference store is keyed by temporal_reference (state.rs:29-38): a picture that re-uses the key of the
    picture it predicts from must still read the OLD picture and then replace it"""
    w, h = 100, 60
    st = h263mi.H263State()
    imbs, ico = recgen.intra_picture(w, h, seed=5)
    st.submit_picture(w, h, imbs, ico, h263mi.PICTURE_I, temporal_reference=7)
    rc, ref = orc.decode_picture(w, h, imbs, ico, None)
    for f in range(4):
        mbs, co = recgen.inter_picture(w, h, seed=30 + f, mv_range=60, p_4v=0.5, p_coded=0.5, quant=11)
        st.submit_picture(w, h, mbs, co, h263mi.PICTURE_P, temporal_reference=7)
        rc, ref = orc.decode_picture(w, h, mbs, co, ref)
        pic = st.get_last_picture()
        assert pic.temporal_reference == 7
        assert_planes_equal(pic.as_yuv(), ref, "P%d with equal TR" % f)
    st.close()


# ---------------------------------------------------------------------------------------------
# batch error semantics (state.rs:142, 464-487: an error leaves the state unchanged)
# ---------------------------------------------------------------------------------------------
def _upload(arr):
    d = h263mi.DeviceBuffer(max(arr.nbytes, 16))
    if arr.nbytes:
        d.upload(arr)
    return d


def test_batch_device_error_restores_the_previous_picture():
    w, h = 64, 48
    n_mb = 12
    b = h263mi.Batch(1, w, h)
    # first picture rejected by the device: no picture exists afterwards
    pm, pc = recgen.inter_picture(w, h, seed=1, p_coded=0.5)
    d_pm, d_pc = _upload(pm), _upload(pc)
    b.decode(h263mi.PICTURE_P, d_pm.ptr, d_pc.ptr, None, len(pc))
    with pytest.raises(h263mi.H263Error) as e:
        b.sync()
    assert e.value.code == h263mi.ERR_UNCODED_IFRAME_BLOCKS
    with pytest.raises(h263mi.H263Error) as e:
        b.copy_yuv(0)
    assert e.value.code == h263mi.ERR_NO_PICTURE
    # a good I picture, then a P picture with a coded block outside the pool: the I picture stays the last picture
    im, ic = recgen.intra_picture(w, h, seed=2)
    d_im, d_ic = _upload(im), _upload(ic)
    b.decode(h263mi.PICTURE_I, d_im.ptr, d_ic.ptr, None, len(ic))
    b.sync()
    rc, ref = orc.decode_picture(w, h, im, ic, None)
    assert_planes_equal(b.copy_yuv(0), ref, "I")
    bad = pm.copy()
    bad[5]["cbp"] = 0x3F
    bad[5]["coeff_index"] = len(pc) - 2                         # blocks len-2 .. len+3: four of them outside
    d_bad = _upload(bad)
    b.decode(h263mi.PICTURE_P, d_bad.ptr, d_pc.ptr, None, len(pc))
    with pytest.raises(h263mi.H263Error) as e:
        b.sync()
    assert e.value.code == h263mi.ERR_INVALID_ARGUMENT
    assert_planes_equal(b.copy_yuv(0), ref, "after the rejected P picture")
    # and it is still the reference: the valid P picture decodes from it
    b.decode(h263mi.PICTURE_P, d_pm.ptr, d_pc.ptr, None, len(pc))
    b.sync()
    rc, want = orc.decode_picture(w, h, pm, pc, ref)
    assert_planes_equal(b.copy_yuv(0), want, "P after the rejected one")
    # two pictures in flight when the error surfaces: nothing survives
    b.decode(h263mi.PICTURE_P, d_bad.ptr, d_pc.ptr, None, len(pc))
    b.decode(h263mi.PICTURE_P, d_pm.ptr, d_pc.ptr, None, len(pc))
    with pytest.raises(h263mi.H263Error):
        b.sync()
    with pytest.raises(h263mi.H263Error) as e:
        b.copy_yuv(0)
    assert e.value.code == h263mi.ERR_NO_PICTURE
    b.sync()                                                     # the error has been consumed
    b.close()
    assert n_mb == len(pm)


def test_batch_submit_host_validates_records():
    w, h, n = 64, 48, 2
    b = h263mi.Batch(n, w, h)
    ims = [recgen.intra_picture(w, h, seed=s) for s in range(n)]
    b.submit_host(h263mi.PICTURE_I, [m for m, c in ims], [c for m, c in ims])
    b.sync()
    before = [b.copy_yuv(s) for s in range(n)]
    pms = [recgen.inter_picture(w, h, seed=10 + s, p_coded=0.5) for s in range(n)]
    for field, value in (("coeff_index", 10 ** 6), ("quant", 0), ("mb_type", 9), ("cbp", 0x7F)):
        bad = pms[1][0].copy()
        bad[3][field] = value
        if field == "coeff_index":
            bad[3]["cbp"] = 1
        with pytest.raises(h263mi.H263Error) as e:
            b.submit_host(h263mi.PICTURE_P, [pms[0][0], bad], [pms[0][1], pms[1][1]])
        assert e.value.code == h263mi.ERR_INVALID_ARGUMENT, field
    b.sync()
    for s in range(n):
        assert_planes_equal(b.copy_yuv(s), before[s], "after rejected host submits")
    # a picture without a single coded block, but a record that claims one: rejected too (pool size 0 is a bound)
    empty = pms[0][0].copy()
    empty["cbp"] = 0
    empty["mb_type"] = 0
    liar = empty.copy()
    liar[2]["cbp"] = 4
    with pytest.raises(h263mi.H263Error):
        b.submit_host(h263mi.PICTURE_P, [empty, liar], [np.zeros((0, 64), np.int16)] * 2)
    b.close()


def test_state_ignores_the_batch_only_overlap_flag():
    """a state created with H263MI_CFG_OVERLAP_POST must not race its RGBA copy against k_post"""
    w, h = 352, 288
    st = h263mi.H263State(cfg_flags=1)
    mbs, co = recgen.intra_picture(w, h, seed=9)
    cw = (w + 1) // 2
    for rep in range(5):
        st.submit_picture(w, h, mbs, co, h263mi.PICTURE_I)
        rc, want = orc.decode_picture(w, h, mbs, co, None)
        filt = tuple(orc.deblock(p, pw, 7) for p, pw in zip(want, (w, cw, cw)))
        assert (st.render_rgba(7) == orc.yuv420_to_rgba(*filt, w)).all()
    st.close()


def test_probe_bandwidth_reports_a_plausible_ceiling():
    copy = h263mi.probe_bandwidth(h263mi.PROBE_COPY, 256 << 20, 5)
    read = h263mi.probe_bandwidth(h263mi.PROBE_READ, 256 << 20, 5)
    write = h263mi.probe_bandwidth(h263mi.PROBE_WRITE, 256 << 20, 5)
    for v in (copy, read, write):
        assert 500.0 < v < 20000.0, (copy, read, write)


# ---------------------------------------------------------------------------------------------
# N x decode_next_picture(bytes) in one call: parser threads -> events -> one launch
# ---------------------------------------------------------------------------------------------
def test_batch_decode_next_pictures_from_bitstreams():
    import sorenson_enc as enc
    from test_bitstream_e2e import make_codable
    w, h, n = 176, 144, 5
    b = h263mi.Batch(n, w, h)
    refs = [None] * n
    d_rgba = h263mi.DeviceBuffer(n * w * h * 4)
    cw = (w + 1) // 2
    for f in range(4):
        datas, recs = [], []
        for s in range(n):
            q = 4 + 3 * s
            if f == 0:
                mbs, co = recgen.intra_picture(w, h, seed=50 + s, max_level=60)
                mbs = make_codable(mbs, q, s, 0)
            else:
                mbs, co = recgen.inter_picture(w, h, seed=100 * f + s, mv_range=32, p_4v=0.3, p_intra=0.1, p_coded=0.4,
                                               quant=q, max_level=60)
                mbs = make_codable(mbs, q, s + f, 1)
            datas.append(enc.encode_picture(w, h, 0 if f == 0 else 1, q, mbs, co, temporal_reference=f))
            recs.append((mbs, co))
        used = b.decode_next_pictures(datas, n_threads=3)
        b.render_rgba(5, d_rgba.ptr)
        b.sync()
        for s in range(n):
            assert 0 < used[s] <= len(datas[s])
            rc, refs[s] = orc.decode_picture(w, h, recs[s][0], recs[s][1], refs[s] if f else None)
            assert rc == 0
            assert_planes_equal(b.copy_yuv(s), refs[s], "frame %d stream %d" % (f, s))
            filt = tuple(orc.deblock(p, pw, 5) for p, pw in zip(refs[s], (w, cw, cw)))
            assert (d_rgba.download(w * h * 4, s * w * h * 4) == orc.yuv420_to_rgba(*filt, w)).all()
    # one broken stream fails the call and changes nothing for any stream
    before = [b.copy_yuv(s) for s in range(n)]
    broken = list(datas)
    broken[3] = broken[3][:5]
    with pytest.raises(h263mi.H263Error):
        b.decode_next_pictures(broken)
    b.sync()
    for s in range(n):
        assert_planes_equal(b.copy_yuv(s), before[s], "after the failed call")
    # a picture of another size is refused
    mbs, co = recgen.intra_picture(128, 96, seed=1, max_level=60)
    other = enc.encode_picture(128, 96, 0, 7, make_codable(mbs, 7, 1, 0), co)
    with pytest.raises(h263mi.H263Error) as e:
        b.decode_next_pictures([other] * n)
    assert e.value.code == h263mi.ERR_PICTURE_FORMAT_INVALID
    b.close()


# ---------------------------------------------------------------------------------------------
# H263MI_CFG_PIPELINE_POST: reconstruction of picture f and post-processing of picture f-1 in one launch (k_frame)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h,n", [(176, 144, 3), (100, 60, 2), (33, 17, 1), (352, 288, 2), (1920, 1080, 2), (64, 36, 2),
                                   (48, 32, 2), (1, 1, 3), (16, 16, 1), (2000, 64, 1),
                                   # 16 pictures and more: a picture is dealt to 4 XCDs, two pictures side by side
                                   # (an odd count leaves the last pair half empty)
                                   (176, 144, 17), (100, 60, 16), (48, 32, 33), (352, 288, 19)])
def test_pipelined_batch_gives_the_same_pictures(w, h, n):
    """every picture's planes, RGBA and filtered planes against the oracle; the mode mixes with the immediate calls
    (submit / render_rgba), with reset, and an I picture in the middle of the chain"""
    b = h263mi.Batch(n, w, h, pipeline_post=True)
    cw, ch = (w + 1) // 2, (h + 1) // 2
    n_frames = 6
    d_rgba = [h263mi.DeviceBuffer(n * w * h * 4) for _ in range(n_frames)]
    d_planes = [h263mi.DeviceBuffer(n * (w * h + 2 * cw * ch)) for _ in range(n_frames)]
    refs = [None] * n
    want = []
    for f in range(n_frames):
        intra = f in (0, 4)
        mbs_all, co_all, base, at = [], [], [], 0
        for s in range(n):
            if intra:
                m, c = recgen.intra_picture(w, h, seed=11 * s + f)
            else:
                m, c = recgen.inter_picture(w, h, seed=100 * f + s, mv_range=40, p_4v=0.3, p_intra=0.1, p_coded=0.4, quant=8)
            rc, refs[s] = orc.decode_picture(w, h, m, c, None if intra else refs[s])
            assert rc == 0
            mbs_all.append(simlib_pad(m, w, h))
            co_all.append(c)
            base.append(at)
            at += len(c)
        strength = (5, 0, 12, 1, 7, 3)[f]
        want.append([(refs[s], strength) for s in range(n)])
        d_m = _upload(np.concatenate(mbs_all))
        d_c = _upload(np.concatenate(co_all) if at else np.zeros((1, 64), np.int16))
        d_b = _upload(np.array(base, np.uint64))
        pt = h263mi.PICTURE_I if intra else h263mi.PICTURE_P
        if f == 3:                                               # the immediate calls in the middle of the pipeline
            b.submit(pt, d_m.ptr, d_c.ptr, d_b.ptr)
            b.render_rgba(strength, d_rgba[f].ptr, d_planes[f].ptr)
        else:
            b.decode(pt, d_m.ptr, d_c.ptr, d_b.ptr, max(at, 1), strength, d_rgba[f].ptr, d_planes[f].ptr)
    b.sync()
    for s in range(n):
        assert_planes_equal(b.copy_yuv(s), refs[s], "last picture, stream %d" % s)
    ny, nc = w * h, cw * ch
    for f in range(n_frames):
        rgba, planes = d_rgba[f].download(), d_planes[f].download()
        for s in range(n):
            ref, strength = want[f][s]
            filt = ref if strength == 0 else tuple(orc.deblock(p, pw, strength) for p, pw in zip(ref, (w, cw, cw)))
            assert (rgba[s * ny * 4:(s + 1) * ny * 4] == orc.yuv420_to_rgba(*filt, w)).all(), (f, s)
            off = s * (ny + 2 * nc)
            got = (planes[off:off + ny], planes[off + ny:off + ny + nc], planes[off + ny + nc:off + ny + 2 * nc])
            assert_planes_equal(got, filt, "filtered planes frame %d stream %d" % (f, s))
    b.close()


def simlib_pad(mbs, w, h):
    import simlib
    return simlib.pad_records(mbs, w, h)


def test_pipelined_batch_rolls_back_a_rejected_picture():
    """pipeline mode + a picture the device rejects: the batch goes back to the previous picture, whose deferred
    post-processing has still been delivered, and carries on from there"""
    w, h = 64, 48
    b = h263mi.Batch(1, w, h, pipeline_post=True)
    cw = (w + 1) // 2
    im, ic = recgen.intra_picture(w, h, seed=21)
    pm, pc = recgen.inter_picture(w, h, seed=22, p_coded=0.5)
    bad = pm.copy()
    bad[2]["cbp"] = 0x3F
    bad[2]["coeff_index"] = len(pc) - 1
    d_im, d_ic, d_pm, d_pc, d_bad = _upload(im), _upload(ic), _upload(pm), _upload(pc), _upload(bad)
    rgba = [h263mi.DeviceBuffer(w * h * 4) for _ in range(3)]
    b.decode(h263mi.PICTURE_I, d_im.ptr, d_ic.ptr, None, len(ic), 5, rgba[0].ptr)
    b.decode(h263mi.PICTURE_P, d_bad.ptr, d_pc.ptr, None, len(pc), 5, rgba[1].ptr)     # launches I's post-processing too
    with pytest.raises(h263mi.H263Error) as e:
        b.sync()
    assert e.value.code == h263mi.ERR_INVALID_ARGUMENT
    rc, ref = orc.decode_picture(w, h, im, ic, None)
    filt = tuple(orc.deblock(p, pw, 5) for p, pw in zip(ref, (w, cw, cw)))
    # the I picture's deferred post-processing ran inside the launch that reconstructed the rejected picture
    assert (rgba[0].download() == orc.yuv420_to_rgba(*filt, w)).all()
    # two pictures were in flight since the last sync, so the batch cannot tell which one was rejected: none survives
    with pytest.raises(h263mi.H263Error) as e:
        b.copy_yuv(0)
    assert e.value.code == h263mi.ERR_NO_PICTURE
    # with a sync between them, the rejected picture alone is dropped
    b.decode(h263mi.PICTURE_I, d_im.ptr, d_ic.ptr, None, len(ic), 5, rgba[0].ptr)
    b.sync()
    b.decode(h263mi.PICTURE_P, d_bad.ptr, d_pc.ptr, None, len(pc), 5, rgba[1].ptr)
    with pytest.raises(h263mi.H263Error):
        b.sync()
    assert_planes_equal(b.copy_yuv(0), ref, "back to the I picture")
    b.decode(h263mi.PICTURE_P, d_pm.ptr, d_pc.ptr, None, len(pc), 5, rgba[2].ptr)
    b.sync()
    rc, want = orc.decode_picture(w, h, pm, pc, ref)
    assert_planes_equal(b.copy_yuv(0), want, "the valid P picture")
    filt = tuple(orc.deblock(p, pw, 5) for p, pw in zip(want, (w, cw, cw)))
    assert (rgba[2].download() == orc.yuv420_to_rgba(*filt, w)).all()
    b.close()


def test_batch_decode_next_pictures_mixed_types_and_lock_step():
    """streams of one batch need not agree on the picture type: one stream restarts with an I picture while the
    others continue with P pictures"""
    import sorenson_enc as enc
    from test_bitstream_e2e import make_codable
    w, h, n = 128, 96, 3
    b = h263mi.Batch(n, w, h)
    refs = [None] * n
    for f in range(4):
        datas = []
        for s in range(n):
            intra = f == 0 or (f == 2 and s == 1)
            if intra:
                mbs, co = recgen.intra_picture(w, h, seed=9 * s + f, max_level=60)
                mbs = make_codable(mbs, 6, s, 0)
            else:
                mbs, co = recgen.inter_picture(w, h, seed=77 * f + s, mv_range=32, p_4v=0.2, p_coded=0.4, quant=6, max_level=60)
                mbs = make_codable(mbs, 6, s + f, 1)
            datas.append(enc.encode_picture(w, h, 0 if intra else 1, 6, mbs, co, temporal_reference=f))
            rc, refs[s] = orc.decode_picture(w, h, mbs, co, None if intra else refs[s])
            assert rc == 0
        b.decode_next_pictures(datas, n_threads=2)
        b.sync()
        for s in range(n):
            assert_planes_equal(b.copy_yuv(s), refs[s], "frame %d stream %d" % (f, s))
    b.close()


# ---------------------------------------------------------------------------------------------
# launch timing: back-to-back launches of one kernel share one pair of events (a chain); the counts say how many
# launches each kernel had, whatever was queued in between
# ---------------------------------------------------------------------------------------------
def test_launch_timing_counts_launches_per_kernel():
    w, h, n = 176, 144, 3
    mbs_pp = 11 * 9
    pics = []
    for f in range(4):
        kind = h263mi.SYNTH_I_MIXED if f == 0 else h263mi.SYNTH_P
        cap = n * mbs_pp * 6
        d = (h263mi.DeviceBuffer(n * mbs_pp * 32), h263mi.DeviceBuffer(cap * 128), h263mi.DeviceBuffer(n * 8))
        h263mi.synth_batch_device(kind, w, h, n, 0, f, d[0].ptr, d[1].ptr, cap, d[2].ptr)
        pics.append((h263mi.PICTURE_I if f == 0 else h263mi.PICTURE_P, d))
    d_rgba = h263mi.DeviceBuffer(n * w * h * 4)
    # frame-pipelined batch: the first picture is a k_recon launch, the next three k_frame launches (one chain), the
    # post-processing of the last picture a k_post launch at the sync
    b = h263mi.Batch(n, w, h, pipeline_post=True)
    b.timing_reserve(16)
    b.timing_begin()
    for ptype, d in pics:
        b.decode(ptype, d[0].ptr, d[1].ptr, d[2].ptr, 0, 5, d_rgba.ptr)
    b.sync()
    kt = b.timing_end()
    assert (kt.recon_launches, kt.frame_launches, kt.post_launches) == (1, 3, 1)
    assert kt.recon_ms > 0 and kt.frame_ms > 0 and kt.post_ms > 0
    assert kt.frame_ms < 50 and kt.recon_ms < 50 and kt.post_ms < 50          # milliseconds of three tiny pictures
    # a copy back to the host between two launches ends the chain: still three launches, in two chains
    b.timing_begin()
    b.decode(*([pics[0][0]] + [x.ptr for x in pics[0][1]]), 0, 5, d_rgba.ptr)
    b.decode(*([pics[1][0]] + [x.ptr for x in pics[1][1]]), 0, 5, d_rgba.ptr)
    b.copy_yuv(0)
    b.decode(*([pics[2][0]] + [x.ptr for x in pics[2][1]]), 0, 5, d_rgba.ptr)
    b.sync()
    kt = b.timing_end()
    assert (kt.recon_launches, kt.frame_launches, kt.post_launches) == (1, 2, 1)
    b.close()
    # two launches per picture: kernels alternate, every launch is a chain of its own
    b = h263mi.Batch(n, w, h)
    b.timing_begin()
    for ptype, d in pics:
        b.submit(ptype, d[0].ptr, d[1].ptr, d[2].ptr)
        b.render_rgba(5, d_rgba.ptr, None)
    b.sync()
    kt = b.timing_end()
    assert (kt.recon_launches, kt.frame_launches, kt.post_launches) == (4, 0, 4)
    assert kt.recon_ms > 0 and kt.post_ms > 0
    b.close()


# ---------------------------------------------------------------------------------------------
# k_post: interior tiles (no bounds handling, floor division throughout) against edge tiles (general form).  Sizes at
# which each term of post_tile_is_interior decides for some tile: widths / heights that are not multiples of 8 (the
# last 8-aligned column / row starts the truncating region, deblock.rs:99-127 vs 29-42), chroma planes whose own
# 8-aligned limit ends the interior before the luma one does, pictures one tile wide or high (no interior tile at all).
# Every strength incl. 0; random planes with low contrast, so that the filters act on most edges.
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h", [(524, 300), (516, 292), (1924, 1084), (640, 360), (260, 68), (132, 36), (1028, 44)])
def test_post_interior_and_edge_tiles_agree_with_the_oracle(w, h):
    rng = np.random.default_rng(w * 13 + h)
    cw, ch = (w + 1) // 2, (h + 1) // 2
    n = 2
    b = h263mi.Batch(n, w, h)
    # a decoded picture whose planes are what we want to filter: intra macroblocks with DC only would be too smooth, so
    # decode random intra pictures (their planes are arbitrary bytes as far as the post-processing is concerned)
    refs = []
    mbs_l, co_l = [], []
    for s in range(n):
        mbs, co = recgen.intra_picture(w, h, seed=int(rng.integers(1 << 30)), max_level=6)
        mbs_l.append(mbs)
        co_l.append(co)
        rc, ref = orc.decode_picture(w, h, mbs, co, None)
        assert rc == 0
        refs.append(ref)
    b.submit_host(h263mi.PICTURE_I, mbs_l, co_l)
    d_rgba = h263mi.DeviceBuffer(n * w * h * 4)
    d_planes = h263mi.DeviceBuffer(n * (w * h + 2 * cw * ch))
    for strength in (0, 1, 5, 9, 12):
        want = []
        for s in range(n):
            filt = refs[s] if strength == 0 else tuple(orc.deblock(p, pw, strength) for p, pw in zip(refs[s], (w, cw, cw)))
            want.append((filt, orc.yuv420_to_rgba(*filt, w)))
        # RGBA only: interior tiles take the instantiations without bounds handling
        b.render_rgba(strength, d_rgba.ptr, None)
        b.sync()
        for s in range(n):
            got = d_rgba.download(w * h * 4, s * w * h * 4)
            bad = np.flatnonzero(got != want[s][1])
            assert bad.size == 0, "RGBA %dx%d strength %d stream %d: %d bytes differ, first at pixel %s" % (
                w, h, strength, s, bad.size, divmod(int(bad[0]) // 4, w))
        # RGBA + filtered planes: every tile takes the general form; same pixels
        b.render_rgba(strength, d_rgba.ptr, d_planes.ptr)
        b.sync()
        for s in range(n):
            assert (d_rgba.download(w * h * 4, s * w * h * 4) == want[s][1]).all(), (strength, s)
            got = d_planes.download(w * h + 2 * cw * ch, s * (w * h + 2 * cw * ch))
            assert (got == np.concatenate(want[s][0])).all(), (strength, s)
    b.close()
