"""GPU tests added in round 6 (ABI 7).

* The post-filter strength belongs to the PICTURE: the reference exports QUANT_TO_STRENGTH (deblock.rs:5-8) and hands out
  each picture's quantiser and USE_DEBLOCKER flag (picture.rs:61-64, types.rs:94-96, 216; set at parser/picture.rs:322) so
  that the consumer picks it per picture.  64 independent streams have 64 quantisers: the batch and mixed-set entry points
  take one strength per stream, or derive it from the header their own parser has just read
  (H263MI_STRENGTH_FROM_HEADER = use_deblocker ? QUANT_TO_STRENGTH[pquant] : 0), and the post-processing waves read their
  picture's value from the stream's word.
* Device arrays of a caller are CHECKED BY DEFAULT: without counts, the allocation a pointer lies in bounds what the waves
  read; H263MI_CFG_TRUSTED_ARRAYS is the explicit opt-out.
* Where the host side of a batch was placed (NUMA node of the device, pool CPUs, staging memory).

Everything goes through the C ABI and is compared bit for bit with the oracle."""
import os

import numpy as np
import pytest

import h263mi
import recgen
import simlib
import sorenson_enc as enc
from oracle import oracle as orc
from test_bitstream_e2e import make_codable

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
Q2S = [int(v) for v in orc.quant_to_strength()]


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if h263mi.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu-marked tests must run on the MI355X box")


def assert_planes_equal(got, want, what=""):
    for g, e, name in zip(got, want, ("Y", "Cb", "Cr")):
        bad = np.flatnonzero(np.asarray(g) != np.asarray(e))
        assert bad.size == 0, "%s %s: %d bytes differ, first at %s" % (what, name, bad.size, bad[:8])


def want_rgba(planes, w, strength):
    cw = (w + 1) // 2
    if strength:
        planes = tuple(orc.deblock(p, pw, strength) for p, pw in zip(planes, (w, cw, cw)))
    return orc.yuv420_to_rgba(*planes, w)


def header_strength(q, flag):
    return Q2S[q] if flag else 0


class _Streams:
    """n streams of w x h Sorenson pictures, each picture with a PQUANT and a deblocking flag of its own; the oracle decodes
    the same records beside them"""

    def __init__(self, n, w, h, seed):
        self.n, self.w, self.h = n, w, h
        self.rng = np.random.default_rng(seed)
        self.refs = [None] * n
        self.q = [0] * n
        self.flag = [0] * n
        self.f = 0

    def pictures(self, intra, only=None):
        datas = [None] * self.n
        for s in range(self.n):
            if only is not None and s not in only:
                continue
            q, flag = int(self.rng.integers(1, 32)), int(self.rng.integers(0, 2))
            sd = int(self.rng.integers(0, 1 << 30))
            if intra:
                mbs, co = recgen.intra_picture(self.w, self.h, seed=sd, max_level=40)
            else:
                mbs, co = recgen.inter_picture(self.w, self.h, seed=sd, mv_range=20, p_4v=0.2, p_intra=0.1, p_coded=0.4, max_level=30)
            mbs = make_codable(mbs, q, sd, 0 if intra else 1)
            datas[s] = enc.encode_picture(self.w, self.h, 0 if intra else 1, q, mbs, co, temporal_reference=self.f & 255,
                                          deblock_flag=flag)
            rc, self.refs[s] = orc.decode_picture(self.w, self.h, mbs, co, None if intra else self.refs[s])
            assert rc == 0
            self.q[s], self.flag[s] = q, flag
        self.f += 1
        return datas


@pytest.mark.parametrize("pipeline", [False, True], ids=["plain", "pipelined"])
def test_64_streams_each_rendered_with_the_strength_its_own_header_asks_for(pipeline):
    """64 streams with random PQUANT 1..31 and random deblocking flags through
    h263mi_batch_decode_next_pictures_ex with H263MI_STRENGTH_FROM_HEADER: every stream's RGBA is
    yuv420_to_rgba(deblock(planes, QUANT_TO_STRENGTH[its q])) -- or the plain conversion where its flag is clear."""
    n, w, h = 64, 176, 144
    st = _Streams(n, w, h, seed=601)
    b = h263mi.Batch(n, w, h, pipeline_post=pipeline)
    d_rgba = [h263mi.DeviceBuffer(n * w * h * 4) for _ in range(2)]
    seen = set()
    for f in range(4):
        datas = st.pictures(intra=f == 0)
        used, rcs = b.decode_next_pictures_ex(datas, n_threads=4, strength=h263mi.STRENGTH_FROM_HEADER, d_rgba=d_rgba[f & 1].ptr)
        assert not any(rcs), rcs
        want = [want_rgba(st.refs[s], w, header_strength(st.q[s], st.flag[s])) for s in range(n)]
        seen |= {header_strength(st.q[s], st.flag[s]) for s in range(n)}
        if pipeline and f < 3:
            continue                                     # (delivered by the next call's launch: checked after the last one)
        b.sync()
        for s in range(n):
            got = d_rgba[f & 1].download(w * h * 4, s * w * h * 4)
            assert np.array_equal(got, want[s]), "frame %d stream %d (q %d, flag %d)" % (f, s, st.q[s], st.flag[s])
    for s in (0, 17, 63):
        assert_planes_equal(b.copy_yuv(s), st.refs[s], "stream %d" % s)
    assert 0 in seen and len(seen) >= 8                  # the draw really mixed "off" with many strengths
    b.close()


def test_pipelined_rendering_keeps_each_pictures_own_strength_across_calls():
    """k_frame renders picture f - 1 inside the launch that reconstructs picture f: the strength that travels in the
    stream's word must be the one of the picture being RENDERED, not of the one being parsed in the same call."""
    n, w, h = 16, 128, 96
    st = _Streams(n, w, h, seed=602)
    b = h263mi.Batch(n, w, h, pipeline_post=True)
    bufs = [h263mi.DeviceBuffer(n * w * h * 4) for _ in range(3)]
    wants = []
    for f in range(3):
        datas = st.pictures(intra=f == 0)
        used, rcs = b.decode_next_pictures_ex(datas, n_threads=2, strength=h263mi.STRENGTH_FROM_HEADER, d_rgba=bufs[f].ptr)
        assert not any(rcs)
        wants.append([want_rgba(st.refs[s], w, header_strength(st.q[s], st.flag[s])) for s in range(n)])
    b.sync()
    for f in range(3):
        for s in range(n):
            assert np.array_equal(bufs[f].download(w * h * 4, s * w * h * 4), wants[f][s]), (f, s)
    b.close()


def test_streams_that_sit_a_call_out_and_per_stream_strength_arrays():
    """the caller's own choice per stream (strengths array) on the bitstream entry, with streams that have no picture in a
    call: they are neither decoded nor rendered, the others use THEIR entry of the array"""
    n, w, h = 12, 176, 144
    st = _Streams(n, w, h, seed=603)
    b = h263mi.Batch(n, w, h, pipeline_post=True)
    d_rgba = h263mi.DeviceBuffer(n * w * h * 4)
    strengths = [int(v) for v in np.random.default_rng(5).integers(0, 13, n)]
    used, rcs = b.decode_next_pictures_ex(st.pictures(True), n_threads=3, strengths=strengths, d_rgba=d_rgba.ptr)
    assert not any(rcs)
    b.sync()
    for s in range(n):
        assert np.array_equal(d_rgba.download(w * h * 4, s * w * h * 4), want_rgba(st.refs[s], w, strengths[s])), s
    before = [d_rgba.download(w * h * 4, s * w * h * 4).copy() for s in range(n)]
    part = {1, 2, 5, 11}
    strengths2 = [int(v) for v in np.random.default_rng(6).integers(0, 13, n)]
    used, rcs = b.decode_next_pictures_ex(st.pictures(False, only=part), n_threads=3, strengths=strengths2, d_rgba=d_rgba.ptr)
    assert not any(rcs)
    b.sync()
    for s in range(n):
        got = d_rgba.download(w * h * 4, s * w * h * 4)
        if s in part:
            assert np.array_equal(got, want_rgba(st.refs[s], w, strengths2[s])), s
        else:
            assert np.array_equal(got, before[s]), "stream %d sat the call out: its output must not be touched" % s
    # out of range values are refused before anything happens
    with pytest.raises(h263mi.H263Error):
        b.decode_next_pictures_ex(st.pictures(False), strengths=[13] * n, d_rgba=d_rgba.ptr)
    with pytest.raises(h263mi.H263Error):
        b.decode_next_pictures_ex(st.pictures(False), strength=13, d_rgba=d_rgba.ptr)
    b.close()


def _device_records(w, h, n, seed, intra, refs):
    """n pictures as device arrays (records, dense pool, bases) + the oracle's planes"""
    rng = np.random.default_rng(seed)
    mbs_all, co_all, base, at = [], [], [], 0
    for s in range(n):
        sd = int(rng.integers(0, 1 << 30))
        if intra:
            m, c = recgen.intra_picture(w, h, seed=sd, max_level=60)
        else:
            m, c = recgen.inter_picture(w, h, seed=sd, mv_range=30, p_4v=0.2, p_intra=0.1, p_coded=0.5, max_level=40)
        rc, refs[s] = orc.decode_picture(w, h, m, c, None if intra else refs[s])
        assert rc == 0
        mbs_all.append(simlib.pad_records(m, w, h))
        co_all.append(c)
        base.append(at)
        at += len(c)
    co = np.concatenate(co_all) if at else np.zeros((1, 64), np.int16)
    return np.concatenate(mbs_all), co, np.array(base, np.uint64), at


def _dev(arr, pad=0):
    arr = np.ascontiguousarray(arr)
    d = h263mi.DeviceBuffer(max(arr.nbytes + pad, 16))
    if arr.nbytes:
        d.upload(arr)
    return d


@pytest.mark.parametrize("pipeline", [False, True], ids=["plain", "pipelined"])
@pytest.mark.parametrize("events", [False, True], ids=["dense", "events"])
def test_device_record_entries_take_one_strength_per_stream(pipeline, events):
    """h263mi_batch_decode_ps / _decode_events_ps / _render_rgba_ps: strengths[s] for stream s"""
    n, w, h = 9, 100, 60
    refs = [None] * n
    b = h263mi.Batch(n, w, h, pipeline_post=pipeline)
    outs, wants, keep = [], [], []
    rng = np.random.default_rng(77)
    for f in range(3):
        mbs, co, base, blocks = _device_records(w, h, n, 900 + f, f == 0, refs)
        strengths = [int(v) for v in rng.integers(0, 13, n)]
        if f == 2:
            strengths = [7] * n                          # all equal: takes the uniform path inside, same answer
        d_out = h263mi.DeviceBuffer(n * w * h * 4)
        d_m, d_b = _dev(mbs), _dev(base)
        if events:
            intra_blk = np.zeros(len(co), bool)
            at = 0
            for s in range(n):
                for r in mbs[s * (len(mbs) // n):(s + 1) * (len(mbs) // n)]:
                    k = int(base[s]) + int(r["coeff_index"])
                    if int(r["mb_type"]) in (3, 4):
                        intra_blk[k:k + bin(int(r["cbp"])).count("1")] = True
            first, ev = h263mi.events_from_dense(co, intra_blk if blocks else None)
            d_f, d_e = _dev(first), _dev(np.concatenate([ev, np.zeros(8, np.uint32)]))
            keep.append((d_m, d_b, d_f, d_e))
            b.decode_events(h263mi.PICTURE_I if f == 0 else h263mi.PICTURE_P, d_m.ptr, d_f.ptr, d_e.ptr, d_b.ptr,
                            d_rgba=d_out.ptr, strengths=strengths)
        else:
            d_c = _dev(co)
            keep.append((d_m, d_b, d_c))
            b.decode(h263mi.PICTURE_I if f == 0 else h263mi.PICTURE_P, d_m.ptr, d_c.ptr, d_b.ptr, d_rgba=d_out.ptr,
                     strengths=strengths)
        outs.append(d_out)
        wants.append([want_rgba(refs[s], w, strengths[s]) for s in range(n)])
    b.sync()
    for f in range(3):
        for s in range(n):
            assert np.array_equal(outs[f].download(w * h * 4, s * w * h * 4), wants[f][s]), (f, s)
    # every stream's LAST picture again with other strengths, and the filtered planes beside the RGBA
    strengths = [int(v) for v in rng.integers(0, 13, n)]
    cw, ch = (w + 1) // 2, (h + 1) // 2
    per = w * h + 2 * cw * ch
    d_out, d_pl = h263mi.DeviceBuffer(n * w * h * 4), h263mi.DeviceBuffer(n * per)
    b.render_rgba(0, d_out.ptr, d_pl.ptr, strengths=strengths)
    b.sync()
    for s in range(n):
        assert np.array_equal(d_out.download(w * h * 4, s * w * h * 4), want_rgba(refs[s], w, strengths[s])), s
        pl = d_pl.download(per, s * per)
        want = refs[s] if strengths[s] == 0 else tuple(orc.deblock(p, pw, strengths[s]) for p, pw in zip(refs[s], (w, cw, cw)))
        assert np.array_equal(pl, np.concatenate(want)), s
    # FROM_HEADER has no meaning where the library saw no header
    with pytest.raises(h263mi.H263Error):
        b.render_rgba(h263mi.STRENGTH_FROM_HEADER, d_out.ptr)
    b.close()


@pytest.mark.parametrize("pipeline", [False, True], ids=["plain", "pipelined"])
def test_mixed_set_renders_every_stream_with_its_own_header_strength(pipeline):
    """h263mi_mixed_decode_next_pictures with H263MI_STRENGTH_FROM_HEADER over streams of three sizes, random PQUANT and
    deblocking flags, streams that skip calls -- and the caller's own array on the _ps form"""
    sizes = [(176, 144), (352, 288), (96, 80)]
    n = 14
    rng = np.random.default_rng(606)
    size = [sizes[int(rng.integers(0, 3))] for _ in range(n)]
    m = h263mi.MixedBatch(n, pipeline_post=pipeline)
    rgba = [h263mi.DeviceBuffer(w * h * 4) for (w, h) in size]
    refs = [None] * n
    last_want = [None] * n
    for call in range(4):
        datas, exp = [None] * n, {}
        own = [int(v) for v in rng.integers(0, 13, n)] if call == 3 else None
        for s in range(n):
            if call and rng.random() < 0.25:
                continue
            w, h = size[s]
            q, flag = int(rng.integers(1, 32)), int(rng.integers(0, 2))
            sd = int(rng.integers(0, 1 << 30))
            intra = call == 0 or refs[s] is None
            if intra:
                mbs, co = recgen.intra_picture(w, h, seed=sd, max_level=40)
            else:
                mbs, co = recgen.inter_picture(w, h, seed=sd, mv_range=12, p_4v=0.1, p_intra=0.1, p_coded=0.4, max_level=20)
            mbs = make_codable(mbs, q, sd, 0 if intra else 1)
            datas[s] = enc.encode_picture(w, h, 0 if intra else 1, q, mbs, co, temporal_reference=call, deblock_flag=flag)
            rc, refs[s] = orc.decode_picture(w, h, mbs, co, None if intra else refs[s])
            assert rc == 0
            exp[s] = own[s] if own else header_strength(q, flag)
        used, rcs, descs = m.decode_next_pictures(datas, n_threads=3, strength=h263mi.STRENGTH_FROM_HEADER if own is None else 0,
                                                  rgba=rgba, strengths=own)
        assert not any(rcs), rcs
        assert not any(m.sync())
        for s in range(n):
            if s in exp:
                last_want[s] = want_rgba(refs[s], size[s][0], exp[s])
                assert descs[s].pquant >= 1
            if last_want[s] is not None:
                got = rgba[s].download(size[s][0] * size[s][1] * 4)
                assert np.array_equal(got, last_want[s]), "call %d stream %d size %s strength %s" % (call, s, size[s], exp.get(s))
    for s in range(n):
        if refs[s] is not None:
            assert_planes_equal(m.copy_yuv(s), refs[s], "stream %d" % s)
    m.close()


def test_state_renders_with_the_strength_of_its_last_pictures_header():
    """h263mi_render_rgba(H263MI_STRENGTH_FROM_HEADER) on one H263State: as_header().quantizer and USE_DEBLOCKER of the
    last picture decide (picture.rs:61-64)"""
    w, h = 176, 144
    st = h263mi.H263State()
    ref = None
    for f, (q, flag) in enumerate([(31, 1), (4, 1), (19, 0), (10, 1)]):
        intra = f == 0
        mbs, co = (recgen.intra_picture(w, h, seed=50 + f, max_level=40) if intra else
                   recgen.inter_picture(w, h, seed=50 + f, mv_range=10, p_4v=0.2, p_intra=0.1, p_coded=0.5, max_level=20))
        mbs = make_codable(mbs, q, f, 0 if intra else 1)
        st.decode_next_picture(enc.encode_picture(w, h, 0 if intra else 1, q, mbs, co, temporal_reference=f, deblock_flag=flag))
        rc, ref = orc.decode_picture(w, h, mbs, co, None if intra else ref)
        pic = st.get_last_picture()
        assert (pic.pquant, pic.use_deblocker) == (q, flag)
        assert np.array_equal(st.render_rgba(h263mi.STRENGTH_FROM_HEADER), want_rgba(ref, w, header_strength(q, flag))), (q, flag)
    pinned = h263mi.PinnedBuffer(w * h * 4)
    assert np.array_equal(st.render_rgba_pinned(h263mi.STRENGTH_FROM_HEADER, pinned), want_rgba(ref, w, Q2S[10]))
    pinned.free()
    st.close()


# ---- device arrays are checked by default -----------------------------------------------------------------------------
def _one_intra_picture_on_device(w, h, seed):
    mbs, co = recgen.intra_picture(w, h, seed=seed, max_level=60)
    rc, ref = orc.decode_picture(w, h, mbs, co, None)
    assert rc == 0
    intra_blk = np.ones(len(co), bool)
    first, ev = h263mi.events_from_dense(co, intra_blk)
    return mbs, co, first, ev, ref


def test_hostile_event_offsets_are_rejected_without_the_caller_saying_any_size():
    """ABI 7: h263mi_batch_decode_events with coeff_pool_blocks = 0 and n_events = 0 on an ordinary batch -- the allocations
    bound what is read.  Offsets that run backwards, that point gigabytes beyond the events, a coded block index far outside
    the pool: the picture is rejected at the sync (nothing outside the caller's arrays is read), and the batch goes on."""
    w, h = 176, 144
    mbs, co, first, ev, ref = _one_intra_picture_on_device(w, h, 31)
    b = h263mi.Batch(1, w, h)
    d_m, d_f, d_e = _dev(mbs), _dev(first), _dev(ev)            # (exact sizes: not one word to spare behind the arrays)
    b.decode_events(h263mi.PICTURE_I, d_m.ptr, d_f.ptr, d_e.ptr)
    b.sync()
    assert_planes_equal(b.copy_yuv(0), ref, "sizes taken from the allocations")
    hostile = []
    f1 = first.copy(); f1[len(f1) // 2] = 0xfffffff0; hostile.append(("offset beyond the events", mbs, f1))
    f2 = first.copy(); f2[3], f2[4] = f2[4] + 5, f2[3]; hostile.append(("offsets not ascending", mbs, f2))
    f3 = first.copy(); f3[1:] += 1 << 28; hostile.append(("every block 2^28 words further on", mbs, f3))
    m4 = mbs.copy(); m4["coeff_index"][len(m4) // 2] = 1 << 24; hostile.append(("coded block far outside the pool", m4, first))
    for what, mm, ff in hostile:
        d_m2, d_f2 = _dev(mm), _dev(ff)
        b.decode_events(h263mi.PICTURE_I, d_m2.ptr, d_f2.ptr, d_e.ptr)
        rcs = b.sync_streams()
        assert rcs == [h263mi.ERR_INVALID_ARGUMENT], (what, rcs)
        assert_planes_equal(b.copy_yuv(0), ref, "after '%s' the good picture is the last picture again" % what)
        b.decode_events(h263mi.PICTURE_I, d_m.ptr, d_f.ptr, d_e.ptr)     # and the batch goes on
        b.sync()
        assert_planes_equal(b.copy_yuv(0), ref, what)
    b.close()


def test_arrays_too_small_for_the_batch_are_refused_before_anything_is_queued():
    w, h = 176, 144
    mbs, co, first, ev, ref = _one_intra_picture_on_device(w, h, 32)
    b = h263mi.Batch(2, w, h)                                    # TWO streams: one picture's worth of records is too little
    d_m, d_c, d_f, d_e = _dev(mbs), _dev(co), _dev(first), _dev(ev)
    for call in (lambda: b.decode(h263mi.PICTURE_I, d_m.ptr, d_c.ptr),
                 lambda: b.decode_events(h263mi.PICTURE_I, d_m.ptr, d_f.ptr, d_e.ptr),
                 lambda: b.submit(h263mi.PICTURE_I, d_m.ptr, d_c.ptr)):
        with pytest.raises(h263mi.H263Error) as e:
            call()
        assert e.value.code == h263mi.ERR_INVALID_ARGUMENT
    assert not b.stream_has_picture(0)
    # sizes the caller claims beyond its allocations are refused too
    both = np.concatenate([mbs, mbs])
    d_m2 = _dev(both)
    base = _dev(np.array([0, 0], np.uint64))
    with pytest.raises(h263mi.H263Error):
        b.decode(h263mi.PICTURE_I, d_m2.ptr, d_c.ptr, base.ptr, coeff_pool_blocks=len(co) + 1)
    with pytest.raises(h263mi.H263Error):
        b.decode_events(h263mi.PICTURE_I, d_m2.ptr, d_f.ptr, d_e.ptr, base.ptr, n_events=len(ev) + 1)
    # a pointer into the MIDDLE of an allocation is bounded by what is left of it
    b.decode(h263mi.PICTURE_I, d_m2.ptr, d_c.ptr, base.ptr)
    b.sync()
    assert_planes_equal(b.copy_yuv(1), ref, "both streams read the same pool")
    half = h263mi.DeviceBuffer(both.nbytes * 2)
    half.upload(both, both.nbytes)
    b.decode(h263mi.PICTURE_I, half.at(both.nbytes), d_c.ptr, base.ptr)
    b.sync()
    with pytest.raises(h263mi.H263Error):
        b.decode(h263mi.PICTURE_I, half.at(both.nbytes + 32), d_c.ptr, base.ptr)        # one record short
    # ... and so are output buffers that cannot hold what the launch writes for the batch's two streams
    small = h263mi.DeviceBuffer(2 * w * h * 4 - 4)
    small_planes = h263mi.DeviceBuffer(2 * (w * h + 2 * ((w + 1) // 2) * ((h + 1) // 2)) - 4)
    for call in (lambda: b.decode(h263mi.PICTURE_I, d_m2.ptr, d_c.ptr, base.ptr, strength=3, d_rgba=small.ptr),
                 lambda: b.render_rgba(3, small.ptr),
                 lambda: b.render_rgba(3, None, small_planes.ptr)):
        with pytest.raises(h263mi.H263Error) as e:
            call()
        assert e.value.code == h263mi.ERR_INVALID_ARGUMENT
    ok = h263mi.DeviceBuffer(2 * w * h * 4)
    b.render_rgba(3, ok.ptr)
    b.sync()
    assert np.array_equal(ok.download(w * h * 4, w * h * 4), want_rgba(ref, w, 3))
    b.close()


def test_trusted_arrays_is_the_explicit_opt_out():
    """H263MI_CFG_TRUSTED_ARRAYS: sizes not given = the caller vouches, nothing is looked up or checked (valid arrays
    decode exactly as on a checked batch)"""
    w, h = 176, 144
    mbs, co, first, ev, ref = _one_intra_picture_on_device(w, h, 33)
    for trusted in (False, True):
        b = h263mi.Batch(1, w, h, pipeline_post=True, trusted_arrays=trusted)
        d_rgba = h263mi.DeviceBuffer(w * h * 4)
        d_m, d_f, d_e = _dev(mbs), _dev(first), _dev(np.concatenate([ev, np.zeros(64, np.uint32)]))
        b.decode_events(h263mi.PICTURE_I, d_m.ptr, d_f.ptr, d_e.ptr, strength=6, d_rgba=d_rgba.ptr)
        b.sync()
        assert_planes_equal(b.copy_yuv(0), ref, "trusted %s" % trusted)
        assert np.array_equal(d_rgba.download(), want_rgba(ref, w, 6))
        b.close()


def test_host_placement_report():
    """h263mi_batch_host_placement: whatever the box looks like, the report is consistent -- pool CPUs lie inside the
    process's affinity mask, and when the device's NUMA node is known the staging memory lies on it"""
    n, w, h = 8, 176, 144
    st = _Streams(n, w, h, seed=610)
    b = h263mi.Batch(n, w, h)
    node, mem_node, cpus = b.host_placement()
    assert mem_node == -1 and cpus == []                 # nothing made yet
    used, rcs = b.decode_next_pictures_ex(st.pictures(True), n_threads=4)
    assert not any(rcs)
    b.sync()
    node, mem_node, cpus = b.host_placement()
    assert set(cpus) <= set(os.sched_getaffinity(0))
    if node >= 0 and os.environ.get("H263MI_NUMA", "1") != "0":
        assert mem_node in (node, -1), (node, mem_node)
        assert cpus, "a known node confines the pool's threads"
    b.close()


# ---- hand-derived bit strings with content, decoded on the GPU ---------------------------------------------------------
def _fixture_records(pic):
    """the EXPECTED records and coefficient blocks of a fixture picture (tests/golden/macroblock_content_known_answers.json):
    typed by hand from the reference's text, never parsed"""
    n = len(pic["macroblocks"])
    mbs = np.zeros(n, orc.MB_RECORD_DTYPE)
    blocks = []
    quant = pic["quant"]
    for k, mb in enumerate(pic["macroblocks"]):
        e = mb["expect"]
        quant = e.get("quant", quant)
        mbs[k]["mb_type"], mbs[k]["quant"], mbs[k]["cbp"], mbs[k]["kill"] = e["mb_type"], quant, e["cbp"], e["kill"]
        mbs[k]["mv"] = e["mv"]
        mbs[k]["intradc"] = e.get("intradc", [0] * 6)
        mbs[k]["coeff_index"] = len(blocks)
        for blk in e["blocks"]:
            c = np.zeros(64, np.int16)
            for p, v in (blk or {}).items():
                c[int(p)] = v
            blocks.append(c)
    return mbs, np.array(blocks, np.int16).reshape(-1, 64)


def test_hand_derived_pictures_with_content_decode_to_what_the_reference_text_gives():
    """pictures A (I) and B (P on A) of the fixture -- literal bit strings -- through
    h263mi_decode_next_picture on the MI355X.  The pixels must equal the oracle's decode of the EXPECTED records (typed by
    hand beside the bits), so a symmetric error of encoder and parser cannot hide here: there is no encoder.  Where a block
    is DC-only the answer is checked by the closed form as well: an intra block with INTRADC code c and no TCOEF is c in
    every pixel (0xFF: 128) (rle.rs:94-109, idct.rs:114-130); a block whose run overflowed is empty, its INTRADC included."""
    from test_parser import _content_fixture, fixture_picture_bytes
    pics = {p["name"][0]: p for p in _content_fixture()["pictures"]}
    a, b = pics["A"], pics["B"]
    w, h = a["width"], a["height"]
    st = h263mi.H263State()
    ref = None
    for pic in (a, b):
        data = fixture_picture_bytes(pic)
        used = st.decode_next_picture(data)
        assert used in (len(data) - 1, len(data))      # (reader.commit() drains WHOLE bytes: the last byte may be half padding)
        mbs, co = _fixture_records(pic)
        rc, ref = orc.decode_picture(w, h, simlib.pad_records(mbs, w, h), co, ref)
        assert rc == 0
        got = st.get_last_picture()
        assert (got.picture_type, got.pquant, got.use_deblocker) == (pic["picture_type"], pic["quant"], pic["use_deblocker"])
        assert_planes_equal(got.as_yuv(), ref, pic["name"])
        if pic is a:
            y = got.as_luma().reshape(h, w)
            cb, cr = got.as_chroma_b().reshape(h // 2, w // 2), got.as_chroma_r().reshape(h // 2, w // 2)
            for k, mb in enumerate(pic["macroblocks"]):
                e = mb["expect"]
                mx, my = (k % 3) * 16, (k // 3) * 16
                for blk in range(4):
                    if not (e["cbp"] >> blk) & 1:
                        code = e["intradc"][blk]
                        tile = y[my + 8 * (blk >> 1):my + 8 * (blk >> 1) + 8, mx + 8 * (blk & 1):mx + 8 * (blk & 1) + 8]
                        assert (tile == (128 if code == 255 else code)).all(), (k, blk, code, int(tile[0, 0]))
                for blk, plane in ((4, cb), (5, cr)):
                    tile = plane[my // 2:my // 2 + 8, mx // 2:mx // 2 + 8]
                    if (e["kill"] >> blk) & 1:
                        assert (tile == 0).all(), "a block whose run overflowed contributes nothing, INTRADC included"
                    elif not (e["cbp"] >> blk) & 1:
                        assert (tile == e["intradc"][blk]).all(), (k, blk)
        # rendered with what the picture's own header asks for
        assert np.array_equal(st.render_rgba(h263mi.STRENGTH_FROM_HEADER),
                              want_rgba(ref, w, header_strength(pic["quant"], pic["use_deblocker"])))
    # picture D: ITU-T H.263 (PLUSPTYPE, custom format 32 x 16, 8-bit escape) on a state WITHOUT the Sorenson option
    d = pics["D"]
    std = h263mi.H263State(decoder_options=0)
    std.decode_next_picture(fixture_picture_bytes(d))
    mbs, co = _fixture_records(d)
    rc, want = orc.decode_picture(d["width"], d["height"], mbs, co, None)
    assert rc == 0
    got = std.get_last_picture()
    assert (got.width, got.height, got.pquant) == (32, 16, 6)
    assert_planes_equal(got.as_yuv(), want, d["name"])
    assert (got.as_luma().reshape(16, 32)[0:8, 8:16] == 101).all() and (got.as_chroma_b() [:8] == 129).all()
    std.close()
    # the same two pictures through the batch entry (sparse records, events parsed straight into pinned staging)
    bt = h263mi.Batch(3, w, h, pipeline_post=True)
    refs = None
    for pic in (a, b):
        data = fixture_picture_bytes(pic)
        used, rcs = bt.decode_next_pictures_ex([data, data, data], n_threads=2)
        assert rcs == [0, 0, 0] and all(u in (len(data) - 1, len(data)) for u in used)
    bt.sync()
    for s in range(3):
        assert_planes_equal(bt.copy_yuv(s), ref, "batch stream %d" % s)
    bt.close()
    st.close()


def test_more_streams_than_fit_the_kernel_arguments_take_their_words_from_device_memory():
    """a launch of up to 64 pictures carries the streams' words in its kernel arguments; a batch of 80 streams puts them into
    device memory (a small copy in front of the launch): same results"""
    n, w, h = 80, 48, 32
    st = _Streams(n, w, h, seed=611)
    b = h263mi.Batch(n, w, h, pipeline_post=True)
    bufs = [h263mi.DeviceBuffer(n * w * h * 4) for _ in range(2)]
    wants = []
    for f in range(2):
        used, rcs = b.decode_next_pictures_ex(st.pictures(f == 0), n_threads=4, strength=h263mi.STRENGTH_FROM_HEADER, d_rgba=bufs[f].ptr)
        assert not any(rcs)
        wants.append([want_rgba(st.refs[s], w, header_strength(st.q[s], st.flag[s])) for s in range(n)])
    b.sync()
    for f in range(2):
        for s in range(n):
            assert np.array_equal(bufs[f].download(w * h * 4, s * w * h * 4), wants[f][s]), (f, s)
    b.close()


def test_the_bench_path_with_64_strengths_at_1080p_matches_the_oracle_on_sampled_streams():
    """the headline launch (64 x 1080p, events, k_frame) with 64 DIFFERENT strengths through h263mi_batch_decode_events_ps
    and _decode_ps: planes and RGBA of sampled streams after a chain of P pictures against the oracle"""
    import bench
    n, frames = 64, 4
    strengths = [int(h263mi.quant_to_strength()[1 + (5 * s) % 31]) if s % 7 else 0 for s in range(n)]
    wl = bench.Workload(h263mi, n, frames, 0, 0, None, events=True)
    b = h263mi.Batch(n, bench.W, bench.H, pipeline_post=True)
    d_rgba = h263mi.DeviceBuffer(n * bench.RGBA_BYTES)
    for fr in wl.frames:
        if fr.get("first") is not None:
            b.decode_events(fr["ptype"], fr["mbs"].ptr, fr["first"].ptr, fr["ev"].ptr, fr["base"].ptr, fr["blocks"], 0, d_rgba.ptr, None,
                            n_events=fr["n_events"], strengths=strengths)
        else:
            b.decode(fr["ptype"], fr["mbs"].ptr, fr["co"].ptr, fr["base"].ptr, fr["blocks"], 0, d_rgba.ptr, None, strengths=strengths)
    b.sync()
    for s in (0, 7, 33, 63):
        ref = None
        for f in range(frames):
            mbs, co = h263mi.synth_picture_host(h263mi.SYNTH_I_MIXED if f == 0 else h263mi.SYNTH_P, bench.W, bench.H, s, f)
            rc, ref = orc.decode_picture(bench.W, bench.H, mbs, co, ref)
            assert rc == 0
        assert_planes_equal(b.copy_yuv(s), ref, "stream %d" % s)
        assert np.array_equal(d_rgba.download(bench.RGBA_BYTES, s * bench.RGBA_BYTES), want_rgba(ref, bench.W, strengths[s])), s
    b.close()
