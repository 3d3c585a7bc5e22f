"""GPU tests added in round 5.

* The dequantiser where the reference's i16 product overflows (rle.rs:130-133, Sorenson's 11-bit escape LEVELs at
  quantisers from 16 up): a release build of the reference wraps, and so must the reconstruction waves
  (recon_kernel.inl: dequant_pair_wrap, chosen per IDCT round by one ballot on "some LEVEL outside [-512, 511]").
  Hand-derived known answers (tests/golden/dequant_i16_wrap_known_answers.json) through the Dc class's closed form,
  every 11-bit LEVEL at every quantiser as intra and as inter blocks through the dense AND the event transport, the
  i16 extremes a record can carry, all against the oracle -- whose dequantiser tests/test_dequant_i16_wrap.py pins.

Everything goes through the C ABI and is compared bit for bit."""
import json
import os

import numpy as np
import pytest

import h263mi
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if h263mi.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu-marked tests must run on the MI355X box")


def assert_planes_equal(got, want, what=""):
    for g, e, name in zip(got, want, ("Y", "Cb", "Cr")):
        bad = np.flatnonzero(np.asarray(g) != np.asarray(e))
        assert bad.size == 0, "%s %s: %d bytes differ, first at %s" % (what, name, bad.size, bad[:8])


def _flat_state(w, h, code):
    """a state whose last picture is flat: every pixel = code (DC-only intra blocks)"""
    st = h263mi.H263State()
    mbw, mbh = (w + 15) // 16, (h + 15) // 16
    flat = np.zeros(mbw * mbh, orc.MB_RECORD_DTYPE)
    flat["mb_type"] = 3
    flat["quant"] = 1
    flat["intradc"] = code
    st.submit_picture(w, h, flat, np.zeros((0, 64), np.int16), h263mi.PICTURE_I)
    assert all((p == code).all() for p in st.get_last_picture().as_yuv())
    return st


def _submit(st, w, h, mbs, coeffs, ptype, events):
    if events:
        # one flag per CODED block, in pool order: is its macroblock intra?
        flags = []
        for m in mbs:
            flags += [m["mb_type"] in (3, 4)] * bin(int(m["cbp"]) & 0x3f).count("1")
        first, ev = h263mi.events_from_dense(coeffs, np.array(flags, bool))
        st.submit_picture_events(w, h, mbs, first, ev, ptype)
    else:
        st.submit_picture(w, h, mbs, coeffs, ptype)


def test_hand_derived_wrap_values_through_the_dc_class():
    """An inter block whose only coefficient sits at zigzag 0 is Dc(value) (rle.rs:151-160): every pixel of the block
    becomes clamp(prediction + trunc(value * 0.5 / 4 + sign * 0.5)) (idct.rs:114-130).  The known answers of the fixture
    -- worked out by hand from rle.rs:130-133 with wrapping i16 arithmetic -- therefore show directly in the pixels."""
    cases = json.load(open(os.path.join(HERE, "golden", "dequant_i16_wrap_known_answers.json")))["cases"]
    w, h = 16 * len(cases), 16
    for pred in (40, 215):
        st = _flat_state(w, h, pred)
        mbs = np.zeros(len(cases), orc.MB_RECORD_DTYPE)
        mbs["mb_type"] = 0
        mbs["quant"] = [c["quant"] for c in cases]
        mbs["cbp"] = 1                                                  # the first luma block only
        mbs["coeff_index"] = np.arange(len(cases))
        co = np.zeros((len(cases), 64), np.int16)
        co[:, 0] = [c["level"] for c in cases]
        st.submit_picture(w, h, mbs, co, h263mi.PICTURE_P)
        y = st.get_last_picture().as_luma().reshape(h, w)
        for k, c in enumerate(cases):
            v = np.float32(c["value"])
            r = int(np.trunc(np.float32(np.float32(v * np.float32(0.5)) / np.float32(4.0)) + np.float32(np.sign(v) * 0.5)))
            want = min(255, max(0, pred + max(-256, min(255, r))))
            blk = y[0:8, 16 * k:16 * k + 8]
            assert (blk == want).all(), (c, pred, int(blk[0, 0]), want)
            assert (y[8:16, 16 * k:16 * k + 16] == pred).all()         # the uncoded blocks keep the prediction
        ref = tuple(np.full(n, pred, np.uint8) for n in (w * h, w * h // 4, w * h // 4))
        rc, want_planes = orc.decode_picture(w, h, mbs, co, ref)
        assert rc == 0
        assert_planes_equal(st.get_last_picture().as_yuv(), want_planes, "pred %d" % pred)
        st.close()


@pytest.mark.parametrize("events", [False, True], ids=["dense", "events"])
def test_every_11_bit_level_at_every_quantiser_like_a_release_build(events):
    """All 2 047 non-zero LEVELs of Sorenson's 11-bit escape (-1024..1023, parser/block.rs:694-708) at every quantiser --
    16..31 are where the reference's i16 product wraps -- as the AC coefficients of Full-class intra blocks and as inter
    blocks over a flat prediction, narrow and wide LEVELs mixed inside the same blocks, rounds and waves."""
    w, h = 176, 144                                            # 99 macroblocks = 594 blocks
    levels = np.array([v for v in range(-1024, 1024) if v != 0], np.int16)
    n_blocks = 99 * 6
    rng = np.random.default_rng(11)
    seen_wrapped = 0
    for q in range(1, 32):
        # every LEVEL at least once; the rest of the picture: mostly small LEVELs with wide ones sprinkled in, and some
        # blocks without any wide LEVEL at all (their rounds take the fast form beside the wide rounds of other waves)
        fill = rng.integers(-40, 41, n_blocks * 63 - len(levels)).astype(np.int16)
        wide_at = rng.random(fill.size) < 0.03
        fill[wide_at] = rng.choice(levels, int(wide_at.sum()))
        vals = np.concatenate([levels, fill])
        rng.shuffle(vals)
        co = np.zeros((n_blocks, 64), np.int16)
        co[:, 1:] = vals.reshape(n_blocks, 63)
        co[rng.random(n_blocks) < 0.2, 9:] = 0                 # sparser blocks: fewer columns / rows in their rounds
        co[:, 1][co[:, 1] == 0] = 1
        co[:, 8][co[:, 8] == 0] = -1                           # (Full class: something off the first row and column)
        for ptype in (h263mi.PICTURE_I, h263mi.PICTURE_P):
            mbs = np.zeros(99, orc.MB_RECORD_DTYPE)
            mbs["quant"] = q
            mbs["cbp"] = 0x3f
            mbs["coeff_index"] = np.arange(99) * 6
            c = co.copy()
            if ptype == h263mi.PICTURE_I:
                mbs["mb_type"] = 3
                mbs["intradc"] = 100
                ref, st = None, h263mi.H263State()
            else:
                mbs["mb_type"] = 0
                c[:, 0] = vals[:n_blocks]                      # an inter block's first coefficient is a LEVEL like any other
                st = _flat_state(w, h, 120)
                ref = tuple(np.full(n, 120, np.uint8) for n in (w * h, w * h // 4, w * h // 4))
            _submit(st, w, h, mbs, c, ptype, events)
            rc, want = orc.decode_picture(w, h, mbs, c, ref)
            assert rc == 0
            assert_planes_equal(st.get_last_picture().as_yuv(), want, "q %d type %d" % (q, ptype))
            st.close()
        l64 = levels.astype(np.int64)
        seen_wrapped += int((q * (2 * np.abs(l64) + 1) > 32767).sum())
    assert seen_wrapped > 5000                                 # (the sweep did visit the overflowing products)


def test_the_i16_extremes_a_record_can_carry():
    """LEVELs are int16 in the records; the reference's arithmetic is defined (by wrapping) for all of them, including
    i16::abs(-32768).  Dense transport, inter blocks (Dc class and Full class)."""
    w, h = 64, 16
    extremes = [-32768, -32767, -16385, -16384, -2048, -1025, 1024, 2047, 16383, 16384, 32766, 32767, 1057, -1057, 1365]
    for q in (1, 2, 4, 5, 8, 15, 16, 24, 31):
        st = _flat_state(w, h, 128)
        mbs = np.zeros(4, orc.MB_RECORD_DTYPE)
        mbs["quant"] = q
        mbs["cbp"] = 0x3f
        mbs["coeff_index"] = np.arange(4) * 6
        co = np.zeros((24, 64), np.int16)
        for k in range(24):
            if k < len(extremes):
                co[k, 0] = extremes[k]                         # alone: Dc class
            else:
                co[k, [0, 1, 8, 9, 63]] = [extremes[(k + j) % len(extremes)] for j in range(5)]
        st.submit_picture(w, h, mbs, co, h263mi.PICTURE_P)
        ref = tuple(np.full(n, 128, np.uint8) for n in (w * h, w * h // 4, w * h // 4))
        rc, want = orc.decode_picture(w, h, mbs, co, ref)
        assert rc == 0
        assert_planes_equal(st.get_last_picture().as_yuv(), want, "q %d" % q)
        st.close()


# ---------------------------------------------------------------------------------------------
# h263mi_mixed under fault injection: a failure at EVERY HIP call of a call that renders into per-stream
# buffers.  Whatever fails, (i) a stream reports H263MI_OK exactly when it decoded its picture and advanced, (ii) its
# pictures stay the oracle's, (iii) a later call renders again -- the ring of per-stream output pointers is made on first
# use, and a failure half-way through making it used to leave the class unable to render for good.
# ---------------------------------------------------------------------------------------------
def _rgba_want(planes, strength, w):
    cw = (w + 1) // 2
    filt = planes if strength == 0 else tuple(orc.deblock(p, pw, strength) for p, pw in zip(planes, (w, cw, cw)))
    return orc.yuv420_to_rgba(*filt, w)


@pytest.mark.parametrize("pipeline", [False, True], ids=["plain", "pipelined"])
def test_mixed_set_survives_a_failure_at_any_hip_call(pipeline):
    import recgen
    import sorenson_enc as enc
    from test_bitstream_e2e import make_codable
    q, strength = 6, 5
    sizes = [(176, 144), (352, 288), (176, 144)]
    n = len(sizes)
    m = h263mi.MixedBatch(n, pipeline_post=pipeline)
    refs, gen = [None] * n, [0] * n
    rgba = [h263mi.DeviceBuffer(352 * 288 * 4) for _ in range(n)]

    def picture(s):
        w, h = sizes[s]
        f = gen[s]
        if f == 0:
            mbs, co = recgen.intra_picture(w, h, seed=500 + s, max_level=60)
            mbs = make_codable(mbs, q, s, 0)
        else:
            mbs, co = recgen.inter_picture(w, h, seed=1000 * f + s, mv_range=32, p_4v=0.2, p_coded=0.4, quant=q, max_level=60)
            mbs = make_codable(mbs, q, s + f, 1)
        return enc.encode_picture(w, h, 0 if f == 0 else 1, q, mbs, co, temporal_reference=f % 256), mbs, co

    def advance(s, mbs, co):
        w, h = sizes[s]
        rc, refs[s] = orc.decode_picture(w, h, mbs, co, None if gen[s] == 0 else refs[s])
        assert rc == 0
        gen[s] += 1

    def check_pictures(what):
        assert not any(m.sync())
        for s in range(n):
            if refs[s] is not None:
                assert_planes_equal(m.copy_yuv(s), refs[s], "%s, stream %d" % (what, s))

    failures = through = 0
    # three sweeps, each from the first HIP call on until a call goes through untouched: the call that creates the classes
    # and the pointer ring, the first P pictures on what exists, and once more
    for sweep in range(3):
        for nth in range(1, 400):
            pics = [picture(s) for s in range(n)]
            h263mi.debug_fail_nth_hip_call(nth)
            used, rcs, descs, call_rc = m.decode_next_pictures([p[0] for p in pics], n_threads=2, strength=strength, rgba=rgba,
                                                               raise_on_error=False)
            fired = h263mi.debug_fail_nth_hip_call(0) <= 0
            for s in range(n):
                assert (rcs[s] == 0) == (used[s] > 0), (nth, s, rcs[s], used[s])
                if rcs[s] == 0:
                    advance(s, pics[s][1], pics[s][2])
            check_pictures("sweep %d, after a failure at HIP call %d" % (sweep, nth) if fired else "after a clean call")
            if fired:
                failures += 1
                assert call_rc != 0 or any(rcs), "HIP call %d failed and nobody was told" % nth
                continue
            # the call went through: everything decoded, and it RENDERED (also right after a failed ring set-up)
            assert call_rc == 0 and not any(rcs)
            for s in range(n):
                w, h = sizes[s]
                assert (rgba[s].download(w * h * 4) == _rgba_want(refs[s], strength, w)).all(), (sweep, nth, s)
            through += 1
            break
        else:
            pytest.fail("sweep %d never went through" % sweep)
    assert failures >= 8 and through == 3, (failures, through)
    m.close()


# ---------------------------------------------------------------------------------------------
# h263mi_mixed: slots by membership.  A class owns as many slots as it has members (a power of two):
# 63 QCIF streams and one 1080p stream hold 64 x 2 QCIF frames and 1 x 2 1080p frames -- round 4 held 64 x 2 of both --;
# a class grows by doubling when streams join it (the members' frames move into the new store: the P pictures that follow
# must predict from them) and gives room back when three quarters of it stand empty.
# ---------------------------------------------------------------------------------------------
def _frame_bytes(w, h):
    mbw, mbh = (w + 15) // 16, (h + 15) // 16
    pitch_c = (mbw * 8 + 63) // 64 * 64
    raw = 2 * pitch_c * mbh * 16 + 2 * pitch_c * mbh * 8 + 256
    return (raw + 255) // 256 * 256


class _MixedChains:
    """n streams through a MixedBatch, every one against its own oracle chain"""

    def __init__(self, n, pipeline):
        import recgen
        import sorenson_enc as enc
        from test_bitstream_e2e import make_codable
        self.recgen, self.enc, self.make_codable = recgen, enc, make_codable
        self.n, self.q = n, 6
        self.m = h263mi.MixedBatch(n, pipeline_post=pipeline)
        self.refs, self.size, self.count = [None] * n, [None] * n, 0

    def picture(self, s, intra, size):
        w, h = size
        self.count += 1
        if intra:
            mbs, co = self.recgen.intra_picture(w, h, seed=31 * self.count + s, max_level=60)
            mbs = self.make_codable(mbs, self.q, s, 0)
        else:
            mbs, co = self.recgen.inter_picture(w, h, seed=31 * self.count + s, mv_range=24, p_4v=0.2, p_coded=0.4, quant=self.q, max_level=60)
            mbs = self.make_codable(mbs, self.q, s + self.count, 1)
        rc, self.refs[s] = orc.decode_picture(w, h, mbs, co, None if intra else self.refs[s])
        assert rc == 0
        self.size[s] = size
        return self.enc.encode_picture(w, h, 0 if intra else 1, self.q, mbs, co, temporal_reference=self.count % 256)

    def call(self, plan):
        """plan: {stream: ('I' | 'P', size)}; streams not named sit the call out"""
        datas = [None] * self.n
        for s, (kind, size) in plan.items():
            datas[s] = self.picture(s, kind == "I", size)
        used, rcs, descs = self.m.decode_next_pictures(datas, n_threads=3)
        assert not any(rcs), rcs
        assert not any(self.m.sync())

    def check(self, what):
        for s in range(self.n):
            if self.refs[s] is not None:
                assert self.m.stream_size(s) == self.size[s]
                assert_planes_equal(self.m.copy_yuv(s), self.refs[s], "%s, stream %d" % (what, s))


def test_mixed_frame_store_follows_the_membership():
    n = 64
    ch = _MixedChains(n, pipeline=True)
    qcif, hd = (176, 144), (1920, 1080)
    ch.call({s: ("I", hd if s == 17 else qcif) for s in range(n)})
    want = 2 * (64 * _frame_bytes(*qcif) + 1 * _frame_bytes(*hd))        # 63 members in 64 slots, 1 member in 1 slot
    assert ch.m.frame_store_bytes() == want, (ch.m.frame_store_bytes(), want)
    assert want < 2 * n * _frame_bytes(*hd) // 20                        # (round 4: 64 x 2 frames of BOTH sizes, 410 MB)
    ch.call({s: ("P", ch.size[s]) for s in range(n)})
    ch.check("one P picture each")
    ch.m.close()


@pytest.mark.parametrize("pipeline", [False, True], ids=["plain", "pipelined"])
def test_mixed_class_grows_by_doubling_and_shrinks_and_keeps_its_pictures(pipeline):
    n = 20
    ch = _MixedChains(n, pipeline)
    cif, qcif = (352, 288), (176, 144)
    fb = _frame_bytes(*cif)
    ch.call({0: ("I", cif)})
    assert ch.m.frame_store_bytes() == 2 * 1 * fb
    # streams join one or two at a time: 1 -> 2 -> 4 -> 8 -> 16 slots, every member predicting across every move
    joined = 1
    for step, add in enumerate((1, 1, 2, 3, 1, 4, 3)):                   # 2, 3, 5, 8, 9, 13, 16 members
        plan = {s: ("P", cif) for s in range(joined)}
        for s in range(joined, joined + add):
            plan[s] = ("I", cif)
        joined += add
        ch.call(plan)
        slots = 1
        while slots < joined:
            slots *= 2
        assert ch.m.frame_store_bytes() == 2 * slots * fb, (step, joined, ch.m.frame_store_bytes() // (2 * fb))
        ch.check("after %d streams have joined" % joined)
    assert joined == 16 and ch.m.size_classes() == 1
    # twelve of the sixteen move to QCIF (an I picture each): the CIF class stands three quarters empty and gives the room back
    plan = {s: ("I", qcif) for s in range(4, 16)}
    plan.update({s: ("P", cif) for s in range(4)})
    ch.call(plan)
    ch.check("twelve streams have left")
    ch.call({s: ("P", ch.size[s]) for s in range(16)})                   # (the class is rebuilt in front of its next launch)
    ch.check("after the shrink")
    assert ch.m.size_classes() == 2
    assert ch.m.frame_store_bytes() == 2 * (4 * fb + 16 * _frame_bytes(*qcif)), ch.m.frame_store_bytes()
    # a stream that is reset gives its slot up; the next joiner takes it
    ch.m.reset_stream(2)
    ch.refs[2] = None
    ch.call({18: ("I", cif), 0: ("P", cif)})
    assert ch.m.frame_store_bytes() == 2 * (4 * fb + 16 * _frame_bytes(*qcif))
    ch.call({s: ("P", ch.size[s]) for s in (0, 1, 3, 18)})
    ch.check("a freed slot taken again")
    ch.m.close()


# ---------------------------------------------------------------------------------------------
# sparse RECORDS over the link (round 5): the batch entry that parses bitstreams sends records for the coded macroblocks only
# plus one index word per group of 8 (ReconArgs::mb_group_index).  Every end-to-end test of the suite runs that way now; this
# one holds the two forms against each other on content with long runs of uncoded macroblocks, whole waves of them, groups
# with a single coded macroblock, a short picture -- and keeps the dense form (H263MI_SPARSE_RECORDS=0) covered.
# ---------------------------------------------------------------------------------------------
_SPARSE_PROBE = r'''
import sys, numpy as np
sys.path[:0] = [%r, %r, %r]
import h263mi, recgen, sorenson_enc as enc
from oracle import oracle as orc
from test_bitstream_e2e import make_codable
w, h, n, q = 352, 288, 6, 8
b = h263mi.Batch(n, w, h, pipeline_post=True)
d_rgba = h263mi.DeviceBuffer(n * w * h * 4)
refs = [None] * n
for f in range(5):
    datas = []
    for s in range(n):
        if f == 0:
            mbs, co = recgen.realistic_intra_picture(w, h, 50 + s, quant=q)
        else:
            mbs, co = recgen.realistic_inter_picture(w, h, 100 * f + s, p_skip=(0.3, 0.7, 0.9, 0.97, 1.0, 0.6)[s], p_coded=0.2, quant=q)
        mbs = make_codable(mbs, q, s + f, 0 if f == 0 else 1)
        if f == 3 and s == 1:
            mbs = mbs[:len(mbs) - 37]                      # a picture that ends early: the rest is padded as not coded
        datas.append(enc.encode_picture(w, h, 0 if f == 0 else 1, q, mbs, co, temporal_reference=f))
        rc, refs[s] = orc.decode_picture(w, h, mbs, co, None if f == 0 else refs[s])
        assert rc == 0
    used, rcs = b.decode_next_pictures_ex(datas, n_threads=3, strength=5, d_rgba=d_rgba.ptr)
    assert not any(rcs), rcs
# a call in which one stream has no picture and another one's data ends inside the picture: both keep what they had, the others
# go on (the staging slot holds whatever the failed parse left in its part)
datas = []
for s in range(n):
    mbs, co = recgen.realistic_inter_picture(w, h, 900 + s, p_skip=0.5, p_coded=0.3, quant=q)
    mbs = make_codable(mbs, q, s + 9, 1)
    data = enc.encode_picture(w, h, 1, q, mbs, co, temporal_reference=5)
    if s == 2:
        data = data[:len(data) // 2]
    elif s == 4:
        data = None
    else:
        rc, refs[s] = orc.decode_picture(w, h, mbs, co, refs[s])
        assert rc == 0
    datas.append(data)
used, rcs = b.decode_next_pictures_ex(datas, n_threads=2, strength=5, d_rgba=d_rgba.ptr)
assert rcs[2] != 0 and not any(rc for s, rc in enumerate(rcs) if s != 2), rcs
b.sync()
for s in range(n):
    got = b.copy_yuv(s)
    assert all((g == e).all() for g, e in zip(got, refs[s])), s
    cw = (w + 1) // 2
    filt = tuple(orc.deblock(p, pw, 5) for p, pw in zip(refs[s], (w, cw, cw)))
    assert (d_rgba.download(w * h * 4, s * w * h * 4) == orc.yuv420_to_rgba(*filt, w)).all(), s      # (2 and 4: their last picture's)
b.close()
print("sparse-probe-ok")
'''


# (H263MI_DIRECT_WORDS, round 5: the parser writes events, block offsets and group index straight into the staging slot, at
# pitches that hold the worst case of the call's pictures -- the default -- or into its own vectors, packed afterwards)
@pytest.mark.parametrize("sparse,direct", [("1", "1"), ("1", "0"), ("0", "1")], ids=["sparse_records_direct_words", "sparse_records_packed_words", "dense_records"])
def test_sparse_and_dense_record_transport_decode_alike(sparse, direct):
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    code = _SPARSE_PROBE % (root, os.path.join(root, "h263-rs_amd"), HERE)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, H263MI_SPARSE_RECORDS=sparse, H263MI_DIRECT_WORDS=direct),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "sparse-probe-ok" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


# ---------------------------------------------------------------------------------------------
# parser threads that PARK between calls (HostThreadPlan: what a call gets under a CPU-time quota when it runs more threads than
# the quota has CPUs).  Forced here with H263MI_SPIN_US=0, whatever the host's quota is: 40 threads for 48 streams, woken and
# parked once per call over a GOP, every stream against the oracle.
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_parked_parser_threads_decode_like_spinning_ones():
    import recgen
    import sorenson_enc as enc
    from test_bitstream_e2e import make_codable
    w, h, n, q = 176, 144, 48, 9
    os.environ["H263MI_SPIN_US"] = "0"
    try:
        b = h263mi.Batch(n, w, h, pipeline_post=True)
        d_rgba = h263mi.DeviceBuffer(n * w * h * 4)
        refs = [None] * n
        for f in range(7):
            datas = []
            for s in range(n):
                if f == 0:
                    mbs, co = recgen.realistic_intra_picture(w, h, 500 + s % 5, quant=q)
                else:
                    mbs, co = recgen.realistic_inter_picture(w, h, 1000 * f + s % 7, p_skip=0.5, p_coded=0.25, quant=q)
                mbs = make_codable(mbs, q, s % 5 + f, 0 if f == 0 else 1)
                datas.append(enc.encode_picture(w, h, 0 if f == 0 else 1, q, mbs, co, temporal_reference=f))
                rc, refs[s] = orc.decode_picture(w, h, mbs, co, None if f == 0 else refs[s])
                assert rc == 0
            used, rcs = b.decode_next_pictures_ex(datas, n_threads=40, strength=4, d_rgba=d_rgba.ptr)
            assert not any(rcs), rcs
            assert all(len(d) - 1 <= u <= len(d) for u, d in zip(used, datas))     # (whole bytes drained, reader.rs commit())
        b.sync()
        cw = (w + 1) // 2
        for s in range(n):
            assert all((g == e).all() for g, e in zip(b.copy_yuv(s), refs[s])), s
            filt = tuple(orc.deblock(p, pw, 4) for p, pw in zip(refs[s], (w, cw, cw)))
            assert (d_rgba.download(w * h * 4, s * w * h * 4) == orc.yuv420_to_rgba(*filt, w)).all(), s
        b.close()
    finally:
        del os.environ["H263MI_SPIN_US"]
