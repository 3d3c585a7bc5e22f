"""GPU parity tests proper: the HIP path, called through the C ABI, against the oracle on the
same seeded inputs (bit-exact for the YUV planes AND for RGBA -- the +-1 LSB the north star
allows for RGBA is not needed), against the reference's own golden fixtures, and at the
BASELINE 1080p size through size-independent properties plus full-size oracle comparison of
sampled streams."""
import hashlib
import json
import os

import numpy as np
import pytest

import h263mi
import recgen
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if h263mi.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu-marked tests must run on the MI355X box")


def assert_planes_equal(got, want, what=""):
    for g, e, name in zip(got, want, ("Y", "Cb", "Cr")):
        bad = np.flatnonzero(np.asarray(g) != np.asarray(e))
        assert bad.size == 0, "%s %s: %d bytes differ, first at %s" % (what, name, bad.size, bad[:8])


# ---------------------------------------------------------------------------------------------
# deblock crate
# ---------------------------------------------------------------------------------------------
def test_deblock_reference_golden_image():                       # deblock.rs:442-558
    img = json.load(open(os.path.join(GOLD, "deblock_reference_tests.json")))["image"]
    for s in ("4", "8", "12"):
        out = h263mi.deblock(np.array(img["data"], np.uint8), img["width"], int(s))
        assert out.tolist() == img["expected"][s], s


def test_deblock_process_table_through_a_4x16_image():            # deblock.rs:352-439
    # one vertical-edge quartet per row: columns 6..9 of a 16-wide, 1-row-per-case image would
    # have no SIMD rows, so the scalar semantics of `process` apply (rows >= 8*floor(h/8)).
    rows = json.load(open(os.path.join(GOLD, "deblock_reference_tests.json")))["process_rows"]
    for r in rows:
        img = np.zeros((1, 16), np.uint8)
        img[0, 6:10] = r["in"]
        img[0, :6] = r["in"][0]
        img[0, 10:] = r["in"][3]
        out = h263mi.deblock(img, 16, r["strength"]).reshape(1, 16)
        assert out[0, 6:10].tolist() == r["out"], r


@pytest.mark.parametrize("w,h", [(11, 17), (16, 16), (9, 9), (10, 10), (8, 2), (1, 1), (100, 60), (200, 37),
                                 (960, 540), (1920, 1080)])
def test_deblock_matches_oracle(w, h):
    rng = np.random.default_rng(w * 7 + h)
    data = rng.integers(0, 256, w * h, dtype=np.uint8)
    for s in (1, 5, 12):
        assert (h263mi.deblock(data, w, s) == orc.deblock(data, w, s)).all(), (w, h, s)


def test_deblock_floor_vs_trunc_regions():
    # appendix B.9: inputs where A-4B+4C-D < 0 and not a multiple of 8, at 960x540 (rows 536..539
    # use the scalar semantics on vertical edges) and at a width with w % 8 != 0
    for (w, h) in ((960, 540), (964, 24)):
        rng = np.random.default_rng(w)
        data = np.clip(rng.normal(128, 6, w * h), 0, 255).astype(np.uint8)
        assert (h263mi.deblock(data, w, 9) == orc.deblock(data, w, 9)).all()


# ---------------------------------------------------------------------------------------------
# yuv crate
# ---------------------------------------------------------------------------------------------
def test_bt601_reference_golden():                                # bt601.rs:199-225, 329-483
    bt = json.load(open(os.path.join(GOLD, "bt601_reference_tests.json")))
    for c in bt["single_pixel"]:
        y, cb, cr = c["yuv"]
        out = h263mi.yuv420_to_rgba([y] * 4, [cb] * 2, [cr] * 2, 4).reshape(4, 4)
        assert (out == np.array(c["rgb"] + [255], np.uint8)).all(), c
    for p in bt["pictures"]:
        out = h263mi.yuv420_to_rgba(np.array(p["y"], np.uint8), np.array(p["cb"], np.uint8),
                                    np.array(p["cr"], np.uint8), p["y_width"])
        assert out.tolist() == p["rgba"], p["y_width"]


@pytest.mark.parametrize("w,h", [(1, 1), (2, 2), (3, 3), (5, 4), (100, 60), (133, 35), (176, 144), (1920, 1080)])
def test_bt601_matches_oracle(w, h):
    planes = recgen.random_planes(w, h, w * 3 + h)
    assert (h263mi.yuv420_to_rgba(*planes, w) == orc.yuv420_to_rgba(*planes, w)).all()


def test_bt601_every_yuv_triple_sampled():
    # all 256 Y x a 64x64 lattice of (Cb, Cr), laid out as one picture with 2x2 constant quads
    cbs, crs = np.meshgrid(np.arange(0, 256, 4), np.arange(0, 256, 4))
    for y in range(0, 256, 5):
        cb = cbs.astype(np.uint8).ravel()
        cr = crs.astype(np.uint8).ravel()
        yy = np.full(128 * 128, y, np.uint8)
        assert (h263mi.yuv420_to_rgba(yy, cb, cr, 128) == orc.yuv420_to_rgba(yy, cb, cr, 128)).all()


# ---------------------------------------------------------------------------------------------
# H263State: record-level decode_next_picture
# ---------------------------------------------------------------------------------------------
SIZES = [(16, 16), (48, 32), (100, 60), (5, 4), (1, 1), (33, 17), (176, 144), (320, 240), (136, 40)]


@pytest.mark.parametrize("w,h", SIZES)
def test_state_intra_then_inter_chain(w, h):
    st = h263mi.H263State()
    assert st.get_last_picture() is None and st.is_sorenson()
    mbs, coeffs = recgen.intra_picture(w, h, seed=w * 31 + h)
    st.submit_picture(w, h, mbs, coeffs, h263mi.PICTURE_I, temporal_reference=0)
    rc, want = orc.decode_picture(w, h, mbs, coeffs, None)
    assert rc == 0
    pic = st.get_last_picture()
    assert (pic.width, pic.height, pic.chroma_width, pic.chroma_height) == (w, h, (w + 1) // 2, (h + 1) // 2)
    assert_planes_equal(pic.as_yuv(), want, "I")
    for f in range(1, 4):
        mbs, coeffs = recgen.inter_picture(w, h, seed=f * 1000 + w + h, mv_range=70, p_4v=0.3, p_intra=0.15, quant=0)
        st.submit_picture(w, h, mbs, coeffs, h263mi.PICTURE_P, temporal_reference=f)
        rc, want = orc.decode_picture(w, h, mbs, coeffs, want)
        assert rc == 0
        pic = st.get_last_picture()
        assert pic.temporal_reference == f
        assert_planes_equal(pic.as_yuv(), want, "P%d" % f)
    # the consumer's post-processing: deblock each plane, then convert
    cw = (w + 1) // 2
    for strength in (0, 5):
        planes = want if strength == 0 else tuple(orc.deblock(p, pw, strength) for p, pw in zip(want, (w, cw, cw)))
        assert (st.render_rgba(strength) == orc.yuv420_to_rgba(*planes, w)).all(), strength
    st.close()


@pytest.mark.parametrize("w,h,mv_range", [(176, 144, 600), (100, 60, 200), (16, 16, 1100), (320, 240, 1100), (1920, 1080, 300)])
def test_state_far_vectors_and_large_levels(w, h, mv_range):
    """vectors far outside the picture (every tap clamps to an edge pixel) and 11-bit levels at quantiser 31"""
    st = h263mi.H263State()
    mbs, coeffs = recgen.intra_picture(w, h, seed=3)
    st.submit_picture(w, h, mbs, coeffs, h263mi.PICTURE_I)
    rc, want = orc.decode_picture(w, h, mbs, coeffs, None)
    mbs, coeffs = recgen.inter_picture(w, h, seed=mv_range + w, mv_range=mv_range, p_4v=0.4, p_intra=0.1, p_coded=0.5,
                                       quant=31, max_level=1023, sparse_low=False)
    st.submit_picture(w, h, mbs, coeffs, h263mi.PICTURE_P)
    rc, want = orc.decode_picture(w, h, mbs, coeffs, want)
    assert rc == 0
    assert_planes_equal(st.get_last_picture().as_yuv(), want, "P")
    st.close()


def test_state_errors_leave_state_unchanged():
    w, h = 64, 48
    st = h263mi.H263State()
    mbs, coeffs = recgen.inter_picture(w, h, seed=1)
    with pytest.raises(h263mi.H263Error) as e:                  # gather.rs:149
        st.submit_picture(w, h, mbs, coeffs, h263mi.PICTURE_P)
    assert e.value.code == h263mi.ERR_UNCODED_IFRAME_BLOCKS
    assert st.get_last_picture() is None and not st.has_reference_picture()
    imbs, icoeffs = recgen.intra_picture(w, h, seed=2)
    with pytest.raises(h263mi.H263Error):                       # a short I picture pads with Inter macroblocks
        st.submit_picture(w, h, imbs[:5], icoeffs, h263mi.PICTURE_I)
    st.submit_picture(w, h, imbs, icoeffs, h263mi.PICTURE_I)
    before = st.get_last_picture().as_yuv()
    bad = mbs.copy()
    bad[3]["coeff_index"] = 10 ** 6
    bad[3]["cbp"] = 1
    with pytest.raises(h263mi.H263Error) as e:
        st.submit_picture(w, h, bad, coeffs, h263mi.PICTURE_P)
    assert e.value.code == h263mi.ERR_INVALID_ARGUMENT
    assert_planes_equal(st.get_last_picture().as_yuv(), before, "after failed decode")
    with pytest.raises(h263mi.H263Error) as e:
        st.decode_next_picture(b"\x00\x00\x80")              # a bare start code: the header read hits EOF
    assert e.value.code == -16
    assert_planes_equal(st.get_last_picture().as_yuv(), before, "after failed bitstream decode")
    st.close()


def test_state_short_picture_padding_kill_and_reset():
    w, h = 64, 48
    st = h263mi.H263State()
    imbs, icoeffs = recgen.intra_picture(w, h, seed=4, classes=("full_sparse", "dc", "vert", "horiz"))
    imbs[0]["kill"] = 0b100101
    imbs[1]["intradc"][:] = 255
    st.submit_picture(w, h, imbs, icoeffs, h263mi.PICTURE_I)
    rc, ref = orc.decode_picture(w, h, imbs, icoeffs, None)
    assert_planes_equal(st.get_last_picture().as_yuv(), ref, "I with kill")
    mbs, coeffs = recgen.inter_picture(w, h, seed=5, n_mbs=7)   # state.rs:421-427
    st.submit_picture(w, h, mbs, coeffs, h263mi.PICTURE_P)
    rc, want = orc.decode_picture(w, h, mbs, coeffs, ref)
    assert_planes_equal(st.get_last_picture().as_yuv(), want, "short P")
    st.reset()
    assert st.get_last_picture() is None
    with pytest.raises(h263mi.H263Error):
        st.submit_picture(w, h, mbs, coeffs, h263mi.PICTURE_P)
    # a new size on an I picture re-creates the frame store; on a P picture it is an error
    imbs2, icoeffs2 = recgen.intra_picture(32, 32, seed=6)
    st.submit_picture(32, 32, imbs2, icoeffs2, h263mi.PICTURE_I)
    rc, ref2 = orc.decode_picture(32, 32, imbs2, icoeffs2, None)
    assert_planes_equal(st.get_last_picture().as_yuv(), ref2, "resized I")
    with pytest.raises(h263mi.H263Error) as e:
        st.submit_picture(w, h, mbs, coeffs, h263mi.PICTURE_P)
    assert e.value.code == h263mi.ERR_PICTURE_FORMAT_INVALID
    st.close()


def test_state_dc_sweeps():
    # appendix B.1/B.6: inter blocks whose only coefficient sits at zigzag 0 take the Dc class with an
    # arbitrary dequantised value; sweep levels x quantisers over near-black and near-white
    # predictions so that both ends of the final clamp(0, 255) are hit
    w, h = 128, 128
    for code in (1, 254):
        st = h263mi.H263State()
        flat = np.zeros(64, orc.MB_RECORD_DTYPE)
        flat["mb_type"] = 3
        flat["quant"] = 1
        flat["intradc"] = code
        st.submit_picture(w, h, flat, np.zeros((0, 64), np.int16), h263mi.PICTURE_I)
        rc, ref = orc.decode_picture(w, h, flat, np.zeros((0, 64), np.int16), None)
        assert (ref[0] == code).all()
        assert_planes_equal(st.get_last_picture().as_yuv(), ref, "flat I")
        rng = np.random.default_rng(code)
        mbs = np.zeros(64, orc.MB_RECORD_DTYPE)
        mbs["quant"] = rng.integers(1, 32, 64)
        mbs["cbp"] = 0x3F
        mbs["coeff_index"] = np.arange(64) * 6
        coeffs = np.zeros((64 * 6, 64), np.int16)
        coeffs[:, 0] = rng.integers(-127, 128, 64 * 6)
        st.submit_picture(w, h, mbs, coeffs, h263mi.PICTURE_P)
        rc, want = orc.decode_picture(w, h, mbs, coeffs, ref)
        assert_planes_equal(st.get_last_picture().as_yuv(), want, "dc sweep")
        st.close()


def test_state_intradc_closed_form():
    # QCIF I picture of DC-only intra blocks => every block is flat = code (255 -> 128)
    w, h = 176, 144
    rng = np.random.default_rng(0)
    mbs = np.zeros(99, orc.MB_RECORD_DTYPE)
    mbs["mb_type"] = 3
    mbs["quant"] = 8
    codes = recgen.random_intradc(rng, 99 * 6).reshape(99, 6)
    codes[0, 0] = 255
    mbs["intradc"] = codes
    st = h263mi.H263State()
    st.submit_picture(w, h, mbs, np.zeros((0, 64), np.int16), h263mi.PICTURE_I)
    y, cb, cr = st.get_last_picture().as_yuv()
    Y = y.reshape(h, w)
    for i in range(99):
        px, py = (i % 11) * 16, (i // 11) * 16
        for b, (ox, oy) in enumerate(((0, 0), (8, 0), (0, 8), (8, 8))):
            want = 128 if codes[i, b] == 255 else codes[i, b]
            assert (Y[py + oy:py + oy + 8, px + ox:px + ox + 8] == want).all()
    st.close()


# ---------------------------------------------------------------------------------------------
# synthetic record generator: device == host, bit for bit
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", [h263mi.SYNTH_I_DENSE, h263mi.SYNTH_I_MIXED, h263mi.SYNTH_P])
def test_synth_device_matches_host(kind):
    w, h, n = 176, 144, 3
    mbs_pp = 99
    d_mbs = h263mi.DeviceBuffer(n * mbs_pp * 32)
    d_co = h263mi.DeviceBuffer(n * mbs_pp * 6 * 128)
    d_base = h263mi.DeviceBuffer(n * 8)
    total = h263mi.synth_batch_device(kind, w, h, n, 10, 4, d_mbs.ptr, d_co.ptr, n * mbs_pp * 6, d_base.ptr)
    mbs = d_mbs.download(dtype=h263mi.MB_RECORD_DTYPE)
    co = d_co.download(total * 128, dtype=np.int16).reshape(-1, 64)
    base = d_base.download(dtype=np.uint64)
    for p in range(n):
        hm, hc = h263mi.synth_picture_host(kind, w, h, 10 + p, 4)
        assert mbs[p * mbs_pp:(p + 1) * mbs_pp].tobytes() == hm.tobytes()
        assert co[int(base[p]):int(base[p]) + hc.shape[0]].tobytes() == hc.tobytes()


# ---------------------------------------------------------------------------------------------
# BASELINE size (1920x1080): batch path, full-size oracle comparison of sampled streams,
# and size-independent properties
# ---------------------------------------------------------------------------------------------
W, H = 1920, 1080
MBS_PP = 120 * 68


class SynthStream:
    """n streams x frames of device-resident synthetic records (frame 0 = I, then P)."""

    def __init__(self, n, frames, i_kind):
        self.n = n
        self.frames = []
        for f in range(frames):
            kind = i_kind if f == 0 else h263mi.SYNTH_P
            cap = n * MBS_PP * (6 if kind != h263mi.SYNTH_P else 3)
            d_mbs = h263mi.DeviceBuffer(n * MBS_PP * 32)
            d_co = h263mi.DeviceBuffer(cap * 128)
            d_base = h263mi.DeviceBuffer(n * 8)
            h263mi.synth_batch_device(kind, W, H, n, 0, f, d_mbs.ptr, d_co.ptr, cap, d_base.ptr)
            self.frames.append((kind, d_mbs, d_co, d_base))


@pytest.fixture(scope="module")
def streams4():
    return SynthStream(4, 3, h263mi.SYNTH_I_MIXED)


def run_batch(s, strength, want_rgba=True):
    b = h263mi.Batch(s.n, W, H)
    d_rgba = h263mi.DeviceBuffer(s.n * W * H * 4)
    d_planes = h263mi.DeviceBuffer(s.n * (W * H + 2 * 960 * 540))
    outs = []
    for f, (kind, d_mbs, d_co, d_base) in enumerate(s.frames):
        b.submit(h263mi.PICTURE_I if f == 0 else h263mi.PICTURE_P, d_mbs.ptr, d_co.ptr, d_base.ptr)
        b.render_rgba(strength, d_rgba.ptr, d_planes.ptr)
        b.sync()
        outs.append(([b.copy_yuv(i) for i in range(s.n)], d_rgba.download(), d_planes.download()))
    b.close()
    return outs


def test_1080p_batch_matches_oracle_on_sampled_streams(streams4):
    outs = run_batch(streams4, 5)
    for stream in (0, 3):
        ref = None
        for f, (yuvs, rgba, planes) in enumerate(outs):
            kind = h263mi.SYNTH_I_MIXED if f == 0 else h263mi.SYNTH_P
            mbs, coeffs = h263mi.synth_picture_host(kind, W, H, stream, f)
            rc, want = orc.decode_picture(W, H, mbs, coeffs, ref)
            assert rc == 0
            assert_planes_equal(yuvs[stream], want, "1080p stream %d frame %d" % (stream, f))
            filt = tuple(orc.deblock(p, pw, 5) for p, pw in zip(want, (W, 960, 960)))
            n_y, n_c = W * H, 960 * 540
            off = stream * (n_y + 2 * n_c)
            got_planes = (planes[off:off + n_y], planes[off + n_y:off + n_y + n_c], planes[off + n_y + n_c:off + n_y + 2 * n_c])
            assert_planes_equal(got_planes, filt, "deblocked planes")
            assert (rgba[stream * n_y * 4:(stream + 1) * n_y * 4] == orc.yuv420_to_rgba(*filt, W)).all()
            ref = want


def test_1080p_determinism_and_stream_independence(streams4):
    h1 = [hashlib.sha256(o[1].tobytes() + o[2].tobytes()).hexdigest() for o in run_batch(streams4, 5)]
    h2 = [hashlib.sha256(o[1].tobytes() + o[2].tobytes()).hexdigest() for o in run_batch(streams4, 5)]
    assert h1 == h2                                              # appendix B.10
    # a batch of one stream gives the same frames as that stream inside a batch of four
    solo = SynthStream(1, 2, h263mi.SYNTH_I_MIXED)
    o4, o1 = run_batch(streams4, 0), run_batch(solo, 0)
    for f in range(2):
        assert_planes_equal(o1[f][0][0], o4[f][0][0], "solo vs batch frame %d" % f)


def test_1080p_dense_iframe_matches_oracle():
    s = SynthStream(2, 1, h263mi.SYNTH_I_DENSE)                  # BASELINE config 2, every block Full
    (yuvs, rgba, _), = run_batch(s, 0)
    mbs, coeffs = h263mi.synth_picture_host(h263mi.SYNTH_I_DENSE, W, H, 1, 0)
    rc, want = orc.decode_picture(W, H, mbs, coeffs, None)
    assert_planes_equal(yuvs[1], want, "dense I")
    assert (rgba[W * H * 4:2 * W * H * 4] == orc.yuv420_to_rgba(*want, W)).all()


def test_1080p_zero_motion_uncoded_picture_is_an_exact_copy(streams4):
    b = h263mi.Batch(2, W, H)
    kind, d_mbs, d_co, d_base = SynthStream(2, 1, h263mi.SYNTH_I_MIXED).frames[0]
    b.submit(h263mi.PICTURE_I, d_mbs.ptr, d_co.ptr, d_base.ptr)
    b.sync()
    first = [b.copy_yuv(i) for i in range(2)]
    empty = np.zeros(2 * MBS_PP, h263mi.MB_RECORD_DTYPE)         # Inter, mv 0, cbp 0
    empty["quant"] = 1
    d_empty = h263mi.DeviceBuffer(empty.nbytes)
    d_empty.upload(empty)
    for _ in range(3):
        b.submit(h263mi.PICTURE_P, d_empty.ptr, d_co.ptr, None)
    b.sync()
    for i in range(2):
        assert_planes_equal(b.copy_yuv(i), first[i], "copy chain")
    b.close()


def test_batch_reports_inter_without_reference():
    b = h263mi.Batch(1, 64, 48)
    rec = np.zeros(12, h263mi.MB_RECORD_DTYPE)
    rec["quant"] = 1
    d = h263mi.DeviceBuffer(rec.nbytes)
    d.upload(rec)
    b.submit(h263mi.PICTURE_P, d.ptr, d.ptr, None)
    with pytest.raises(h263mi.H263Error) as e:
        b.sync()
    assert e.value.code == h263mi.ERR_UNCODED_IFRAME_BLOCKS
    b.close()


def test_batch_submit_host_matches_the_oracle_per_stream():
    """h263mi_batch_submit_host: per-stream host record arrays (one short, one without coded blocks), packed and
    decoded in one launch; the second picture reuses the other staging slot, the third the first again."""
    w, h, n = 176, 144, 4
    b = h263mi.Batch(n, w, h)
    refs = [None] * n
    for f in range(4):
        mbs, cos = [], []
        for s in range(n):
            if f == 0:
                m, c = recgen.intra_picture(w, h, seed=10 * s + 1)
            else:
                m, c = recgen.inter_picture(w, h, seed=100 * f + s, mv_range=40, p_4v=0.3, p_intra=0.1, p_coded=0.4,
                                            quant=7)
                if s == 1:
                    m = m[:37]                                   # short picture: the rest is padded (state.rs:421-427)
                    c = c[:int(m["coeff_index"][-1]) + bin(int(m["cbp"][-1])).count("1")]
                if s == 2:
                    m = m.copy()
                    m["cbp"] = 0                                 # nothing coded at all
                    m["mb_type"] = np.where(np.isin(m["mb_type"], (3, 4)), 0, m["mb_type"])
                    c = c[:0]
            mbs.append(m)
            cos.append(c)
        pt = h263mi.PICTURE_I if f == 0 else h263mi.PICTURE_P
        if f & 1:                                                # odd pictures through the sparse transport
            fe = []
            for m, c in zip(mbs, cos):
                intra = np.zeros(len(c), bool)
                for r in m:
                    if r["mb_type"] in (3, 4):
                        k = bin(int(r["cbp"])).count("1")
                        intra[int(r["coeff_index"]):int(r["coeff_index"]) + k] = True
                fe.append(h263mi.events_from_dense(c, intra))
            b.submit_host_events(pt, mbs, [x[0] for x in fe], [x[1] for x in fe])
        else:
            b.submit_host(pt, mbs, cos)
        b.sync()
        for s in range(n):
            rc, refs[s] = orc.decode_picture(w, h, mbs[s], cos[s], refs[s])
            assert rc == 0
            assert_planes_equal(b.copy_yuv(s), refs[s], "frame %d stream %d" % (f, s))
    b.close()


@pytest.mark.parametrize("w,h", [(176, 144), (100, 60), (16, 16), (320, 240)])
def test_submit_picture_events_equals_dense_submit(w, h):
    """sparse coefficient transport (events read by the reconstruction waves) against the oracle, incl. blocks without any event, an intra block
    whose element 0 is set in the dense form (ignored there, absent here), and duplicate positions"""
    st = h263mi.H263State()
    mbs, co = recgen.intra_picture(w, h, seed=w + 1, max_level=127)
    intra_blk = np.ones(len(co), bool)
    first, ev = h263mi.events_from_dense(co, intra_blk)
    st.submit_picture_events(w, h, mbs, first, ev, h263mi.PICTURE_I)
    rc, want = orc.decode_picture(w, h, mbs, co, None)
    assert_planes_equal(st.get_last_picture().as_yuv(), want, "I")
    for f in range(3):
        mbs, co = recgen.inter_picture(w, h, seed=f + h, mv_range=50, p_4v=0.3, p_intra=0.2, p_coded=0.5, quant=9,
                                       max_level=1023, sparse_low=bool(f & 1))
        # which dense blocks belong to intra macroblocks
        intra_blk = np.zeros(len(co), bool)
        for m in mbs:
            if m["mb_type"] in (3, 4):
                n = bin(int(m["cbp"])).count("1")
                intra_blk[int(m["coeff_index"]):int(m["coeff_index"]) + n] = True
        first, ev = h263mi.events_from_dense(co, intra_blk)
        st.submit_picture_events(w, h, mbs, first, ev, h263mi.PICTURE_P, temporal_reference=f + 1)
        rc, want = orc.decode_picture(w, h, mbs, co, want)
        assert_planes_equal(st.get_last_picture().as_yuv(), want, "P%d" % f)
    # malformed offsets are rejected before anything is touched
    bad = first.copy()
    if len(bad) > 2:
        bad[1] = bad[-1] + 5
        with pytest.raises(h263mi.H263Error):
            st.submit_picture_events(w, h, mbs, bad, ev, h263mi.PICTURE_P)
        assert_planes_equal(st.get_last_picture().as_yuv(), want, "after error")
    st.close()


def test_batch_overlap_mode_gives_the_same_pictures():
    """H263MI_CFG_OVERLAP_POST: k_post on a second stream under the next k_recon; the frame-set dependencies are HIP
    events.  Several pictures back to back without a sync in between, RGBA of every picture checked."""
    w, h, n = 176, 144, 3
    b = h263mi.Batch(n, w, h, overlap_post=True)
    cw = (w + 1) // 2
    refs = [None] * n
    d_rgba = [h263mi.DeviceBuffer(n * w * h * 4) for _ in range(5)]
    want_rgba = []
    for f in range(5):
        mbs, cos = [], []
        for s in range(n):
            if f == 0:
                m, c = recgen.intra_picture(w, h, seed=7 * s + 2)
            else:
                m, c = recgen.inter_picture(w, h, seed=50 * f + s, mv_range=40, p_4v=0.3, p_coded=0.4, quant=6)
            mbs.append(m)
            cos.append(c)
            rc, refs[s] = orc.decode_picture(w, h, m, c, refs[s])
        b.submit_host(h263mi.PICTURE_I if f == 0 else h263mi.PICTURE_P, mbs, cos)
        b.render_rgba(5, d_rgba[f].ptr)
        want_rgba.append([orc.yuv420_to_rgba(*(orc.deblock(p, pw, 5) for p, pw in zip(refs[s], (w, cw, cw))), w)
                          for s in range(n)])
    b.sync()
    for f in range(5):
        got = d_rgba[f].download().reshape(n, -1)
        for s in range(n):
            assert (got[s] == want_rgba[f][s].reshape(-1)).all(), (f, s)
    for s in range(n):
        assert_planes_equal(b.copy_yuv(s), refs[s], "stream %d" % s)
    b.close()


def test_states_on_different_host_threads():
    """include/h263mi.h: one h263mi_state per stream, not thread-safe per object, but distinct states may be driven
    from different host threads (ctypes releases the GIL inside the calls)."""
    import threading
    w, h, n_threads, n_frames = 176, 144, 6, 6
    errors = []

    def worker(tid):
        try:
            st = h263mi.H263State()
            ref = None
            for f in range(n_frames):
                if f == 0:
                    mbs, co = recgen.intra_picture(w, h, seed=100 + tid)
                    pt = h263mi.PICTURE_I
                else:
                    mbs, co = recgen.inter_picture(w, h, seed=1000 * tid + f, mv_range=40, p_4v=0.3, p_coded=0.4, quant=5 + tid)
                    pt = h263mi.PICTURE_P
                st.submit_picture(w, h, mbs, co, pt, temporal_reference=f)
                rc, ref = orc.decode_picture(w, h, mbs, co, ref if f else None)
                got = st.get_last_picture().as_yuv()
                for g, e in zip(got, ref):
                    if not (np.asarray(g) == e).all():
                        errors.append((tid, f))
                cw = (w + 1) // 2
                planes = tuple(orc.deblock(p, pw, 4) for p, pw in zip(ref, (w, cw, cw)))
                if not (st.render_rgba(4) == orc.yuv420_to_rgba(*planes, w)).all():
                    errors.append((tid, f, "rgba"))
            st.close()
        except Exception as e:          # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(n_threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
