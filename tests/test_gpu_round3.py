"""GPU tests added in round 3.

* the kernel of the headline number (k_frame) at the headline geometry: 64 x 1080p, frame-pipelined, the 4-band deal
  with two pictures side by side, both directions of the walk over the pictures, I and P pictures, dense I pictures;
* NAMED known answers for the IDCT class corner cases (idct.rs:113-169, rle.rs:130-133) on the MI355X: the reachable
  Vert columns of tests/golden/vert_named_columns.json and EVERY reachable Dc value -- INTRADC codes 1..255 without 128,
  and inter blocks whose only coefficient sits at zigzag 0 for every quantiser and LEVEL -- over predictions 0 and 255;
* the RCCL leg of bench.py on one GPU.

Everything goes through the C ABI and is compared with the oracle (or a closed form) bit for bit."""
import json
import os
import subprocess
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


def _upload(arr):
    d = h263mi.DeviceBuffer(max(arr.nbytes, 16))
    if arr.nbytes:
        d.upload(arr)
    return d


def _rgba_want(planes, strength, w=W):
    cw = (w + 1) // 2
    filt = planes if strength == 0 else tuple(orc.deblock(p, pw, strength) for p, pw in zip(planes, (w, cw, cw)))
    return orc.yuv420_to_rgba(*filt, w)


# ---------------------------------------------------------------------------------------------
# k_frame at the geometry of the bench line: Batch(64, 1920, 1080, pipeline_post) -- every launch but the first is a
# k_frame (reconstruction of picture f + deblock / RGBA of picture f - 1), a picture is dealt to 4 XCDs with two
# pictures side by side, and the direction in which the 64 pictures are walked alternates from launch to launch
# ---------------------------------------------------------------------------------------------
def test_k_frame_at_the_headline_geometry_64x1080p_i_plus_5p():
    n, first_stream, n_frames, strength = 64, 11, 6, 5
    check = (0, 31, 63)
    b = h263mi.Batch(n, W, H, pipeline_post=True)
    d_rgba = [h263mi.DeviceBuffer(n * W * H * 4) for _ in range(n_frames)]
    bufs = []
    b.timing_reserve(2 * n_frames)
    b.timing_begin()
    for f in range(n_frames):
        kind = h263mi.SYNTH_I_MIXED if f == 0 else h263mi.SYNTH_P
        cap = n * MBS_PP * (6 if f == 0 else 2)
        d = (h263mi.DeviceBuffer(n * MBS_PP * 32), h263mi.DeviceBuffer(cap * 128), h263mi.DeviceBuffer(n * 8))
        total = h263mi.synth_batch_device(kind, W, H, n, first_stream, f, d[0].ptr, d[1].ptr, cap, d[2].ptr)
        bufs.append(d)
        b.decode(h263mi.PICTURE_I if f == 0 else h263mi.PICTURE_P, d[0].ptr, d[1].ptr, d[2].ptr, total, strength,
                 d_rgba[f].ptr)
    b.sync()
    kt = b.timing_end()
    # the first picture has nothing to post-process beside it (k_recon), the last one's post-processing runs at the sync
    assert (kt.recon_launches, kt.frame_launches, kt.post_launches) == (1, n_frames - 1, 1)
    for s in check:
        ref = None
        for f in range(n_frames):
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


def test_k_frame_dense_i_pictures_1080p_x16():
    """BASELINE configs[1] through the fused launch: dense I pictures (every block Full, 6 coded blocks per macroblock),
    16 pictures = the smallest batch that takes the 4-band deal; no deblocking (strength 0 = plain yuv420_to_rgba) on the
    second picture, strength 9 on the others"""
    n, first_stream, n_frames = 16, 3, 3
    check = (0, 7, 15)
    b = h263mi.Batch(n, W, H, pipeline_post=True)
    d_rgba = [h263mi.DeviceBuffer(n * W * H * 4) for _ in range(n_frames)]
    strengths = (9, 0, 9)
    bufs = []
    for f in range(n_frames):
        cap = n * MBS_PP * 6
        d = (h263mi.DeviceBuffer(n * MBS_PP * 32), h263mi.DeviceBuffer(cap * 128), h263mi.DeviceBuffer(n * 8))
        total = h263mi.synth_batch_device(h263mi.SYNTH_I_DENSE, W, H, n, first_stream, f, d[0].ptr, d[1].ptr, cap, d[2].ptr)
        assert total == cap
        bufs.append(d)
        b.decode(h263mi.PICTURE_I, d[0].ptr, d[1].ptr, d[2].ptr, total, strengths[f], d_rgba[f].ptr)
    b.sync()
    for s in check:
        for f in range(n_frames):
            mbs, co = h263mi.synth_picture_host(h263mi.SYNTH_I_DENSE, W, H, first_stream + s, f)
            rc, ref = orc.decode_picture(W, H, mbs, co, None)
            assert rc == 0
            got = d_rgba[f].download(W * H * 4, s * W * H * 4)
            assert (got == _rgba_want(ref, strengths[f])).all(), "RGBA stream %d frame %d" % (s, f)
        assert_planes_equal(b.copy_yuv(s), ref, "last picture of stream %d" % s)
    b.close()


# ---------------------------------------------------------------------------------------------
# named Vert columns (idct.rs:152-169): tests/golden/vert_named_columns.json, made by tools/gen_vert_columns.py
# ---------------------------------------------------------------------------------------------
def _flat_state(w, h, code):
    """a state whose last picture is flat: every pixel = code (DC-only intra blocks: the closed form of SURVEY 8c)"""
    st = h263mi.H263State()
    mbw, mbh = (w + 15) // 16, (h + 15) // 16
    flat = np.zeros(mbw * mbh, orc.MB_RECORD_DTYPE)
    flat["mb_type"] = 3
    flat["quant"] = 1
    flat["intradc"] = code
    st.submit_picture(w, h, flat, np.zeros((0, 64), np.int16), h263mi.PICTURE_I)
    got = st.get_last_picture().as_yuv()
    assert all((p == code).all() for p in got)
    return st


def test_named_vert_columns_take_the_vert_arithmetic():
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "vert_named_columns.json")))["columns"]
    w, h, base = 64, 32, 100                                   # 8 macroblocks, 48 blocks
    st = _flat_state(w, h, base)
    mbs = np.zeros(8, orc.MB_RECORD_DTYPE)                     # inter, zero vectors: prediction = the flat picture
    mbs["quant"] = 1
    placed = []                                                # (macroblock, block, golden entry)
    for k, g in enumerate(gold):
        for blk in (k % 6, (k + 3) % 6):                       # every column in two block positions (luma and chroma)
            m = (2 * k + (blk > k % 6)) % 8
            if not mbs[m]["cbp"] & (1 << blk):
                mbs[m]["cbp"] |= 1 << blk
                placed.append((m, blk, g))
    # coded blocks follow each other in macroblock order, block order inside a macroblock
    coeffs = []
    for m in range(8):
        mbs[m]["coeff_index"] = len(coeffs)
        for mm, blk, g in sorted(placed, key=lambda t: (t[0], t[1])):
            if mm == m:
                c = np.zeros(64, np.int16)
                for r, level in g["levels_at_quant_1"].items():
                    c[8 * int(r)] = level                      # raster x + 8y: column 0, row r
                coeffs.append(c)
    coeffs = np.array(coeffs, np.int16)
    st.submit_picture(w, h, mbs, coeffs, h263mi.PICTURE_P)
    y, cb, cr = st.get_last_picture().as_yuv()
    y, cb, cr = y.reshape(h, w), cb.reshape(h // 2, w // 2), cr.reshape(h // 2, w // 2)
    for m, blk, g in placed:
        mx, my = m % 4, m // 4
        if blk < 4:
            tile = y[my * 16 + 8 * (blk >> 1):, mx * 16 + 8 * (blk & 1):][:8, :8]
        else:
            tile = (cb if blk == 4 else cr)[my * 8:, mx * 8:][:8, :8]
        want = base + np.array(g["vert"])[:, None] * np.ones((1, 8), int)
        assert (tile == want).all(), "column %s in macroblock %d block %d: got rows %s, Vert gives %s, Full would give %s" % (
            g["column"], m, blk, (tile[:, 0].astype(int) - base).tolist(), g["vert"], g["full"])
    # and the oracle agrees with the whole picture (untouched blocks stay flat)
    rc, ref = orc.decode_picture(w, h, mbs, coeffs, tuple(np.full(n, base, np.uint8) for n in (w * h, w * h // 4, w * h // 4)))
    assert rc == 0
    assert_planes_equal((y.ravel(), cb.ravel(), cr.ravel()), ref, "named Vert columns")
    st.close()


# ---------------------------------------------------------------------------------------------
# every reachable Dc value (idct.rs:113-131: r = ((dc * 0.5 / 4.0 + signum(dc) * 0.5) as i16).clamp(-256, 255))
# ---------------------------------------------------------------------------------------------
def _dc_residual(dc):
    """idct.rs:119-120 in exact arithmetic: dc/8 is exact in binary32 for |dc| <= 2048, so is the sum with +-0.5"""
    v = abs(dc) / 8.0 + 0.5
    r = int(v)                                                 # truncation toward zero of a positive number
    r = r if dc > 0 else -r
    return max(-256, min(255, r))


def test_every_intradc_code_gives_its_closed_form():
    """INTRADC codes 1..255 without 128 (types.rs:930-936), blocks without TCOEF: Dc(level), level = code << 3, 0xFF ->
    1024 (types.rs:955-961); every pixel of the block = clamp(residual) = code (255 -> 128).  Through the state API
    (k_recon) and through a frame-pipelined batch (k_frame)."""
    codes = [c for c in range(1, 256) if c != 128]
    w, h = 112, 112                                            # 49 macroblocks = 294 blocks >= 254 codes
    mbs = np.zeros(49, orc.MB_RECORD_DTYPE)
    mbs["mb_type"] = 3
    mbs["quant"] = np.arange(49) % 31 + 1                      # the quantiser must not matter
    flat = np.array((codes + codes)[:294], np.uint8).reshape(49, 6)
    mbs["intradc"] = flat
    st = h263mi.H263State()
    st.submit_picture(w, h, mbs, np.zeros((0, 64), np.int16), h263mi.PICTURE_I)
    planes = st.get_last_picture().as_yuv()
    st.close()
    b = h263mi.Batch(2, w, h, pipeline_post=True)
    d_m, d_c = _upload(np.concatenate([mbs, mbs])), _upload(np.zeros((1, 64), np.int16))
    d_rgba = h263mi.DeviceBuffer(2 * w * h * 4)
    for rep in range(2):                                       # the second call is a k_frame launch
        b.decode(h263mi.PICTURE_I, d_m.ptr, d_c.ptr, None, 1, 0, d_rgba.ptr)
    b.sync()
    for got in (planes, b.copy_yuv(0), b.copy_yuv(1)):
        y, cb, cr = got[0].reshape(h, w), got[1].reshape(h // 2, w // 2), got[2].reshape(h // 2, w // 2)
        for m in range(49):
            mx, my = m % 7, m // 7
            for blk in range(6):
                code = int(flat[m, blk])
                want = 128 if code == 255 else code
                assert want == min(255, max(0, _dc_residual(1024 if code == 255 else code << 3)))
                if blk < 4:
                    tile = y[my * 16 + 8 * (blk >> 1):, mx * 16 + 8 * (blk & 1):][:8, :8]
                else:
                    tile = (cb if blk == 4 else cr)[my * 8:, mx * 8:][:8, :8]
                assert (tile == want).all(), (code, m, blk, tile[0, :4])
    b.close()


@pytest.mark.parametrize("pred", [0, 255])
def test_every_reachable_inter_dc_value_over_prediction(pred):
    """inter blocks whose only coefficient sits at zigzag 0: Dc(v), v = sign(L) * (q * (2|L| + 1) - (q even)) clamped to
    [-2048, 2047] (rle.rs:130-133) -- every quantiser 1..31 with every LEVEL +-1..127, and +-1023 where the record
    contract q * (2|L| + 1) <= 32767 allows it (q <= 16) -- on a flat prediction of 0 and of 255, so that both ends of
    the final clamp(0, 255) (idct.rs:127-130) are hit.  Expected pixels from the closed form AND from the oracle."""
    w, h = 1024, 512                                           # 2 048 macroblocks
    mbw = w // 16
    # a flat picture of exactly 0 / 255: flat 1 / 254 from INTRADC (codes 0 and 255 do not give 0 / 255), then a P picture
    # whose Dc blocks push every pixel over the end of the range
    st = _flat_state(w, h, 1 if pred == 0 else 254)
    push = np.zeros(mbw * (h // 16), orc.MB_RECORD_DTYPE)
    push["quant"] = 8
    push["cbp"] = 0x3F
    push["coeff_index"] = np.arange(len(push)) * 6
    pc = np.zeros((len(push) * 6, 64), np.int16)
    pc[:, 0] = -20 if pred == 0 else 20
    st.submit_picture(w, h, push, pc, h263mi.PICTURE_P)
    got = st.get_last_picture().as_yuv()
    assert all((p == pred).all() for p in got)
    # the sweep
    cases = []                                                 # (quant, level)
    for q in range(1, 32):
        levels = list(range(1, 128)) + ([1023] if q * 2047 <= 32767 else [])
        per_q = [(q, s * lv) for lv in levels for s in (1, -1)]
        per_q += [(q, 1)] * (-len(per_q) % 6)                  # whole macroblocks per quantiser
        cases += per_q
    n_mb = len(cases) // 6
    assert n_mb <= len(push)
    mbs = np.zeros(n_mb, orc.MB_RECORD_DTYPE)
    mbs["quant"] = [cases[6 * m][0] for m in range(n_mb)]
    mbs["cbp"] = 0x3F
    mbs["coeff_index"] = np.arange(n_mb) * 6
    co = np.zeros((n_mb * 6, 64), np.int16)
    co[:, 0] = [lv for q, lv in cases]
    st.submit_picture(w, h, mbs, co, h263mi.PICTURE_P)        # (the macroblocks behind n_mb are padded: Inter, mv 0)
    y, cb, cr = st.get_last_picture().as_yuv()
    ref = tuple(np.full(n, pred, np.uint8) for n in (w * h, w * h // 4, w * h // 4))
    rc, want = orc.decode_picture(w, h, mbs, co, ref)
    assert rc == 0
    assert_planes_equal((y, cb, cr), want, "inter Dc sweep over prediction %d" % pred)
    y, cb, cr = y.reshape(h, w), cb.reshape(h // 2, w // 2), cr.reshape(h // 2, w // 2)
    seen = set()
    for k, (q, lv) in enumerate(cases):
        m, blk = divmod(k, 6)
        mx, my = m % mbw, m // mbw
        v = q * (2 * abs(lv) + 1) - (1 if q % 2 == 0 else 0)
        v = max(-2048, min(2047, v if lv > 0 else -v))
        seen.add(v)
        expect = max(0, min(255, pred + _dc_residual(v)))
        if blk < 4:
            tile = y[my * 16 + 8 * (blk >> 1):, mx * 16 + 8 * (blk & 1):][:8, :8]
        else:
            tile = (cb if blk == 4 else cr)[my * 8:, mx * 8:][:8, :8]
        assert (tile == expect).all(), "q %d LEVEL %d (Dc %d) over %d: got %d, closed form %d" % (q, lv, v, pred, tile[0, 0], expect)
    assert len(seen) > 1800 and 2047 in seen and -2048 in seen
    st.close()


# ---------------------------------------------------------------------------------------------
# the RCCL leg of bench.py in a driver-run record: one rank, process group "nccl" (= RCCL) forced on, the barrier, the
# max-over-ranks reduction of the elapsed time, the sum of the pictures and the parity-gate verdict all go through it
# ---------------------------------------------------------------------------------------------
def _bench_child(extra_args, extra_env):
    env = dict(os.environ)
    env.update(extra_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = "29533"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-extra",
           "--no-cpu-baseline"] + extra_args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_bench_rccl_leg_on_one_gpu():
    out = _bench_child([], {"H263MI_FORCE_DIST": "1"})
    assert out["parity_gate"] == "ok" and out["n_gpus"] == 1 and out["scaling"] == "weak"
    assert out["value"] > 1000.0 and out["config"]["streams_per_gpu"] == 64
    assert out["roofline"]["kernel"] == "k_frame" and out["roofline"]["achieved"] > 0


def test_bench_rccl_leg_strong_scaling_form():
    out = _bench_child(["--total-streams", "64"], {"H263MI_FORCE_DIST": "1"})
    assert out["parity_gate"] == "ok" and out["n_gpus"] == 1 and out["scaling"] == "strong"
    assert out["config"]["streams_per_gpu"] == 64 and out["config"]["pictures_per_step"] == 64 * 124


# ---------------------------------------------------------------------------------------------
# every stream of a batch is its own H263State (state.rs:16-50, 138-142, 464-483): one stream gets a corrupt picture and
# keeps its state, one restarts with an I picture, one has no picture in a call, one is reset and must wait for an I
# picture -- all 64 compared with 64 independent oracle chains after every call, through the frame-pipelined launch
# ---------------------------------------------------------------------------------------------
def test_batch_streams_are_independent_states():
    import sorenson_enc as enc
    from test_bitstream_e2e import make_codable
    w, h, n, q = 176, 144, 64, 6
    cw = (w + 1) // 2
    b = h263mi.Batch(n, w, h, pipeline_post=True)
    refs = [None] * n                                            # the oracle chain of every stream
    d_rgba = [h263mi.DeviceBuffer(n * w * h * 4) for _ in range(7)]
    rendered = []                                                # (call, stream, expected RGBA)

    def picture(s, f, intra):
        if intra:
            mbs, co = recgen.intra_picture(w, h, seed=1000 * f + s, max_level=60)
            mbs = make_codable(mbs, q, s, 0)
        else:
            mbs, co = recgen.inter_picture(w, h, seed=1000 * f + s, mv_range=32, p_4v=0.2, p_intra=0.05, p_coded=0.4, quant=q,
                                           max_level=60)
            mbs = make_codable(mbs, q, s + f, 1)
        return enc.encode_picture(w, h, 0 if intra else 1, q, mbs, co, temporal_reference=f), mbs, co

    def step(f, plan, expect_rc):
        """plan[s]: 'I', 'P', 'corrupt', None (no data).  expect_rc[s]: the code stream s must report."""
        datas = []
        for s in range(n):
            kind = plan.get(s, "P")
            if kind is None:
                datas.append(None)
                continue
            data, mbs, co = picture(s, f, kind == "I")
            if kind == "corrupt":
                datas.append(data[:4])                           # not even a whole picture header (a picture cut off behind
                                                                 # its header would decode: state.rs:411 ends a picture at EOF)
                continue
            datas.append(data)
            if expect_rc.get(s, 0) == 0:
                rc, refs[s] = orc.decode_picture(w, h, mbs, co, None if kind == "I" else refs[s])
                assert rc == 0
                rendered.append((f, s, _rgba_want(refs[s], 5, w)))
            
        used, rcs = b.decode_next_pictures_ex(datas, n_threads=4, strength=5, d_rgba=d_rgba[f].ptr)
        for s in range(n):
            want_rc = expect_rc.get(s, 0)
            assert (rcs[s] == want_rc) if want_rc != "any error" else rcs[s] < 0, "call %d stream %d: rc %d" % (f, s, rcs[s])
            assert (used[s] > 0) == (plan.get(s, "P") is not None and rcs[s] == 0)
        assert all(rc == 0 for rc in b.sync_streams())
        for s in range(n):
            if refs[s] is None:
                assert not b.stream_has_picture(s)
            else:
                assert_planes_equal(b.copy_yuv(s), refs[s], "after call %d, stream %d" % (f, s))

    step(0, {s: "I" for s in range(n)}, {})
    step(1, {}, {})
    # stream 5: corrupt data -> its error, its state untouched; stream 9 restarts with an I picture; stream 20 sits the call out
    step(2, {5: "corrupt", 9: "I", 20: None}, {5: "any error"})
    step(3, {}, {})                                              # 5 and 20 predict from their picture of call 1
    b.reset_stream(33)
    refs[33] = None
    assert not b.stream_has_picture(33)
    step(4, {}, {33: h263mi.ERR_UNCODED_IFRAME_BLOCKS})          # a reset stream needs an I picture (gather.rs:149)
    step(5, {33: "I", 40: None}, {})
    step(6, {}, {})
    b.sync()
    # the RGBA of every picture that was decoded, delivered by the launch of the following call (or the last sync)
    for f, s, want in rendered:
        got = d_rgba[f].download(w * h * 4, s * w * h * 4)
        assert (got == want).all(), "RGBA of call %d stream %d" % (f, s)
    b.close()


def test_batch_device_error_rolls_back_one_stream_only():
    """record-level batch API: the device rejects the picture of ONE stream (a coded block outside the pool); that stream
    goes back to its previous picture, the others keep the new one, and the next P pictures predict per stream from the
    right frame set (the streams' ping-pong positions now differ)"""
    w, h, n = 96, 64, 6
    mbs_pp = 6 * 4
    b = h263mi.Batch(n, w, h, pipeline_post=True)
    refs = [None] * n
    d_rgba = h263mi.DeviceBuffer(n * w * h * 4)

    def submit(f, intra, bad=None):
        mbs_all, co_all, base, at = [], [], [], 0
        for s in range(n):
            m, c = (recgen.intra_picture(w, h, seed=50 * f + s) if intra else
                    recgen.inter_picture(w, h, seed=50 * f + s, mv_range=40, p_4v=0.3, p_coded=0.5, quant=7))
            if s == bad:
                m = m.copy()
                m[3]["cbp"] = 0x3F
                m[3]["coeff_index"] = 10 ** 6                    # far outside the pool
            else:
                rc, refs[s] = orc.decode_picture(w, h, m, c, None if intra else refs[s])
                assert rc == 0
            mbs_all.append(m)
            co_all.append(c)
            base.append(at)
            at += len(c)
        d = (_upload(np.concatenate(mbs_all)), _upload(np.concatenate(co_all) if at else np.zeros((1, 64), np.int16)),
             _upload(np.array(base, np.uint64)))
        b.decode(h263mi.PICTURE_I if intra else h263mi.PICTURE_P, d[0].ptr, d[1].ptr, d[2].ptr, max(at, 1), 5, d_rgba.ptr)
        return d

    keep = [submit(0, True)]
    assert all(rc == 0 for rc in b.sync_streams())
    keep.append(submit(1, False, bad=2))
    rcs = b.sync_streams()
    assert rcs == [0, 0, h263mi.ERR_INVALID_ARGUMENT, 0, 0, 0]
    for s in range(n):
        assert_planes_equal(b.copy_yuv(s), refs[s], "after the rejected picture, stream %d" % s)      # refs[2] is still its I picture
    for f in (2, 3, 4):
        keep.append(submit(f, False))
        assert all(rc == 0 for rc in b.sync_streams())
        for s in range(n):
            assert_planes_equal(b.copy_yuv(s), refs[s], "frame %d stream %d" % (f, s))
        got = d_rgba.download()
        for s in range(n):
            assert (got[s * w * h * 4:(s + 1) * w * h * 4] == _rgba_want(refs[s], 5, w)).all(), (f, s)
    # a stream that sits calls out keeps its picture while the others advance
    b.set_active([s != 4 for s in range(n)])
    frozen = refs[4]
    keep.append(submit(5, False))
    refs[4] = frozen
    b.sync()
    b.set_active(None)
    keep.append(submit(6, False))
    b.sync()
    for s in range(n):
        assert_planes_equal(b.copy_yuv(s), refs[s], "after the skipped call, stream %d" % s)
    b.close()
    assert mbs_pp == len(recgen.intra_picture(w, h, seed=1)[0])


# ---------------------------------------------------------------------------------------------
# h263mi_batch_decode_events: records + sparse events in device memory (the transport form of the host parser), read
# by the reconstruction waves as they are -- intra pictures of every block class (up to 63 events per block), P pictures
# with sparse and with full residuals, through k_recon and through the frame-pipelined launch
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h,n,pipeline", [(176, 144, 3, False), (100, 60, 17, True), (352, 288, 2, True)])
def test_batch_decode_events_matches_the_oracle(w, h, n, pipeline):
    b = h263mi.Batch(n, w, h, pipeline_post=pipeline)
    refs = [None] * n
    d_rgba = h263mi.DeviceBuffer(n * w * h * 4)
    keep = []
    for f in range(5):
        intra = f in (0, 3)
        mbs_all, co_all, intra_blocks, base, at = [], [], [], [], 0
        for s in range(n):
            if intra:
                m, c = recgen.intra_picture(w, h, seed=31 * s + f)
            else:
                m, c = recgen.inter_picture(w, h, seed=200 * f + s, mv_range=40, p_4v=0.3, p_intra=0.15, p_coded=0.5, quant=0,
                                            max_level=127, sparse_low=bool(s & 1))
            rc, refs[s] = orc.decode_picture(w, h, m, c, None if intra else refs[s])
            assert rc == 0
            m = simlib_pad(m, w, h)
            blk_intra = np.zeros(len(c), bool)
            for r in m:
                if int(r["mb_type"]) in (3, 4):
                    k = int(r["coeff_index"])
                    blk_intra[k:k + bin(int(r["cbp"])).count("1")] = True
            mbs_all.append(m)
            co_all.append(c)
            intra_blocks.append(blk_intra)
            base.append(at)
            at += len(c)
        first, ev = h263mi.events_from_dense(np.concatenate(co_all) if at else np.zeros((0, 64), np.int16),
                                             np.concatenate(intra_blocks) if at else None)
        ev = np.concatenate([ev, np.zeros(8, np.uint32)])
        d = (_upload(np.concatenate(mbs_all)), _upload(first), _upload(ev), _upload(np.array(base, np.uint64)))
        keep.append(d)
        b.decode_events(h263mi.PICTURE_I if intra else h263mi.PICTURE_P, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, max(at, 1), 5,
                        d_rgba.ptr)
        b.sync()
        for s in range(n):
            assert_planes_equal(b.copy_yuv(s), refs[s], "frame %d stream %d" % (f, s))
            assert (d_rgba.download(w * h * 4, s * w * h * 4) == _rgba_want(refs[s], 5, w)).all(), (f, s)
    b.close()


def simlib_pad(mbs, w, h):
    import simlib
    return simlib.pad_records(mbs, w, h)


# ---------------------------------------------------------------------------------------------
# The dequantiser at every LEVEL and every quantiser.  The kernels compute 16 x the dequantised value with a saturating
# 16-bit multiply-add whose saturation IS the reference's clamp to [-2048, 2047] (rle.rs:130-133), and take the factor
# out again through the row pass's basis table (recon_kernel.inl: dequant_pair_i16, kBasisSixteenth): every one of the
# 4 094 non-zero 12-bit LEVELs at every quantiser 1..31 -- all overflow and saturation cases included -- as the AC
# coefficients of Full-class intra blocks, and as inter blocks over a flat prediction, against the oracle.
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("span", [511, 2047], ids=["within_9_bits", "12_bits"])
def test_every_level_at_every_quantiser_dequantises_like_the_oracle(span):
    """span = 511: no LEVEL outside [-512, 511] anywhere, so every IDCT round takes the saturating 16 x form (round 5: a round
    that holds a wider LEVEL takes the wrapping form instead, tests/test_gpu_round5.py) -- its saturation at 12 bits is what
    the dequant mutant breaks; span = 2047: every 12-bit LEVEL, the reference's i16 wrap included (the oracle wraps too)."""
    w, h = 176, 144                                            # 99 macroblocks = 594 blocks of 63 AC coefficients
    levels = np.array([v for v in range(-span, span + 1) if v != 0], np.int16)
    n_blocks = 99 * 6
    rng = np.random.default_rng(7)
    for q in range(1, 32):
        vals = np.concatenate([levels, rng.choice(levels, n_blocks * 63 - len(levels))])
        rng.shuffle(vals)
        co = np.zeros((n_blocks, 64), np.int16)
        co[:, 1:] = vals.reshape(n_blocks, 63)
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
            st.submit_picture(w, h, mbs, c, ptype)
            rc, want = orc.decode_picture(w, h, mbs, c, ref)
            assert rc == 0
            assert_planes_equal(st.get_last_picture().as_yuv(), want, "q %d type %d" % (q, ptype))
            st.close()
