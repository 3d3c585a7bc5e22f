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
