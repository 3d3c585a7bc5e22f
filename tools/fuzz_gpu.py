#!/usr/bin/env python3
"""Randomised differential test on the GPU box: random picture sizes, record mixes, vectors, quantisers and filter
strengths through the C ABI (state API) against the oracle.  Usage: python tools/fuzz_gpu.py [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h263-rs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np

import h263mi
import recgen
import sorenson_enc as enc
from oracle import oracle as orc
from test_bitstream_e2e import make_codable


class FuzzMismatch(AssertionError):
    pass


Q2S = [int(v) for v in orc.quant_to_strength()]


def _dev(arr):
    arr = np.ascontiguousarray(arr)
    d = h263mi.DeviceBuffer(max(arr.nbytes, 16))
    if arr.nbytes:
        d.upload(arr)
    return d


def fuzz_batch(rng, w, h, seed, n_pic, n_px):
    import simlib
    n = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 24]))
    pipeline, events = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    frames = int(rng.integers(2, 6))
    b = h263mi.Batch(n, w, h, pipeline_post=pipeline)
    d_rgba = [h263mi.DeviceBuffer(n * w * h * 4) for _ in range(frames)]
    refs, keep, want = [None] * n, [], []
    cw = (w + 1) // 2
    what = (w, h, n, pipeline, events, seed)
    for f in range(frames):
        intra = f == 0 or rng.random() < 0.2
        mbs_all, co_all, blk_intra_all, base, at = [], [], [], [], 0
        for s_ in range(n):
            sd = int(rng.integers(0, 1 << 30))
            if intra:
                m, c = recgen.intra_picture(w, h, seed=sd, max_level=int(rng.choice([40, 127, 1023])))
            else:
                m, c = recgen.inter_picture(w, h, seed=sd, mv_range=int(rng.choice([2, 32, 70])), p_4v=float(rng.choice([0.0, 0.3])),
                                            p_intra=float(rng.choice([0.0, 0.15])), p_coded=float(rng.choice([0.05, 0.3, 0.9])),
                                            quant=int(rng.choice([0, 1, 10, 31])), max_level=int(rng.choice([3, 60, 1023])),
                                            sparse_low=bool(rng.integers(0, 2)))
            rc, refs[s_] = orc.decode_picture(w, h, m, c, None if intra else refs[s_])
            assert rc == 0
            m = simlib.pad_records(m, w, h)
            bi = np.zeros(len(c), bool)
            for r in m:
                if int(r["mb_type"]) in (3, 4):
                    k = int(r["coeff_index"])
                    bi[k:k + bin(int(r["cbp"])).count("1")] = True
            mbs_all.append(m)
            co_all.append(c)
            blk_intra_all.append(bi)
            base.append(at)
            at += len(c)
        co = np.concatenate(co_all) if at else np.zeros((0, 64), np.int16)
        strength = int(rng.integers(0, 13))
        # (ABI 7) one strength per stream in half of the frames; the sizes of the arrays given, or left to the allocations
        strengths = [int(v) for v in rng.integers(0, 13, n)] if rng.random() < 0.5 else None
        told = rng.random() < 0.5
        pt = h263mi.PICTURE_I if intra else h263mi.PICTURE_P
        d_m, d_b = _dev(np.concatenate(mbs_all)), _dev(np.array(base, np.uint64))
        if events:
            first, ev = h263mi.events_from_dense(co, np.concatenate(blk_intra_all) if at else None)
            d_f, d_e = _dev(first), _dev(np.concatenate([ev, np.zeros(8, np.uint32)]))
            keep.append((d_m, d_b, d_f, d_e))
            b.decode_events(pt, d_m.ptr, d_f.ptr, d_e.ptr, d_b.ptr, max(at, 1) if told else 0, strength, d_rgba[f].ptr,
                            n_events=len(ev) if told else 0, strengths=strengths)
        else:
            d_c = _dev(co if at else np.zeros((1, 64), np.int16))
            keep.append((d_m, d_b, d_c))
            b.decode(pt, d_m.ptr, d_c.ptr, d_b.ptr, max(at, 1) if told else 0, strength, d_rgba[f].ptr, strengths=strengths)
        # a pipelined batch renders picture f inside the launch that reconstructs picture f + 1 -- unless a sync comes
        # first, which renders it with a launch of its own: both orders occur
        if rng.random() < 0.3:
            b.sync()
        want.append([])
        for s_ in range(n):
            st_ = strengths[s_] if strengths else strength
            planes = refs[s_] if st_ == 0 else tuple(orc.deblock(p, pw, st_) for p, pw in zip(refs[s_], (w, cw, cw)))
            want[f].append(orc.yuv420_to_rgba(*planes, w))
        n_pic += n
        n_px += n * w * h
    b.sync()
    for f in range(frames):
        for s_ in range(n):
            if not (d_rgba[f].download(w * h * 4, s_ * w * h * 4) == want[f][s_]).all():
                raise FuzzMismatch("batch rgba: %r frame %d stream %d" % (what, f, s_))
    for s_ in sorted(set([0, n - 1, int(rng.integers(0, n))])):
        for g, e, name in zip(b.copy_yuv(s_), refs[s_], "Y Cb Cr".split()):
            if not (np.asarray(g) == e).all():
                raise FuzzMismatch("batch planes: %r stream %d %s" % (what, s_, name))
    b.close()
    for d in d_rgba:
        d.free()
    for t in keep:
        for d in t:
            d.free()
    return n_pic, n_px


def fuzz_hostile(rng, w, h, seed, n_pic, n_px):
    """HOSTILE device arrays with no sizes given (ABI 7: checked by default, the allocations bound what is read): the block
    offsets, the events' positions in the pool, coded-block indices and per-stream bases of SOME streams of a batch are
    overwritten with garbage (huge values, descending offsets, values just past the end).  Nothing may fault; a stream whose
    arrays were left alone decodes to the oracle's planes; a stream that was hit is either rejected (its previous picture is
    its last picture again) or -- garbage that happens to stay inside the arrays -- decodes SOMETHING; and the batch decodes
    a clean picture afterwards."""
    import simlib
    n = int(rng.choice([1, 2, 5, 8]))
    b = h263mi.Batch(n, w, h, pipeline_post=bool(rng.integers(0, 2)))
    what = ("hostile", w, h, n, seed)
    recs, refs, at, base = [], [], 0, []
    for s_ in range(n):
        m, c = recgen.intra_picture(w, h, seed=int(rng.integers(0, 1 << 30)), max_level=int(rng.choice([40, 1023])))
        rc, ref = orc.decode_picture(w, h, m, c, None)
        assert rc == 0
        recs.append((simlib.pad_records(m, w, h), c))
        refs.append(ref)
        base.append(at)
        at += len(c)
    mbs = np.concatenate([r[0] for r in recs])
    co = np.concatenate([r[1] for r in recs])
    first, ev = h263mi.events_from_dense(co, np.ones(len(co), bool))
    good = (_dev(mbs), _dev(first), _dev(ev if len(ev) else np.zeros(4, np.uint32)), _dev(np.array(base, np.uint64)))

    def clean():
        b.decode_events(h263mi.PICTURE_I, good[0].ptr, good[1].ptr, good[2].ptr, good[3].ptr)
        if any(b.sync_streams()):
            raise FuzzMismatch("clean picture rejected: %r" % (what,))
        for s_ in range(n):
            for g, e, name in zip(b.copy_yuv(s_), refs[s_], "Y Cb Cr".split()):
                if not (np.asarray(g) == e).all():
                    raise FuzzMismatch("clean picture differs: %r stream %d %s" % (what, s_, name))

    clean()
    per = len(mbs) // n
    for attempt in range(int(rng.integers(2, 6))):
        hit = set(int(v) for v in rng.choice(n, size=int(rng.integers(1, n + 1)), replace=False))
        m2, f2, b2 = mbs.copy(), first.copy(), np.array(base, np.uint64)
        for s_ in hit:
            lo, hi = base[s_], (base[s_ + 1] if s_ + 1 < n else at)
            kind = int(rng.integers(0, 5))
            junk = [0xffffffff, 0xfffffffe, 0xfffffff9, 0xfffffff0, len(ev), len(ev) + 1, len(ev) + 64, 1 << 28, 0x7fffffff]
            # (entries lo and hi of the offsets are shared with the neighbouring streams' blocks: only lo + 1 .. hi - 1 are this
            # stream's own)
            if kind in (0, 1, 4) and hi - lo < 4:
                kind = 2
            if kind == 0:                        # garbage offsets inside this stream's blocks
                k = rng.integers(lo + 1, hi, size=min(4, hi - lo - 1))
                f2[k] = rng.choice(junk, size=len(k))
            elif kind == 1:                      # a descending pair
                k = int(rng.integers(lo + 1, hi - 1))
                f2[k], f2[k + 1] = f2[k + 1] + 7, f2[k]
            elif kind == 2:                      # coded-block indices far outside / just outside the pool
                k = s_ * per + rng.integers(0, per, size=3)
                m2["coeff_index"][k] = rng.choice([at, at + 1, 1 << 24, 0xffffffff], size=3)
                m2["cbp"][k] |= 1
            elif kind == 3:                      # the stream's base beyond the pool
                b2[s_] = [at, at + 5, 1 << 40, (1 << 64) - (1 << 20), (1 << 64) - 1][int(rng.integers(0, 5))]
            else:                                # every offset of the stream shifted far out
                f2[lo + 1:hi] = (f2[lo + 1:hi].astype(np.uint64) + int(rng.choice([len(ev), 1 << 27]))).astype(np.uint32)
        d = (_dev(m2), _dev(f2), _dev(b2))
        b.decode_events(h263mi.PICTURE_I, d[0].ptr, d[1].ptr, good[2].ptr, d[2].ptr)
        rcs = b.sync_streams()
        for s_ in range(n):
            if s_ not in hit:
                if rcs[s_] != 0:
                    raise FuzzMismatch("untouched stream %d rejected (%d): %r attempt %d" % (s_, rcs[s_], what, attempt))
                for g, e, name in zip(b.copy_yuv(s_), refs[s_], "Y Cb Cr".split()):
                    if not (np.asarray(g) == e).all():
                        raise FuzzMismatch("untouched stream %d differs: %r attempt %d %s" % (s_, what, attempt, name))
            elif rcs[s_] not in (0, h263mi.ERR_INVALID_ARGUMENT):
                raise FuzzMismatch("hit stream %d: verdict %d: %r" % (s_, rcs[s_], what))
        for x in d:
            x.free()
        clean()
        n_pic += 2 * n
        n_px += 2 * n * w * h
    b.close()
    for x in good:
        x.free()
    return n_pic, n_px


def fuzz_garbage_fields(rng, w, h, seed, n_pic, n_px):
    """GARBAGE record fields and event / coefficient words (tests/sim/garbage_fields_replay.py is the same on the CPU under
    AddressSanitizer): macroblock types 6..255, quantisers 0 and 32..255, any coded-block-pattern and kill byte, vectors over
    the whole int16 range, INTRADC codes that never occur, event words with positions beyond 63 and LEVELs over the whole int16
    range -- in P pictures on a reference and in I pictures, both transports, checked launches with no sizes given.  Nothing
    may fault; streams left alone decode to the oracle's planes; a stream that was hit is rejected or decodes SOMETHING; the
    batch decodes clean pictures afterwards."""
    import simlib
    n = int(rng.choice([1, 2, 5, 8]))
    inter, events = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    b = h263mi.Batch(n, w, h, pipeline_post=bool(rng.integers(0, 2)))
    what = ("garbage fields", w, h, n, inter, events, seed)

    def picture(make):
        recs, at, base = [], 0, []
        for s_ in range(n):
            m, c = make(s_)
            m = simlib.pad_records(m, w, h)
            bi = np.zeros(len(c), bool)
            for r in m:
                if int(r["mb_type"]) in (3, 4):
                    k = int(r["coeff_index"])
                    bi[k:k + bin(int(r["cbp"])).count("1")] = True
            recs.append((m, c, bi))
            base.append(at)
            at += len(c)
        mbs = np.concatenate([r[0] for r in recs])
        co = np.concatenate([r[1] for r in recs]) if at else np.zeros((0, 64), np.int16)
        first, ev = h263mi.events_from_dense(co, np.concatenate([r[2] for r in recs]) if at else None)
        return {"mbs": mbs, "co": co if at else np.zeros((1, 64), np.int16), "first": first,
                "ev": ev if len(ev) else np.zeros(4, np.uint32), "base": base, "at": at}

    seeds = [int(v) for v in rng.integers(0, 1 << 30, 2 * n)]
    key = picture(lambda s_: recgen.intra_picture(w, h, seed=seeds[s_], max_level=int(rng.choice([40, 1023]))))
    key_out = []
    for s_ in range(n):
        lo, hi = key["base"][s_], (key["base"][s_ + 1] if s_ + 1 < n else key["at"])
        per = len(key["mbs"]) // n
        m = key["mbs"][s_ * per:(s_ + 1) * per]                  # (block indices are relative to the stream's base)
        rc, out = orc.decode_picture(w, h, m, key["co"][lo:hi] if hi > lo else np.zeros((0, 64), np.int16), None)
        assert rc == 0
        key_out.append(out)
    target, want = key, key_out
    if inter:
        target = picture(lambda s_: recgen.inter_picture(w, h, seed=seeds[n + s_], mv_range=32, p_4v=0.3, p_intra=0.15, p_coded=0.5,
                                                          quant=0, max_level=60, sparse_low=False))
        want = []
        per = len(target["mbs"]) // n
        for s_ in range(n):
            lo, hi = target["base"][s_], (target["base"][s_ + 1] if s_ + 1 < n else target["at"])
            rc, out = orc.decode_picture(w, h, target["mbs"][s_ * per:(s_ + 1) * per],
                                         target["co"][lo:hi] if hi > lo else np.zeros((0, 64), np.int16), key_out[s_])
            assert rc == 0
            want.append(out)

    def submit(pic, ptype, mbs=None, co=None, ev=None):
        d = [_dev(pic["mbs"] if mbs is None else mbs), _dev(np.array(pic["base"], np.uint64))]
        if events:
            d += [_dev(pic["first"]), _dev(pic["ev"] if ev is None else ev)]
            b.decode_events(ptype, d[0].ptr, d[2].ptr, d[3].ptr, d[1].ptr)
        else:
            d += [_dev(pic["co"] if co is None else co)]
            b.decode(ptype, d[0].ptr, d[2].ptr, d[1].ptr)
        rcs = b.sync_streams()
        for x in d:
            x.free()
        return rcs

    def clean():
        if any(submit(key, h263mi.PICTURE_I)):
            raise FuzzMismatch("clean key picture rejected: %r" % (what,))

    def check(rcs, hit, attempt):
        for s_ in range(n):
            if s_ in hit:
                if rcs[s_] not in (0, h263mi.ERR_INVALID_ARGUMENT, h263mi.ERR_UNCODED_IFRAME_BLOCKS):
                    raise FuzzMismatch("hit stream %d: verdict %d: %r" % (s_, rcs[s_], what))
                continue
            if rcs[s_] != 0:
                raise FuzzMismatch("untouched stream %d rejected (%d): %r attempt %d" % (s_, rcs[s_], what, attempt))
            for g, e, name in zip(b.copy_yuv(s_), want[s_], "Y Cb Cr".split()):
                if not (np.asarray(g) == e).all():
                    raise FuzzMismatch("untouched stream %d differs: %r attempt %d %s" % (s_, what, attempt, name))

    ptype = h263mi.PICTURE_P if inter else h263mi.PICTURE_I
    per = len(target["mbs"]) // n
    for attempt in range(int(rng.integers(2, 6))):
        clean()
        if attempt == 0:
            check(submit(target, ptype), set(), -1)              # the picture as it is: every stream decodes
            clean()
        hit = set(int(v) for v in rng.choice(n, size=int(rng.integers(1, n + 1)), replace=False))
        m2, e2, c2 = target["mbs"].copy(), target["ev"].copy(), target["co"].copy()
        first, base, at = target["first"], target["base"], target["at"]
        for s_ in hit:
            kind = int(rng.integers(0, 7))
            k = s_ * per + rng.integers(0, per, size=max(1, per // 3))
            lo, hi = base[s_], (base[s_ + 1] if s_ + 1 < n else at)
            if kind == 0:
                m2["mb_type"][k] = rng.integers(6, 256, size=len(k))
            elif kind == 1:
                m2["quant"][k] = rng.choice([0, 32, 127, 128, 255], size=len(k))
            elif kind == 2:                      # (more coded blocks than the macroblock has: into the next one's, or past the pool)
                m2["cbp"][k] = rng.integers(0, 256, size=len(k))
                m2["kill"][k] = rng.integers(0, 256, size=len(k))
            elif kind == 3:
                m2["mv"][k] = rng.choice([-32768, -32767, -4097, -1025, 1024, 4096, 32766, 32767], size=(len(k), 4, 2))
            elif kind == 4:
                m2["intradc"][k] = rng.choice([0, 128, 255], size=(len(k), 6))
                m2["reserved"][k] = 255
            elif events and first[hi] > first[lo]:
                j = rng.integers(int(first[lo]), int(first[hi]), size=16)
                e2[j] = rng.integers(0, 1 << 32, size=16, dtype=np.uint64).astype(np.uint32)
            elif hi > lo:
                j = rng.integers(lo, hi, size=4)
                c2[j] = rng.integers(-32768, 32768, size=(4, 64))
        check(submit(target, ptype, m2, c2, e2), hit, attempt)
        n_pic += 2 * n
        n_px += 2 * n * w * h
    clean()
    b.close()
    return n_pic, n_px


def fuzz_mixed(rng, seed, n_pic, n_px):
    """A set of streams of different (and changing) picture sizes behind one call (h263mi_mixed): per call every stream
    does one of -- nothing, a P picture, a key frame of its size, a key frame of ANOTHER size (the stream moves), a P picture
    of another size (refused: PICTURE_FORMAT_INVALID, state kept), a reset -- and every plane and every RGBA picture is the
    oracle's; the number of size classes stays within two per stream (the one it is in and the one it left)."""
    n = int(rng.integers(2, 8))
    pipeline = bool(rng.integers(0, 2))
    strength = int(rng.integers(0, 13))
    from_header = bool(rng.integers(0, 2))       # (ABI 7) every picture rendered with what its own header asks for
    palette = [(176, 144), (352, 288), (96, 80), (128, 96), (int(rng.integers(17, 200)), int(rng.integers(17, 150))),
               (4 * int(rng.integers(5, 60)), 4 * int(rng.integers(5, 40)))]
    m = h263mi.MixedBatch(n, pipeline_post=pipeline)
    refs, size = [None] * n, [None] * n
    calls = int(rng.integers(3, 8))
    q = int(rng.integers(2, 20))
    rgba = [[h263mi.DeviceBuffer(max(w * h for w, h in palette) * 4) for _ in range(n)] for _ in range(calls)]
    want = []
    what = (n, pipeline, strength, seed)
    for c in range(calls):
        datas, expect = [], []
        for s_ in range(n):
            act = "key" if size[s_] is None and rng.random() < 0.8 else str(rng.choice(["idle", "p", "p", "p", "key", "move", "badp", "reset"]))
            if act == "reset":
                m.reset_stream(s_)
                refs[s_], size[s_] = None, None
                act = "idle"
            if act == "idle":
                datas.append(None)
                expect.append(0)
                continue
            wh = size[s_] if act in ("p", "key") and size[s_] is not None else palette[int(rng.integers(0, len(palette)))]
            if act == "badp" and (size[s_] is None or wh == size[s_]):
                act = "p" if size[s_] is not None else "key"
                wh = size[s_] if size[s_] is not None else wh
            sd = int(rng.integers(0, 1 << 30))
            intra = act in ("key", "move")
            if intra:
                mbs, co = recgen.intra_picture(wh[0], wh[1], seed=sd, max_level=int(rng.choice([40, 127, 1023])))
            else:
                mbs, co = recgen.inter_picture(wh[0], wh[1], seed=sd, mv_range=32, p_4v=float(rng.choice([0.0, 0.3])), p_intra=0.05,
                                               p_coded=float(rng.choice([0.1, 0.5])), quant=q, max_level=int(rng.choice([60, 1023])))
            mbs = make_codable(mbs, q, sd, 0 if intra else 1)
            flag = int(rng.integers(0, 2))
            datas.append(enc.encode_picture(wh[0], wh[1], 0 if intra else 1, q, mbs, co, temporal_reference=c, deblock_flag=flag))
            if not intra and refs[s_] is None:
                expect.append(h263mi.ERR_UNCODED_IFRAME_BLOCKS)
            elif not intra and wh != size[s_]:
                expect.append(h263mi.ERR_PICTURE_FORMAT_INVALID)
            else:
                rc, refs[s_] = orc.decode_picture(wh[0], wh[1], mbs, co, None if intra else refs[s_])
                assert rc == 0
                size[s_] = wh
                expect.append(0)
                cw = (wh[0] + 1) // 2
                st_ = (Q2S[q] if flag else 0) if from_header else strength
                planes = refs[s_] if st_ == 0 else tuple(orc.deblock(p, pw, st_) for p, pw in zip(refs[s_], (wh[0], cw, cw)))
                want.append((c, s_, wh, orc.yuv420_to_rgba(*planes, wh[0])))
                n_pic += 1
                n_px += wh[0] * wh[1]
        used, rcs, descs = m.decode_next_pictures(datas, n_threads=int(rng.integers(1, 4)),
                                                  strength=h263mi.STRENGTH_FROM_HEADER if from_header else strength, rgba=rgba[c])
        if list(rcs) != expect:
            raise FuzzMismatch("mixed set: call %d return codes %s, expected %s: %r" % (c, list(rcs), expect, what))
        if m.size_classes() > 2 * n:
            raise FuzzMismatch("mixed set: %d size classes for %d streams: %r" % (m.size_classes(), n, what))
        if rng.random() < 0.5 and any(m.sync()):
            raise FuzzMismatch("mixed set: device verdict: %r" % (what,))
    if any(m.sync()):
        raise FuzzMismatch("mixed set: device verdict at the end: %r" % (what,))
    for s_ in range(n):
        if refs[s_] is None:
            if m.stream_size(s_) != (0, 0):
                raise FuzzMismatch("mixed set: stream %d should have no picture: %r" % (s_, what))
            continue
        if m.stream_size(s_) != size[s_]:
            raise FuzzMismatch("mixed set: stream %d size %s, expected %s: %r" % (s_, m.stream_size(s_), size[s_], what))
        for g, e, name in zip(m.copy_yuv(s_), refs[s_], "Y Cb Cr".split()):
            if not (np.asarray(g) == e.ravel()).all():
                raise FuzzMismatch("mixed set: stream %d %s: %r" % (s_, name, what))
    for c, s_, wh, expect_rgba in want:
        if not (rgba[c][s_].download(wh[0] * wh[1] * 4) == expect_rgba.ravel()).all():
            raise FuzzMismatch("mixed set: RGBA of call %d stream %d %s: %r" % (c, s_, wh, what))
    m.close()
    for row in rgba:
        for d in row:
            d.free()
    return n_pic, n_px


def run(budget=60.0, seed=1, verbose=True):
    """random pictures for `budget` seconds; returns (pictures, pixels); raises FuzzMismatch on the first difference"""
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    n_pic = n_px = 0
    t_note = time.time() + 60.0
    while time.time() < t_end:
        if verbose and time.time() >= t_note:                    # (a long run must not look hung to the GPU box's watchdog)
            print("  ... %d pictures so far" % n_pic, flush=True)
            t_note = time.time() + 60.0
        # (the last entries: sizes with several interior tiles of k_post per row and column, widths that are / are not
        # multiples of 4 and 8 -- the wrap of the left picture columns into the last tile and the floor / truncation
        # regions of the deblocking filter depend on them)
        w = int(rng.choice([rng.integers(1, 64), rng.integers(1, 420), 16 * rng.integers(1, 30), 176, 352,
                            rng.integers(260, 800), 4 * rng.integers(65, 200)]))
        h = int(rng.choice([rng.integers(1, 64), rng.integers(1, 300), 16 * rng.integers(1, 20), 144, 288,
                            rng.integers(64, 420), 4 * rng.integers(16, 100)]))
        if rng.random() < (1.0 if os.environ.get("H263MI_FUZZ_MIXED_ONLY") else 0.08):      # (the switch: a run of mixed sets only)
            n_pic, n_px = fuzz_mixed(rng, seed, n_pic, n_px)
            continue
        if rng.random() < (1.0 if os.environ.get("H263MI_FUZZ_HOSTILE_ONLY") else 0.07):      # (the switch: hostile arrays only)
            n_pic, n_px = fuzz_hostile(rng, min(max(w, 17), 300), min(max(h, 17), 200), seed, n_pic, n_px)
            continue
        if rng.random() < (1.0 if os.environ.get("H263MI_FUZZ_GARBAGE_ONLY") else 0.05):      # (the switch: garbage fields only)
            n_pic, n_px = fuzz_garbage_fields(rng, min(max(w, 17), 300), min(max(h, 17), 200), seed, n_pic, n_px)
            continue
        if rng.random() < 0.2:
            # a BATCH of streams in lock step (h263mi_batch_decode / _decode_events), plain or frame-pipelined: the launch
            # that the bench times, at random sizes, stream counts, transports and filter strengths; RGBA of every
            # picture of every stream, planes of a few
            n_pic, n_px = fuzz_batch(rng, min(w, 400), min(h, 300), seed, n_pic, n_px)
            continue
        if rng.random() < 0.35:
            # through the bitstream: records -> test encoder -> h263mi_decode_next_picture (host parser, sparse transport)
            standard = bool(rng.integers(0, 2))
            if standard:
                w, h = [(128, 96), (176, 144), (352, 288), (4 * int(rng.integers(4, 60)), 4 * int(rng.integers(4, 40)))][int(rng.integers(0, 4))]
            else:
                w, h = min(w, 255), min(h, 255)
            st = h263mi.H263State(0 if standard else h263mi.SORENSON_SPARK_BITSTREAM)
            std = None
            if standard:
                std = {} if (w, h) in enc.STD_FORMATS and rng.random() < 0.5 else {"plus": True}
            ref = None
            for f in range(int(rng.integers(1, 4))):
                s = int(rng.integers(0, 1 << 30))
                q = int(rng.integers(1, 32))
                lvl = 127 if standard else int(rng.choice([60, 1023]))
                if f == 0:
                    mbs, co = recgen.intra_picture(w, h, seed=s, max_level=lvl)
                    mbs = make_codable(mbs, q, s, 0)
                    pt = 0
                else:
                    mbs, co = recgen.inter_picture(w, h, seed=s, mv_range=32, p_coded=float(rng.choice([0.1, 0.5])),
                                                   p_4v=float(rng.choice([0.0, 0.4])), p_intra=0.1, quant=q, max_level=lvl,
                                                   sparse_low=bool(rng.integers(0, 2)))
                    mbs = make_codable(mbs, q, s, 1)
                    pt = 1
                try:
                    data = enc.encode_picture(w, h, pt, q, mbs, co, temporal_reference=f, standard=std)
                except (AssertionError, KeyError) as e:      # a record mix the little test encoder cannot express
                    if verbose:
                        print("encoder skipped:", w, h, pt, q, s, lvl, standard, repr(e)[:80])
                    break
                st.decode_next_picture(data)
                rc, ref = orc.decode_picture(w, h, mbs, co, ref if pt else None)
                assert rc == 0
                for g, e, name in zip(st.get_last_picture().as_yuv(), ref, "Y Cb Cr".split()):
                    if not (np.asarray(g) == e).all():
                        raise FuzzMismatch("bitstream path: %r" % ((w, h, f, s, name, standard, seed),))
                n_pic += 1
                n_px += w * h
            st.close()
            continue
        st = h263mi.H263State()
        ref = None
        for f in range(int(rng.integers(1, 5))):
            s = int(rng.integers(0, 1 << 30))
            if f == 0 or rng.random() < 0.15:
                mbs, co = recgen.intra_picture(w, h, seed=s, max_level=int(rng.choice([3, 40, 127, 1023])))
                pt = h263mi.PICTURE_I
                want_ref = None
            else:
                mbs, co = recgen.inter_picture(w, h, seed=s, mv_range=int(rng.choice([2, 32, 70, 300])),
                                               p_coded=float(rng.choice([0.05, 0.3, 0.9])), p_4v=float(rng.choice([0.0, 0.3, 1.0])),
                                               p_intra=float(rng.choice([0.0, 0.2])), quant=int(rng.choice([0, 1, 10, 31])),
                                               max_level=int(rng.choice([3, 60, 1023])), sparse_low=bool(rng.integers(0, 2)))
                pt = h263mi.PICTURE_P
                want_ref = ref
            if want_ref is not None and rng.random() < 0.2 and len(mbs) > 2:     # a picture that ends early (padded as inter)
                mbs = mbs[:int(rng.integers(1, len(mbs)))]
            st.submit_picture(w, h, mbs, co, pt, temporal_reference=f)
            rc, ref = orc.decode_picture(w, h, mbs, co, want_ref)
            assert rc == 0
            got = st.get_last_picture().as_yuv()
            for g, e, name in zip(got, ref, "Y Cb Cr".split()):
                if not (np.asarray(g) == e).all():
                    raise FuzzMismatch("recon: %r" % ((w, h, f, s, name, seed),))
            strength = int(rng.integers(0, 13))
            cw = (w + 1) // 2
            planes = ref if strength == 0 else tuple(orc.deblock(p, pw, strength) for p, pw in zip(ref, (w, cw, cw)))
            if not (st.render_rgba(strength) == orc.yuv420_to_rgba(*planes, w)).all():
                raise FuzzMismatch("rgba: %r" % ((w, h, f, s, strength, seed),))
            n_pic += 1
            n_px += w * h
        st.close()
    if verbose:
        print("fuzz ok: %d pictures, %.1f MP, seed %d, %.0f s" % (n_pic, n_px / 1e6, seed, budget))
    return n_pic, n_px


if __name__ == "__main__":
    try:
        run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    except FuzzMismatch as e:
        print("MISMATCH", e)
        sys.exit(1)
