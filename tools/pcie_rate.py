#!/usr/bin/env python3
"""PCIe-inclusive rate of the single-stream path (never the headline `value`): host records through
h263mi_submit_picture (pinned staging + async H2D + k_recon) and render_rgba with D2H of the frame, one 1080p
stream."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h263-rs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np

import h263mi

W, H = 1920, 1080
st = h263mi.H263State()
frames = []
for f in range(8):
    kind = h263mi.SYNTH_I_MIXED if f == 0 else h263mi.SYNTH_P
    frames.append((h263mi.PICTURE_I if f == 0 else h263mi.PICTURE_P,) + h263mi.synth_picture_host(kind, W, H, 0, f))


def run(n, rgba):
    t0 = time.perf_counter()
    for i in range(n):
        pt, mbs, co = frames[i % 8]
        st.submit_picture(W, H, mbs, co, pt)
        if rgba:
            st.render_rgba(5)
    if not rgba:
        st.get_last_picture()
    return (time.perf_counter() - t0) / n


run(8, False)
dt = run(64, False)
rec_mb = np.mean([m.nbytes + c.nbytes for _, m, c in frames]) / 1e6
print("submit_picture (host records, %.2f MB/picture mean): %.3f ms/picture = %.0f pictures/s = %.1f MP/s, H2D %.1f GB/s"
      % (rec_mb, dt * 1e3, 1 / dt, W * H / 1e6 / dt, rec_mb / 1e3 / dt))
# the same pictures with sparse coefficient transport (h263mi_submit_picture_events)
ev_frames = []
for pt, mbs, co in frames:
    intra = np.zeros(len(co.reshape(-1, 64)), bool)
    if pt == h263mi.PICTURE_I:
        intra[:] = True
    first, ev = h263mi.events_from_dense(co, intra)
    ev_frames.append((pt, mbs, first, ev))


def run_events(n):
    t0 = time.perf_counter()
    for i in range(n):
        pt, mbs, first, ev = ev_frames[i % 8]
        st.submit_picture_events(W, H, mbs, first, ev, pt)
    st.get_last_picture()
    return (time.perf_counter() - t0) / n


run_events(8)
dt = run_events(64)
ev_mb = np.mean([m.nbytes + f.nbytes + e.nbytes for _, m, f, e in ev_frames]) / 1e6
print("submit_picture_events (sparse coefficients, %.2f MB/picture mean): %.3f ms/picture = %.0f pictures/s = %.1f MP/s, "
      "H2D %.1f GB/s" % (ev_mb, dt * 1e3, 1 / dt, W * H / 1e6 / dt, ev_mb / 1e3 / dt))
dt = run(32, True)
print("submit_picture + render_rgba(5) + D2H of 8.3 MB RGBA: %.3f ms/picture = %.0f pictures/s = %.1f MP/s"
      % (dt * 1e3, 1 / dt, W * H / 1e6 / dt))
dense = h263mi.synth_picture_host(h263mi.SYNTH_I_DENSE, W, H, 0, 0)
t0 = time.perf_counter()
for i in range(32):
    st.submit_picture(W, H, dense[0], dense[1], h263mi.PICTURE_I)
st.get_last_picture()
dt = (time.perf_counter() - t0) / 32
print("dense I pictures (%.2f MB of records each): %.3f ms/picture = %.0f pictures/s, H2D %.1f GB/s"
      % ((dense[0].nbytes + dense[1].nbytes) / 1e6, dt * 1e3, 1 / dt, (dense[0].nbytes + dense[1].nbytes) / 1e9 / dt))

# ---- many streams: h263mi_batch_submit_host, 32 streams x (I + 7 P) from per-stream host record arrays -------------
import ctypes as C   # noqa: E402

n = 32
b = h263mi.Batch(n, W, H)
per_frame = []
for f in range(8):
    kind = h263mi.SYNTH_I_MIXED if f == 0 else h263mi.SYNTH_P
    recs = [h263mi.synth_picture_host(kind, W, H, s, f) for s in range(4)]          # 4 distinct streams, reused
    mbs = [recs[s % 4][0] for s in range(n)]
    cos = [recs[s % 4][1].reshape(-1, 64) for s in range(n)]
    pm = (C.c_void_p * n)(*[m.ctypes.data for m in mbs])
    pc = (C.c_void_p * n)(*[c.ctypes.data for c in cos])
    nm = (C.c_uint32 * n)(*[len(m) for m in mbs])
    nc = (C.c_uint32 * n)(*[len(c) for c in cos])
    nbytes = sum(m.nbytes + c.nbytes for m, c in zip(mbs, cos))
    per_frame.append((h263mi.PICTURE_I if f == 0 else h263mi.PICTURE_P, pm, nm, pc, nc, nbytes, (mbs, cos)))
L = h263mi.lib()


def run_batch(steps):
    t0 = time.perf_counter()
    for i in range(steps):
        pt, pm, nm, pc, nc, _, _ = per_frame[i % 8]
        rc = L.h263mi_batch_submit_host(b._h, pt, pm, nm, pc, nc)
        assert rc == 0, rc
    b.sync()
    return (time.perf_counter() - t0) / steps


run_batch(8)
dt = run_batch(32)
mb = np.mean([p[5] for p in per_frame]) / 1e6
print("batch_submit_host, %d streams (%.1f MB of records per step): %.3f ms/step = %.0f pictures/s = %.1f MP/s, "
      "pack + H2D %.1f GB/s" % (n, mb, dt * 1e3, n / dt, n * W * H / 1e6 / dt, mb / 1e3 / dt))

# ---- the same with sparse coefficient transport -------------------------------------------------------------------
L.h263mi_batch_submit_host_events.argtypes = [C.c_void_p, C.c_uint8] + [C.c_void_p] * 6
per_frame_ev = []
for f in range(8):
    pt, _, _, _, _, _, (mbs, cos) = per_frame[f]
    fe = []
    for s4 in range(4):
        intra = np.zeros(len(cos[s4]), bool)
        if pt == h263mi.PICTURE_I:
            intra[:] = True
        fe.append(h263mi.events_from_dense(cos[s4], intra))
    firsts = [fe[s % 4][0] for s in range(n)]
    evs = [fe[s % 4][1] for s in range(n)]
    pm = (C.c_void_p * n)(*[m.ctypes.data for m in mbs])
    pf = (C.c_void_p * n)(*[x.ctypes.data for x in firsts])
    pe = (C.c_void_p * n)(*[x.ctypes.data for x in evs])
    nm = (C.c_uint32 * n)(*[len(m) for m in mbs])
    nb = (C.c_uint32 * n)(*[len(x) - 1 for x in firsts])
    ne = (C.c_uint32 * n)(*[len(x) for x in evs])
    nbytes = sum(m.nbytes for m in mbs) + sum(x.nbytes for x in firsts) + sum(x.nbytes for x in evs)
    per_frame_ev.append((pt, pm, nm, pf, nb, pe, ne, nbytes, (mbs, firsts, evs)))


def run_batch_ev(steps):
    t0 = time.perf_counter()
    for i in range(steps):
        pt, pm, nm, pf, nb, pe, ne, _, _ = per_frame_ev[i % 8]
        rc = L.h263mi_batch_submit_host_events(b._h, pt, pm, nm, pf, nb, pe, ne)
        assert rc == 0, rc
    b.sync()
    return (time.perf_counter() - t0) / steps


run_batch_ev(8)
dt = run_batch_ev(32)
mb = np.mean([p[7] for p in per_frame_ev]) / 1e6
print("batch_submit_host_events, %d streams (%.1f MB of records + events per step): %.3f ms/step = %.0f pictures/s = "
      "%.1f MP/s, pack + H2D %.1f GB/s" % (n, mb, dt * 1e3, n / dt, n * W * H / 1e6 / dt, mb / 1e3 / dt))
