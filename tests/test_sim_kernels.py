"""Kernel phase logic (h263-rs_amd/csrc/*_kernel.inl) run on the CPU by tests/sim against the
oracle: same tiling, LDS indexing, clamping and edge handling as the GPU build, without a
GPU.  Bit-exact for every plane and RGBA byte.  (The GPU parity tests are in test_gpu_*.py.)
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import recgen
import simlib
from oracle import oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SIZES = [(16, 16), (48, 32), (100, 60), (5, 4), (1, 1), (33, 17), (176, 144), (320, 240), (136, 40),
         (128, 32), (256, 48)]   # the last two: width == row pitch (no padding columns), as at 1920


@pytest.mark.parametrize("w,h", SIZES)
def test_recon_intra(w, h):
    mbs, coeffs = recgen.intra_picture(w, h, seed=w * 31 + h)
    rc, want = orc.decode_picture(w, h, mbs, coeffs, None)
    st, got = simlib.recon(w, h, mbs, coeffs, None)
    assert rc == 0 and st == 0
    for g, e, name in zip(got, want, "Y Cb Cr".split()):
        assert (g == e).all(), (name, np.flatnonzero(g != e)[:10])


@pytest.mark.parametrize("w,h", SIZES)
def test_recon_inter(w, h):
    ref = recgen.random_planes(w, h, 77)
    mbs, coeffs = recgen.inter_picture(w, h, seed=w * 13 + h, mv_range=70, p_4v=0.3, p_intra=0.15, quant=0)
    rc, want = orc.decode_picture(w, h, mbs, coeffs, ref)
    st, got = simlib.recon(w, h, mbs, coeffs, ref)
    assert rc == 0 and st == 0
    for g, e, name in zip(got, want, "Y Cb Cr".split()):
        assert (g == e).all(), (name, np.flatnonzero(g != e)[:10])


@pytest.mark.parametrize("w,h,mv_range", [(176, 144, 600), (100, 60, 200), (16, 16, 1100), (48, 32, 64), (320, 240, 1100)])
def test_recon_far_vectors_and_large_levels(w, h, mv_range):
    """vectors far outside the picture (Annex D ranges and beyond: every tap clamps to an edge pixel), 11-bit
    Sorenson levels at quantiser 31"""
    ref = recgen.random_planes(w, h, 5)
    mbs, coeffs = recgen.inter_picture(w, h, seed=mv_range + w, mv_range=mv_range, p_4v=0.4, p_intra=0.1, p_coded=0.5,
                                       quant=31, max_level=1023, sparse_low=False)
    rc, want = orc.decode_picture(w, h, mbs, coeffs, ref)
    st, got = simlib.recon(w, h, mbs, coeffs, ref)
    assert rc == 0 and st == 0
    for g, e, name in zip(got, want, "Y Cb Cr".split()):
        assert (g == e).all(), (name, np.flatnonzero(g != e)[:10])


def test_recon_short_picture_is_padded_and_errors_without_reference():
    w, h = 64, 48
    ref = recgen.random_planes(w, h, 3)
    mbs, coeffs = recgen.inter_picture(w, h, seed=5, n_mbs=7)
    rc, want = orc.decode_picture(w, h, mbs, coeffs, ref)
    st, got = simlib.recon(w, h, mbs, coeffs, ref)
    assert rc == 0 and st == 0
    for g, e in zip(got, want):
        assert (g == e).all()
    st, _ = simlib.recon(w, h, mbs, coeffs, None)
    assert st & 1                                   # STATUS_INTER_WITHOUT_REFERENCE (gather.rs:149)


def test_recon_kill_and_dc_special_cases():
    w, h = 32, 16
    mbs, coeffs = recgen.intra_picture(w, h, 9, classes=("full_sparse", "dc", "vert", "horiz"))
    mbs[0]["kill"] = 0b100101
    mbs[1]["intradc"][:] = 255                      # level 1024
    rc, want = orc.decode_picture(w, h, mbs, coeffs, None)
    st, got = simlib.recon(w, h, mbs, coeffs, None)
    assert rc == 0 and st == 0
    for g, e in zip(got, want):
        assert (g == e).all()


def test_recon_all_dc_values_through_inter_dc_class():
    # inter block whose only coefficient sits at zigzag 0 -> Dc(dequantised) (appendix B.6):
    # sweep levels x quants over the whole dequantised range, prediction 0 and 255
    w, h = 16 * 8, 16 * 8
    for pred in (0, 255):
        ref = tuple(np.full(n, pred, np.uint8) for n in (w * h, (w // 2) * (h // 2), (w // 2) * (h // 2)))
        rng = np.random.default_rng(pred)
        mbs = np.zeros(64, orc.MB_RECORD_DTYPE)
        mbs["mb_type"] = 0
        mbs["quant"] = rng.integers(1, 32, 64)
        mbs["cbp"] = 0x3F
        mbs["coeff_index"] = np.arange(64) * 6
        coeffs = np.zeros((64 * 6, 64), np.int16)
        coeffs[:, 0] = rng.integers(-127, 128, 64 * 6)
        rc, want = orc.decode_picture(w, h, mbs, coeffs, ref)
        st, got = simlib.recon(w, h, mbs, coeffs, ref)
        assert rc == 0 and st == 0
        for g, e in zip(got, want):
            assert (g == e).all()


@pytest.mark.parametrize("w,h", [(11, 17), (16, 16), (100, 60), (9, 9), (10, 10), (8, 2), (1, 1), (200, 37),
                                 (136, 40), (960, 20)])
@pytest.mark.parametrize("strength", [1, 5, 12])
def test_post_deblock_planes(w, h, strength):
    planes = recgen.random_planes(w, h, w + h + strength)
    _, got = simlib.post(w, h, planes, strength, want_rgba=False)
    cw, ch = (w + 1) // 2, (h + 1) // 2
    for g, p, pw in zip(got, planes, (w, cw, cw)):
        assert (g == orc.deblock(p, pw, strength)).all()


def test_post_deblock_reference_image_luma_only():
    img = json.load(open(os.path.join(GOLD, "deblock_reference_tests.json")))["image"]
    data = np.array(img["data"], np.uint8)
    for s in ("4", "8", "12"):
        _, got = simlib.post(11, 17, (data, None, None), int(s), want_rgba=False, luma_only=True)
        assert got[0].tolist() == img["expected"][s]


@pytest.mark.parametrize("w,h", [(1, 1), (2, 2), (3, 2), (3, 3), (4, 4), (5, 4), (100, 60), (133, 35), (176, 144)])
@pytest.mark.parametrize("strength", [0, 7])
def test_post_rgba(w, h, strength):
    planes = recgen.random_planes(w, h, w * h)
    rgba, _ = simlib.post(w, h, planes, strength, want_planes=False)
    cw = (w + 1) // 2
    if strength:
        planes = tuple(orc.deblock(p, pw, strength) for p, pw in zip(planes, (w, cw, cw)))
    assert (rgba == orc.yuv420_to_rgba(*planes, w)).all()


@pytest.mark.parametrize("w,h", [(524, 300), (516, 292), (260, 68), (132, 36), (392, 100)])
@pytest.mark.parametrize("strength", [0, 3, 11])
def test_post_rgba_interior_and_edge_tiles(w, h, strength):
    """sizes at which each term of post_tile_is_interior decides for some tile (widths / heights that are not multiples
    of 8, chroma limits that end the interior before the luma ones, pictures one tile wide or high); low-contrast planes,
    so that the filters act on most edges; RGBA only (interior instantiations) and RGBA + planes (general form)"""
    rng = np.random.default_rng(w * 5 + h + strength)
    cw, ch = (w + 1) // 2, (h + 1) // 2
    planes = tuple(np.clip(rng.normal(128, 7, n), 0, 255).astype(np.uint8) for n in (w * h, cw * ch, cw * ch))
    want = planes if strength == 0 else tuple(orc.deblock(p, pw, strength) for p, pw in zip(planes, (w, cw, cw)))
    rgba, _ = simlib.post(w, h, planes, strength, want_planes=False)
    assert (rgba == orc.yuv420_to_rgba(*want, w)).all()
    rgba, got = simlib.post(w, h, planes, strength)
    assert (rgba == orc.yuv420_to_rgba(*want, w)).all()
    for g, e in zip(got, want):
        assert (g == e).all()


def test_post_rgba_reference_pictures():
    bt = json.load(open(os.path.join(GOLD, "bt601_reference_tests.json")))
    for p in bt["pictures"]:
        if not p["y"]:
            continue
        w = p["y_width"]
        h = len(p["y"]) // w
        rgba, _ = simlib.post(w, h, (np.array(p["y"], np.uint8), np.array(p["cb"], np.uint8),
                                     np.array(p["cr"], np.uint8)), 0, want_planes=False)
        assert rgba.tolist() == p["rgba"]


def test_synth_records_decode_identically_in_sim_and_oracle():
    w, h = 176, 144
    ref = None
    for frame, kind in enumerate((1, 2, 2)):                 # mixed I, then two P pictures
        mbs, coeffs = simlib.synth_picture(kind, w, h, 3, frame)
        rc, want = orc.decode_picture(w, h, mbs, coeffs, ref)
        st, got = simlib.recon(w, h, mbs, coeffs, ref)
        assert rc == 0 and st == 0
        for g, e in zip(got, want):
            assert (g == e).all()
        ref = want
    mbs, coeffs = simlib.synth_picture(0, 48, 32, 0, 0)      # dense I
    assert (mbs["cbp"] == 0x3F).all() and coeffs.shape[0] == 6 * 6
    rc, want = orc.decode_picture(48, 32, mbs, coeffs, None)
    st, got = simlib.recon(48, 32, mbs, coeffs, None)
    assert all((g == e).all() for g, e in zip(got, want))


def test_asan_build_runs_clean():
    """The same phases under AddressSanitizer + UBSan (child process, libasan preloaded)."""
    import subprocess
    import sys
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    code = (
        "import sys; sys.path[:0]=[%r,%r]\n"
        "import numpy as np, recgen, simlib\n"
        "for (w,h) in [(5,4),(33,17),(100,60),(136,40)]:\n"
        "    ref=recgen.random_planes(w,h,1)\n"
        "    mbs,co=recgen.inter_picture(w,h,seed=2,mv_range=90,p_4v=0.3,p_intra=0.2,quant=0)\n"
        "    simlib.recon(w,h,mbs,co,ref,asan=True)\n"
        "    mbs,co=recgen.intra_picture(w,h,seed=3)\n"
        "    simlib.recon(w,h,mbs,co,None,asan=True)\n"
        "    simlib.post(w,h,ref,6,asan=True)\n"
        "    simlib.post(w,h,ref,0,asan=True)\n"
        "print('asan-ok')\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "asan-ok" in out.stdout, out.stderr[-3000:]


@pytest.mark.parametrize("floor_sem", [0, 1])
def test_quartet_arithmetic_matches_the_oracle_for_every_difference(floor_sem):
    """deblock_quartet (medians, biased shifts) vs the reference-shaped process functions of the oracle: every
    (A - D, C - B), at both ends of the byte range, strengths 1..12 -- 12.5 M quartets per division semantics."""
    import ctypes as C
    from oracle import oracle as orc_mod
    L = simlib.lib()
    ref = getattr(orc_mod.lib(), "orc_deblock_process_simd_lane" if floor_sem else "orc_deblock_process_scalar")
    first = (C.c_int * 5)()
    L.sim_quartet_sweep.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.sim_quartet_sweep.restype = C.c_int
    bad = L.sim_quartet_sweep(C.cast(ref, C.c_void_p), floor_sem, first)
    assert bad == 0, (bad, list(first))


def test_kernel_phases_on_the_mutation_sensitive_blocks():
    """the committed blocks on which a fused or re-associated IDCT changes a pixel (tests/golden/
    idct_sensitive_blocks.json) through the kernel phase functions on the CPU: same pixels as the reference arithmetic"""
    import mutation_probe
    blocks = mutation_probe.fixture()
    w, h, intra, inter, coeffs = mutation_probe.records(blocks)
    st, flat = simlib.recon(w, h, intra, np.zeros((0, 64), np.int16))
    assert st == 0 and all((p == 128).all() for p in flat)
    st, got = simlib.recon(w, h, inter, coeffs, ref=flat)
    assert st == 0
    want = np.full((h, w), 128, np.int32)
    for k, b in enumerate(blocks):
        px, py = (k % mutation_probe.MB_COLS) * 16, (k // mutation_probe.MB_COLS) * 16
        want[py:py + 8, px:px + 8] += np.array(b["residual"], np.int32)
    assert (got[0].reshape(h, w) == np.clip(want, 0, 255)).all()
    rc, oracle_planes = orc.decode_picture(w, h, inter, coeffs, flat)
    assert rc == 0 and (oracle_planes[0] == got[0]).all()


@pytest.mark.parametrize("w,h", [(176, 144), (100, 60), (48, 32), (16, 16)])
def test_recon_from_events_equals_recon_from_dense_blocks(w, h):
    """sparse coefficient transport consumed by the reconstruction wave itself (recon_kernel.inl: coeff_row_from_events):
    same pictures as from dense blocks -- intra pictures with every block class (dense blocks = 63 events), P pictures
    with sparse residuals, mixed intra / inter"""
    mbs, coeffs = recgen.intra_picture(w, h, seed=w + 7 * h)
    rc, want = orc.decode_picture(w, h, mbs, coeffs, None)
    st, got = simlib.recon(w, h, mbs, coeffs, None, events=True)
    assert rc == 0 and st == 0
    for g, e, name in zip(got, want, "Y Cb Cr".split()):
        assert (g == e).all(), ("intra", name, np.flatnonzero(g != e)[:10])
    ref = want
    for seed in (1, 2):
        mbs, coeffs = recgen.inter_picture(w, h, seed=seed * 91 + w, mv_range=50, p_4v=0.3, p_intra=0.2, p_coded=0.5, quant=0,
                                           max_level=127, sparse_low=seed == 1)
        rc, want = orc.decode_picture(w, h, mbs, coeffs, ref)
        st, got = simlib.recon(w, h, mbs, coeffs, ref, events=True)
        assert rc == 0 and st == 0
        for g, e, name in zip(got, want, "Y Cb Cr".split()):
            assert (g == e).all(), ("inter", seed, name, np.flatnonzero(g != e)[:10])
        ref = want


@pytest.mark.parametrize("w,h", [(256, 48), (176, 144), (384, 32)])
def test_recon_static_waves_take_the_copy_path(w, h):
    """pictures shaped like real content (tests/recgen.py: realistic_inter_picture): runs of macroblocks that are not
    coded and do not move -- whole waves of them take recon_phase_copy -- beside moving and coded ones, and the all-static
    picture (every wave copies)"""
    ref = recgen.random_planes(w, h, 11)
    for seed in (1, 2, 3):
        mbs, coeffs = recgen.realistic_inter_picture(w, h, seed, p_skip=0.8 if seed == 3 else 0.6)
        rc, want = orc.decode_picture(w, h, mbs, coeffs, ref)
        st, got = simlib.recon(w, h, mbs, coeffs, ref)
        assert rc == 0 and st == 0
        for g, e, name in zip(got, want, "Y Cb Cr".split()):
            assert (g == e).all(), (seed, name, np.flatnonzero(g != e)[:10])
    mbs = np.zeros(((w + 15) // 16) * ((h + 15) // 16), orc.MB_RECORD_DTYPE)
    mbs["quant"] = 5
    st, got = simlib.recon(w, h, mbs, np.zeros((0, 64), np.int16), ref)
    assert st == 0
    for g, e in zip(got, ref):
        assert (g == e).all()


def test_dequantiser_every_level_every_quantiser_is_sixteen_times_the_clamped_value():
    """recon_kernel.inl: dequant_pair_i16 hands on 16 x sign(L) * clamp(q * (2|L| + 1) - (q even), ..) (rle.rs:130-133
    with the clamp to [-2048, 2047]) for every 16-bit LEVEL a record can carry, at every quantiser -- the portable form
    of the device's saturating multiply-add (the device form itself: tests/test_gpu_round3.py and the dequant mutant)."""
    import ctypes as C
    L = simlib.lib()
    L.sim_dequant_pairs.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lv = np.arange(-2047, 2048, dtype=np.int16)
    lv = np.concatenate([lv, np.zeros(1, np.int16)])            # an even count: pairs
    packed = np.ascontiguousarray(lv).view(np.uint32)
    out = np.empty_like(packed)
    for q in range(1, 32):
        L.sim_dequant_pairs(packed.ctypes.data, len(packed), q, out.ctypes.data)
        got = out.view(np.int16).astype(np.int64)
        l64 = lv.astype(np.int64)
        want = np.sign(l64) * (q * (2 * np.abs(l64) + 1) - (1 if q % 2 == 0 else 0))
        want = np.clip(want, -2048, 2047)
        assert (got == 16 * want).all(), q


@pytest.mark.parametrize("w,h,events", [(176, 144, False), (352, 288, True), (100, 60, True), (136, 40, False)])
def test_recon_from_sparse_records_equals_recon_from_dense_records(w, h, events):
    """ReconArgs::mb_group_index (round 5): records for the coded macroblocks only + one index word per group of 8; a
    macroblock without a record is not coded.  Real content -- runs of COD = 1, whole waves of them (the copy path is then
    taken without a record being read), groups with one coded macroblock, ragged right edges -- against the oracle, and a
    picture of nothing but uncoded macroblocks."""
    ref = recgen.random_planes(w, h, 11)
    for seed, p_skip in ((1, 0.7), (2, 0.95), (3, 0.3)):
        mbs, coeffs = recgen.realistic_inter_picture(w, h, seed, p_skip=p_skip, p_coded=0.2)
        rc, want = orc.decode_picture(w, h, mbs, coeffs, ref)
        st, got = simlib.recon(w, h, mbs, coeffs, ref, events=events, sparse_records=True)
        assert rc == 0 and st == 0
        for g, e, name in zip(got, want, "Y Cb Cr".split()):
            assert (g == e).all(), (seed, name, np.flatnonzero(g != e)[:10])
    mbs, coeffs = recgen.realistic_inter_picture(w, h, 4, p_skip=1.0, p_coded=0.0)
    st, got = simlib.recon(w, h, mbs, coeffs, ref, events=events, sparse_records=True)
    assert st == 0 and all((g == e).all() for g, e in zip(got, ref))
    # an inter picture without a reference is still an error (gather.rs:149), also when nothing has a record
    st, _ = simlib.recon(w, h, mbs, coeffs, None, events=events, sparse_records=True)
    assert st & 1
    # intra macroblocks in the mix, a short picture (macroblocks the bitstream does not reach have no record either)
    mbs, coeffs = recgen.inter_picture(w, h, seed=9, mv_range=20, p_4v=0.2, p_intra=0.2, p_coded=0.4, n_mbs=max(1, ((w + 15) // 16) * ((h + 15) // 16) - 5))
    rc, want = orc.decode_picture(w, h, mbs, coeffs, ref)
    st, got = simlib.recon(w, h, mbs, coeffs, ref, events=events, sparse_records=True)
    assert rc == 0 and st == 0 and all((g == e).all() for g, e in zip(got, want))


def _hostile_base_script(root):
    return (
        "import sys; sys.path[:0]=[%r,%r]\n"
        "import numpy as np, ctypes as C, simlib, recgen\n"
        "w,h=64,48\n"
        "mbs,co=recgen.intra_picture(w,h,seed=3,max_level=40)\n"
        "mbs=simlib.pad_records(mbs,w,h); L=simlib.layout(w,h)\n"
        "co=np.ascontiguousarray(co,np.int16).reshape(-1,64)\n"
        "nz=co!=0; nz[:,0]=False\n"
        "first=np.zeros(len(co)+1,np.uint32); np.cumsum(nz.sum(axis=1),out=first[1:])\n"
        "blk,pos=np.nonzero(nz); ev=((co[blk,pos].astype(np.uint16).astype(np.uint32)<<16)|pos.astype(np.uint32))\n"
        "lib=simlib.lib(asan=True)\n"
        "for base in BASES:\n"
        "    for events in (False, True):\n"
        "        cur=np.full(L.frame_bytes,0xC3,np.uint8); status=np.zeros(1,np.uint32); b=np.array([base],np.uint64)\n"
        "        dummy=np.zeros((1,64),np.int16)\n"
        "        if events:\n"
        "            rc=lib.sim_recon_ex(w,h,1,simlib._p(mbs),simlib._p(dummy),len(co),simlib._p(b),None,0,simlib._p(cur),simlib._p(status),simlib._p(first),simlib._p(ev))\n"
        "        else:\n"
        "            rc=lib.sim_recon(w,h,1,simlib._p(mbs),simlib._p(co),len(co),simlib._p(b),None,0,simlib._p(cur),simlib._p(status))\n"
        "        assert rc==0 and (int(status[0]) & 2), (hex(base), events, rc, int(status[0]))\n"
        "print('hostile-base-ok')\n" % (root, os.path.join(root, "tests"))).replace(
            "BASES", "[len(co), len(co)+5, 1<<40, (1<<63)+7, (1<<64)-(1<<20), (1<<64)-2, (1<<64)-1]")


def test_a_hostile_coefficient_base_reads_nothing_outside_the_pool():
    """ReconArgs::coeff_base comes out of a caller's DEVICE memory (h263mi_batch_decode[_events]): nobody has seen its values.
    A base at, beyond or astronomically beyond the end of the pool -- up to 2^64 - 1, where `pool - base` wraps to a small
    positive number -- must make every coded block of the picture "outside the pool" (status bit 2) and read NOTHING: run
    under AddressSanitizer, where a read in front of the pool or of the block offsets is a report.  (Round 6: the GPU fuzzer's
    hostile arrays found a memory access fault here -- base 2^64 - 1 made block 0 read 128 bytes in FRONT of the pool.)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    out = subprocess.run([sys.executable, "-c", _hostile_base_script(root)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "hostile-base-ok" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])


def test_the_gpu_fuzzers_hostile_arrays_replayed_under_address_sanitizer():
    """tests/sim/hostile_replay.py: what tools/fuzz_gpu.py's hostile cases hand to a CHECKED launch -- garbage block offsets
    (0xffffffff: `first + lane` wrapped and the loop walked off the events, a memory access fault on the MI355X in round 6),
    descending pairs, block indices and bases far outside the pool (2^64 - 2^20: the other fault) -- through the kernel phases
    on the CPU with exact-size arrays: no read outside them, untouched streams decode, hit streams are rejected or decode
    something."""
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sim", "hostile_replay.py")
    for seed in ("6", "7", "8", "9"):
        out = subprocess.run([sys.executable, script, seed, "8"], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "hostile-cpu-ok" in out.stdout, (seed, out.stdout[-800:], out.stderr[-3000:])


def test_garbage_record_fields_and_event_words_under_address_sanitizer():
    """tests/sim/garbage_fields_replay.py: WHAT a checked launch reads may be anything -- macroblock types 6..255, quantisers 0
    and 32..255, any coded-block-pattern and kill byte, vectors over the whole int16 range, INTRADC codes that never occur,
    event words with positions beyond 63 -- P pictures on a reference and I pictures, both transports: no access outside the
    caller's arrays and the frame store (exact-size arrays under AddressSanitizer, array indices under UBSan), and the streams
    left alone decode to the oracle's planes.  (tools/fuzz_gpu.py: fuzz_garbage_fields is the same on the MI355X.)"""
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sim", "garbage_fields_replay.py")
    for seed in ("11", "12", "13"):
        out = subprocess.run([sys.executable, script, seed, "10"], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "garbage-fields-cpu-ok" in out.stdout, (seed, out.stdout[-800:], out.stderr[-3000:])
