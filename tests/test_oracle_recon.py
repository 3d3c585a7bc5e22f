"""Oracle checks for dequant / IDCT / motion compensation (the part of the hot path the
reference itself does not test: "parity unpinned").  Three kinds of evidence:

 1. hand-derivable identities from the reference text;
 2. agreement of the C oracle with the independent numpy restatement on seeded inputs;
 3. the numerical facts SURVEY section 0 lists (Dc != Full on 204 DCs, Horiz == Full, ...).
"""
import numpy as np
import pytest

import recgen
from oracle import np_restatement as npr
from oracle import oracle as orc


def test_basis_table_bit_patterns():
    want = np.array([int(x, 16) for x in npr.BASIS_HEX.split()], dtype=np.uint32).reshape(8, 8)
    assert (npr.BASIS.view(np.uint32) == want).all()
    assert (orc.basis_table().view(np.uint32) == want).all()


def test_intradc_and_halfpel_helpers():
    L = orc.lib()
    assert L.orc_intradc_into_level(0xFF) == 1024
    assert [L.orc_intradc_into_level(c) for c in (1, 2, 127, 254)] == [8, 16, 1016, 2032]
    # SURVEY A.4 sample values of average_sum_of_mvs
    table = {-20: -3, -18: -2, -14: -2, -13: -1, -3: -1, -2: 0, 0: 0, 2: 0, 3: 1, 13: 1, 14: 2, 18: 2, 19: 3}
    for s, want in table.items():
        assert L.orc_average_sum_of_mvs(s) == want, s
        assert npr.chroma_mv(s) == want, s
    for s in range(-300, 300):
        assert L.orc_average_sum_of_mvs(s) == npr.chroma_mv(s)
    import ctypes as C
    for hp in range(-70, 70):
        d, i = C.c_int16(), C.c_int()
        L.orc_lerp_parameters(hp, C.byref(d), C.byref(i))
        assert (d.value, i.value) == (hp >> 1, hp & 1) == (int(np.floor(hp / 2)), hp % 2), hp


def test_inverse_rle_classes():
    # empty intra block -> Dc(level); code 0xFF -> 1024 (rle.rs:94-109, types.rs:955-961)
    assert orc.inverse_rle(True, 10, [], [], 5)[0] == orc.ORC_DC
    tag, v = orc.inverse_rle(True, 0xFF, [], [], 5)
    assert (tag, v[0]) == (orc.ORC_DC, 1024.0)
    assert orc.inverse_rle(False, 0, [], [], 5)[0] == orc.ORC_ZERO
    # inter block, single coefficient at zigzag 0 -> Dc(dequantised) (appendix B.6)
    tag, v = orc.inverse_rle(False, 0, [0], [3], 6)          # 6*(2*3+1)-1 = 41
    assert (tag, v[0]) == (orc.ORC_DC, 41.0)
    tag, v = orc.inverse_rle(False, 0, [0], [-3], 5)         # -(5*7) = -35
    assert (tag, v[0]) == (orc.ORC_DC, -35.0)
    # clamp to [-2048, 2047] (rle.rs:133)
    assert orc.inverse_rle(False, 0, [0], [127], 31)[1][0] == 2047.0
    assert orc.inverse_rle(False, 0, [0], [-127], 31)[1][0] == -2048.0
    # row-only -> Horiz, column-only -> Vert, both -> Full
    assert orc.inverse_rle(True, 10, [0], [1], 4)[0] == orc.ORC_HORIZ      # zigzag 1 = (1,0)
    assert orc.inverse_rle(True, 10, [1], [1], 4)[0] == orc.ORC_VERT       # zigzag 2 = (0,1)
    assert orc.inverse_rle(True, 10, [0, 0], [1, 1], 4)[0] == orc.ORC_FULL
    # run overflow: block left Zero, DC discarded (rle.rs:125-127, appendix B.4)
    assert orc.inverse_rle(True, 10, [0, 63], [1, 1], 4)[0] == orc.ORC_ZERO
    assert orc.inverse_rle(True, 10, [63], [1], 4)[0] == orc.ORC_ZERO      # 1 + 63 = 64
    assert orc.inverse_rle(False, 0, [63], [1], 4)[0] == orc.ORC_FULL      # inter: lands on zigzag 63


def test_dc_only_intra_block_gives_flat_code_value():
    # DC-only intra block with INTRADC code c => all pixels c (255 => 128)
    for code in list(range(1, 128)) + list(range(129, 256)):
        out = orc.idct_blocks([orc.ORC_DC], [[orc.lib().orc_intradc_into_level(code)] + [0] * 63],
                              np.zeros(64, np.uint8), 1, 8)
        want = 128 if code == 255 else code
        assert (out == want).all(), code


def _dc_path(dc):
    r = npr.idct_residual(npr.DC, np.array([[dc] + [0] * 7] + [[0] * 8] * 7, np.float32))
    return int(r[0, 0])


def _full_path(blk):
    return npr.idct_residual(npr.FULL, np.asarray(blk, np.float32))


def test_dc_class_differs_from_full_on_204_values():
    # 204 of the 4096 DC values, all with dc = 4 mod 8
    diff = []
    for dc in range(-2048, 2048):
        if dc == 0:
            continue
        blk = np.zeros((8, 8), np.float32)
        blk[0, 0] = dc
        f = _full_path(blk)
        assert (f == f[0, 0]).all()
        if int(f[0, 0]) != _dc_path(dc):
            diff.append(dc)
    assert len(diff) == 204
    assert all(d % 8 == 4 for d in diff)
    assert -1972 in diff


def test_all_dc_values_c_vs_numpy_with_pred_0_and_255():       # appendix B.1
    for pred in (0, 255):
        for dc in range(-2048, 2048):
            out = orc.idct_blocks([orc.ORC_DC], [[dc] + [0] * 63], np.full(64, pred, np.uint8), 1, 8)
            want = np.clip(pred + _dc_path(dc), 0, 255) if dc != 0 else pred
            assert (out == want).all(), (dc, pred)


def test_horiz_equals_full_vert_differs_rarely():
    rng = np.random.default_rng(3)
    vert_diff = 0
    for _ in range(3000):
        row = np.zeros((8, 8), np.float32)
        row[0, :] = rng.integers(-300, 300, 8)
        assert (npr.idct_residual(npr.HORIZ, row) == _full_path(row)).all()
        col = np.zeros((8, 8), np.float32)
        col[:, 0] = rng.integers(-60, 60, 8) * (rng.random(8) < 0.4)
        vert_diff += not (npr.idct_residual(npr.VERT, col) == _full_path(col)).all()
    assert vert_diff > 0            # Vert must keep its own rounding order
    # the concrete columns of appendix B.2
    col = np.zeros((8, 8), np.float32)
    col[:, 0] = [19, 0, 0, 0, -7, 0, 0, 0]
    assert npr.idct_residual(npr.VERT, col)[:, 0].tolist() == [1, 3, 3, 1, 1, 3, 3, 1]
    assert _full_path(col)[:, 0].tolist() == [2, 3, 3, 1, 2, 3, 3, 2]
    col[:, 0] = [-38, 0, 0, 0, -2, 0, 0, 0]
    assert npr.idct_residual(npr.VERT, col)[:, 0].tolist() == [-5, -5, -5, -5, -5, -4, -5, -5]


def test_c_idct_matches_numpy_on_every_class():
    rng = np.random.default_rng(11)
    for trial in range(400):
        cls = ["dc", "horiz", "vert", "full_dense", "full_sparse"][trial % 5]
        intra = trial % 2 == 0
        q = int(rng.integers(1, 32))
        coeff = recgen.fill_block(rng, cls, 127, intra)
        code = int(recgen.random_intradc(rng, 1)[0])
        coded = cls != "dc"
        tag, blk = npr.classify_dense(coeff, intra, code, coded, False, q)
        # the C oracle, through the run/level Block form
        mbs = np.zeros(1, orc.MB_RECORD_DTYPE)
        mbs[0]["mb_type"] = 3 if intra else 0
        mbs[0]["quant"] = q
        mbs[0]["cbp"] = 1 if coded else 0
        mbs[0]["intradc"][0] = code
        ref = recgen.random_planes(16, 16, trial)
        rc, (y, cb, cr) = orc.decode_picture(16, 16, mbs, coeff.reshape(1, 64), ref)
        assert rc == 0
        res = npr.idct_residual(tag, blk)
        pred = np.zeros((8, 8), np.int64) if intra else ref[0].reshape(16, 16)[:8, :8].astype(np.int64)
        assert (y.reshape(16, 16)[:8, :8] == np.clip(pred + res, 0, 255)).all(), (trial, cls)


@pytest.mark.parametrize("w,h", [(16, 16), (48, 32), (100, 60), (5, 4), (1, 1), (176, 144)])
def test_c_picture_matches_numpy_intra(w, h):
    mbs, coeffs = recgen.intra_picture(w, h, seed=w * 1000 + h)
    rc, got = orc.decode_picture(w, h, mbs, coeffs, None)
    rc2, want = npr.decode_picture(w, h, mbs, coeffs, None)
    assert rc == rc2 == 0
    for g, e in zip(got, want):
        assert (g == e).all()


@pytest.mark.parametrize("w,h", [(16, 16), (48, 32), (100, 60), (5, 4), (33, 17), (176, 144)])
def test_c_picture_matches_numpy_inter(w, h):
    ref = recgen.random_planes(w, h, 5)
    # mv range +-80 half-pels pushes all four taps outside each border and corner on small pictures
    mbs, coeffs = recgen.inter_picture(w, h, seed=w * 7 + h, mv_range=80, p_4v=0.3, p_intra=0.15, quant=0)
    rc, got = orc.decode_picture(w, h, mbs, coeffs, ref)
    rc2, want = npr.decode_picture(w, h, mbs, coeffs, ref)
    assert rc == rc2 == 0
    for g, e in zip(got, want):
        assert (g == e).all()


def test_zero_mv_uncoded_picture_is_a_copy_and_padding_semantics():
    w, h = 100, 60
    ref = recgen.random_planes(w, h, 9)
    mbs = np.zeros(0, orc.MB_RECORD_DTYPE)              # every MB padded as Inter / mv 0 (state.rs:421-427)
    rc, got = orc.decode_picture(w, h, mbs, np.zeros((0, 64), np.int16), ref)
    assert rc == 0
    for g, e in zip(got, ref):
        assert (g == e).all()
    rc, _ = orc.decode_picture(w, h, mbs, np.zeros((0, 64), np.int16), None)
    assert rc == orc.ERR_UNCODED_IFRAME_BLOCKS              # gather.rs:149


def test_halfpel_identities():
    w, h = 32, 32
    ref = recgen.random_planes(w, h, 21)
    Y = ref[0].reshape(h, w).astype(np.int64)
    mbs = np.zeros(4, orc.MB_RECORD_DTYPE)
    for mv, fn in (((1, 0), lambda a, b, c, d: (a + b + 1) >> 1), ((0, 1), lambda a, b, c, d: (a + c + 1) >> 1),
                   ((1, 1), lambda a, b, c, d: (a + b + c + d + 2) >> 2), ((2, 2), None)):
        mbs["mv"] = np.array(mv, np.int16)
        rc, (y, _, _) = orc.decode_picture(w, h, mbs, np.zeros((0, 64), np.int16), ref)
        assert rc == 0
        y = y.reshape(h, w)
        xs = np.clip(np.arange(w) + 1, 0, w - 1)
        ys = np.clip(np.arange(h) + 1, 0, h - 1)
        a, b, c, d = Y, Y[:, xs], Y[ys, :], Y[np.ix_(ys, xs)]
        want = d if fn is None else fn(a, b, c, d)
        assert (y == want).all(), mv


def test_kill_bit_discards_block_and_dc():
    mbs, coeffs = recgen.intra_picture(16, 16, 1, classes=("full_sparse",))
    mbs[0]["kill"] = 0b000101
    rc, (y, cb, cr) = orc.decode_picture(16, 16, mbs, coeffs, None)
    rc2, (y2, _, _) = npr.decode_picture(16, 16, mbs, coeffs, None)
    assert rc == rc2 == 0 and (y == y2).all()
    Y = y.reshape(16, 16)
    assert (Y[:8, :8] == 0).all() and (Y[8:, :8] == 0).all()       # blocks 0 and 2 are Zero
    assert Y[:8, 8:].any() and Y[8:, 8:].any()


def test_named_vert_columns_golden_file():
    """tests/golden/vert_named_columns.json (tools/gen_vert_columns.py): reachable first-coefficient columns on which the
    Vert class and the Full arithmetic differ; the C oracle (through the whole record path: LEVELs at quantiser 1 ->
    inverse_rle -> idct_channel), the numpy restatement and the soft-float model must all give the file's `vert` rows.
    tests/test_gpu_round3.py decodes the same columns on the MI355X."""
    import json
    import os
    from oracle import softfloat_idct as sf
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vert_named_columns.json")
    gold = json.load(open(path))["columns"]
    assert len(gold) >= 5 and gold[0]["column"] == [19, 0, 0, 0, -7, 0, 0, 0]
    for g in gold:
        col = g["column"]
        assert all(v == 0 or v % 2 for v in col)               # rle.rs:130-133 only produces odd values
        # the record path: a P macroblock at quantiser 1 whose block 0 carries the LEVELs, over a flat prediction
        mbs = np.zeros(1, orc.MB_RECORD_DTYPE)
        mbs["quant"] = 1
        mbs["cbp"] = 1
        co = np.zeros((1, 64), np.int16)
        for r, level in g["levels_at_quant_1"].items():
            co[0, 8 * int(r)] = level
            assert sf.dequant(level, 1) == col[int(r)]
        ref = (np.full(256, 100, np.uint8), np.full(64, 100, np.uint8), np.full(64, 100, np.uint8))
        rc, out = orc.decode_picture(16, 16, mbs, co, ref)
        assert rc == 0
        rows = out[0].reshape(16, 16)[:8, :8].astype(int) - 100
        assert (rows == np.array(g["vert"])[:, None]).all(), col
        dense = np.zeros((8, 8), np.float32)
        dense[:, 0] = col
        assert npr.idct_residual(npr.VERT, dense)[:, 0].tolist() == g["vert"]
        assert _full_path(dense)[:, 0].tolist() == g["full"]
        block = [0] * 64
        for r, v in enumerate(col):
            block[8 * r] = v
        assert [row[0] for row in sf.block_residual(block)] == g["vert"]
        assert [row[0] for row in sf.block_residual(block, force_full=True)] == g["full"]
