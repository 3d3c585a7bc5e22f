"""Host bitstream parser (h263-rs_amd/host/bitstream.cpp) against every known answer the
reference's in-source parser tests hold: bit reader (reader.rs:448-559), all code words of the MCBPC / CBPY /
MVD / TCOEF tables (macroblock.rs:559-1010, block.rs:766-1705) and the eight hand-built block bitstreams
incl. the Sorenson 7- and 11-bit escapes (block.rs:1707-2124).  Data: tests/golden/parser_reference_tests.json
(tools/extract_golden.py); the reader cases are transcribed below (inputs and expected values only)."""
import json
import os

import numpy as np
import pytest

import parselib as pl

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "parser_reference_tests.json")))
MBTYPE = {"Inter": 0, "InterQ": 1, "Inter4V": 2, "Intra": 3, "IntraQ": 4, "Inter4Vq": 5}
READ, PEEK, SKIP, READ_SIGNED, START_CODE = range(5)


# ---- reader.rs:448-559 ---------------------------------------------------------------------------------
def test_reader_unaligned_and_signed_reads():
    assert pl.reader_script([0xFF, 0x72, 0x1C, 0x1F], [(READ, 3), (READ, 6), (READ, 23), (READ, 1)]) == \
        [(0, 0x07), (0, 0x3E), (0, 0x721C1F), (pl.EOF_ERR, 0)]
    r = pl.reader_script([0xFF, 0x40, 0x72, 0x1C, 0x1F], [(READ_SIGNED, 3), (READ_SIGNED, 6), (READ_SIGNED, 8),
                                                          (READ_SIGNED, 23), (READ, 1)])
    assert r == [(0, -1), (0, -2), (0, -0x80), (0, -0xDE3E1), (pl.EOF_ERR, 0)]
    r = pl.reader_script([0xFF, 0x72, 0x1C, 0x1F], [(PEEK, 3), (PEEK, 6), (PEEK, 23)])
    assert r == [(0, 0x07), (0, 0x3F), (0, 0x7FB90E)]


def test_reader_bytes_and_words():
    assert pl.reader_script([0xFE, 0x73, 0xF3], [(READ, 8)] * 3) == [(0, 0xFE), (0, 0x73), (0, 0xF3)]
    assert pl.reader_script([0xFE, 0x73, 0xF3], [(SKIP, 2), (READ, 8), (READ, 8), (READ, 8)]) == \
        [(0, 0), (0, 0xF9), (0, 0xCF), (pl.EOF_ERR, 0)]
    assert pl.reader_script([0xFE, 0x73, 0x50, 0xF3], [(READ, 16), (READ, 16)]) == [(0, 0xFE73), (0, 0x50F3)]
    assert pl.reader_script([0xFE, 0x73, 0x50, 0xF3], [(READ, 32)]) == [(0, 0xFE7350F3)]


def test_reader_start_codes():
    assert pl.reader_script([0x00, 0x00, 0x80, 0x00], [(START_CODE, 0)]) == [(0, 0)]
    # stuffed: None (-1) at the aligned position, 3 bits ahead once one bit has been consumed
    assert pl.reader_script([0x00, 0x00, 0x08, 0x00], [(START_CODE, 0), (SKIP, 1), (START_CODE, 0)]) == \
        [(0, -1), (0, 0), (0, 3)]
    assert pl.reader_script([0x13, 0x80, 0x00, 0x40, 0x00], [(START_CODE, 1)]) == [(0, 9)]


# ---- code tables ---------------------------------------------------------------------------------------------
def test_tcoef_table_every_code_word():
    t = GOLD["tcoef_table"]
    got = pl.read_vlc(pl.TCOEF, t["bytes"], len(t["expected"]))
    assert len(got) == len(t["expected"]) == 102
    for g, e in zip(got, t["expected"]):
        assert g[0] == 1 and [bool(g[1]), int(g[2]), int(g[3])] == e, (g, e)
    # ESCAPE: 0000 011
    g = pl.read_vlc(pl.TCOEF, [0b00000110], 1)[0]
    assert g[0] == 1 and g[1] == -1


def _check_mcbpc(table, key):
    t = GOLD[key]
    got = pl.read_vlc(table, t["bytes"], len(t["expected"]))
    assert len(got) == len(t["expected"])
    for g, e in zip(got, t["expected"]):
        if e == "stuffing":
            assert g[0] == 1 and g[1] == -1, (g, e)
        elif e == "invalid":
            assert g[0] == 0, (g, e)
        else:
            assert g[0] == 1 and [int(g[1]), bool(g[2]), bool(g[3])] == [MBTYPE[e[0]], e[1], e[2]], (g, e)


def test_mcbpc_tables_every_code_word():
    _check_mcbpc(pl.MCBPC_I, "mcbpc_i")
    _check_mcbpc(pl.MCBPC_P, "mcbpc_p")


def test_cbpy_table_every_code_word():
    t = GOLD["cbpy"]
    got = pl.read_vlc(pl.CBPY, t["bytes"], len(t["expected"]))
    assert len(got) == len(t["expected"])
    for g, e in zip(got, t["expected"]):
        if e is None:
            assert g[0] == 0, (g, e)
        else:
            assert g[0] == 1 and [bool(g[1] & 8), bool(g[1] & 4), bool(g[1] & 2), bool(g[1] & 1)] == e, (g, e)


def test_mvd_table_every_code_word():
    t = GOLD["mvd"]
    got = pl.read_vlc(pl.MVD, t["bytes"], len(t["expected"]))
    assert len(got) == len(t["expected"])
    for g, e in zip(got, t["expected"]):
        if e is None:
            assert g[0] == 0, (g, e)
        else:
            assert g[0] == 1 and g[1] == int(e * 2), (g, e)       # HalfPel::from(f32) = floor(v * 2)


# ---- whole blocks (block.rs:1707-2124) ------------------------------------------------------------------------
@pytest.mark.parametrize("case", GOLD["blocks"], ids=[b["name"] for b in GOLD["blocks"]])
def test_decode_block_reference_cases(case):
    rc, has_dc, code, tcoef, _ = pl.decode_block(case["bytes"], case["sorenson"], case["version"], case["intra"],
                                                 case["tcoef_present"])
    assert rc == 0
    if case["intradc_level"] is None:
        assert not has_dc
    else:
        assert has_dc and (1024 if code == 255 else code << 3) == case["intradc_level"]
    assert [(bool(s), r, l) for s, r, l in tcoef] == [(bool(s), r, l) for s, r, l in case["tcoef"]]


def test_decode_block_errors():
    assert pl.decode_block([0x00], False, None, True, False)[0] == -5        # INTRADC 0 is illegal (types.rs:930-936)
    assert pl.decode_block([0x80], False, None, True, False)[0] == -5        # INTRADC 128 too
    assert pl.decode_block([0x00, 0x00], False, None, False, True)[0] == -6  # 0000 0000 0...: no TCOEF starts so
    assert pl.decode_block([0b00000110, 0, 0], False, None, False, True)[0] == -7   # escape with LEVEL 0
    assert pl.decode_block([0b10], False, None, False, True)[0] == pl.EOF_ERR  # data ends inside the code word


def test_mv_prediction_and_sorenson_header_known_answers_derived_from_the_reference_text():
    """tests/golden/mv_prediction_known_answers.json: literal bit strings of a Sorenson picture header and of macroblock
    headers, with the vectors predict_candidate / halfpel_decode must produce worked out by hand from mvd_pred.rs:27-117,
    picture.rs:619-659 and macroblock.rs:445-549 (the derivation is written next to every macroblock).  Nothing in it
    comes from this repository's encoder or parser."""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mv_prediction_known_answers.json")))
    for pic in gold["pictures"]:
        bits = "".join(pic["header_bits"]) + "".join("".join(mb["bits"]) for mb in pic["macroblocks"])
        bits += "0" * (-len(bits) % 8)
        data = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
        pl.context_reset()
        rc, d, mbs, co, used = pl.parse_picture(data, options=1)
        assert rc == 0, pic["name"]
        assert (d.width, d.height, d.picture_type, d.pquant, d.temporal_reference, d.use_deblocker) == (
            pic["width"], pic["height"], pic["picture_type"], pic["quant"], pic["temporal_reference"], pic["use_deblocker"])
        assert len(mbs) == len(pic["macroblocks"]) == ((pic["width"] + 15) // 16) * ((pic["height"] + 15) // 16)
        assert len(co) == 0
        for k, mb in enumerate(pic["macroblocks"]):
            e = mb["expect"]
            assert int(mbs[k]["mb_type"]) == e["mb_type"], (pic["name"], k)
            assert mbs[k]["mv"].tolist() == e["mv"], "%s, macroblock %d: got %s, the reference text gives %s (%s)" % (
                pic["name"], k, mbs[k]["mv"].tolist(), e["mv"], mb["derivation"])
            assert int(mbs[k]["cbp"]) == 0 and int(mbs[k]["quant"]) == pic["quant"]


def _content_fixture():
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "macroblock_content_known_answers.json")))


def fixture_picture_bytes(pic):
    """the literal bit strings of a fixture picture, concatenated and zero-padded to a byte boundary -- nothing else"""
    bits = "".join(pic["header_bits"]) + "".join("".join(mb["bits"]) for mb in pic["macroblocks"])
    assert set(bits) <= {"0", "1"}
    bits += "0" * (-len(bits) % 8)
    return bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))


def test_macroblocks_with_content_known_answers_derived_from_the_reference_text():
    """tests/golden/macroblock_content_known_answers.json: hand-derived bit strings that carry
    CONTENT behind the macroblock header -- non-zero CBPY in both senses (the inversion of macroblock.rs:479-489), every
    MCBPC chroma pattern, DQUANT with the clamp of state.rs:226-227 at both ends, four-vector candidates (mvd_pred.rs:27-67),
    TCOEF events incl. LAST, the 7-, 11- and 8-bit escapes (block.rs:694-708), a run that overflows (rle.rs:125-127),
    stuffing, a picture that ends with its data in the middle of a row (state.rs:411, 421-427).  The parser's records, kill
    flags and coefficient blocks must be what the reference's text gives.  tests/sorenson_enc.py is not involved."""
    gold = _content_fixture()
    for pic in gold["pictures"]:
        data = fixture_picture_bytes(pic)
        pl.context_reset()
        options = pic.get("decoder_options", 1)          # (1 = SORENSON_SPARK_BITSTREAM; picture D is ITU-T H.263)
        rc, d, mbs, co, used = pl.parse_picture(data, options=options)
        assert rc == 0, pic["name"]
        assert (d.width, d.height, d.picture_type, d.pquant, d.temporal_reference, d.use_deblocker) == (
            pic["width"], pic["height"], pic["picture_type"], pic["quant"], pic["temporal_reference"], pic["use_deblocker"])
        assert len(mbs) == pic["n_records"] == len(pic["macroblocks"]), (pic["name"], len(mbs))
        at = 0
        for k, mb in enumerate(pic["macroblocks"]):
            e, r, what = mb["expect"], mbs[k], "%s, macroblock %d (%s)" % (pic["name"], k, mb["derivation"])
            assert int(r["mb_type"]) == e["mb_type"], what
            assert int(r["cbp"]) == e["cbp"], what
            assert int(r["kill"]) == e["kill"], what
            assert r["mv"].tolist() == e["mv"], what
            if "quant" in e:
                assert int(r["quant"]) == e["quant"], what
            if "intradc" in e:
                assert r["intradc"].tolist() == e["intradc"], what
            intra = e["mb_type"] in (3, 4)
            assert len(e["blocks"]) == bin(e["cbp"]).count("1"), what
            if e["blocks"]:
                assert int(r["coeff_index"]) == at, what
            for j, blk in enumerate(e["blocks"]):
                got = co[at + j]
                if blk is not None:                              # (null: a killed block, its coefficients do not count)
                    have = {int(p): int(got[p]) for p in np.flatnonzero(got) if not (intra and p == 0)}
                    assert have == {int(p): v for p, v in blk.items()}, "%s: block %d: %s" % (what, j, have)
            at += len(e["blocks"])
        assert len(co) == at
        # ... and the EVENTS the product's form of the parse emits (what crosses to the device): block k's events are exactly the
        # expected (raster position, LEVEL) pairs -- an intra block's DC is not an event
        rc, first, ev = pl.parse_picture_events(data, options=options)
        assert rc == 0 and len(first) == at + 1 and first[0] == 0 and first[-1] == len(ev)
        k = 0
        for mb in pic["macroblocks"]:
            for blk in mb["expect"]["blocks"]:
                if blk is not None:
                    got = {int(e & 63): int(np.int16(np.uint16(e >> 16))) for e in ev[first[k]:first[k + 1]]}
                    assert got == {int(p): v for p, v in blk.items()}, (pic["name"], k, got)
                k += 1
        # the windowed fast paths and the field-by-field transcription agree on these bytes too
        assert pl.compare_parser_paths(data, options)[0] == 0, pic["name"]
