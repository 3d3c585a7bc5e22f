"""ITU-T H.263 (non-Sorenson) picture layer and the decode loop around it -- SURVEY section 8 row f-4.

The reference has no tests for parser/picture.rs; every expectation below is derived from its code and cites the
lines.  Streams come from tests/sorenson_enc.py (standard=...), records from tests/recgen.py."""
import numpy as np
import pytest

import parselib as pl
import recgen
import sorenson_enc as enc
from oracle import oracle as orc
from test_bitstream_e2e import assert_records_equal, make_codable

ERR_MIDDLE, ERR_INVALID_MB_HEADER, ERR_INVALID_PTYPE, ERR_INVALID_PLUSPTYPE = -2, -3, -9, -10
ERR_FORMAT_MISSING, ERR_FORMAT_INVALID, ERR_INVALID_BITSTREAM, ERR_EOF, ERR_UNIMPLEMENTED = -13, -14, -12, -16, -17
ERR_INVALID_MVD = -8
OPT_UMV, OPT_SAC, OPT_AP, OPT_MQ, OPT_RPR, OPT_RT1 = 1 << 3, 1 << 4, 1 << 5, 1 << 12, 1 << 13, 1 << 15


def bits(s):
    s = s.replace(" ", "")
    s += "0" * (-len(s) % 8)
    return bytes(int(s[i:i + 8], 2) for i in range(0, len(s), 8))


def header_bytes(*a, **k):
    bw = enc.BitWriter()
    enc.write_standard_header(bw, *a, **k)
    return bw.tobytes() + b"\x00\x00\x00"


def test_error_codes_are_the_ones_named_here():
    import re
    txt = open(pl.os.path.join(pl.HERE, "..", "include", "h263mi.h")).read()
    want = {"INVALID_MACROBLOCK_HEADER": ERR_INVALID_MB_HEADER, "INVALID_MVD": ERR_INVALID_MVD, "INVALID_PTYPE": ERR_INVALID_PTYPE,
            "INVALID_PLUS_PTYPE": ERR_INVALID_PLUSPTYPE, "PICTURE_FORMAT_MISSING": ERR_FORMAT_MISSING,
            "PICTURE_FORMAT_INVALID": ERR_FORMAT_INVALID, "INVALID_BITSTREAM": ERR_INVALID_BITSTREAM,
            "UNHANDLED_IO_ERROR": ERR_EOF, "UNIMPLEMENTED_DECODING": ERR_UNIMPLEMENTED, "MIDDLE_OF_BITSTREAM": ERR_MIDDLE}
    for name, code in want.items():
        m = re.search(r"#define H263MI_ERR_%s\s+\((-?\d+)\)" % name, txt)
        assert m and int(m.group(1)) == code, name


# --- reader.rs:298-324 -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("code,value", [("1", 0), ("0 00", 1), ("0 10", -1), ("0 01 00", 2), ("0 11 00", 3),
                                        ("0 11 10", -3), ("0 01 01 00", 4), ("0 11 01 11 10", -13)])
def test_read_umv_table_d3(code, value):
    rc, v, used = pl.read_umv(bits(code + "1111"))
    assert (rc, v, used) == (0, value, len(code.replace(" ", "")))


def test_read_umv_limits_and_writer_round_trip():
    # 12 continuation pairs bring `bulk` to 4096: InvalidMvd (reader.rs:321)
    assert pl.read_umv(bits("0" + "01" * 12 + "00"))[0] == ERR_INVALID_MVD
    assert pl.read_umv(bits("0" + "11" * 11 + "00"))[:2] == (0, 4095)
    assert pl.read_umv(b"")[0] == ERR_EOF
    assert pl.read_umv(bits("0 01 01 01 0"))[0] == ERR_EOF             # exactly one byte: the data ends inside the code
    for v in list(range(-70, 71)) + [-4095, 4095, 2048, -2049]:
        bw = enc.BitWriter()
        enc.write_umv(bw, v)
        n = sum(len(b) for b in bw.bits)
        assert pl.read_umv(bw.tobytes())[:3] == (0, v, n)


# --- decode_ptype (picture.rs:21-79) ---------------------------------------------------------------------------------
@pytest.mark.parametrize("size,kind", [((128, 96), 1), ((176, 144), 2), ((352, 288), 3), ((704, 576), 4), ((1408, 1152), 5)])
def test_ptype_source_formats(size, kind):
    h = pl.parse_header(header_bytes(size[0], size[1], 0, 13, temporal_reference=77, pei=(0xAB, 0xCD)))
    assert (h["rc"], h["is_picture"], h["width"], h["height"], h["format_kind"]) == (0, 1, size[0], size[1], kind)
    assert (h["picture_type"], h["quantizer"], h["temporal_reference"], h["n_extra"]) == (0, 13, 77, 2)
    assert not h["has_plusptype"] and not h["has_opptype"] and h["mv_range"] == 0
    # 17 start + 5 GN + 8 TR + 13 PTYPE + 5 PQUANT + 1 CPM + 2 * 9 PEI/PSUPP + 1 PEI
    assert h["bits_used"] == 17 + 5 + 8 + 13 + 5 + 1 + 18 + 1


def test_ptype_flags_and_errors():
    h = pl.parse_header(header_bytes(176, 144, 1, 5, ptype_low=0x8 | 0x4 | 0x2, ptype_high_flags=0b101))
    assert h["rc"] == 0 and h["picture_type"] == 1
    assert h["options"] == (0b101 | OPT_UMV | OPT_SAC | OPT_AP)     # split screen + freeze release; document camera clear
    assert pl.parse_header(header_bytes(176, 144, 0, 5, pb=True))["picture_type"] == 4        # PbFrame; TRB + DBQUANT consumed
    assert pl.parse_header(header_bytes(176, 144, 0, 5, pb=True))["bits_used"] == 17 + 5 + 8 + 13 + 5 + 1 + 5 + 1
    assert pl.parse_header(header_bytes(176, 144, 0, 5, cpm=1))["bits_used"] == 17 + 5 + 8 + 13 + 5 + 3 + 1
    # the two leading PTYPE bits must be "10"; source format 000 is forbidden (picture.rs:31-33, 48)
    raw = bytearray(header_bytes(176, 144, 0, 5))
    bad = bytes(raw[:3]) + bytes([raw[3] ^ 0x20]) + bytes(raw[4:])               # bit 31 of the stream = first PTYPE bit?  checked below
    assert pl.parse_header(header_bytes(176, 144, 0, 5, ptype_format=0))["rc"] == ERR_INVALID_PTYPE
    bw = enc.BitWriter()
    bw.put(1, 17); bw.put(0, 5); bw.put(0, 8); bw.put(0, 2); bw.put(0, 3); bw.put(2, 3); bw.put(0x10, 5); bw.put(5, 5); bw.put(0, 2)
    assert pl.parse_header(bw.tobytes() + b"\0\0")["rc"] == ERR_INVALID_PTYPE
    del bad
    # reserved source format 110 parses, but has no dimensions (types.rs:175, state.rs:169-171)
    res = header_bytes(176, 144, 0, 5, ptype_format=6)
    assert pl.parse_header(res)["rc"] == 0 and pl.parse_header(res)["format_kind"] == 6
    assert pl.parse_picture(res, options=0)[0] == ERR_FORMAT_INVALID
    # GOB number != 0 is not a picture: Ok(None), nothing consumed (picture.rs:660-662)
    bw = enc.BitWriter()
    bw.put(1, 17); bw.put(3, 5); bw.put(0, 24)
    h = pl.parse_header(bw.tobytes())
    assert (h["rc"], h["is_picture"], h["bits_used"]) == (0, 0, 0)
    assert pl.parse_picture(bw.tobytes(), options=0)[0] == ERR_MIDDLE
    # header cut short: EOF, nothing consumed
    cut = pl.parse_header(header_bytes(176, 144, 0, 5)[:4])
    assert cut["rc"] == ERR_EOF and cut["bits_used"] == 0


# --- PLUSPTYPE and its followers (picture.rs:135-268, 335-596, 662-808) ---------------------------------------------------
def test_plusptype_custom_format_and_followers():
    h = pl.parse_header(header_bytes(200, 120, 1, 9, plus=True, temporal_reference=3))
    assert (h["rc"], h["width"], h["height"], h["format_kind"], h["picture_type"]) == (0, 200, 120, 7, 1)
    assert h["has_plusptype"] and h["has_opptype"] and h["quantizer"] == 9 and h["temporal_reference"] == 3
    assert h["bits_used"] == 17 + 5 + 8 + 8 + 3 + 18 + 9 + 1 + 23 + 5 + 1
    # standard format restated through OPPTYPE, rounding type + UMV with "unlimited" UUI
    h = pl.parse_header(header_bytes(352, 288, 1, 9, plus=True, opptype=enc.OPP_UMV, uui="01", mpptype_flags=0x008))
    assert (h["width"], h["height"], h["format_kind"], h["mv_range"]) == (352, 288, 3, 2)
    assert h["options"] == OPT_UMV | OPT_RT1
    assert pl.parse_header(header_bytes(352, 288, 1, 9, plus=True, opptype=enc.OPP_UMV, uui="1"))["mv_range"] == 1
    assert pl.parse_header(header_bytes(352, 288, 1, 9, plus=True, opptype=enc.OPP_UMV, uui="00"))["rc"] == ERR_INVALID_BITSTREAM
    # custom picture clock: CPCFC, and two more TR bits (picture.rs:711-723)
    h = pl.parse_header(header_bytes(176, 144, 0, 4, plus=True, opptype=enc.OPP_CUSTOM_PCF, temporal_reference=0x12, etr=3))
    assert h["rc"] == 0 and h["temporal_reference"] == 0x312
    # extended pixel aspect ratio: two more bytes, neither may be zero (picture.rs:370-377)
    ok = header_bytes(64, 48, 0, 4, plus=True, par=15, epar=(4, 3))
    assert pl.parse_header(ok)["rc"] == 0 and pl.parse_header(ok)["bits_used"] == 17 + 5 + 8 + 8 + 3 + 18 + 9 + 1 + 23 + 16 + 5 + 1
    assert pl.parse_header(header_bytes(64, 48, 0, 4, plus=True, par=15, epar=(0, 3)))["rc"] == ERR_FORMAT_INVALID
    assert pl.parse_header(header_bytes(64, 48, 0, 4, plus=True, par=0))["rc"] == ERR_FORMAT_INVALID
    # slice structured submode, reference picture selection followers
    h = pl.parse_header(header_bytes(176, 144, 1, 4, plus=True, opptype=enc.OPP_SS | enc.OPP_RPS))
    assert h["rc"] == 0 and h["bits_used"] == 17 + 5 + 8 + 8 + 3 + 18 + 9 + 1 + 2 + 3 + 1 + 2 + 5 + 1
    # ELNUM / RLNUM only with the decoder's scalability option (picture.rs:732-736)
    sc = header_bytes(176, 144, 1, 4, plus=True, scalability=True)
    assert pl.parse_header(sc, options=2)["bits_used"] == 17 + 5 + 8 + 8 + 3 + 18 + 9 + 1 + 8 + 5 + 1
    # picture types of MPPTYPE
    for code, want in [(0, 0), (1, 1), (2, 5), (3, 6), (4, 7), (5, 8), (6, 9), (7, 9)]:
        assert pl.parse_header(header_bytes(176, 144, 0, 4, plus=True, mpp_type=code))["picture_type"] == want


def test_plusptype_errors():
    assert pl.parse_header(header_bytes(176, 144, 0, 4, plus=True, ufep=2))["rc"] == ERR_INVALID_PLUSPTYPE
    # OPPTYPE must end in 1000, MPPTYPE in 001 (picture.rs:163-165, 238-240)
    good = bytearray(header_bytes(176, 144, 0, 4, plus=True))
    start = 17 + 5 + 8 + 8 + 3
    for bit in (start + 14, start + 18 + 8):
        bad = bytearray(good)
        bad[bit // 8] ^= 0x80 >> (bit % 8)
        assert pl.parse_header(bytes(bad))["rc"] == ERR_INVALID_PLUSPTYPE
    # CPFMT marker bit (picture.rs:357-359)
    cp = bytearray(header_bytes(200, 120, 0, 4, plus=True))
    bit = start + 18 + 9 + 1 + 13
    cp[bit // 8] ^= 0x80 >> (bit % 8)
    assert pl.parse_header(bytes(cp))["rc"] == ERR_FORMAT_INVALID
    # reference picture resampling is a stub in the reference (picture.rs:540-545)
    assert pl.parse_header(header_bytes(176, 144, 1, 4, plus=True, mpptype_flags=0x020))["rc"] == ERR_UNIMPLEMENTED
    # back-channel messages likewise (picture.rs:523-524): BCI = 1
    bw = enc.BitWriter()
    enc.write_standard_header(bw, 176, 144, 1, 4, plus=True, opptype=enc.OPP_RPS)
    s = "".join(bw.bits)
    i = 17 + 5 + 8 + 8 + 3 + 18 + 9 + 1 + 3 + 1          # ... RPSMF, TRPI, then BCI
    assert s[i:i + 2] == "01"
    assert pl.parse_header(bits(s[:i] + "1" + s[i + 2:]))["rc"] == ERR_UNIMPLEMENTED
    assert pl.parse_header(bits(s[:i] + "00" + s[i + 2:]))["rc"] == ERR_INVALID_BITSTREAM


def test_format_carry_over_rules_of_the_state():
    """state.rs:157-167 and picture.rs:760-769, including the reference's consequence that a PLUSPTYPE picture without
    OPPTYPE can never follow a picture whose header stated a format."""
    w, h = 176, 144
    im, ic = recgen.intra_picture(w, h, seed=5, max_level=30)
    im = make_codable(im, 6, 5, 0)
    i_pic = enc.encode_picture(w, h, 0, 6, im, ic, standard={})
    p_noformat = header_bytes(w, h, 1, 6, plus=True, ufep=0)
    i_noformat = header_bytes(w, h, 0, 6, plus=True, ufep=0)
    pl.context_reset()
    # no format anywhere: PictureFormatMissing for I and for a P without a last picture
    assert pl.parse_picture(i_noformat, options=0, use_context=True)[0] == ERR_FORMAT_MISSING
    assert pl.parse_picture(p_noformat, options=0, use_context=True)[0] == ERR_FORMAT_MISSING
    assert pl.parse_picture(i_pic, options=0, use_context=True)[0] == 0
    # now the previous header has Some(QCIF) and this one None: "format changed" -> RPRP -> UnimplementedDecoding
    assert pl.parse_picture(p_noformat, options=0, use_context=True)[0] == ERR_UNIMPLEMENTED
    # the same format restated is fine; a different one is a format change again
    pm, pc = recgen.inter_picture(w, h, seed=6, mv_range=20, quant=6, max_level=40)
    pm = make_codable(pm, 6, 6, 1)
    assert pl.parse_picture(enc.encode_picture(w, h, 1, 6, pm, pc, standard={}), options=0, use_context=True)[0] == 0
    assert pl.parse_picture(header_bytes(128, 96, 1, 6), options=0, use_context=True)[0] == ERR_UNIMPLEMENTED
    # SubQcif through PTYPE vs 128x96 through CPFMT are different SourceFormat values (derived PartialEq)
    pl.context_reset()
    assert pl.parse_picture(header_bytes(128, 96, 0, 6), options=0, use_context=True)[0] == 0
    assert pl.parse_picture(header_bytes(128, 96, 1, 6, plus=True, custom_format=True), options=0, use_context=True)[0] == ERR_UNIMPLEMENTED
    # an error leaves the context alone: the QCIF-less state still accepts SubQcif
    assert pl.parse_picture(header_bytes(128, 96, 1, 6), options=0, use_context=True)[0] == 0
    pl.context_reset()


# --- whole pictures ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h,std", [(176, 144, {}), (128, 96, {}), (352, 288, {"plus": True}), (200, 120, {"plus": True}),
                                     (64, 48, {"plus": True, "par": 2})])
def test_standard_pictures_round_trip_through_the_parser(w, h, std):
    im, ic = recgen.intra_picture(w, h, seed=w, max_level=120)            # 8-bit escape levels (block.rs:699)
    im = make_codable(im, 7, w, 0)
    rc, d, got, gco, used = pl.parse_picture(enc.encode_picture(w, h, 0, 7, im, ic, temporal_reference=9, standard=std), options=0)
    assert rc == 0 and (d.width, d.height, d.picture_type, d.pquant, d.use_deblocker, d.temporal_reference) == (w, h, 0, 7, 0, 9)
    assert_records_equal(got, im)
    want = ic.copy()
    want[:, 0] = 0
    assert (gco == want).all()
    pm, pc = recgen.inter_picture(w, h, seed=h, mv_range=32, p_4v=0.3, p_intra=0.1, p_coded=0.4, quant=9, max_level=127,
                                  sparse_low=False)
    pm = make_codable(pm, 9, h, 1)
    rc, d, got, _, _ = pl.parse_picture(enc.encode_picture(w, h, 1, 9, pm, pc, standard=std), options=0)
    assert rc == 0 and d.picture_type == 1
    assert_records_equal(got, pm)


@pytest.mark.parametrize("uui,mv_range", [("01", 32), ("1", 64)])
def test_annex_d_vectors(uui, mv_range):
    """OPPTYPE UMV bit: vectors are read with read_umv (macroblock.rs:424-430); the result still wraps at the
    standard range unless UUI says "extended" (mvd_pred.rs:84-112: 64 half-pels up to CIF width/height)."""
    w, h = 176, 144
    pm, pc = recgen.inter_picture(w, h, seed=21, mv_range=mv_range, p_4v=0.3, p_coded=0.3, quant=8, max_level=30)
    pm = make_codable(pm, 8, 3, 1)
    std = {"plus": True, "opptype": enc.OPP_UMV, "uui": uui, "umv": True}
    rc, d, got, _, _ = pl.parse_picture(enc.encode_picture(w, h, 1, 8, pm, pc, standard=std), options=0)
    assert rc == 0
    assert_records_equal(got, pm)
    if mv_range == 64:
        assert np.abs(pm["mv"]).max() > 32                               # the extended range was really used


def test_plain_ptype_option_bits_never_reach_the_macroblock_layer():
    """state.rs:147-155 with `running_options` never written: UMV / SAC / AP of a plain PTYPE are dropped, so the
    vectors of such a picture are ordinary Table 14 codes."""
    w, h = 176, 144
    pm, pc = recgen.inter_picture(w, h, seed=4, mv_range=32, p_coded=0.3, quant=8, max_level=30)
    pm = make_codable(pm, 8, 4, 1)
    data = enc.encode_picture(w, h, 1, 8, pm, pc, standard={"ptype_low": 0x8 | 0x2})
    rc, _, got, _, _ = pl.parse_picture(data, options=0)
    assert rc == 0
    assert_records_equal(got, pm)


def test_unimplemented_picture_kinds():
    w, h = 176, 144
    pm, pc = recgen.inter_picture(w, h, seed=4, mv_range=16, p_coded=0.3, quant=8, max_level=30)
    pm = make_codable(pm, 8, 4, 1)
    pm[0]["mv"] = 2                                                         # make sure the first macroblock is not a COD = 1 one
    # PB pictures: MCBPC of anything but I / P is UnimplementedDecoding (macroblock.rs:461-465)
    assert pl.parse_picture(enc.encode_picture(w, h, 1, 8, pm, pc, standard={"pb": True}), options=0)[0] == ERR_UNIMPLEMENTED
    assert pl.parse_picture(enc.encode_picture(w, h, 1, 8, pm, pc, standard={"plus": True, "mpp_type": 3}), options=0)[0] == ERR_UNIMPLEMENTED
    # modified quantisation (Annex T): macroblock.rs:497-498
    assert pl.parse_picture(enc.encode_picture(w, h, 1, 8, pm, pc, standard={"plus": True, "opptype": enc.OPP_MQ}), options=0)[0] == ERR_UNIMPLEMENTED
    # ... but a B picture made of COD = 1 macroblocks only never reaches MCBPC and decodes as a copy.  93 of them end
    # the data on a byte boundary (75 header bits + 93): the reference has no "picture full" exit, padding bits would
    # be read as one more macroblock (state.rs:193-417)
    skip = np.zeros(93, pm.dtype)
    skip["quant"] = 8
    data = enc.encode_picture(w, h, 1, 8, skip, pc[:0], standard={"plus": True, "mpp_type": 3})
    assert len(data) * 8 == 75 + 93
    rc, d, got, _, _ = pl.parse_picture(data, options=0)
    assert rc == 0 and d.picture_type == 6 and len(got) == 93 and (got["cbp"] == 0).all()


def test_macroblock_errors_resynchronise_only_in_standard_mode():
    """state.rs:387-408 + gob.rs:20-41: InvalidMacroblockHeader looks for a start code; a picture start code (or none at
    all) ends the picture, a real GOB header is UnimplementedDecoding.  Sorenson streams fail right away."""
    w, h = 128, 96
    im, ic = recgen.intra_picture(w, h, seed=8, max_level=30)
    im = make_codable(im, 6, 8, 0)
    good = enc.encode_picture(w, h, 0, 6, im[:5], ic, standard={})

    n_bits = pl.parse_picture(good, options=0)[4]
    good_bits = "".join(format(b, "08b") for b in good)[:n_bits]

    def stream(tail_bits):
        return bits(good_bits + tail_bits)

    # (a) nine zeros are no MCBPC-I code (Table 7); garbage follows, no start code within the realignment window of
    # the failed macroblock: InvalidGobHeader, which ends the picture
    rc, _, got, _, used = pl.parse_picture(stream("000000000" + "1011011101" * 4), options=0)
    assert rc == 0 and len(got) == 5 and used == n_bits
    # a start code further away than that window is not found either (recognize_start_code(false), reader.rs:244-262)
    far = "000000000" + "1" * 8 + "0" * (-(n_bits + 17) % 8) + "0" * 16 + "1" + "00011" + "0" * 16
    assert pl.parse_picture(stream(far), options=0)[0] == 0
    # (b) stuffing zeros up to the byte boundary and a picture start code: the zeros are the invalid MCBPC, the
    # resynchronisation finds the code and ends the picture
    pad = "0" * (-n_bits % 8)
    rc, _, got, _, used = pl.parse_picture(stream(pad + "0" * 16 + "1" + "00000" + "0" * 16), options=0)
    assert rc == 0 and len(got) == 5 and used == n_bits
    # (c) GOB number 15 is treated like a picture start (gob.rs:34); (d) any other GOB header is a stub
    assert pl.parse_picture(stream(pad + "0" * 16 + "1" + "01111" + "0" * 16), options=0)[0] == 0
    assert pl.parse_picture(stream(pad + "0" * 16 + "1" + "00011" + "0" * 16), options=0)[0] == ERR_UNIMPLEMENTED
    # (e) the data ends while looking: EOF ends the picture
    assert pl.parse_picture(stream("000000000"), options=0)[0] == 0
    # Sorenson: the same damage is an error (state.rs:387 `!self.is_sorenson()`)
    sgood = enc.encode_picture(w, h, 0, 6, im[:5], ic)
    sbits = pl.parse_picture(sgood)[4]
    bw = enc.BitWriter()
    bw.bits.append("".join(format(b, "08b") for b in sgood)[:sbits])
    bw.code("000000000" + "1011011101" * 4)
    assert pl.parse_picture(bw.tobytes())[0] == ERR_INVALID_MB_HEADER


# --- GPU: the same streams through H263State::decode_next_picture -------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("w,h,std", [(176, 144, {}), (200, 120, {"plus": True})])
def test_standard_stream_decodes_like_the_oracle(w, h, std):
    import h263mi
    st = h263mi.H263State(0)
    assert not st.is_sorenson()
    im, ic = recgen.intra_picture(w, h, seed=w, max_level=120)
    im = make_codable(im, 8, 1, 0)
    data = enc.encode_picture(w, h, 0, 8, im, ic, temporal_reference=1, standard=std)
    hdr = st.parse_picture(data)
    assert (hdr.width, hdr.height, hdr.picture_type, hdr.pquant) == (w, h, 0, 8)
    st.decode_next_picture(data)
    rc, ref = orc.decode_picture(w, h, im, ic, None)
    for g, e in zip(st.get_last_picture().as_yuv(), ref):
        assert (g == e).all()
    for f in range(2, 4):
        pm, pc = recgen.inter_picture(w, h, seed=f + h, mv_range=32, p_4v=0.3, p_intra=0.1, p_coded=0.3, quant=8, max_level=100)
        pm = make_codable(pm, 8, f, 1)
        st.decode_next_picture(enc.encode_picture(w, h, 1, 8, pm, pc, temporal_reference=f, standard=std))
        rc, ref = orc.decode_picture(w, h, pm, pc, ref)
        pic = st.get_last_picture()
        assert pic.temporal_reference == f
        for g, e in zip(pic.as_yuv(), ref):
            assert (g == e).all()
    # a B picture made of COD = 1 macroblocks only never reaches the unimplemented MCBPC branch: it decodes as a copy of
    # the last picture and, not being "disposable", becomes the reference like any other (state.rs:464-480)
    bstd = dict(std, plus=True, mpp_type=3)       # OPPTYPE restates the same SourceFormat: no "format change"
    hdr = enc.BitWriter()
    enc.write_standard_header(hdr, w, h, 1, 8, 9, **bstd)
    hdr_bits = sum(len(x) for x in hdr.bits)
    skip = np.zeros(8 + (-hdr_bits) % 8, pm.dtype)  # end on a byte boundary: padding bits would be read as a coded macroblock
    skip["quant"] = 8
    st.decode_next_picture(enc.encode_picture(w, h, 1, 8, skip, pc[:0], temporal_reference=9, standard=bstd))
    pic = st.get_last_picture()
    assert pic.picture_type == h263mi.PICTURE_B and pic.temporal_reference == 9
    for g, e2 in zip(pic.as_yuv(), ref):
        assert (g == e2).all()
    # a format change is the reference's RPRP stub; the state keeps its picture
    with pytest.raises(h263mi.H263Error) as e:
        st.decode_next_picture(enc.encode_picture(128, 96, 1, 8, pm[:1], pc, standard={}))
    assert e.value.code == h263mi.ERR_UNIMPLEMENTED_DECODING
    for g, e2 in zip(st.get_last_picture().as_yuv(), ref):
        assert (g == e2).all()
    st.close()
