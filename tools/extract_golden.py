#!/usr/bin/env python3
"""Extract the known-answer DATA held by the reference's own in-source tests into
small JSON fixtures under tests/golden/.

Run in the build container only (it reads /root/reference, which does not exist on
the GPU box); the JSON it writes is committed.  Only inputs and expected outputs are
extracted -- no reference source text is kept.

Sources:
  yuv/src/bt601.rs:198-483      -> tests/golden/bt601_reference_tests.json
  deblock/src/deblock.rs:320-558 -> tests/golden/deblock_reference_tests.json
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def strip_comments(src):
    return re.sub(r"//[^\n]*", "", src)


def ints(text):
    return [int(v) for v in re.findall(r"-?\d+", re.sub(r"u8|usize|i16", "", text))]


def balanced(src, start, open_ch, close_ch):
    """Return (inner_text, end_index) of the bracket group opening at src[start]."""
    assert src[start] == open_ch, (src[start - 10:start + 10])
    depth = 0
    for i in range(start, len(src)):
        if src[i] == open_ch:
            depth += 1
        elif src[i] == close_ch:
            depth -= 1
            if depth == 0:
                return src[start + 1:i], i
    raise ValueError("unbalanced")


def extract_bt601():
    src = strip_comments(open(os.path.join(REF, "yuv/src/bt601.rs")).read())
    test_src = src[src.index("fn test_yuv_to_rgb()"):]
    out = {"source": "yuv/src/bt601.rs:198-483 (ruffle-rs/h263-rs @ 2025-04-10)"}

    # single-pixel known answers: assert_eq!(yuv_to_rgb((y, cb, cr)), (r, g, b));
    single = []
    for m in re.finditer(r"assert_eq!\(\s*yuv_to_rgb\(\((\d+),\s*(\d+),\s*(\d+)\)\),\s*\((\d+),\s*(\d+),\s*(\d+)\)\s*\)", test_src):
        v = [int(x) for x in m.groups()]
        single.append({"yuv": v[:3], "rgb": v[3:]})
    out["single_pixel"] = single

    # the test-only inverse: assert_eq!(rgb_to_yuv((r,g,b)), (y,u,v))
    inv = []
    for m in re.finditer(r"assert_eq!\(\s*rgb_to_yuv\(\((\d+),\s*(\d+),\s*(\d+)\)\),\s*\((\d+),\s*(\d+),\s*(\d+)\)", test_src):
        v = [int(x) for x in m.groups()]
        inv.append({"rgb": v[:3], "yuv": v[3:]})
    out["rgb_to_yuv"] = inv

    # round trips: assert_eq!(yuv_to_rgb(rgb_to_yuv((r,g,b))), (r2,g2,b2))
    rt = []
    for m in re.finditer(r"assert_eq!\(\s*yuv_to_rgb\(rgb_to_yuv\(\((\d+),\s*(\d+),\s*(\d+)\)\)\),\s*\((\d+),\s*(\d+),\s*(\d+)\)", test_src):
        v = [int(x) for x in m.groups()]
        rt.append({"rgb_in": v[:3], "rgb_out": v[3:]})
    out["roundtrip_exact"] = rt

    # tab10 palette (tolerance +-1 round trip)
    pal_start = test_src.index("for rgb in [")
    pal_txt, _ = balanced(test_src, test_src.index("[", pal_start), "[", "]")
    pal = ints(pal_txt)
    out["roundtrip_pm1_palette"] = [pal[i:i + 3] for i in range(0, len(pal), 3)]

    # whole-picture cases: assert_eq!( yuv420_to_rgba(&[..], &[..], &[..], w), vec![..] );
    pics = []
    pos = 0
    while True:
        k = test_src.find("yuv420_to_rgba(", pos)
        if k < 0:
            break
        args, end = balanced(test_src, k + len("yuv420_to_rgba"), "(", ")")
        pos = end
        # three &[...] groups then width
        groups = []
        p = 0
        for _ in range(3):
            b = args.index("[", p)
            inner, e = balanced(args, b, "[", "]")
            groups.append(ints(inner))
            p = e + 1
        width = ints(args[p:])[0]
        # expected: next "vec![" after the call
        v = test_src.index("vec!", end)
        exp_txt, vend = balanced(test_src, test_src.index("[", v), "[", "]")
        if ";" in exp_txt:  # vec![0u8; 0]
            val, cnt = exp_txt.split(";")
            expected = ints(val) * ints(cnt)[0]
        else:
            expected = ints(exp_txt)
        pos = vend
        pics.append({"y": groups[0], "cb": groups[1], "cr": groups[2], "y_width": width, "rgba": expected})
    out["pictures"] = pics
    return out


def extract_deblock():
    src = strip_comments(open(os.path.join(REF, "deblock/src/deblock.rs")).read())
    out = {"source": "deblock/src/deblock.rs:320-558 (ruffle-rs/h263-rs @ 2025-04-10)"}
    tbl = re.search(r"QUANT_TO_STRENGTH: \[u8; 32\] = \[(.*?)\];", src, re.S)
    out["quant_to_strength"] = ints(tbl.group(1))

    t = src[src.index("fn test_process()"):src.index("fn test_deblock()")]
    rows = []
    for m in re.finditer(r"\(\((\d+),\s*(\d+),\s*(\d+),\s*(\d+)\),\s*(\d+),\s*\((\d+),\s*(\d+),\s*(\d+),\s*(\d+)\)\)", t):
        v = [int(x) for x in m.groups()]
        rows.append({"in": v[:4], "strength": v[4], "out": v[5:]})
    out["process_rows"] = rows

    t = src[src.index("fn test_deblock()"):]

    def arr(name):
        k = t.index("let %s:" % name)
        b = t.index("= &[", k) + 3
        inner, _ = balanced(t, b, "[", "]")
        return ints(inner)

    out["image"] = {"width": 11, "data": arr("data"),
                    "expected": {"4": arr("expected_4"), "8": arr("expected_8"), "12": arr("expected_12")}}
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    bt = extract_bt601()
    db = extract_deblock()
    assert len(bt["single_pixel"]) == 11, len(bt["single_pixel"])   # 10 in test_yuv_to_rgb + 1 in tiny test
    assert len(bt["pictures"]) == 10, len(bt["pictures"])
    assert len(db["process_rows"]) == 37, len(db["process_rows"])
    assert len(db["image"]["data"]) == 11 * 17
    for k in ("4", "8", "12"):
        assert len(db["image"]["expected"][k]) == 11 * 17
    for p in bt["pictures"]:
        assert len(p["rgba"]) == 4 * len(p["y"]), p
    with open(os.path.join(OUT, "bt601_reference_tests.json"), "w") as f:
        json.dump(bt, f, indent=1)
    with open(os.path.join(OUT, "deblock_reference_tests.json"), "w") as f:
        json.dump(db, f, indent=1)
    print("bt601: %d single, %d inv, %d roundtrip, %d palette, %d pictures" % (
        len(bt["single_pixel"]), len(bt["rgb_to_yuv"]), len(bt["roundtrip_exact"]),
        len(bt["roundtrip_pm1_palette"]), len(bt["pictures"])))
    print("deblock: %d process rows, image %d px" % (len(db["process_rows"]), len(db["image"]["data"])))


if __name__ == "__main__":
    main()


# ---------------------------------------------------------------------------------------------
# parser known answers: h263/src/parser/{reader,macroblock,block}.rs tests
# ---------------------------------------------------------------------------------------------
def _fn_body(src, name):
    k = src.index("fn %s()" % name)
    b = src.index("{", k)
    inner, _ = balanced(src, b, "{", "}")
    return inner


def _byte_array(body, var):
    k = body.index("let %s" % var)
    b = body.index("[", body.index("=", k))
    inner, _ = balanced(body, b, "[", "]")
    out = []
    for tok in re.findall(r"0b[01_]+|0x[0-9A-Fa-f]+|\d+", inner):
        out.append(int(tok.replace("_", ""), 0))
    return out


def extract_parser():
    out = {"source": "h263/src/parser/{reader,macroblock,block}.rs in-source tests (ruffle-rs/h263-rs @ 2025-04-10)"}
    blk = strip_comments(open(os.path.join(REF, "h263/src/parser/block.rs")).read())
    mb = strip_comments(open(os.path.join(REF, "h263/src/parser/macroblock.rs")).read())

    # --- TCOEF table: every code word of Table 16/H.263 (block.rs:768-1705)
    body = _fn_body(blk, "tcoef_table")
    seq = []
    for m in re.finditer(r"assert_eq!\(\s*reader\.read_vlc\(&TCOEF_TABLE\)\.unwrap\(\),\s*(.*?)\s*\);", body, re.S):
        e = re.sub(r"\s+", " ", m.group(1))
        r = re.search(r"last: (true|false), run: (\d+), level: (\d+)", e)
        if r:
            seq.append([r.group(1) == "true", int(r.group(2)), int(r.group(3))])
        elif "EscapeToLong" in e:
            seq.append("escape")
        elif e.strip() == "None":
            seq.append(None)
        else:
            raise ValueError(e)
    out["tcoef_table"] = {"bytes": _byte_array(body, "bit_pattern"), "expected": seq}

    # --- whole-block decodes (block.rs:1707-2124)
    blocks = []
    for name in ("empty_inter_block", "empty_intra_block", "long_coded_inter_block", "long_coded_intra_block",
                 "short_coded_inter_block", "short_coded_intra_block", "sorenson_long_coded_intra_block",
                 "sorenson_xlong_coded_intra_block"):
        body = _fn_body(blk, name)
        case = {"name": name, "bytes": _byte_array(body, "bitstream")}
        case["version"] = int(re.search(r"version:\s*(None|Some\((\d+)\))", body).group(2) or -1) \
            if "Some(" in re.search(r"version:\s*(None|Some\(\d+\))", body).group(1) else None
        case["sorenson"] = "SORENSON_SPARK_BITSTREAM" in body
        case["intra"] = "MacroblockType::Intra" in body
        m = re.search(r"decode_block\((.*?)\)\s*\.unwrap", body, re.S)
        args = [a.strip() for a in re.sub(r"\s+", " ", m.group(1)).split(",")]
        case["tcoef_present"] = args[-1] == "true"
        m = re.search(r"intradc:\s*(None|IntraDc::from_level\((0x[0-9A-Fa-f]+|\d+)\)|IntraDc::from_u8\((0x[0-9A-Fa-f]+|\d+)\))", body)
        if m.group(1) == "None":
            case["intradc_level"] = None
        elif m.group(2):
            case["intradc_level"] = int(m.group(2), 0)
        else:
            c = int(m.group(3), 0)
            case["intradc_level"] = 1024 if c == 255 else c << 3
        case["tcoef"] = [[t[0] == "true", int(t[1]), int(t[2])] for t in
                         re.findall(r"is_short:\s*(true|false),\s*run:\s*(\d+),\s*level:\s*(-?\d+)", body)]
        blocks.append(case)
    out["blocks"] = blocks

    # --- macroblock layer tables (macroblock.rs:559-1010)
    def table(fn, tbl, conv):
        body = _fn_body(mb, fn)
        seq = []
        for m in re.finditer(r"assert_eq!\(\s*reader\.read_vlc\(&%s(?:\[\.\.\])?\)\.unwrap\(\),\s*(.*?)\s*\);" % tbl, body, re.S):
            seq.append(conv(re.sub(r"\s+", " ", m.group(1)).strip()))
        return {"bytes": _byte_array(body, "bit_pattern"), "expected": seq}

    def conv_mcbpc(e):
        if "Stuffing" in e:
            return "stuffing"
        if "Invalid" in e:
            return "invalid"
        r = re.search(r"MacroblockType::(\w+), (true|false), (true|false)", e)
        return [r.group(1), r.group(2) == "true", r.group(3) == "true"]

    def conv_cbpy(e):
        if e == "None":
            return None
        return [v == "true" for v in re.findall(r"true|false", e)]

    def conv_mvd(e):
        if e == "None":
            return None
        return float(re.search(r"Some\((-?[\d.]+)\)", e).group(1))

    out["mcbpc_i"] = table("macroblock_mcbpc_iframe", "MCBPC_I_TABLE", conv_mcbpc)
    out["mcbpc_p"] = table("macroblock_mcbpc_pframe", "MCBPC_P_TABLE", conv_mcbpc)
    out["cbpy"] = table("macroblock_cbpy_table", "CBPY_TABLE_INTRA", conv_cbpy)
    out["mvd"] = table("macroblock_mvd_table", "MVD_TABLE", conv_mvd)
    return out


def main_parser():
    p = extract_parser()
    assert len(p["tcoef_table"]["expected"]) >= 100, len(p["tcoef_table"]["expected"])
    assert len(p["blocks"]) == 8
    assert len(p["mcbpc_i"]["expected"]) == 10 and len(p["mcbpc_p"]["expected"]) >= 22
    assert len(p["cbpy"]["expected"]) >= 16 and len(p["mvd"]["expected"]) >= 64
    with open(os.path.join(OUT, "parser_reference_tests.json"), "w") as f:
        json.dump(p, f, indent=1)
    print("parser: tcoef %d, blocks %d, mcbpc_i %d, mcbpc_p %d, cbpy %d, mvd %d" % (
        len(p["tcoef_table"]["expected"]), len(p["blocks"]), len(p["mcbpc_i"]["expected"]),
        len(p["mcbpc_p"]["expected"]), len(p["cbpy"]["expected"]), len(p["mvd"]["expected"])))


if __name__ == "__main__":
    main_parser()
