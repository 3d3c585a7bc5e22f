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
