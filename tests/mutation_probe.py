"""Decodes the mutation fixture (tests/golden/idct_sensitive_blocks.json) with whichever build of the library
H263MI_LIB names and saves the luma plane.  Run as a child process by tests/test_gpu_mutation.py: a process can load
one build of the library only.

Picture: one macroblock per fixture block.  Frame 0: intra, INTRADC code 255 everywhere (level 1024 -> every pixel
128).  Frame 1: inter, zero vectors, block Y0 of macroblock k carries fixture block k (cbp = 1, its own quantiser):
that 8x8 area becomes clamp(128 + residual)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "h263-rs_amd"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
import h263mi  # noqa: E402

MB_COLS = 10


def fixture():
    return json.load(open(os.path.join(HERE, "golden", "idct_sensitive_blocks.json")))["blocks"]


def picture_size(n_blocks):
    rows = (n_blocks + MB_COLS - 1) // MB_COLS
    return MB_COLS * 16, rows * 16


def records(blocks):
    w, h = picture_size(len(blocks))
    total = (w // 16) * (h // 16)
    intra = np.zeros(total, h263mi.MB_RECORD_DTYPE)
    intra["mb_type"] = 3
    intra["quant"] = 1
    intra["intradc"] = 255
    inter = np.zeros(total, h263mi.MB_RECORD_DTYPE)
    inter["quant"] = 1
    coeffs = np.zeros((len(blocks), 64), np.int16)
    for k, b in enumerate(blocks):
        inter[k]["quant"] = b["quant"]
        inter[k]["cbp"] = 1
        inter[k]["coeff_index"] = k
        coeffs[k] = b["levels"]
    return w, h, intra, inter, coeffs


def decode_luma():
    blocks = fixture()
    w, h, intra, inter, coeffs = records(blocks)
    st = h263mi.H263State()
    st.submit_picture(w, h, intra, np.zeros((0, 64), np.int16), h263mi.PICTURE_I)
    flat = st.get_last_picture().as_yuv()
    assert all((p == 128).all() for p in flat), "the flat prediction is not 128"
    st.submit_picture(w, h, inter, coeffs, h263mi.PICTURE_P, temporal_reference=1)
    y = st.get_last_picture().as_luma().reshape(h, w).copy()
    st.close()
    return y


if __name__ == "__main__":
    np.save(sys.argv[1], decode_luma())
