#!/usr/bin/env python3
"""Named known answers for the Vert class of the IDCT (idct.rs:152-169) that a macroblock record can REACH.

SURVEY appendix B.2 names three first-coefficient columns on which the Vert arithmetic and the Full arithmetic of the
reference differ.  Only the first of them, [19,0,0,0,-7,0,0,0], can come out of the dequantiser: rle.rs:130-133 gives
q*(2|L|+1) - (q even), which is ODD for every quantiser and LEVEL, and an intra DC is a multiple of 8 (types.rs:955-961),
so the columns with -38, -34 and -2 cannot be built from LEVELs.  This script searches the reachable neighbourhood
instead -- two odd values in column 0, quantiser 1 (value = 2|L|+1) -- for columns on which the two classes differ, and
writes a handful of them with BOTH expectations to tests/golden/vert_named_columns.json:

    levels      the LEVELs (quantiser 1) at (x=0, y=0) and (x=0, y=r)
    column      the dequantised first column
    vert        the eight residuals (one per pixel row) of the Vert class  -- what the decoder must produce
    full        the eight residuals the Full arithmetic would give         -- what a decoder that ignores the class produces

The expectations come from the C oracle AND the FPU-free soft-float model (oracle/softfloat_idct.py); the script refuses
to write a column on which the two disagree.  tests/test_oracle_recon.py re-checks the file against both on the CPU,
tests/test_gpu_round3.py decodes the columns on the MI355X.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc            # noqa: E402
from oracle import softfloat_idct as sf     # noqa: E402


def residuals(tag, col):
    v = np.zeros((1, 64), np.float32)
    if tag == orc.ORC_VERT:
        v[0, :8] = col
    else:
        v[0, ::8] = col
    out = orc.idct_blocks([tag], v, np.full(64, 128, np.uint8), 1, 8).astype(int) - 128
    rows = out.reshape(8, 8)
    assert (rows == rows[:, :1]).all()                # a first-column block: every pixel of a row gets the same value
    return rows[:, 0].tolist()


def softfloat_rows(col, force_full=False):
    co = [0] * 64
    for r, v in enumerate(col):
        co[8 * r] = int(v)                             # raster x + 8y: column 0
    res = sf.block_residual(co, force_full=force_full)
    assert all(len(set(row)) == 1 for row in res)
    return [int(row[0]) for row in res]


def main():
    named = [[19, 0, 0, 0, -7, 0, 0, 0]]               # appendix B.2, reachable: LEVELs 9 and -3 at quantiser 1
    for a, b in ((-37, -33), (-39, 37), (-41, -3), (-35, 15), (-37, 9), (-39, 5)):
        col = [0] * 8
        col[0], col[4] = a, b
        named.append(col)
    out = []
    for col in named:
        vert, full = residuals(orc.ORC_VERT, col), residuals(orc.ORC_FULL, col)
        assert vert != full, col
        assert vert == softfloat_rows(col), (col, vert, softfloat_rows(col))
        assert full == softfloat_rows(col, force_full=True), (col, full)
        assert all(v == 0 or (v % 2 and abs(v) >= 3) for v in col), col      # |q(2|L|+1) - (q even)| >= 3
        levels = {str(r): (abs(v) - 1) // 2 * (1 if v > 0 else -1) for r, v in enumerate(col) if v}
        out.append({"levels_at_quant_1": levels, "column": col, "vert": vert, "full": full})
    path = os.path.join(ROOT, "tests", "golden", "vert_named_columns.json")
    with open(path, "w") as f:
        json.dump({"source": "tools/gen_vert_columns.py: C oracle (idct.rs:152-169 restated) cross-checked with the "
                             "soft-float model; prediction-free residuals per pixel row",
                   "columns": out}, f, indent=1)
    print("wrote", path, len(out), "columns")


if __name__ == "__main__":
    main()
