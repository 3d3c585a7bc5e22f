#!/usr/bin/env python3
"""Finds 8x8 blocks on which an arithmetic MUTATION of the IDCT changes a reconstructed pixel, with the FPU-free
soft-float model (oracle/softfloat_idct.py), and writes them to tests/golden/idct_sensitive_blocks.json.

The reference IDCT (idct.rs:52-65) rounds every product and every partial sum to binary32, in the order of the
frequency index.  Two implementations a GPU port is tempted by give slightly different sums: fusing the multiply
into the add (v_fma_f32 / v_pk_fma_f32 / MFMA: -ffp-contract=fast) and summing the eight products as a tree.  The
final `(v / 4 + signum(v) / 2) as i16` hides almost all of those differences; this search finds the blocks where it
does not, so that tests/test_gpu_mutation.py can prove that the GPU parity suite would catch either mutation.

Blocks are inter blocks (LEVELs + quantiser, dequantised per rle.rs:130-133) meant to be decoded over a flat
prediction of 128, so residuals in [-128, 127] are visible in the output.

usage: python tools/find_sensitive_blocks.py [n_blocks=400000] [seed=1]     (several minutes on 8 cores)
"""
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import softfloat_idct as sf  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "idct_sensitive_blocks.json")


def make_blocks(rng, n):
    levels = np.zeros((n, 64), np.int64)
    quant = rng.integers(1, 32, n)
    for b in range(n):
        k = int(rng.integers(2, 14))
        pos = rng.choice(64, k, replace=False)
        amp = int(rng.choice([3, 8, 20, 40]))
        lv = rng.integers(-amp, amp + 1, k)
        lv[lv == 0] = 1
        levels[b, pos] = lv
        # make sure it is a Full block: something off the first row and off the first column
        if not (levels[b].reshape(8, 8)[1:, 1:] != 0).any():
            levels[b, 9 + int(rng.integers(0, 7))] = 1 + int(rng.integers(0, amp))
    return levels, quant


def dequant_all(levels, quant):
    q = quant[:, None]
    m = q * (2 * np.abs(levels) + 1) - (q % 2 == 0)
    return np.clip(np.where(levels > 0, m, np.where(levels < 0, -m, 0)), -2048, 2047)


def scan(args):
    seed, n = args
    rng = np.random.default_rng(seed)
    levels, quant = make_blocks(rng, n)
    co = dequant_all(levels, quant)
    res, out = sf.vfull_residual(co)
    # t = v / 4 + signum(v) / 2 in binary32; near-tie = within 6 ulp of an integer, |t| in (1, 127)
    quarter = np.full(out.shape, sf.QUARTER, np.uint32)
    t = sf.vf32_add(sf.vf32_mul(out, quarter), (out & np.uint32(0x80000000)) | np.uint32(sf.HALF))
    e = ((t >> np.uint32(23)) & np.uint32(0xFF)).astype(np.int64)
    m = ((t & np.uint32(0x7FFFFF)) | np.uint32(0x800000)).astype(np.int64)
    fb = np.clip(150 - e, 1, 30)
    frac = m & ((np.int64(1) << fb) - 1)
    dist = np.minimum(frac, (np.int64(1) << fb) - frac)
    near = (dist <= 6) & (e >= 127) & (e <= 133)
    found = []
    for b in np.flatnonzero(near.any(axis=(1, 2))):
        c = [int(x) for x in co[b]]
        ref = sf.block_residual(c, "reference")
        assert ref == res[b].tolist()
        entry = None
        for mode in ("fma", "pairwise"):
            mut = sf.block_residual(c, mode)
            diff = [(y, x) for y in range(8) for x in range(8) if mut[y][x] != ref[y][x] and -128 <= ref[y][x] <= 127 and
                    -128 <= mut[y][x] <= 127]
            if diff:
                entry = entry or {"quant": int(quant[b]), "levels": [int(v) for v in levels[b]], "residual": ref, "detects": {}}
                entry["detects"][mode] = [[y, x, mut[y][x]] for y, x in diff]
        if entry:
            found.append(entry)
    return found


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    chunk = 5000
    jobs = [(seed * 100000 + i, chunk) for i in range(n // chunk)]
    found = []
    with ProcessPoolExecutor(8) as ex:
        for part in ex.map(scan, jobs):
            found += part
            print("blocks so far: %d (fma %d, pairwise %d)" % (len(found), sum("fma" in f["detects"] for f in found),
                                                                 sum("pairwise" in f["detects"] for f in found)), flush=True)
    fma = [f for f in found if "fma" in f["detects"]][:48]
    pw = [f for f in found if "pairwise" in f["detects"] and f not in fma][:48]
    doc = {"note": "Full-class inter blocks (raster-order LEVELs + quantiser) on which a fused multiply-add (`fma`) or a "
                   "pairwise summation (`pairwise`) in idct_1d changes the clipped residual of at least one pixel; "
                   "`residual`[y][x] is the reference arithmetic's (idct.rs:52-65, 171-196), `detects`[mutation] lists "
                   "[y, x, mutated residual].  Found by tools/find_sensitive_blocks.py with the FPU-free soft-float "
                   "model (oracle/softfloat_idct.py), seed %d, %d random blocks." % (seed, n),
           "blocks": fma + pw}
    json.dump(doc, open(OUT, "w"), separators=(",", ":"))
    print("wrote %s: %d blocks (%d detect fma, %d detect pairwise)" % (OUT, len(doc["blocks"]),
          sum("fma" in f["detects"] for f in doc["blocks"]), sum("pairwise" in f["detects"] for f in doc["blocks"])))


if __name__ == "__main__":
    main()
