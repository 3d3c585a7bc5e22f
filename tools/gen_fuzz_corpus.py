#!/usr/bin/env python3
"""Seed corpus of the parser fuzzer (tests/parser/fuzz_parser.cpp) -> tests/golden/parser_fuzz_corpus.bin.

Coded pictures made by the test encoder (tests/sorenson_enc.py) from seeded synthetic records: Sorenson Spark I and P
pictures (custom 8- and 16-bit sizes and the fixed formats, INTER4V, intra macroblocks in P pictures,
DQUANT, stuffing, runs past zigzag 63, runs of COD = 1, 11-bit escape LEVELs), and ITU-T H.263 pictures (plain PTYPE,
PLUSPTYPE with a custom format, Annex D vectors, PEI bytes).  Small pictures: the fuzzer parses every mutated input
several times under AddressSanitizer.

File format: u32 count, then per seed: u32 decoder_options, u32 n_bytes, bytes.  All little endian.
The file is DATA produced by this repository's own encoder; regenerate with `python tools/gen_fuzz_corpus.py`."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "h263-rs_amd")]
import recgen  # noqa: E402
import sorenson_enc as enc  # noqa: E402
from test_bitstream_e2e import make_codable  # noqa: E402

SORENSON, STANDARD = 1, 0


def seeds():
    out = []
    for w, h, seed in ((48, 32, 1), (64, 48, 2), (100, 60, 3), (16, 16, 6), (33, 17, 7), (260, 20, 8)):
        q = 5 + seed
        mbs, co = recgen.intra_picture(w, h, seed=seed, max_level=120)
        mbs = make_codable(mbs, q, seed, 0)
        out.append((SORENSON, enc.encode_picture(w, h, 0, q, mbs, co, temporal_reference=seed)))
        mbs, co = recgen.inter_picture(w, h, seed=seed + 10, mv_range=32, p_4v=0.3, p_intra=0.15, p_coded=0.4, quant=q,
                                       max_level=100, sparse_low=False)
        mbs = make_codable(mbs, q, seed, 1)
        out.append((SORENSON, enc.encode_picture(w, h, 1, q, mbs, co)))
        out.append((SORENSON, enc.encode_picture(w, h, 1, q, mbs, co, stuffing_every=3)))
        coded = np.flatnonzero(mbs["cbp"] & 1)
        if coded.size:
            out.append((SORENSON, enc.encode_picture(w, h, 1, q, mbs, co, overflow_blocks={(int(coded[0]), 0)})))
        mbs, co = recgen.realistic_inter_picture(w, h, seed + 20, p_skip=0.75, p_coded=0.15)
        out.append((SORENSON, enc.encode_picture(w, h, 1, 9, make_codable(mbs, 9, seed, 1), co)))
    # 11-bit escape LEVELs at a large quantiser (parser/block.rs:694-708)
    mbs, co = recgen.intra_picture(64, 48, seed=31, max_level=1023, quant=31)
    out.append((SORENSON, enc.encode_picture(64, 48, 0, 31, make_codable(mbs, 31, 1, 0), co)))
    # a picture of nothing but COD = 1
    mbs, co = recgen.realistic_inter_picture(176, 144, 44, p_skip=1.0, p_coded=0.0)
    out.append((SORENSON, enc.encode_picture(176, 144, 1, 9, make_codable(mbs, 9, 2, 1), co)))
    mbs, co = recgen.inter_picture(176, 144, seed=52, mv_range=32, p_4v=0.2, p_intra=0.05, p_coded=0.2, quant=10, max_level=40)
    out.append((SORENSON, enc.encode_picture(176, 144, 1, 10, make_codable(mbs, 10, 3, 1), co)))
    # the densest stream there is: every block of every macroblock 64 coefficients of +-1 (run 0, level 1: the 3-bit code
    # words) -- 3.03 bits per event, next to the bound the caller's event arrays are sized by (bits::event_words_bound)
    for ptype in (1, 0):
        rng = np.random.default_rng(70 + ptype)
        mbs, _ = (recgen.inter_picture(48, 32, seed=71, mv_range=4, p_4v=0.0, p_intra=0.0, p_coded=1.0, quant=4, max_level=1)
                  if ptype else recgen.intra_picture(48, 32, seed=72, max_level=1, quant=4))
        mbs["cbp"] = 0x3f
        mbs["coeff_index"] = np.arange(len(mbs), dtype=np.uint32) * 6
        co = rng.choice(np.array([-1, 1], np.int16), (len(mbs) * 6, 64))
        out.append((SORENSON, enc.encode_picture(48, 32, 0 if ptype == 0 else 1, 4, make_codable(mbs, 4, 7, ptype), co)))
    # ITU-T H.263: Annex D vectors (OPPTYPE UMV, both UUI forms)
    for uui, rng_ in (("01", 32), ("1", 64)):
        mbs, co = recgen.inter_picture(176, 144, seed=21, mv_range=rng_, p_4v=0.3, p_coded=0.2, quant=8, max_level=30)
        std = {"plus": True, "opptype": enc.OPP_UMV, "uui": uui, "umv": True}
        out.append((STANDARD, enc.encode_picture(176, 144, 1, 8, make_codable(mbs, 8, 3, 1), co, standard=std)))
    # ITU-T H.263
    for w, h, std in (((128, 96), None, {"pei": [7, 200]}), ((64, 48), None, {"plus": True}), ((128, 96), None, {}),
                      ((64, 48), None, {"plus": True, "par": 2})):
        ww, hh = w
        for ptype in (0, 1):
            if ptype == 0:
                mbs, co = recgen.intra_picture(ww, hh, seed=61, max_level=100)
                mbs = make_codable(mbs, 8, 5, 0)
            else:
                mbs, co = recgen.inter_picture(ww, hh, seed=62, mv_range=30, p_4v=0.2, p_intra=0.1, p_coded=0.4, quant=8,
                                               max_level=100, sparse_low=False)
                mbs = make_codable(mbs, 8, 5, 1)
            try:
                out.append((STANDARD, enc.encode_picture(ww, hh, ptype, 8, mbs, co, standard=dict(std))))
            except Exception as e:          # (a header form the writer does not support at this size: not a seed)
                print("skipped a standard seed:", ww, hh, std, e)
    return out


def main():
    s = seeds()
    path = os.path.join(ROOT, "tests", "golden", "parser_fuzz_corpus.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("<I", len(s)))
        for opt, data in s:
            f.write(struct.pack("<II", opt, len(data)))
            f.write(data)
    print("%d seeds, %d bytes -> %s" % (len(s), os.path.getsize(path), path))


if __name__ == "__main__":
    main()
