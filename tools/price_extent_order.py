"""Pricing of one candidate, on the CPU: would ordering a wave's active blocks by coefficient extent before compaction
(recon_kernel.inl: recon_phase_compact) make the IDCT rounds' n_cols / n_rows maxima tighter?

A reconstruction wave owns 8 macroblocks of a row; its active blocks are compacted in task order (the 16 blocks of the upper
luma row, the 16 of the lower, 8 Cb, 8 Cr) and taken 8 to a round; a round's row pass runs to the LARGEST column extent of its
eight blocks and its column pass to the largest row extent (idct_1d_pairs: `n` terms, wave-uniform).  This script takes the
pictures of bench.py's workloads (the synthetic P distribution, the realistic streams of extra.e2e_bitstream_realistic),
computes every block's extents, and sums the terms the rounds execute (a) as the kernel orders them, (b) with the wave's blocks
sorted by extent first, (c) the floor: every block with its own extent.  Terms beyond the first cost 4 packed multiplies + 4
packed adds each (8 of the ~190 vector instructions of a round).
usage: python tools/price_extent_order.py"""
import os, sys
import numpy as np
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))     # (analysis tooling: the test generators)
import recgen

W, H = 1920, 1088
MBW, MBH = recgen.mb_dims(W, H)


def extents(co):
    """(n_rows, n_cols) per block: 1 + the largest row / column index that holds a non-zero coefficient (0 for none)."""
    nz = co.reshape(-1, 8, 8) != 0
    rows = nz.any(axis=2)
    cols = nz.any(axis=1)
    n_rows = np.where(rows.any(axis=1), 8 - np.argmax(rows[:, ::-1], axis=1), 0)
    n_cols = np.where(cols.any(axis=1), 8 - np.argmax(cols[:, ::-1], axis=1), 0)
    return n_rows, n_cols


def price(name, mbs, co, intra_dc):
    n_rows, n_cols = extents(co)
    as_is = sorted_ = floor = rounds = 0
    for y in range(MBH):
        for x0 in range(0, MBW, 8):
            ms = [y * MBW + x for x in range(x0, min(x0 + 8, MBW))]
            blocks = []                                     # in task order
            for group in ((0, 1), (2, 3), (4,), (5,)):      # upper luma row, lower luma row, Cb, Cr
                for m in ms:
                    rec = mbs[m]
                    base = int(rec["coeff_index"])
                    cbp = int(rec["cbp"])
                    for b in group:
                        if cbp >> b & 1:
                            k = base + bin(cbp & ((1 << b) - 1)).count("1")
                            blocks.append((max(int(n_rows[k]), 1), max(int(n_cols[k]), 1)))
                        elif intra_dc:
                            blocks.append((1, 1))           # an uncoded intra block is its INTRADC: a block of the Dc class
            if not blocks:
                continue
            def terms(bl):
                t = 0
                for i in range(0, len(bl), 8):
                    rnd = bl[i:i + 8]
                    t += max(r for r, _ in rnd) + max(c for _, c in rnd)
                return t
            as_is += terms(blocks)
            sorted_ += terms(sorted(blocks, key=lambda rc: (max(rc), rc)))
            floor += sum(r + c for r, c in blocks) / 8.0
            rounds += (len(blocks) + 7) // 8
    print("%-44s rounds %6d  terms as ordered %7d  sorted by extent %7d (%+.1f %%)  floor %9.0f (%+.1f %%)" % (
        name, rounds, as_is, sorted_, 100.0 * (sorted_ - as_is) / as_is, floor, 100.0 * (floor - as_is) / as_is))
    return as_is, sorted_


if __name__ == "__main__":
    mbs, co = recgen.realistic_inter_picture(W, H, 7001)
    price("realistic P picture (e2e_bitstream_realistic)", mbs, co, False)
    mbs, co = recgen.realistic_intra_picture(W, H, 300)
    price("realistic key frame", mbs, co, True)
    mbs, co = recgen.inter_picture(W, H, 11)
    price("mixed-class test P picture (tests)", mbs, co, False)
    # bench.py's timed P workload: 4 coefficients at zigzag positions < 16 in every coded block
    rng = np.random.default_rng(5)
    mbs, _ = recgen.inter_picture(W, H, 12, p_coded=0.4)
    nb = int(sum(bin(int(c)).count("1") for c in mbs["cbp"]))
    co = np.zeros((nb, 64), np.int16)
    for k in range(4):
        co[np.arange(nb), recgen.ZIGZAG_RASTER[rng.integers(0, 16, nb)]] = 3
    price("bench P distribution (4 coefficients, zz < 16)", mbs, co, False)
