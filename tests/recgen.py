"""Seeded builders of macroblock-record pictures for the parity tests (numpy RNG).

These make inputs only; expected outputs always come from the oracle.
"""
import numpy as np

from oracle.oracle import MB_RECORD_DTYPE
from oracle.np_restatement import ZIGZAG_RASTER

INTER, INTERQ, INTER4V, INTRA, INTRAQ, INTER4VQ = range(6)
BLOCK_CLASSES = ("dc", "horiz", "vert", "full_dense", "full_sparse")


def mb_dims(w, h):
    return (w + 15) // 16, (h + 15) // 16


def random_intradc(rng, n):
    c = rng.integers(1, 255, n)        # 1..254
    c[c == 128] = 129
    return c.astype(np.uint8)


def fill_block(rng, cls, max_level=127, intra=True):
    """int16[64] raster coefficient block of the requested class."""
    c = np.zeros(64, np.int16)

    def lv(n):
        v = rng.integers(1, max_level + 1, n) * rng.choice([-1, 1], n)
        return v.astype(np.int16)

    if cls == "horiz":
        pos = np.arange(1, 8)
        pos = pos[rng.random(7) < 0.7]
        if pos.size == 0:
            pos = np.array([1])
        c[pos] = lv(pos.size)
    elif cls == "vert":
        pos = np.arange(1, 8) * 8
        pos = pos[rng.random(7) < 0.7]
        if pos.size == 0:
            pos = np.array([8])
        c[pos] = lv(pos.size)
    elif cls == "full_dense":
        c[1:] = lv(63)
        if not intra:
            c[0] = lv(1)[0]
    elif cls == "full_sparse":
        pos = rng.choice(np.arange(0 if not intra else 1, 64), 4, replace=False)
        c[pos] = lv(4)
    elif cls == "dc_coeff":              # inter block whose only coefficient sits at zigzag 0
        c[0] = lv(1)[0]
    return c


def intra_picture(w, h, seed, classes=BLOCK_CLASSES, max_level=127, quant=None, n_mbs=None):
    """All-intra picture (config 1 / config 2 'mixed'); returns (mbs, coeffs)."""
    rng = np.random.default_rng(seed)
    mbw, mbh = mb_dims(w, h)
    n = mbw * mbh if n_mbs is None else n_mbs
    mbs = np.zeros(n, MB_RECORD_DTYPE)
    coeffs = []
    for i in range(n):
        m = mbs[i]
        m["mb_type"] = INTRA if rng.random() < 0.8 else INTRAQ
        m["quant"] = quant if quant else rng.integers(1, 32)
        m["intradc"] = random_intradc(rng, 6)
        m["coeff_index"] = len(coeffs)
        cbp = 0
        for b in range(6):
            cls = classes[rng.integers(0, len(classes))]
            if cls == "dc":
                continue
            cbp |= 1 << b
            # keep q*(2|L|+1) inside i16 (SURVEY 8a1 contract)
            coeffs.append(fill_block(rng, cls, max_level, intra=True))
        m["cbp"] = cbp
    c = np.array(coeffs, np.int16).reshape(-1, 64) if coeffs else np.zeros((0, 64), np.int16)
    return mbs, c


def inter_picture(w, h, seed, mv_range=32, p_coded=0.25, p_4v=0.1, p_intra=0.0, quant=10,
                  max_level=31, n_mbs=None, sparse_low=True):
    """P picture (config 3): random half-pel MVs, sparse residuals; returns (mbs, coeffs)."""
    rng = np.random.default_rng(seed)
    mbw, mbh = mb_dims(w, h)
    n = mbw * mbh if n_mbs is None else n_mbs
    mbs = np.zeros(n, MB_RECORD_DTYPE)
    coeffs = []
    for i in range(n):
        m = mbs[i]
        intra = rng.random() < p_intra
        m["quant"] = quant if quant else rng.integers(1, 32)
        m["coeff_index"] = len(coeffs)
        if intra:
            m["mb_type"] = INTRA
            m["intradc"] = random_intradc(rng, 6)
        else:
            four = rng.random() < p_4v
            m["mb_type"] = INTER4V if four else INTER
            if four:
                m["mv"] = rng.integers(-mv_range, mv_range, (4, 2))
            else:
                m["mv"] = np.tile(rng.integers(-mv_range, mv_range, (1, 2)), (4, 1))
        cbp = 0
        for b in range(6):
            if rng.random() >= p_coded:
                continue
            cbp |= 1 << b
            c = np.zeros(64, np.int16)
            lo = 1 if intra else 0
            zz = rng.choice(np.arange(lo, 16 if sparse_low else 64), 4, replace=False)
            c[ZIGZAG_RASTER[zz]] = (rng.integers(1, max_level + 1, 4) * rng.choice([-1, 1], 4)).astype(np.int16)
            coeffs.append(c)
        m["cbp"] = cbp
    c = np.array(coeffs, np.int16).reshape(-1, 64) if coeffs else np.zeros((0, 64), np.int16)
    return mbs, c


def random_planes(w, h, seed):
    rng = np.random.default_rng(seed)
    cw, ch = (w + 1) // 2, (h + 1) // 2
    return (rng.integers(0, 256, w * h, dtype=np.uint8), rng.integers(0, 256, cw * ch, dtype=np.uint8),
            rng.integers(0, 256, cw * ch, dtype=np.uint8))


def realistic_inter_picture(w, h, seed, p_skip=0.6, p_coded=0.15, quant=10, p_halfpel=0.3):
    """A P picture shaped like real Sorenson Spark content (bench.py extra.e2e_bitstream_realistic): most macroblocks
    are not coded at all (COD = 1: Inter, zero vector, nothing coded), the others follow one slow global motion with a
    little jitter -- mostly integer-pel --, few blocks carry residuals, and those have two or three small low-frequency
    coefficients.  About 3 Mbit/s at 1080p30.  Returns (mbs, coeffs)."""
    rng = np.random.default_rng(seed)
    mbw, mbh = mb_dims(w, h)
    n = mbw * mbh
    mbs = np.zeros(n, MB_RECORD_DTYPE)
    mbs["quant"] = quant
    gmv = rng.integers(-3, 4, 2) * 2                          # global motion, whole pixels
    skip = rng.random(n) < p_skip
    # skipped macroblocks come in runs (backgrounds): smooth the mask along the rows
    skip = (np.convolve(skip.astype(np.float32), np.ones(5) / 5, mode="same") > 0.5) if n >= 5 else skip
    coeffs = []
    for i in range(n):
        m = mbs[i]
        m["coeff_index"] = len(coeffs)
        if skip[i]:
            continue
        mv = gmv + rng.integers(-1, 2, 2) * 2
        if rng.random() < p_halfpel:
            mv = mv + rng.integers(0, 2, 2)
        m["mv"] = np.tile(np.clip(mv, -31, 31), (4, 1))
        cbp = 0
        for b in range(6):
            if rng.random() >= p_coded:
                continue
            cbp |= 1 << b
            c = np.zeros(64, np.int16)
            k = int(rng.integers(2, 4))
            zz = rng.choice(np.arange(0, 6), k, replace=False)
            c[ZIGZAG_RASTER[zz]] = (rng.integers(1, 4, k) * rng.choice([-1, 1], k)).astype(np.int16)
            coeffs.append(c)
        m["cbp"] = cbp
    c = np.array(coeffs, np.int16).reshape(-1, 64) if coeffs else np.zeros((0, 64), np.int16)
    return mbs, c


def realistic_intra_picture(w, h, seed, quant=10, p_coded=0.7):
    """An I picture shaped like real content at a moderate quantiser (the key frame of bench.py's
    extra.e2e_bitstream_realistic): every block has its INTRADC, 70 % of the blocks carry one to six small
    low-frequency coefficients (two on average), none has a dense spectrum.  About 110 KB at 1080p, against the 2.3 MB of
    the mixed-class test picture.  Returns (mbs, coeffs)."""
    rng = np.random.default_rng(seed)
    mbw, mbh = mb_dims(w, h)
    n = mbw * mbh
    mbs = np.zeros(n, MB_RECORD_DTYPE)
    mbs["mb_type"] = INTRA
    mbs["quant"] = quant
    # INTRADC: a smooth field plus a little texture, kept inside the legal codes 1..254 without 128
    gy, gx = np.divmod(np.arange(n), mbw)
    field = 128 + 70 * np.sin(gx / 9.0 + seed) * np.cos(gy / 7.0) + rng.normal(0, 6, n)
    dc = np.clip(field[:, None] + rng.normal(0, 4, (n, 6)), 1, 254).astype(np.int64)
    dc[dc == 128] = 129
    dc[:, 4:] = np.clip(128 + (dc[:, 4:] - 128) // 4, 1, 254)
    dc[dc == 128] = 127
    mbs["intradc"] = dc.astype(np.uint8)
    coded = rng.random((n, 6)) < p_coded
    mbs["cbp"] = (coded * (1 << np.arange(6))).sum(axis=1).astype(np.uint8)
    n_blocks = int(coded.sum())
    counts = coded.sum(axis=1)
    mbs["coeff_index"] = (np.cumsum(counts) - counts).astype(np.uint32)
    co = np.zeros((n_blocks, 64), np.int16)
    k = np.minimum(rng.geometric(0.5, n_blocks), 6)
    for j in range(6):
        sel = np.flatnonzero(k > j)
        zz = rng.integers(1, 11, sel.size)                        # zigzag 1..10 (0 is the DC, carried by INTRADC)
        lv = (rng.integers(1, 4, sel.size) * rng.choice([-1, 1], sel.size)).astype(np.int16)
        co[sel, ZIGZAG_RASTER[zz]] = lv
    return mbs, co
