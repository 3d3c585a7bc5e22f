"""Soft-float restatement of the reference's dequantisation + block classification + IDCT.

TEST INFRASTRUCTURE ONLY (like everything under oracle/).  A THIRD implementation of
h263/src/decoder/cpu/rle.rs:112-171 and idct.rs:39-201, beside the C oracle (h263_oracle.c) and the numpy
restatement (np_restatement.py), that does not touch the host FPU at all: IEEE-754 binary32 multiply and add are
carried out on integers with an explicit round-to-nearest-even step.  Two forms:

  * scalar, on Python ints of unlimited width (exact products and exact aligned sums, then ONE rounding) -- the
    specification; also offers the two arithmetic MUTATIONS the parity tests must be able to detect: a fused
    multiply-add (`fma`: round(a*b + c) once, what -ffp-contract=fast or an MFMA would do) and a pairwise
    (tree) summation of the eight products of idct_1d;
  * vectorised over blocks with numpy INTEGER arrays (uint64 products, guard/round/sticky alignment) -- the same
    arithmetic fast enough for 10^5 blocks; no numpy float type is used anywhere.

The reference cannot be built here (no Rust toolchain) and holds no test for these functions, so agreement of three
independently written implementations -- one of them FPU-free -- is what pins the reconstruction half of the oracle.
"""
import numpy as np

# ---- idct.rs:39-48 BASIS_TABLE as binary32 bit patterns -----------------------------------
BASIS_BITS = [
    [0x3F3504F3] * 8,
    [0x3F7B14BE, 0x3F54DB31, 0x3F0E39D9, 0x3E47C5BC, 0xBE47C5C2, 0xBF0E39DC, 0xBF54DB32, 0xBF7B14BF],
    [0x3F6C835E, 0x3EC3EF15, 0xBEC3EF18, 0xBF6C8360, 0xBF6C835E, 0xBEC3EF0B, 0x3EC3EF1B, 0x3F6C835F],
    [0x3F54DB31, 0xBE47C5C2, 0xBF7B14BF, 0xBF0E39D6, 0x3F0E39D7, 0x3F7B14BE, 0x3E47C5B1, 0xBF54DB34],
    [0x3F3504F3, 0xBF3504F3, 0xBF3504F1, 0x3F3504F7, 0x3F3504F3, 0xBF3504FB, 0xBF3504EF, 0x3F3504F4],
    [0x3F0E39D9, 0xBF7B14BF, 0x3E47C5C8, 0x3F54DB2D, 0xBF54DB34, 0xBE47C57C, 0x3F7B14BF, 0xBF0E39D7],
    [0x3EC3EF15, 0xBF6C835E, 0x3F6C8362, 0xBEC3EF25, 0xBEC3EF23, 0x3F6C835B, 0xBF6C8362, 0x3EC3EF25],
    [0x3E47C5BC, 0xBF0E39D6, 0x3F54DB2D, 0xBF7B14BD, 0x3F7B14BE, 0xBF54DB3A, 0x3F0E39E9, 0xBE47C596],
]
HALF, QUARTER, ONE, ZERO = 0x3F000000, 0x3E800000, 0x3F800000, 0x00000000

# rle.rs:6-71 DEZIGZAG_MAPPING as raster index x + 8*y per zigzag position
DEZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
            28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
            54, 47, 55, 62, 63]


# =================================================================================================================
# scalar binary32 on Python ints
# =================================================================================================================
def _unpack(bits):
    """bits -> (sign, m, e) with |value| = m * 2**e, m an integer (0 for zeros)"""
    sign, exp, frac = bits >> 31, (bits >> 23) & 0xFF, bits & 0x7FFFFF
    if exp == 0xFF:
        raise ValueError("inf / nan never occurs on this path")
    if exp == 0:
        return sign, frac, -149
    return sign, frac | 0x800000, exp - 150


def _round_pack(sign, m, e):
    """round-to-nearest-even of (-1)**sign * m * 2**e (m >= 0 any width) to binary32 bits"""
    if m == 0:
        return sign << 31
    shift = max(m.bit_length() - 24, -149 - e)
    if shift > 0:
        q, rem, half = m >> shift, m & ((1 << shift) - 1), 1 << (shift - 1)
        if rem > half or (rem == half and (q & 1)):
            q += 1
        if q == 1 << 24:
            q, shift = q >> 1, shift + 1
    else:
        q = m << -shift
    x = e + shift                                        # value = q * 2**x, q < 2**24
    if q >= 1 << 23:
        if x + 150 >= 255:
            raise OverflowError("binary32 overflow never occurs on this path")
        return (sign << 31) | ((x + 150) << 23) | (q & 0x7FFFFF)
    assert x == -149                                      # subnormal (or zero after rounding)
    return (sign << 31) | q


def f32_mul(a, b):
    sa, ma, ea = _unpack(a)
    sb, mb, eb = _unpack(b)
    return _round_pack(sa ^ sb, ma * mb, ea + eb)


def _signed_sum(terms):
    """exact sum of (sign, m, e) terms -> (sign, m, e); an exact zero is +0 unless every term is -0 (IEEE, RNE)"""
    e = min(t[2] for t in terms)
    total = sum((-t[1] if t[0] else t[1]) << (t[2] - e) for t in terms)
    if total == 0:
        return (1 if all(t[0] and t[1] == 0 for t in terms) else 0), 0, e
    return (1, -total, e) if total < 0 else (0, total, e)


def f32_add(a, b):
    return _round_pack(*_signed_sum([_unpack(a), _unpack(b)]))


def f32_fma(a, b, c):
    """round(a*b + c) with ONE rounding"""
    sa, ma, ea = _unpack(a)
    sb, mb, eb = _unpack(b)
    return _round_pack(*_signed_sum([(sa ^ sb, ma * mb, ea + eb), _unpack(c)]))


def f32_from_int(n):
    return _round_pack(1 if n < 0 else 0, abs(n), 0)


def f32_signum(a):
    """f32::signum: 1.0 for +0.0 and positive numbers, -1.0 for -0.0 and negative numbers"""
    return ONE | (a & 0x80000000)


def f32_to_i16(a):
    """Rust `as i16`: truncation toward zero, saturating"""
    s, m, e = _unpack(a)
    v = (m << e) if e >= 0 else (m >> -e)
    v = -v if s else v
    return max(-32768, min(32767, v))


def idct_1d(inp, mode="reference"):
    """idct.rs:52-65: out[i] = 0.0; for freq in 0..8 { out[i] += in[freq] * BASIS_TABLE[freq][i] }
    mode "fma": every step fused; mode "pairwise": the eight rounded products summed as a balanced tree"""
    out = []
    for i in range(8):
        if mode == "pairwise":
            p = [f32_mul(inp[f], BASIS_BITS[f][i]) for f in range(8)]
            while len(p) > 1:
                p = [f32_add(p[k], p[k + 1]) for k in range(0, len(p), 2)]
            out.append(f32_add(ZERO, p[0]))
            continue
        acc = ZERO
        for f in range(8):
            acc = f32_fma(inp[f], BASIS_BITS[f][i], acc) if mode == "fma" else f32_add(acc, f32_mul(inp[f], BASIS_BITS[f][i]))
        out.append(acc)
    return out


def _clip(v_bits, pre=None):
    """((x / 4.0 + v.signum() * 0.5) as i16).clamp(-256, 255) with x = v (Full) or v * pre (Dc: 0.5; Horiz / Vert:
    BASIS_TABLE[0][0]); the signum is the un-scaled value's (idct.rs:119-120, 138-140, 160-162, 189-190).  Dividing
    by 4.0 and multiplying by 0.25 are the same binary32 operation (no underflow here)."""
    x = f32_mul(v_bits, pre) if pre is not None else v_bits
    t = f32_add(f32_mul(x, QUARTER), f32_mul(f32_signum(v_bits), HALF))
    return max(-256, min(255, f32_to_i16(t)))


def _wrap_i16(v):
    """i16 arithmetic of a Rust release build (no overflow checks): modulo 2^16, signed"""
    return ((v + 32768) & 0xffff) - 32768


def dequant(level, quant):
    """rle.rs:130-133 the way a release build of the reference executes it: every step an i16 that wraps"""
    m = _wrap_i16(quant * _wrap_i16(_wrap_i16(2 * _wrap_i16(abs(level))) + 1))
    if quant % 2 == 0:
        m = _wrap_i16(m - 1)
    sg = (level > 0) - (level < 0)
    return max(-2048, min(2047, _wrap_i16(sg * m)))


def classify(coeffs):
    """coeffs: 64 dequantised integers in raster order x + 8*y (0 = absent).  rle.rs:138-171 -> (tag, values)"""
    is_horiz = all(coeffs[y * 8 + x] == 0 for y in range(1, 8) for x in range(8))
    is_vert = all(coeffs[y * 8 + x] == 0 for y in range(8) for x in range(1, 8))
    if is_horiz and is_vert:
        return ("zero", None) if coeffs[0] == 0 else ("dc", coeffs[0])
    if is_horiz:
        return "horiz", coeffs[0:8]
    if is_vert:
        return "vert", [coeffs[y * 8] for y in range(8)]
    return "full", coeffs


def block_residual(coeffs, mode="reference", force_full=False):
    """clipped_idct of every pixel of one block as res[y][x] (idct.rs:108-196); coeffs: 64 integers, raster order"""
    tag, val = classify(coeffs)
    if force_full and tag != "zero":
        tag, val = "full", coeffs
    if tag == "zero":
        return [[0] * 8 for _ in range(8)]
    if tag == "dc":
        c = _clip(f32_from_int(val), HALF)
        return [[c] * 8 for _ in range(8)]
    if tag == "horiz":
        row = idct_1d([f32_from_int(v) for v in val], mode)
        return [[_clip(row[x], BASIS_BITS[0][0]) for x in range(8)] for _ in range(8)]
    if tag == "vert":
        col = idct_1d([f32_from_int(v) for v in val], mode)
        return [[_clip(col[y], BASIS_BITS[0][0])] * 8 for y in range(8)]
    inter = [[ZERO] * 8 for _ in range(8)]
    for row in range(8):                                  # idct.rs:171-177 (with the transposition)
        o = idct_1d([f32_from_int(coeffs[row * 8 + x]) for x in range(8)], mode)
        for i in range(8):
            inter[i][row] = o[i]
    out = [idct_1d(inter[row], mode) for row in range(8)]
    # idct.rs:183-196: idct_output[x_offset][y_offset]
    return [[_clip(out[x][y]) for x in range(8)] for y in range(8)]


# =================================================================================================================
# vectorised binary32 on numpy integer arrays (no float dtype anywhere)
# =================================================================================================================
U64, I64 = np.uint64, np.int64


def _vunpack(bits):
    bits = bits.astype(U64)
    sign = (bits >> U64(31)).astype(I64)
    exp = ((bits >> U64(23)) & U64(0xFF)).astype(I64)
    frac = (bits & U64(0x7FFFFF))
    assert not (exp == 255).any()
    m = np.where(exp == 0, frac, frac | U64(0x800000))
    e = np.where(exp == 0, I64(1), exp)                  # value = m * 2**(e - 150)
    return sign, m, e


def _vbitlen(m):
    """bit length of uint64 values < 2**63 without a float: binary search on integer thresholds"""
    n = np.zeros(m.shape, I64)
    x = m.copy()
    for s in (32, 16, 8, 4, 2, 1):
        big = x >= (U64(1) << U64(s))
        n += np.where(big, I64(s), I64(0))
        x = np.where(big, x >> U64(s), x)
    return n + (x > 0).astype(I64)


def _vround_pack(sign, m, e, sticky=None):
    """RNE of m * 2**(e - 150) (m: uint64 < 2**62; `sticky`: bits already shifted out below m) -> bits"""
    L = _vbitlen(m)
    shift = np.maximum(L - 24, 1 - e)                    # keep e + shift >= 1 (subnormal range)
    rs = np.clip(shift, 0, 62).astype(U64)
    q = np.where(shift > 0, m >> rs, m << np.clip(-shift, 0, 62).astype(U64))
    rem = np.where(shift > 0, m & ((U64(1) << rs) - U64(1)), U64(0))
    half = np.where(shift > 0, U64(1) << (np.maximum(rs, U64(1)) - U64(1)), U64(0))
    st = np.zeros(m.shape, bool) if sticky is None else sticky
    up = (shift > 0) & ((rem > half) | ((rem == half) & (st | ((q & U64(1)) == U64(1)))))
    q = q + up.astype(U64)
    carry = q == U64(1 << 24)
    q = np.where(carry, q >> U64(1), q)
    x = e + shift + carry.astype(I64)
    normal = q >= U64(1 << 23)
    assert not (normal & (x >= 255)).any()
    out = np.where(normal, (x.astype(U64) << U64(23)) | (q & U64(0x7FFFFF)), q)
    out = np.where(m == 0, U64(0), out)
    return (out | (sign.astype(U64) << U64(31))).astype(np.uint32)


def vf32_mul(a, b):
    sa, ma, ea = _vunpack(a)
    sb, mb, eb = _vunpack(b)
    # ma*mb < 2**48; value = ma*mb * 2**(ea + eb - 300) = (ma*mb) * 2**((ea + eb - 150) - 150)
    return _vround_pack(sa ^ sb, ma * mb, ea + eb - 150)


def vf32_add(a, b):
    sa, ma, ea = _vunpack(a)
    sb, mb, eb = _vunpack(b)
    swap = (eb > ea) | ((eb == ea) & (mb > ma))           # make a the operand of larger magnitude
    sa, sb = np.where(swap, sb, sa), np.where(swap, sa, sb)
    ma, mb = np.where(swap, mb, ma), np.where(swap, ma, mb)
    ea, eb = np.where(swap, eb, ea), np.where(swap, ea, eb)
    G = U64(30)                                           # guard bits: the smaller operand is shifted inside a 54-bit window
    d = np.minimum(ea - eb, 60).astype(U64)
    big = ma << G
    small_full = mb << G
    small = small_full >> d
    lost = (small_full & ((U64(1) << d) - U64(1))) != U64(0)
    same = sa == sb
    # subtraction with lost bits: borrow one unit so that the sticky information stays below the kept bits
    total = np.where(same, big + small, big - small - (lost & ~same).astype(U64))
    sticky = lost
    # value = total * 2**(ea - 150 - 30)  ->  exponent argument of _vround_pack is e with m * 2**(e - 150)
    bits = _vround_pack(sa, total, ea - 30, sticky)
    # exact zero: +0 unless both operands are -0 (only possible without lost bits)
    zero = (total == 0) & ~lost
    both_neg_zero = (ma == 0) & (mb == 0) & (sa == 1) & (sb == 1)
    return np.where(zero, np.where(both_neg_zero, np.uint32(0x80000000), np.uint32(0)), bits).astype(np.uint32)


def vf32_from_int(n):
    n = np.asarray(n, I64)
    sign = (n < 0).astype(I64)
    return _vround_pack(sign, np.abs(n).astype(U64), np.full(n.shape, 150, I64))


def vf32_to_i16(a):
    s, m, e = _vunpack(a)
    sh = e - 150
    v = np.where(sh >= 0, m << np.clip(sh, 0, 40).astype(U64), m >> np.clip(-sh, 0, 63).astype(U64)).astype(I64)
    v = np.where(-sh > 63, 0, v)
    v = np.where(s == 1, -v, v)
    return np.clip(v, -32768, 32767)


_BASIS_NP = np.array(BASIS_BITS, np.uint32)


def vidct_1d(inp):
    """inp: uint32 bits [..., 8] -> [..., 8]; sequential accumulation from +0.0 like idct.rs:59-63"""
    acc = np.zeros(inp.shape, np.uint32)
    for f in range(8):
        prod = vf32_mul(np.broadcast_to(inp[..., f:f + 1], inp.shape), np.broadcast_to(_BASIS_NP[f], inp.shape))
        acc = vf32_add(acc, prod)
    return acc


def vclip(v_bits, pre=None):
    x = vf32_mul(v_bits, np.full(v_bits.shape, pre, np.uint32)) if pre is not None else v_bits
    sig_half = (v_bits & np.uint32(0x80000000)) | np.uint32(HALF)            # signum(v) * 0.5 is exactly +-0.5
    t = vf32_add(vf32_mul(x, np.full(v_bits.shape, QUARTER, np.uint32)), sig_half)
    return np.clip(vf32_to_i16(t), -256, 255)


def vfull_residual(coeffs):
    """coeffs: int array [n, 64] (raster, dequantised) treated as Full blocks -> residual int array [n, 8(y), 8(x)]
    and the binary32 bits of t = v / 4 + signum(v) * 0.5 per pixel (for near-tie searches)"""
    c = vf32_from_int(np.asarray(coeffs, I64).reshape(-1, 8, 8))               # [n, row, x]
    rows = vidct_1d(c)                                                         # [n, row, i]
    inter = np.swapaxes(rows, 1, 2).copy()                                     # [n, i, row]
    out = vidct_1d(inter)                                                      # [n, x_offset, y_offset]
    res = vclip(out)
    return np.swapaxes(res, 1, 2).copy(), np.swapaxes(out, 1, 2).copy()
