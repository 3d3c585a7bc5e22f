// tests/fixture_enc/fixture_enc.cpp -- a Sorenson Spark bitstream WRITER and a picture generator for fixtures, in C++.
//
// TEST / BENCH INFRASTRUCTURE, never a product path.  The reference ships no sample streams (README.md:23) and has no
// encoder; end-to-end figures over thousands of DISTINCT 1080p pictures (bench.py: extra.e2e_bitstream_distinct) cannot wait
// for the Python writer (tests/sorenson_enc.py: 0.4-0.7 s per picture).  Two entry points:
//   fx_encode   records + coefficient blocks -> bytes.  The same layout, element by element, as tests/sorenson_enc.py's
//               encode_picture (Sorenson flavour): tests/test_fixture_enc.py holds the two to each other BYTE FOR BYTE, and
//               the parser's independent answers (tests/golden/*known_answers.json) pin what both must mean.
//               Layout per the reference's parser: picture header parser/picture.rs:619-659, macroblock layer
//               parser/macroblock.rs:445-549, block layer parser/block.rs:670-755, vectors as differences against the median
//               predictor of decoder/cpu/mvd_pred.rs:27-67.  Code tables: h263-rs_amd/host/vlc_tables.inc.
//   fx_picture  a deterministic picture "shaped like real content" for (seed, stream, frame): records + coefficients, and
//               its bytes through fx_encode.  P pictures: ~2/3 of the macroblocks not coded (in runs), slow global motion
//               with jitter, few small low-frequency residuals, now and then an intra or a four-vector macroblock and a
//               DQUANT; key frames: smooth INTRADC field, 70 % of the blocks with one to six small coefficients.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/h263mi.h"

namespace {

struct VlcCode {
    const char *bits;
    int16_t v0, v1, v2;
};
#include "../../h263-rs_amd/host/vlc_tables.inc"

struct Code {
    uint32_t bits = 0;
    uint8_t len = 0;
};
Code parse_code(const char *s)
{
    Code c;
    for (; *s; s++) {
        c.bits = (c.bits << 1) | (uint32_t)(*s == '1');
        c.len++;
    }
    return c;
}

struct Tables {
    Code tcoef[2][64][13];          // [last][run][|level|] (len 0 = escape)
    Code tcoef_escape;
    Code mcbpc_i[6][2][2], mcbpc_p[6][2][2], stuffing_p, stuffing_i;
    Code cbpy[16];
    Code mvd[64];                   // index = vector + 32
    Tables()
    {
        for (const VlcCode &c : kTcoefCodes) {
            if (c.v0 < 0) tcoef_escape = parse_code(c.bits);
            else if (c.v1 < 64 && c.v2 < 13) tcoef[c.v0][c.v1][c.v2] = parse_code(c.bits);
        }
        for (const VlcCode &c : kMcbpcICodes) {
            if (c.v0 < 0) stuffing_i = parse_code(c.bits);
            else mcbpc_i[c.v0][c.v1][c.v2] = parse_code(c.bits);
        }
        for (const VlcCode &c : kMcbpcPCodes) {
            if (c.v0 < 0) stuffing_p = parse_code(c.bits);
            else mcbpc_p[c.v0][c.v1][c.v2] = parse_code(c.bits);
        }
        for (const VlcCode &c : kCbpyCodes) cbpy[c.v0 & 15] = parse_code(c.bits);
        for (const VlcCode &c : kMvdCodes)
            if (c.v0 >= -32 && c.v0 <= 31) mvd[c.v0 + 32] = parse_code(c.bits);
    }
};
const Tables &tables()
{
    static const Tables t;
    return t;
}

// rle.rs:6-71 as raster indices
const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct BitWriter {
    uint8_t *out;
    size_t cap, n = 0;
    uint64_t acc = 0;
    int fill = 0;
    bool overflow = false;
    BitWriter(uint8_t *o, size_t c) : out(o), cap(c) {}
    void put(uint32_t value, int bits)
    {
        if (!bits) return;
        acc = (acc << bits) | (value & (bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u)));
        fill += bits;
        while (fill >= 8) {
            fill -= 8;
            if (n < cap) out[n] = (uint8_t)(acc >> fill);
            else overflow = true;
            n++;
        }
    }
    void code(const Code &c) { put(c.bits, c.len); }
    void finish()
    {
        if (fill) put(0, 8 - fill);              // zero padding to the byte boundary
    }
};

int median3(int a, int b, int c)
{
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    return c < lo ? lo : (c > hi ? hi : c);
}

typedef int16_t Mv4[4][2];

// mvd_pred.rs:27-67.  pv: the vectors of the macroblocks coded so far (n of them), cur: this macroblock's so far
void predict(const Mv4 *pv, size_t n, const Mv4 &cur, size_t mbw, int index, int out[2])
{
    const size_t col = n % mbw, line = n / mbw;
    int mv1[2] = {0, 0}, mv2[2], mv3[2];
    if (index == 0 || index == 2) {
        if (col != 0) { mv1[0] = pv[n - 1][index + 1][0]; mv1[1] = pv[n - 1][index + 1][1]; }
    } else {
        mv1[0] = cur[index - 1][0]; mv1[1] = cur[index - 1][1];
    }
    const size_t last_line = (line ? line - 1 : 0) * mbw + col;
    if (index <= 1) {
        if (line == 0) { mv2[0] = mv1[0]; mv2[1] = mv1[1]; }
        else { mv2[0] = pv[last_line][index + 2][0]; mv2[1] = pv[last_line][index + 2][1]; }
        if (col == mbw - 1) { mv3[0] = mv3[1] = 0; }
        else if (line == 0) { mv3[0] = mv1[0]; mv3[1] = mv1[1]; }
        else { mv3[0] = pv[last_line + 1][2][0]; mv3[1] = pv[last_line + 1][2][1]; }
    } else {
        mv2[0] = cur[0][0]; mv2[1] = cur[0][1];
        mv3[0] = cur[1][0]; mv3[1] = cur[1][1];
    }
    out[0] = median3(mv1[0], mv2[0], mv3[0]);
    out[1] = median3(mv1[1], mv2[1], mv3[1]);
}

// the difference in [-32, 31] that halfpel_decode (mvd_pred.rs:70-117) maps back to mv (mv and pred in [-32, 31])
int mvd_for(int mv, int pred)
{
    int d = mv - pred;
    if (d < -32) d += 64;
    else if (d > 31) d -= 64;
    return d;
}

void write_block(BitWriter &bw, const int16_t *coeff, bool intra, uint8_t intradc, bool coded)
{
    const Tables &t = tables();
    if (intra) bw.put(intradc, 8);
    if (!coded) return;
    // events: (run, level) in zigzag order; an intra block's scan starts behind the DC
    int runs[64], levels[64], n = 0, last = intra ? 1 : 0;
    for (int z = last; z < 64; z++) {
        const int lv = coeff[kZigzag[z]];
        if (lv) {
            runs[n] = z - last;
            levels[n] = lv;
            n++;
            last = z + 1;
        }
    }
    for (int i = 0; i < n; i++) {
        const int is_last = i == n - 1, run = runs[i], lv = levels[i], mag = lv < 0 ? -lv : lv;
        const Code *c = (mag <= 12) ? &t.tcoef[is_last][run][mag] : nullptr;
        if (c && c->len) {
            bw.code(*c);
            bw.put(lv < 0 ? 1u : 0u, 1);
        } else {
            bw.code(t.tcoef_escape);
            if (lv >= -64 && lv <= 63) {
                bw.put(0, 1); bw.put((uint32_t)is_last, 1); bw.put((uint32_t)run, 6); bw.put((uint32_t)lv, 7);
            } else {
                bw.put(1, 1); bw.put((uint32_t)is_last, 1); bw.put((uint32_t)run, 6); bw.put((uint32_t)lv, 11);
            }
        }
    }
}

// H263MI_OK, or H263MI_ERR_INVALID_ARGUMENT for records the syntax cannot express / a buffer that is too small
int encode(int width, int height, int picture_type, int pquant, int temporal_reference, int deblock_flag,
           const h263mi_mb_record *mbs, size_t n_mbs, const int16_t *coeffs, uint8_t *out, size_t cap, size_t *n_bytes)
{
    const Tables &t = tables();
    BitWriter bw(out, cap);
    bw.put(1, 17);
    bw.put(1, 5);                                    // version 1
    bw.put((uint32_t)temporal_reference, 8);
    static const int kFormats[5][3] = {{352, 288, 2}, {176, 144, 3}, {128, 96, 4}, {320, 240, 5}, {160, 120, 6}};
    int fmt = -1;
    for (const auto &f : kFormats)
        if (f[0] == width && f[1] == height) fmt = f[2];
    if (fmt >= 0) bw.put((uint32_t)fmt, 3);
    else if (width < 256 && height < 256) { bw.put(0, 3); bw.put((uint32_t)width, 8); bw.put((uint32_t)height, 8); }
    else { bw.put(1, 3); bw.put((uint32_t)width, 16); bw.put((uint32_t)height, 16); }
    bw.put((uint32_t)picture_type, 2);
    bw.put((uint32_t)deblock_flag, 1);
    bw.put((uint32_t)pquant, 5);
    bw.put(0, 1);                                    // PEI
    const size_t mbw = (size_t)(width + 15) / 16;
    int quant = pquant;
    std::vector<int16_t> pv_store(n_mbs * 8, 0);
    Mv4 *pv = reinterpret_cast<Mv4 *>(pv_store.data());
    static const int kDquant[5] = {1, 0, -1, 2, 3};  // dq -2 -> "01", -1 -> "00", +1 -> "10", +2 -> "11"
    for (size_t i = 0; i < n_mbs; i++) {
        const h263mi_mb_record &m = mbs[i];
        int type = m.mb_type;
        const int cbp = m.cbp;
        const bool intra = type == 3 || type == 4, four = type == 2 || type == 5;
        bool moving = false;
        for (int k = 0; k < 4; k++) moving = moving || m.mv[k][0] || m.mv[k][1];
        if (picture_type != 0) {
            if (!intra && cbp == 0 && !moving && m.quant == quant && type == 0) {
                bw.put(1, 1);                        // COD = 1: not coded (the vectors it leaves behind are zero)
                continue;
            }
            bw.put(0, 1);
        }
        const int dq = (int)m.quant - quant;
        if (dq < -2 || dq > 2) return H263MI_ERR_INVALID_ARGUMENT;
        type = intra ? (dq ? 4 : 3) : (four ? (dq ? 5 : 2) : (dq ? 1 : 0));
        const int cb = (cbp >> 4) & 1, cr = (cbp >> 5) & 1;
        const Code &mc = (picture_type == 0 ? t.mcbpc_i : t.mcbpc_p)[type][cb][cr];
        if (!mc.len) return H263MI_ERR_INVALID_ARGUMENT;
        bw.code(mc);
        const int luma = ((cbp & 1) << 3) | (((cbp >> 1) & 1) << 2) | (((cbp >> 2) & 1) << 1) | ((cbp >> 3) & 1);
        bw.code(t.cbpy[intra ? luma : (~luma & 15)]);
        if (dq) {
            bw.put((uint32_t)kDquant[dq + 2], 2);
            quant += dq;
        }
        Mv4 &cur = pv[i];
        if (!intra) {
            for (int k = 0; k < (four ? 4 : 1); k++) {
                int p[2];
                predict(pv, i, cur, mbw, k, p);
                for (int c = 0; c < 2; c++) {
                    const int v = m.mv[k][c];
                    if (v < -32 || v > 31) return H263MI_ERR_INVALID_ARGUMENT;
                    bw.code(t.mvd[mvd_for(v, p[c]) + 32]);
                }
                cur[k][0] = m.mv[k][0];
                cur[k][1] = m.mv[k][1];
            }
            if (!four)
                for (int k = 1; k < 4; k++) { cur[k][0] = cur[0][0]; cur[k][1] = cur[0][1]; }
        }
        size_t ci = m.coeff_index;
        for (int b = 0; b < 6; b++) {
            const bool coded = (cbp >> b) & 1;
            write_block(bw, coded ? coeffs + ci * 64 : nullptr, intra, m.intradc[b], coded);
            ci += coded ? 1 : 0;
        }
    }
    bw.finish();
    if (n_bytes) *n_bytes = bw.n;
    return bw.overflow ? H263MI_ERR_INVALID_ARGUMENT : H263MI_OK;
}

// ---- the generator ---------------------------------------------------------------------------------------------------
struct Rng {                                         // splitmix64 (counter based: a picture is a function of its key)
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next()
    {
        uint64_t z = (s += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    uint32_t below(uint32_t n) { return (uint32_t)((next() >> 32) * (uint64_t)n >> 32); }
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    int range(int lo, int hi) { return lo + (int)below((uint32_t)(hi - lo + 1)); }        // inclusive
};

void small_block(Rng &r, int16_t *c, bool intra, int max_events, int reach, int max_level)
{
    memset(c, 0, 64 * sizeof(int16_t));
    const int k = r.range(1, max_events), first = intra ? 1 : 0;
    for (int e = 0; e < k; e++) {
        const int z = first + (int)r.below((uint32_t)reach);
        int lv = r.range(1, max_level);
        if (r.below(2)) lv = -lv;
        c[kZigzag[z]] = (int16_t)lv;             // (a position drawn twice keeps the later level: still one event)
    }
}

}  // namespace

extern "C" {

int fx_encode(int width, int height, int picture_type, int pquant, int temporal_reference, int deblock_flag,
              const h263mi_mb_record *mbs, size_t n_mbs, const int16_t *coeffs, uint8_t *out, size_t cap, size_t *n_bytes)
{
    if (!mbs || !out || width <= 0 || height <= 0 || pquant < 1 || pquant > 31 || picture_type < 0 || picture_type > 2)
        return H263MI_ERR_INVALID_ARGUMENT;
    if (n_mbs > (size_t)((width + 15) / 16) * (size_t)((height + 15) / 16)) return H263MI_ERR_INVALID_ARGUMENT;
    return encode(width, height, picture_type, pquant, temporal_reference, deblock_flag, mbs, n_mbs, coeffs, out, cap, n_bytes);
}

// picture `frame` of stream `stream`: an I picture when intra != 0, else a P picture.  mbs: mbw * mbh records out;
// coeffs: coeff_cap_blocks * 64 levels out (6 * mbw * mbh blocks always suffice); out / cap: the bytes.
int fx_picture(uint64_t seed, uint32_t stream, uint32_t frame, int width, int height, int intra, int pquant, int deblock_flag,
               h263mi_mb_record *mbs, int16_t *coeffs, size_t coeff_cap_blocks, size_t *n_blocks, uint8_t *out, size_t cap,
               size_t *n_bytes)
{
    if (!mbs || !coeffs || !out || width <= 0 || height <= 0 || pquant < 1 || pquant > 31) return H263MI_ERR_INVALID_ARGUMENT;
    const size_t mbw = (size_t)(width + 15) / 16, mbh = (size_t)(height + 15) / 16, n = mbw * mbh;
    Rng r(seed * 0x100000001b3ull + ((uint64_t)stream << 32) + frame * 2654435761ull + (intra ? 1 : 0));
    memset(mbs, 0, n * sizeof *mbs);
    size_t used = 0;
    int quant = pquant;
    if (intra) {
        const double phase = (double)(seed % 997) + stream * 0.37;
        for (size_t i = 0; i < n; i++) {
            h263mi_mb_record &m = mbs[i];
            const size_t gx = i % mbw, gy = i / mbw;
            m.mb_type = H263MI_MB_INTRA;
            if (r.below(40) == 0) {                  // now and then a DQUANT
                const int nq = quant + r.range(-2, 2);
                if (nq >= 1 && nq <= 31 && nq != quant) { quant = nq; m.mb_type = H263MI_MB_INTRA_Q; }
            }
            m.quant = (uint8_t)quant;
            const double field = 128.0 + 70.0 * std::sin(gx / 9.0 + phase) * std::cos(gy / 7.0);
            for (int b = 0; b < 6; b++) {
                int dc = (int)(field + (r.unit() - 0.5) * 24.0);
                if (b >= 4) dc = 128 + (dc - 128) / 4;
                dc = dc < 1 ? 1 : (dc > 254 ? 254 : dc);
                if (dc == 128) dc = 129;
                m.intradc[b] = (uint8_t)dc;
            }
            m.coeff_index = (uint32_t)used;
            for (int b = 0; b < 6; b++) {
                if (r.unit() >= 0.7) continue;
                if (used >= coeff_cap_blocks) return H263MI_ERR_INVALID_ARGUMENT;
                small_block(r, coeffs + used * 64, true, r.below(4) ? 2 : 6, 10, 4);
                m.cbp |= (uint8_t)(1u << b);
                used++;
            }
        }
    } else {
        const int gmx = r.range(-3, 3) * 2, gmy = r.range(-3, 3) * 2;      // global motion, whole pixels
        bool skipping = r.below(3) != 0;
        for (size_t i = 0; i < n; i++) {
            h263mi_mb_record &m = mbs[i];
            m.mb_type = H263MI_MB_INTER;
            m.quant = (uint8_t)quant;
            m.coeff_index = (uint32_t)used;
            // not-coded macroblocks come in runs (backgrounds): a two-state chain, 2/3 of the time in "skip"
            if (skipping ? r.below(12) == 0 : r.below(6) == 0) skipping = !skipping;
            if (skipping) continue;
            if (r.below(50) == 0) {                  // an intra macroblock in a P picture
                m.mb_type = H263MI_MB_INTRA;
                for (int b = 0; b < 6; b++) {
                    int dc = r.range(40, 220);
                    if (dc == 128) dc = 129;
                    m.intradc[b] = (uint8_t)dc;
                }
            } else {
                const bool four = r.below(20) == 0;
                if (four) m.mb_type = H263MI_MB_INTER4V;
                for (int k = 0; k < 4; k++) {
                    if (k && !four) { m.mv[k][0] = m.mv[0][0]; m.mv[k][1] = m.mv[0][1]; continue; }
                    int vx = gmx + r.range(-1, 1) * 2, vy = gmy + r.range(-1, 1) * 2;
                    if (r.below(10) < 3) { vx += (int)r.below(2); vy += (int)r.below(2); }     // half-pel now and then
                    m.mv[k][0] = (int16_t)(vx < -31 ? -31 : (vx > 31 ? 31 : vx));
                    m.mv[k][1] = (int16_t)(vy < -31 ? -31 : (vy > 31 ? 31 : vy));
                }
            }
            const bool is_intra = m.mb_type == H263MI_MB_INTRA;
            for (int b = 0; b < 6; b++) {
                if (r.unit() >= 0.15) continue;
                if (used >= coeff_cap_blocks) return H263MI_ERR_INVALID_ARGUMENT;
                small_block(r, coeffs + used * 64, is_intra, 3, 6, 3);
                m.cbp |= (uint8_t)(1u << b);
                used++;
            }
            if (r.below(30) == 0) {                  // DQUANT on a coded macroblock
                const int nq = quant + r.range(-2, 2);
                if (nq >= 1 && nq <= 31 && nq != quant) {
                    quant = nq;
                    m.quant = (uint8_t)quant;
                    m.mb_type = is_intra ? H263MI_MB_INTRA_Q : (m.mb_type == H263MI_MB_INTER4V ? H263MI_MB_INTER4V_Q : H263MI_MB_INTER_Q);
                }
            }
        }
    }
    if (n_blocks) *n_blocks = used;
    return encode(width, height, intra ? 0 : 1, pquant, (int)(frame & 255u), deblock_flag, mbs, n, coeffs, out, cap, n_bytes);
}

}  // extern "C"
