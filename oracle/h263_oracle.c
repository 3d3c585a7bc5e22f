/*
 * h263_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the hot path of ruffle-rs/h263-rs.  Every function
 * cites the reference file:line it follows.  See h263_oracle.h for the pinning
 * status of each function.  MUST be compiled with -ffp-contract=off and without
 * -ffast-math: the reference IDCT is f32 with separately rounded multiply and
 * add (Rust never contracts), and bit-exactness depends on it.
 */
#include "h263_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* small helpers mirroring Rust semantics                                    */
/* ------------------------------------------------------------------------ */

static inline ptrdiff_t clamp_pd(ptrdiff_t v, ptrdiff_t lo, ptrdiff_t hi)
{
    return v < lo ? lo : (v > hi ? hi : v);
}

static inline int clamp_i(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* i16 arithmetic of a Rust release build: the result modulo 2^16, reinterpreted as signed */
static inline int16_t wrap_i16(int v)
{
    const uint16_t u = (uint16_t)((unsigned)v & 0xffffu);
    return (int16_t)(u >= 0x8000u ? (int)u - 0x10000 : (int)u);
}

/* Rust `f32 as i16`: truncate toward zero, saturate, NaN -> 0. */
static inline int16_t f32_as_i16(float f)
{
    if (f != f) return 0;
    if (f >= 32767.0f) return 32767;
    if (f <= -32768.0f) return -32768;
    return (int16_t)(int32_t)f;
}

/* Rust f32::signum: +1 for +0.0 / positive, -1 for -0.0 / negative. */
static inline float f32_signum(float f)
{
    union { float f; uint32_t u; } b;
    b.f = f;
    if (f != f) return f;
    return (b.u >> 31) ? -1.0f : 1.0f;
}

/* ------------------------------------------------------------------------ */
/* h263/src/types.rs                                                         */
/* ------------------------------------------------------------------------ */

/* types.rs:955-961  IntraDc::into_level */
int16_t orc_intradc_into_level(uint8_t code)
{
    if (code == 0xFF) return 1024;
    return (int16_t)((uint16_t)code << 3);
}

/* types.rs:721-729  HalfPel::into_lerp_parameters (Rust `/` and `%` truncate) */
void orc_lerp_parameters(int16_t hp, int16_t *delta, int *interp)
{
    if (hp % 2 == 0) {
        *delta = (int16_t)(hp / 2);
        *interp = 0;
    } else if (hp < 0) {
        *delta = (int16_t)(hp / 2 - 1);
        *interp = 1;
    } else {
        *delta = (int16_t)(hp / 2);
        *interp = 1;
    }
}

/* types.rs:759-768  HalfPel::average_sum_of_mvs */
int16_t orc_average_sum_of_mvs(int16_t sum)
{
    int16_t whole = (int16_t)((sum >> 4) * 2); /* (self.0 >> 4) << 1, arithmetic shift on i16 */
    int frac = sum & 0x0F;
    if (frac <= 2) return whole;
    if (frac >= 14) return (int16_t)(whole + 2);
    return (int16_t)(whole + 1);
}

/* types.rs:653-658  MacroblockType::is_inter */
static inline int mb_is_inter(uint8_t t) { return t == 0 || t == 1 || t == 2 || t == 5; }
static inline int mb_is_intra(uint8_t t) { return t == 3 || t == 4; }

/* ------------------------------------------------------------------------ */
/* h263/src/decoder/cpu/rle.rs                                               */
/* ------------------------------------------------------------------------ */

/* rle.rs:6-71  DEZIGZAG_MAPPING as (x, y) pairs */
static const uint8_t DEZIGZAG[64][2] = {
    {0,0},{1,0},{0,1},{0,2},{1,1},{2,0},{3,0},{2,1},
    {1,2},{0,3},{0,4},{1,3},{2,2},{3,1},{4,0},{5,0},
    {4,1},{3,2},{2,3},{1,4},{0,5},{0,6},{1,5},{2,4},
    {3,3},{4,2},{5,1},{6,0},{7,0},{6,1},{5,2},{4,3},
    {3,4},{2,5},{1,6},{0,7},{1,7},{2,6},{3,5},{4,4},
    {5,3},{6,2},{7,1},{7,2},{6,3},{5,4},{4,5},{3,6},
    {2,7},{3,7},{4,6},{5,5},{6,4},{7,3},{7,4},{6,5},
    {5,6},{4,7},{5,7},{6,6},{7,5},{7,6},{6,7},{7,7},
};

/* rle.rs:82-172  inverse_rle */
void orc_inverse_rle(const orc_block *eb, orc_dct_block *levels,
                     size_t pos_x, size_t pos_y, size_t blk_per_line, uint8_t quant)
{
    size_t block_id = pos_x / 8 + (pos_y / 8 * blk_per_line);     /* rle.rs:89 */
    orc_dct_block *block = &levels[block_id];

    if (eb->n_tcoef == 0) {                                        /* rle.rs:94-109 */
        if (eb->has_intradc) {
            int16_t dc_level = orc_intradc_into_level(eb->intradc);
            if (dc_level == 0) {
                block->tag = ORC_ZERO;
            } else {
                block->tag = ORC_DC;
                block->v[0] = (float)dc_level;
            }
        } else {
            block->tag = ORC_ZERO;
        }
        return;
    }

    float data[8][8];                                              /* [y][x], rle.rs:112 */
    memset(data, 0, sizeof data);
    int is_horiz = 1, is_vert = 1;
    size_t zz = 0;
    if (eb->has_intradc) {                                         /* rle.rs:118-121 */
        data[0][0] = (float)orc_intradc_into_level(eb->intradc);
        zz += 1;
    }
    for (int t = 0; t < eb->n_tcoef; t++) {                        /* rle.rs:122-149 */
        zz += eb->run[t];
        if (zz >= 64) return;                  /* rle.rs:125-127: block left as it was */
        uint8_t zx = DEZIGZAG[zz][0], zy = DEZIGZAG[zz][1];
        /* rle.rs:130-133 as a RELEASE build executes it (what Ruffle ships; [profile.release] has no
         * overflow-checks, and the dev profile -- which panics on the overflow instead -- is no decoder of such a
         * stream at all): every operation is i16 and wraps.  For quant * (2|L| + 1) > 32767 (Sorenson's 11-bit
         * escapes at quantisers from 16 up, parser/block.rs:694-708) the wrapped product, not the mathematical one,
         * is what reaches the clamp.  i16::abs wraps too (|-32768| = -32768). */
        int16_t level = eb->level[t];
        int16_t alevel = wrap_i16(level < 0 ? -(int)level : (int)level);          /* tcoef.level.abs() */
        int16_t deq = wrap_i16((int)quant * wrap_i16(wrap_i16(2 * (int)alevel) + 1)); /* rle.rs:130 */
        int16_t parity = (quant % 2 == 1) ? 0 : -1;                               /* rle.rs:131 */
        int16_t sg = (int16_t)((level > 0) - (level < 0));                        /* i16::signum */
        int16_t value = wrap_i16((int)sg * wrap_i16((int)deq + parity));          /* rle.rs:133 */
        value = (int16_t)clamp_i(value, -2048, 2047);
        float val = (float)value;
        data[zy][zx] = val;
        zz += 1;
        if (val != 0.0f) {
            if (zy > 0) is_horiz = 0;
            if (zx > 0) is_vert = 0;
        }
    }

    if (is_horiz && is_vert) {                                     /* rle.rs:151-160 */
        if (data[0][0] == 0.0f) {
            block->tag = ORC_ZERO;
        } else {
            block->tag = ORC_DC;
            block->v[0] = data[0][0];
        }
    } else if (is_horiz) {                                         /* rle.rs:161 */
        block->tag = ORC_HORIZ;
        for (int i = 0; i < 8; i++) block->v[i] = data[0][i];
    } else if (is_vert) {                                          /* rle.rs:162-171 */
        block->tag = ORC_VERT;
        for (int i = 0; i < 8; i++) block->v[i] = data[i][0];
    } else {
        block->tag = ORC_FULL;
        memcpy(block->v, data, sizeof data);
    }
}

/* ------------------------------------------------------------------------ */
/* h263/src/decoder/cpu/idct.rs                                              */
/* ------------------------------------------------------------------------ */

/* idct.rs:39-48  BASIS_TABLE[freq][x] -- the literals of the reference (they are
 * not exact cosines; tests check the binary32 bit patterns). */
static const float BASIS[8][8] = {
    { 0.70710677f,  0.70710677f,  0.70710677f,  0.70710677f,  0.70710677f,  0.70710677f,  0.70710677f,  0.70710677f },
    { 0.98078525f,  0.8314696f,   0.5555702f,   0.19509023f, -0.19509032f, -0.55557036f, -0.83146966f, -0.9807853f  },
    { 0.9238795f,   0.38268343f, -0.38268352f, -0.9238796f,  -0.9238795f,  -0.38268313f,  0.3826836f,   0.92387956f },
    { 0.8314696f,  -0.19509032f, -0.9807853f,  -0.55557f,     0.55557007f,  0.98078525f,  0.19509007f, -0.8314698f  },
    { 0.70710677f, -0.70710677f, -0.70710665f,  0.707107f,    0.70710677f, -0.70710725f, -0.70710653f,  0.7071068f  },
    { 0.5555702f,  -0.9807853f,   0.19509041f,  0.83146936f, -0.8314698f,  -0.19508928f,  0.9807853f,  -0.55557007f },
    { 0.38268343f, -0.9238795f,   0.92387974f, -0.3826839f,  -0.38268384f,  0.9238793f,  -0.92387974f,  0.3826839f  },
    { 0.19509023f, -0.55557f,     0.83146936f, -0.9807852f,   0.98078525f, -0.83147013f,  0.55557114f, -0.19508967f },
};

const float *orc_basis_table(void) { return &BASIS[0][0]; }

/* idct.rs:52-65  idct_1d: sequential f32 accumulation, un-fused */
static void idct_1d(const float in[8], float out[8])
{
    for (int i = 0; i < 8; i++) {
        float acc = 0.0f;
        for (int f = 0; f < 8; f++) {
            float p = in[f] * BASIS[f][i];
            acc = acc + p;
        }
        out[i] = acc;
    }
}

static inline uint8_t add_clip(int16_t clipped_idct, uint8_t mocomp)
{
    return (uint8_t)clamp_i((int)clipped_idct + (int)mocomp, 0, 255);  /* idct.rs:127-130 */
}

/* idct.rs:82-201  idct_channel */
void orc_idct_channel(const orc_dct_block *levels, size_t n_levels,
                      uint8_t *output, size_t output_len,
                      size_t blk_per_line, size_t spl)
{
    size_t output_height = output_len / spl;                      /* idct.rs:88 */
    size_t blk_height = n_levels / blk_per_line;                  /* idct.rs:89 */
    float inter[8][8], outp[8][8];
    memset(inter, 0, sizeof inter);
    memset(outp, 0, sizeof outp);

    for (size_t yb = 0; yb < blk_height; yb++) {
        for (size_t xb = 0; xb < blk_per_line; xb++) {
            size_t block_id = xb + yb * blk_per_line;
            if (block_id >= n_levels) continue;
            size_t xs = (size_t)clamp_pd((ptrdiff_t)spl - (ptrdiff_t)xb * 8, 0, 8);          /* idct.rs:106 */
            size_t ys = (size_t)clamp_pd((ptrdiff_t)output_height - (ptrdiff_t)yb * 8, 0, 8); /* idct.rs:107 */
            const orc_dct_block *b = &levels[block_id];

            switch (b->tag) {
            case ORC_ZERO:
                break;
            case ORC_DC: {                                        /* idct.rs:113-132 */
                float dc = b->v[0];
                int16_t ci = f32_as_i16(dc * 0.5f / 4.0f + f32_signum(dc) * 0.5f);
                ci = (int16_t)clamp_i(ci, -256, 255);
                for (size_t yo = 0; yo < ys; yo++)
                    for (size_t xo = 0; xo < xs; xo++) {
                        size_t idx = (xb * 8 + xo) + (yb * 8 + yo) * spl;
                        output[idx] = add_clip(ci, output[idx]);
                    }
                break;
            }
            case ORC_HORIZ: {                                     /* idct.rs:133-151 */
                idct_1d(b->v, inter[0]);
                for (size_t yo = 0; yo < ys; yo++)
                    for (size_t xo = 0; xo < xs; xo++) {
                        float idct = inter[0][xo];
                        int16_t ci = f32_as_i16(idct * BASIS[0][0] / 4.0f + f32_signum(idct) * 0.5f);
                        ci = (int16_t)clamp_i(ci, -256, 255);
                        size_t idx = (xb * 8 + xo) + (yb * 8 + yo) * spl;
                        output[idx] = add_clip(ci, output[idx]);
                    }
                break;
            }
            case ORC_VERT: {                                      /* idct.rs:152-169 */
                idct_1d(b->v, inter[0]);
                for (size_t yo = 0; yo < ys; yo++) {
                    float idct = inter[0][yo];
                    for (size_t xo = 0; xo < xs; xo++) {
                        int16_t ci = f32_as_i16(idct * BASIS[0][0] / 4.0f + f32_signum(idct) * 0.5f);
                        ci = (int16_t)clamp_i(ci, -256, 255);
                        size_t idx = (xb * 8 + xo) + (yb * 8 + yo) * spl;
                        output[idx] = add_clip(ci, output[idx]);
                    }
                }
                break;
            }
            default: {                                            /* ORC_FULL, idct.rs:170-198 */
                for (int row = 0; row < 8; row++) {
                    idct_1d(&b->v[row * 8], outp[row]);
                    for (int i = 0; i < 8; i++) inter[i][row] = outp[row][i]; /* transposition */
                }
                for (int row = 0; row < 8; row++) idct_1d(inter[row], outp[row]);
                for (size_t xo = 0; xo < xs; xo++)
                    for (size_t yo = 0; yo < ys; yo++) {
                        float idct = outp[xo][yo];
                        int16_t ci = f32_as_i16(idct / 4.0f + f32_signum(idct) * 0.5f);
                        ci = (int16_t)clamp_i(ci, -256, 255);
                        size_t idx = (xb * 8 + xo) + (yb * 8 + yo) * spl;
                        output[idx] = add_clip(ci, output[idx]);
                    }
                break;
            }
            }
        }
    }
}

/* ------------------------------------------------------------------------ */
/* h263/src/decoder/cpu/gather.rs                                            */
/* ------------------------------------------------------------------------ */

/* gather.rs:16-31  read_sample */
static inline uint8_t read_sample(const uint8_t *px, size_t spr, size_t rows,
                                  ptrdiff_t x, ptrdiff_t y)
{
    ptrdiff_t xm = spr ? (ptrdiff_t)spr - 1 : 0;
    ptrdiff_t ym = rows ? (ptrdiff_t)rows - 1 : 0;
    x = clamp_pd(x, 0, xm);
    y = clamp_pd(y, 0, ym);
    return px[(size_t)x + (size_t)y * spr];
}

/* gather.rs:34-40  lerp (u16 div_ceil(2)) */
static inline uint8_t lerp8(uint8_t a, uint8_t b, int middle)
{
    if (middle) return (uint8_t)(((unsigned)a + (unsigned)b + 1u) / 2u);
    return a;
}

/* gather.rs:47-126  gather_block */
static void gather_block(const uint8_t *px, size_t px_len, size_t spr,
                         size_t pos_x, size_t pos_y, int16_t mvx, int16_t mvy,
                         uint8_t *target)
{
    int16_t xd, yd;
    int xi, yi;
    orc_lerp_parameters(mvx, &xd, &xi);
    orc_lerp_parameters(mvy, &yd, &yi);

    ptrdiff_t src_x = (ptrdiff_t)pos_x + xd;
    ptrdiff_t src_y = (ptrdiff_t)pos_y + yd;
    size_t array_height = px_len / spr;
    ptrdiff_t cols = clamp_pd((ptrdiff_t)spr - (ptrdiff_t)pos_x, 0, 8);           /* gather.rs:60 */
    ptrdiff_t rows = clamp_pd((ptrdiff_t)array_height - (ptrdiff_t)pos_y, 0, 8);  /* gather.rs:61 */

    if (!xi && !yi) {
        if (cols == 8 && rows == 8 &&
            src_x >= 0 && src_x <= (ptrdiff_t)spr - 8 &&
            src_y >= 0 && src_y <= (ptrdiff_t)array_height - 8) {                  /* gather.rs:66-79 */
            for (int j = 0; j < 8; j++)
                memcpy(&target[pos_x + (pos_y + (size_t)j) * spr],
                       &px[(size_t)src_x + (size_t)(src_y + j) * spr], 8);
        } else {                                                                   /* gather.rs:80-89 */
            for (ptrdiff_t j = 0; j < rows; j++)
                for (ptrdiff_t i = 0; i < cols; i++)
                    target[pos_x + (size_t)i + (pos_y + (size_t)j) * spr] =
                        read_sample(px, spr, array_height, src_x + i, src_y + j);
        }
        return;
    }
    for (ptrdiff_t j = 0; j < rows; j++) {                                         /* gather.rs:90-125 */
        for (ptrdiff_t i = 0; i < cols; i++) {
            ptrdiff_t u = src_x + i, v = src_y + j;
            uint8_t s00 = read_sample(px, spr, array_height, u, v);
            uint8_t s10 = read_sample(px, spr, array_height, u + 1, v);
            uint8_t s01 = read_sample(px, spr, array_height, u, v + 1);
            uint8_t s11 = read_sample(px, spr, array_height, u + 1, v + 1);
            uint8_t s;
            if (xi && yi) {
                s = (uint8_t)(((unsigned)s00 + s10 + s01 + s11 + 2u) / 4u);        /* gather.rs:103-111 */
            } else {
                uint8_t m0 = lerp8(s00, s10, xi);
                uint8_t m1 = lerp8(s01, s11, xi);
                s = lerp8(m0, m1, yi);
            }
            target[pos_x + (size_t)i + (pos_y + (size_t)j) * spr] = s;
        }
    }
}

/* gather.rs:140-204  gather */
int orc_gather(const uint8_t *mb_types, const int16_t (*mvs)[4][2], size_t n_mbs,
               const uint8_t *ref_y, const uint8_t *ref_cb, const uint8_t *ref_cr,
               size_t width, size_t height, size_t mb_per_line,
               uint8_t *new_y, uint8_t *new_cb, uint8_t *new_cr)
{
    size_t cw = (width + 1) / 2, ch = (height + 1) / 2;            /* picture.rs:45-46 */
    size_t y_len = width * height, c_len = cw * ch;
    for (size_t i = 0; i < n_mbs; i++) {
        if (!mb_is_inter(mb_types[i])) continue;
        if (!ref_y) return ORC_ERR_UNCODED_IFRAME_BLOCKS;          /* gather.rs:149 */
        size_t px = (i % mb_per_line) * 16, py = (i / mb_per_line) * 16;
        const int16_t (*mv)[2] = mvs[i];
        gather_block(ref_y, y_len, width, px, py, mv[0][0], mv[0][1], new_y);
        gather_block(ref_y, y_len, width, px + 8, py, mv[1][0], mv[1][1], new_y);
        gather_block(ref_y, y_len, width, px, py + 8, mv[2][0], mv[2][1], new_y);
        gather_block(ref_y, y_len, width, px + 8, py + 8, mv[3][0], mv[3][1], new_y);
        /* gather.rs:182: (mv0+mv1+mv2+mv3).average_sum_of_mvs(), i16 adds */
        int16_t sx = (int16_t)(mv[0][0] + mv[1][0] + mv[2][0] + mv[3][0]);
        int16_t sy = (int16_t)(mv[0][1] + mv[1][1] + mv[2][1] + mv[3][1]);
        int16_t cx = orc_average_sum_of_mvs(sx), cy = orc_average_sum_of_mvs(sy);
        size_t cpx = (i % mb_per_line) * 8, cpy = (i / mb_per_line) * 8;
        gather_block(ref_cb, c_len, cw, cpx, cpy, cx, cy, new_cb);
        gather_block(ref_cr, c_len, cw, cpx, cpy, cx, cy, new_cr);
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* record-level picture reconstruction: h263/src/decoder/state.rs            */
/* ------------------------------------------------------------------------ */

/* Rebuild the parser-level `Block` (types.rs:887-893) from one dense coefficient
 * block of the record format, so that the literal inverse_rle restatement above is
 * what classifies and dequantises it. */
static void block_from_record(orc_block *b, int is_intra, uint8_t intradc_code,
                              int coded, int kill, const int16_t *coeff /* 64, raster x+8y, or NULL */)
{
    b->has_intradc = is_intra;
    b->intradc = intradc_code;
    b->n_tcoef = 0;
    if (!coded) return;
    int last = is_intra ? 1 : 0;    /* next zigzag index to fill (rle.rs:117-121) */
    for (int z = last; z < 64 && coeff; z++) {
        int raster = DEZIGZAG[z][0] + 8 * DEZIGZAG[z][1];
        int16_t lv = coeff[raster];
        if (lv == 0) continue;
        b->run[b->n_tcoef] = (uint8_t)(z - last);
        b->level[b->n_tcoef] = lv;
        b->n_tcoef++;
        last = z + 1;
    }
    if (kill) {                     /* a run that walks past zigzag 63 (rle.rs:125-127) */
        b->run[b->n_tcoef] = 64;
        b->level[b->n_tcoef] = 1;
        b->n_tcoef++;
    }
}

int orc_decode_picture(uint16_t width, uint16_t height,
                       const orc_mb_record *mbs, size_t n_mbs,
                       const int16_t *coeffs, size_t n_coeff_blocks,
                       const uint8_t *ref_y, const uint8_t *ref_cb, const uint8_t *ref_cr,
                       uint8_t *out_y, uint8_t *out_cb, uint8_t *out_cr)
{
    if (!width || !height) return ORC_ERR_INVALID_ARGUMENT;
    size_t w = width, h = height;
    size_t mbw = (w + 15) / 16, mbh = (h + 15) / 16;              /* state.rs:173-174 */
    size_t n_total = mbw * mbh;
    if (n_mbs > n_total) return ORC_ERR_INVALID_ARGUMENT;
    size_t cw = (w + 1) / 2, ch = (h + 1) / 2;                    /* picture.rs:45-46 */

    size_t n_luma = mbw * 2 * mbh * 2, n_chroma = mbw * mbh;      /* state.rs:186-191 */
    orc_dct_block *ll = calloc(n_luma, sizeof *ll);
    orc_dct_block *lb = calloc(n_chroma, sizeof *lb);
    orc_dct_block *lr = calloc(n_chroma, sizeof *lr);
    uint8_t *types = malloc(n_total);
    int16_t (*mvs)[4][2] = calloc(n_total, sizeof *mvs);
    if (!ll || !lb || !lr || !types || !mvs) {
        free(ll); free(lb); free(lr); free(types); free(mvs);
        return ORC_ERR_INVALID_ARGUMENT;
    }

    int rc = ORC_OK;
    for (size_t i = 0; i < n_mbs && rc == ORC_OK; i++) {          /* state.rs:193-417, record form */
        const orc_mb_record *m = &mbs[i];
        size_t px = (i % mbw) * 16, py = (i / mbw) * 16;          /* state.rs:199-202 */
        int intra = mb_is_intra(m->mb_type);
        types[i] = m->mb_type;
        memcpy(mvs[i], m->mv, sizeof m->mv);
        uint32_t ci = m->coeff_index;
        for (int blk = 0; blk < 6; blk++) {
            int coded = (m->cbp >> blk) & 1;
            int kill = coded && ((m->kill >> blk) & 1);
            const int16_t *c = NULL;
            if (coded) {
                if ((size_t)ci >= n_coeff_blocks) { rc = ORC_ERR_INVALID_ARGUMENT; break; }
                c = coeffs + (size_t)ci * 64;
                ci++;
            }
            orc_block b;
            block_from_record(&b, intra, m->intradc[blk], coded, kill, c);
            switch (blk) {                                        /* state.rs:287-381 */
            case 0: orc_inverse_rle(&b, ll, px, py, mbw * 2, m->quant); break;
            case 1: orc_inverse_rle(&b, ll, px + 8, py, mbw * 2, m->quant); break;
            case 2: orc_inverse_rle(&b, ll, px, py + 8, mbw * 2, m->quant); break;
            case 3: orc_inverse_rle(&b, ll, px + 8, py + 8, mbw * 2, m->quant); break;
            case 4: orc_inverse_rle(&b, lb, px / 2, py / 2, mbw, m->quant); break;
            default: orc_inverse_rle(&b, lr, px / 2, py / 2, mbw, m->quant); break;
            }
        }
    }
    for (size_t i = n_mbs; i < n_total; i++) types[i] = 0;        /* state.rs:421-427: Inter, mv 0 */

    if (rc == ORC_OK) {
        memset(out_y, 0, w * h);                                  /* picture.rs:39-58 zeroed planes */
        memset(out_cb, 0, cw * ch);
        memset(out_cr, 0, cw * ch);
        rc = orc_gather(types, (const int16_t (*)[4][2])mvs, n_total, ref_y, ref_cb, ref_cr,
                        w, h, mbw, out_y, out_cb, out_cr);        /* state.rs:432-438 */
    }
    if (rc == ORC_OK) {
        orc_idct_channel(ll, n_luma, out_y, w * h, mbw * 2, w);   /* state.rs:439-444 */
        orc_idct_channel(lb, n_chroma, out_cb, cw * ch, mbw, cw); /* state.rs:446-452 */
        orc_idct_channel(lr, n_chroma, out_cr, cw * ch, mbw, cw); /* state.rs:453-458 */
    }
    free(ll); free(lb); free(lr); free(types); free(mvs);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* deblock/src/deblock.rs                                                    */
/* ------------------------------------------------------------------------ */

/* deblock.rs:5-8  Table J.2 */
const uint8_t orc_quant_to_strength[32] = {
    0, 1, 1, 2, 2, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 7, 7, 8, 8, 8, 9, 9, 9, 10, 10, 10, 11, 11, 11,
    12, 12, 12,
};

static inline int16_t i16_abs(int16_t x) { return (int16_t)(x < 0 ? -x : x); }
static inline int16_t i16_max(int16_t a, int16_t b) { return a > b ? a : b; }
static inline int16_t i16_min(int16_t a, int16_t b) { return a < b ? a : b; }
static inline int16_t i16_signum(int16_t x) { return (int16_t)((x > 0) - (x < 0)); }

/* deblock.rs:13-15 / 66-69  up_down_ramp (identical in both modules) */
static inline int16_t up_down_ramp(int16_t x, int16_t strength)
{
    int16_t ax = i16_abs(x);
    return (int16_t)(i16_signum(x) * i16_max((int16_t)(ax - i16_max((int16_t)(2 * (ax - strength)), 0)), 0));
}

/* deblock.rs:19-21 / 73-76  clipd1 */
static inline int16_t clipd1(int16_t x, int16_t lim)
{
    int16_t la = i16_abs(lim);
    return i16_min(i16_max(x, (int16_t)-la), la);
}

/* deblock.rs:29-42  scalar_impl::process -- Rust `/` truncates toward zero */
void orc_deblock_process_scalar(uint8_t *A, uint8_t *B, uint8_t *C, uint8_t *D, uint8_t strength)
{
    int16_t a = *A, b = *B, c = *C, d16 = *D;
    int16_t d = (int16_t)((a - 4 * b + 4 * c - d16) / 8);
    int16_t d1 = up_down_ramp(d, strength);
    int16_t d2 = clipd1((int16_t)((a - d16) / 4), (int16_t)(d1 / 2));
    *A = (uint8_t)(a - d2);                         /* `as u8`: wraps */
    *B = (uint8_t)clamp_i(b + d1, 0, 255);
    *C = (uint8_t)clamp_i(c - d1, 0, 255);
    *D = (uint8_t)(d16 + d2);
}

/* deblock.rs:99-127  simd_impl::process_simd, one lane -- wide::i16x8 `shr` is an
 * arithmetic shift (floor) */
void orc_deblock_process_simd_lane(uint8_t *A, uint8_t *B, uint8_t *C, uint8_t *D, uint8_t strength)
{
    int16_t a = *A, b = *B, c = *C, d16 = *D;
    int16_t d = (int16_t)((int16_t)(a - 4 * b + 4 * c - d16) >> 3);
    int16_t d1 = up_down_ramp(d, strength);
    int16_t d2 = clipd1((int16_t)((int16_t)(a - d16) >> 2), (int16_t)(d1 >> 1));
    *A = (uint8_t)(a - d2);
    *B = (uint8_t)clamp_i(b + d1, 0, 255);
    *C = (uint8_t)clamp_i(c - d1, 0, 255);
    *D = (uint8_t)(d16 + d2);
}

/* deblock.rs:136-181  deblock_horiz */
static void deblock_horiz(uint8_t *r, size_t len, size_t width, uint8_t strength)
{
    size_t height = len / width;
    if (height < 2) return;   /* reference precondition h >= 2 (usize underflow otherwise) */
    size_t simd_cols = (width / 8) * 8;              /* chunks_exact_mut(8), deblock.rs:150-163 */
    for (size_t edge_y = 8; edge_y <= height - 2; edge_y += 8) {
        uint8_t *ra = r + (edge_y - 2) * width, *rb = ra + width, *rc = rb + width, *rd = rc + width;
        for (size_t x = 0; x < simd_cols; x++)
            orc_deblock_process_simd_lane(&ra[x], &rb[x], &rc[x], &rd[x], strength);
        for (size_t x = simd_cols; x < width; x++)                               /* deblock.rs:165-177 */
            orc_deblock_process_scalar(&ra[x], &rb[x], &rc[x], &rd[x], strength);
    }
}

/* deblock.rs:185-299  deblock_vert */
static void deblock_vert(uint8_t *r, size_t len, size_t width, uint8_t strength)
{
    if (width < 10) return;                                                       /* deblock.rs:228 */
    size_t height = len / width;
    size_t simd_rows = (height / 8) * 8;             /* chunks_exact_mut(width*8), deblock.rs:231 */
    for (size_t y = 0; y < height; y++) {
        uint8_t *row = r + y * width;
        /* row[2..].chunks_exact_mut(8): chunk k = columns 2+8k .. 9+8k; ABCD = chunk[4..8] */
        for (size_t k = 0; 2 + 8 * k + 8 <= width; k++) {
            uint8_t *q = row + 2 + 8 * k + 4;
            if (y < simd_rows)
                orc_deblock_process_simd_lane(&q[0], &q[1], &q[2], &q[3], strength);
            else
                orc_deblock_process_scalar(&q[0], &q[1], &q[2], &q[3], strength);  /* deblock.rs:281-297 */
        }
    }
}

/* deblock.rs:305-315  deblock */
int orc_deblock(const uint8_t *data, size_t len, size_t width, uint8_t strength, uint8_t *out)
{
    if (width == 0 || len % width != 0 || strength < 1 || strength > 12) return ORC_ERR_INVALID_ARGUMENT;
    memcpy(out, data, len);
    deblock_horiz(out, len, width, strength);
    deblock_vert(out, len, width, strength);
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* yuv/src/bt601.rs                                                          */
/* ------------------------------------------------------------------------ */

/* bt601.rs:12-59  yuv_to_rgba_4x, one lane */
static inline void yuv_to_rgba_1(uint8_t yv, uint8_t cbv, uint8_t crv, uint8_t *rgba)
{
    int32_t y = (int32_t)yv - 16, cb = (int32_t)cbv - 128, cr = (int32_t)crv - 128;
    int32_t gray = y * 76309;
    int32_t cr2r = cr * 104597;
    int32_t cr2g = cr * -53279;
    int32_t cb2g = cb * -25675;
    int32_t cb2b = cb * 132201;
    int32_t half = 32768;
    int32_t r = (gray + cr2r + half) >> 16;           /* arithmetic shift */
    int32_t g = (gray + cr2g + cb2g + half) >> 16;
    int32_t b = (gray + cb2b + half) >> 16;
    rgba[0] = (uint8_t)clamp_i(r, 0, 255);
    rgba[1] = (uint8_t)clamp_i(g, 0, 255);
    rgba[2] = (uint8_t)clamp_i(b, 0, 255);
    rgba[3] = 255;
}

/* bt601.rs:105-196  yuv420_to_rgba */
int orc_yuv420_to_rgba(const uint8_t *y, size_t y_len,
                       const uint8_t *cb, const uint8_t *cr, size_t c_len,
                       size_t y_width, uint8_t *rgba)
{
    if (y_len == 0) return ORC_OK;                                 /* bt601.rs:107-112 */
    if (y_width == 0 || y_len % y_width != 0) return ORC_ERR_INVALID_ARGUMENT;
    size_t br_width = (y_width + 1) / 2;                           /* bt601.rs:115 */
    size_t y_height = y_len / y_width;
    if (c_len % br_width != 0 || c_len / br_width != (y_height + 1) / 2) return ORC_ERR_INVALID_ARGUMENT;
    for (size_t row = 0; row < y_height; row++) {                  /* bt601.rs:132-193 */
        size_t crow = row / 2;
        for (size_t x = 0; x < y_width; x++) {
            /* body (bt601.rs:141-165) and remainder (168-192) both pair pixel x with
             * chroma sample x/2 of the row */
            yuv_to_rgba_1(y[row * y_width + x], cb[crow * br_width + x / 2],
                          cr[crow * br_width + x / 2], &rgba[(row * y_width + x) * 4]);
        }
    }
    return ORC_OK;
}
