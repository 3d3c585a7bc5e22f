/*
 * simd_stages.c -- CPU BASELINE ONLY (test / bench infrastructure, NOT product code, NOT the parity oracle).
 *
 * The reference's two post-processing crates are explicit 128-bit SIMD: deblock works on eight i16 lanes
 * (deblock/src/deblock.rs:44-127 `process_simd`, driven by deblock_horiz :136-181 and deblock_vert :185-299), BT.601 on
 * four i32 lanes (yuv/src/bt601.rs:12-59 `yuv_to_rgba_4x`, driven by yuv420_to_rgba :105-196).  The parity oracle
 * (h263_oracle.c) restates their arithmetic one lane at a time and gcc does NOT vectorise those loops (-fopt-info-vec:
 * no line for deblock_horiz / deblock_vert / orc_yuv420_to_rgba), so timing it would under-state what the reference's CPU
 * path does per core.  This file restates the same two functions with the reference's own data-parallel shape, on gcc's
 * portable 128-bit vector types -- eight i16 lanes per quartet group, four i32 lanes per pixel group, the same split
 * between SIMD region and scalar tails -- so that bench.py's cpu_baseline can time BOTH forms and say which is which.
 * Before anything is timed the results are compared byte for byte with the scalar oracle (native_bench.py).
 */
#include <string.h>

#include "h263_oracle.h"

typedef int16_t v8i16 __attribute__((vector_size(16)));
typedef uint8_t v8u8 __attribute__((vector_size(8)));
typedef int32_t v4i32 __attribute__((vector_size(16)));
typedef uint8_t v4u8 __attribute__((vector_size(4)));

static inline v8i16 splat16(int16_t v) { return (v8i16){v, v, v, v, v, v, v, v}; }
static inline v8i16 max16(v8i16 a, v8i16 b) { const v8i16 m = a > b; return (a & m) | (b & ~m); }
static inline v8i16 min16(v8i16 a, v8i16 b) { const v8i16 m = a < b; return (a & m) | (b & ~m); }
static inline v8i16 abs16(v8i16 a) { return max16(a, -a); }
/* deblock.rs:50-55: lt - gt on all-ones masks */
static inline v8i16 signum16(v8i16 x) { return (x < splat16(0)) - (x > splat16(0)); }
/* deblock.rs:65-69 */
static inline v8i16 up_down_ramp16(v8i16 x, int16_t strength)
{
    const v8i16 ax = abs16(x), zero = splat16(0);
    return signum16(x) * max16(ax - max16(2 * (ax - splat16(strength)), zero), zero);
}
/* deblock.rs:72-76 */
static inline v8i16 clipd1_16(v8i16 x, v8i16 lim)
{
    const v8i16 la = abs16(lim);
    return min16(max16(x, -la), la);
}

/* deblock.rs:99-127 process_simd on eight quartets held as i16 lanes */
static inline void process8(v8i16 *a, v8i16 *b, v8i16 *c, v8i16 *d, int16_t strength)
{
    const v8i16 a16 = *a, b16 = *b, c16 = *c, d16 = *d, zero = splat16(0), top = splat16(255);
    const v8i16 dd = (a16 - 4 * b16 + 4 * c16 - d16) >> 3;
    const v8i16 d1 = up_down_ramp16(dd, strength);
    const v8i16 d2 = clipd1_16((a16 - d16) >> 2, d1 >> 1);
    *a = a16 - d2;
    *b = min16(max16(b16 + d1, zero), top);
    *c = min16(max16(c16 - d1, zero), top);
    *d = d16 + d2;
}

static inline v8i16 load8(const uint8_t *p)
{
    v8u8 v;
    memcpy(&v, p, 8);
    return __builtin_convertvector(v, v8i16);
}
static inline void store8(uint8_t *p, v8i16 v)
{
    const v8u8 o = __builtin_convertvector(v, v8u8);       /* `as u8`: keeps the low byte */
    memcpy(p, &o, 8);
}

/* deblock.rs:136-181 */
static void deblock_horiz_simd(uint8_t *r, size_t len, size_t width, uint8_t strength)
{
    const size_t height = len / width;
    if (height < 2) return;
    const size_t simd_cols = (width / 8) * 8;
    for (size_t edge_y = 8; edge_y <= height - 2; edge_y += 8) {
        uint8_t *ra = r + (edge_y - 2) * width, *rb = ra + width, *rc = rb + width, *rd = rc + width;
        for (size_t x = 0; x < simd_cols; x += 8) {
            v8i16 a = load8(ra + x), b = load8(rb + x), c = load8(rc + x), d = load8(rd + x);
            process8(&a, &b, &c, &d, strength);
            store8(ra + x, a); store8(rb + x, b); store8(rc + x, c); store8(rd + x, d);
        }
        for (size_t x = simd_cols; x < width; x++) orc_deblock_process_scalar(&ra[x], &rb[x], &rc[x], &rd[x], strength);
    }
}

/* deblock.rs:185-299: eight rows supply the eight lanes, the columns are extracted and set one value at a time */
static void deblock_vert_simd(uint8_t *r, size_t len, size_t width, uint8_t strength)
{
    if (width < 10) return;
    const size_t height = len / width, simd_rows = (height / 8) * 8;
    for (size_t y0 = 0; y0 < simd_rows; y0 += 8) {
        uint8_t *row[8];
        for (int k = 0; k < 8; k++) row[k] = r + (y0 + (size_t)k) * width;
        for (size_t x = 2; x + 8 <= width; x += 8) {
            v8i16 q[4];
            for (int s = 0; s < 4; s++)
                for (int k = 0; k < 8; k++) q[s][k] = row[k][x + 4 + (size_t)s];
            process8(&q[0], &q[1], &q[2], &q[3], strength);
            for (int s = 0; s < 4; s++)
                for (int k = 0; k < 8; k++) row[k][x + 4 + (size_t)s] = (uint8_t)q[s][k];
        }
    }
    for (size_t y = simd_rows; y < height; y++) {
        uint8_t *rw = r + y * width;
        for (size_t x = 2; x + 8 <= width; x += 8)
            orc_deblock_process_scalar(&rw[x + 4], &rw[x + 5], &rw[x + 6], &rw[x + 7], strength);
    }
}

/* deblock.rs:305-315 */
int orc_deblock_simd(const uint8_t *data, size_t len, size_t width, uint8_t strength, uint8_t *out)
{
    if (width == 0 || len % width != 0 || strength < 1 || strength > 12) return ORC_ERR_INVALID_ARGUMENT;
    memcpy(out, data, len);
    deblock_horiz_simd(out, len, width, strength);
    deblock_vert_simd(out, len, width, strength);
    return ORC_OK;
}

/* bt601.rs:12-59: four pixels, one per i32 lane, chroma samples doubled horizontally */
static inline void yuv_to_rgba_4x(const uint8_t y4[4], const uint8_t cb2[2], const uint8_t cr2[2], uint8_t rgba[16])
{
    const v4i32 y = (v4i32){y4[0], y4[1], y4[2], y4[3]} - 16;
    const v4i32 cb = (v4i32){cb2[0], cb2[0], cb2[1], cb2[1]} - 128;
    const v4i32 cr = (v4i32){cr2[0], cr2[0], cr2[1], cr2[1]} - 128;
    const v4i32 gray = y * 76309, half = {32768, 32768, 32768, 32768}, zero = {0, 0, 0, 0}, top = {255, 255, 255, 255};
    v4i32 r = (gray + cr * 104597 + half) >> 16;
    v4i32 g = (gray + cr * -53279 + cb * -25675 + half) >> 16;
    v4i32 b = (gray + cb * 132201 + half) >> 16;
    v4i32 m;
    m = r > zero; r &= m; m = r < top; r = (r & m) | (top & ~m);
    m = g > zero; g &= m; m = g < top; g = (g & m) | (top & ~m);
    m = b > zero; b &= m; m = b < top; b = (b & m) | (top & ~m);
    const v4i32 px = (r | (g << 8)) | ((b << 16) | (top << 24));          /* little endian: R, G, B, A in memory */
    memcpy(rgba, &px, 16);
}

/* bt601.rs:105-196 */
int orc_yuv420_to_rgba_simd(const uint8_t *y, size_t y_len, const uint8_t *cb, const uint8_t *cr, size_t c_len, size_t y_width,
                            uint8_t *rgba)
{
    if (y_len == 0) return ORC_OK;
    if (y_width == 0 || y_len % y_width != 0) return ORC_ERR_INVALID_ARGUMENT;
    const size_t br_width = (y_width + 1) / 2, y_height = y_len / y_width;
    if (c_len % br_width != 0 || c_len / br_width != (y_height + 1) / 2) return ORC_ERR_INVALID_ARGUMENT;
    const size_t y_rem = y_width % 4, body = y_width - y_rem;
    for (size_t row = 0; row < y_height; row++) {
        const uint8_t *yr = y + row * y_width, *cbr = cb + (row / 2) * br_width, *crr = cr + (row / 2) * br_width;
        uint8_t *out = rgba + row * y_width * 4;
        for (size_t x = 0; x < body; x += 4) yuv_to_rgba_4x(yr + x, cbr + x / 2, crr + x / 2, out + x * 4);
        if (y_rem) {                                              /* bt601.rs:168-192 */
            uint8_t y4[4] = {0, 0, 0, 0}, cb2[2] = {0, 0}, cr2[2] = {0, 0}, px[16];
            for (size_t x = body; x < y_width; x++) {
                y4[x % 4] = yr[x];
                cb2[(x % 4) / 2] = cbr[x / 2];
                cr2[(x % 4) / 2] = crr[x / 2];
            }
            yuv_to_rgba_4x(y4, cb2, cr2, px);
            memcpy(out + body * 4, px, y_rem * 4);
        }
    }
    return ORC_OK;
}
