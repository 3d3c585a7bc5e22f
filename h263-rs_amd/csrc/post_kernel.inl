// post_kernel.inl -- k_post: deblocking post-filter + BT.601 YUV 4:2:0 -> RGBA, fused.
//
// Replaces deblock::deblock (deblock/src/deblock.rs:305-315: deblock_horiz 136-181 then
// deblock_vert 185-299, kernel process/process_simd 29-42 / 99-127) applied to each plane
// and yuv::bt601::yuv420_to_rgba (yuv/src/bt601.rs:105-196, yuv_to_rgba_4x 12-59).
//
// One workgroup owns a 128x32 luma tile (+ the matching 64x16 chroma tiles) whose origin
// is shifted to (4,4) mod 16.  Every filtered pixel group -- rows edge-2..edge+1 of a
// horizontal block edge, columns edge-2..edge+1 of a vertical one -- then lies entirely
// inside one tile for luma (edges at multiples of 8) AND for chroma (tile origin (2,2)
// mod 8), so the two filter passes run in LDS with no halo and no second pass over HBM:
// load tile -> H-edges -> V-edges -> convert -> 16-byte RGBA stores.
//
// The reference mixes two integer semantics by position: its
// SIMD lanes use arithmetic shifts (floor), its scalar tails use `/` (truncation).
//   horizontal edges: floor for columns < 8*floor(w/8), truncation right of that;
//   vertical edges  : floor for rows    < 8*floor(h/8), truncation below that.
#pragma once

#include "dev_common.h"

namespace h263mi {

constexpr int POST_THREADS = 256;
constexpr int POST_TW = 128, POST_TH = 32;          // luma tile
constexpr int POST_CW = 64, POST_CH = 16;           // chroma tile
constexpr int POST_OX = POST_TW - 4, POST_OY = POST_TH - 4;   // tile (tx,ty) starts at tx*TW - OX, ty*TH - OY

struct PostSmem {
    uint8_t y[POST_TH * POST_TW];
    uint8_t c[2][POST_CH * POST_CW];
};

// One A,B,C,D quartet (deblock.rs:29-42 / 99-127).  floor_sem selects the SIMD-lane
// semantics (>>) over the scalar ones (/).
H263_HD void deblock_quartet(int &A, int &B, int &C, int &D, int strength, bool floor_sem)
{
    const int n = A - 4 * B + 4 * C - D;
    const int d = floor_sem ? (n >> 3) : (n / 8);
    const int ad = d < 0 ? -d : d;
    // up_down_ramp (deblock.rs:13-15): signum(d) * max(0, |d| - max(0, 2*(|d| - strength)))
    int t = 2 * (ad - strength);
    t = t < 0 ? 0 : t;
    int mag = ad - t;
    mag = mag < 0 ? 0 : mag;
    const int d1 = d < 0 ? -mag : mag;
    const int half = floor_sem ? (d1 >> 1) : (d1 / 2);
    const int lim = half < 0 ? -half : half;
    const int q = floor_sem ? ((A - D) >> 2) : ((A - D) / 4);
    const int d2 = clampi(q, -lim, lim);                      // clipd1 (deblock.rs:19-21)
    A = (A - d2) & 0xff;                                      // `as u8`: wraps, no clamp
    B = clampi(B + d1, 0, 255);
    C = clampi(C - d1, 0, 255);
    D = (D + d2) & 0xff;
}

// bt601.rs:12-59, one pixel -> packed R | G<<8 | B<<16 | 255<<24
H263_HD uint32_t bt601_pixel(int y, int cb, int cr)
{
    const int gray = (y - 16) * 76309 + 32768;
    const int r = gray + (cr - 128) * 104597;
    const int g = gray + (cr - 128) * -53279 + (cb - 128) * -25675;
    const int b = gray + (cb - 128) * 132201;
    // clamp(v >> 16, 0, 255) == clamp(v, 0, 0xFFFFFF) >> 16 (the shift is monotone).  Written in
    // this order on purpose: hipcc (ROCm 7.2) turns the shift-then-clamp form into the gfx950
    // instruction v_ashr_pk_u8_i32, whose result for negative inputs did not match the 0 the
    // reference expects (caught by the bt601.rs:206-207 golden on an MI355X).
    const uint32_t R = (uint32_t)clampi(r, 0, 0xFFFFFF) >> 16;
    const uint32_t G = (uint32_t)clampi(g, 0, 0xFFFFFF) >> 16;
    const uint32_t B = (uint32_t)clampi(b, 0, 0xFFFFFF) >> 16;
    return R | (G << 8) | (B << 16) | 0xff000000u;
}

// ---- phase 0: tile -> LDS ---------------------------------------------------------------
H263_DEV void post_phase_load(const PostArgs &a, PostSmem &s, int tid, int tile, int pic)
{
    const int tx = tile % (int)a.tiles_x, ty = tile / (int)a.tiles_x;
    const uint8_t *frame = a.frames + (size_t)pic * a.L.frame_bytes;
    const int xl = tx * POST_TW - POST_OX, yl = ty * POST_TH - POST_OY;
    // luma: 32 rows x 128 B = 256 lanes x 16 B (4 dwords; the origin is only 4-byte aligned)
    {
        const int row = tid >> 3, col = (tid & 7) * 16;
        const int gy = yl + row;
        uint32_t v[4] = {0, 0, 0, 0};
        if (gy >= 0 && gy < (int)a.L.rows_y) {
            const uint8_t *src = frame + (size_t)gy * a.L.pitch_y;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int gx = xl + col + 4 * q;
                if (gx >= 0 && gx + 4 <= (int)a.L.pitch_y) v[q] = *reinterpret_cast<const uint32_t *>(src + gx);
            }
        }
        *reinterpret_cast<uint4 *>(&s.y[row * POST_TW + col]) = make_uint4(v[0], v[1], v[2], v[3]);
    }
    if (a.luma_only) return;
    // chroma: 2 planes x 16 rows x 64 B = 256 lanes x 8 B (origin 2-byte aligned: byte-pair loads)
    {
        const int plane = tid >> 7, row = (tid >> 3) & 15, col = (tid & 7) * 8;
        const int cxl = xl / 2, cyl = yl / 2;                 // xl, yl are even (and may be negative)
        const int gy = cyl + row;
        const uint8_t *src = frame + (plane ? a.L.off_cr : a.L.off_cb) + (size_t)(gy < 0 ? 0 : gy) * a.L.pitch_c;
        uint16_t v[4] = {0, 0, 0, 0};
        if (gy >= 0 && gy < (int)a.L.rows_c) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int gx = cxl + col + 2 * q;
                if (gx >= 0 && gx + 2 <= (int)a.L.pitch_c) v[q] = *reinterpret_cast<const uint16_t *>(src + gx);
            }
        }
        uint64_t packed = (uint64_t)v[0] | ((uint64_t)v[1] << 16) | ((uint64_t)v[2] << 32) | ((uint64_t)v[3] << 48);
        *reinterpret_cast<uint64_t *>(&s.c[plane][row * POST_CW + col]) = packed;
    }
}

// filter 4 neighbouring columns of one horizontal edge held in LDS
H263_DEV void hfilter4(uint8_t *t, int pitch, int row_c, int col, int strength, int gx0, int floor_cols, int w)
{
    uint32_t ra = *reinterpret_cast<uint32_t *>(t + (row_c - 2) * pitch + col);
    uint32_t rb = *reinterpret_cast<uint32_t *>(t + (row_c - 1) * pitch + col);
    uint32_t rc = *reinterpret_cast<uint32_t *>(t + (row_c)*pitch + col);
    uint32_t rd = *reinterpret_cast<uint32_t *>(t + (row_c + 1) * pitch + col);
    uint32_t oa = 0, ob = 0, oc = 0, od = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int A = (ra >> (8 * k)) & 0xff, B = (rb >> (8 * k)) & 0xff, C = (rc >> (8 * k)) & 0xff, D = (rd >> (8 * k)) & 0xff;
        int gx = gx0 + k;
        if (gx >= 0 && gx < w) deblock_quartet(A, B, C, D, strength, gx < floor_cols);
        oa |= (uint32_t)A << (8 * k);
        ob |= (uint32_t)B << (8 * k);
        oc |= (uint32_t)C << (8 * k);
        od |= (uint32_t)D << (8 * k);
    }
    *reinterpret_cast<uint32_t *>(t + (row_c - 2) * pitch + col) = oa;
    *reinterpret_cast<uint32_t *>(t + (row_c - 1) * pitch + col) = ob;
    *reinterpret_cast<uint32_t *>(t + (row_c)*pitch + col) = oc;
    *reinterpret_cast<uint32_t *>(t + (row_c + 1) * pitch + col) = od;
}

// ---- phase 1: horizontal block edges (deblock_horiz, deblock.rs:136-181) ------------------
H263_DEV void post_phase_hedges(const PostArgs &a, PostSmem &s, int tid, int tile)
{
    const int tx = tile % (int)a.tiles_x, ty = tile / (int)a.tiles_x;
    const int xl = tx * POST_TW - POST_OX, yl = ty * POST_TH - POST_OY;
    const int strength = (int)a.strength;
    if (tid < 128) {
        // luma: 4 edges (tile rows 4, 12, 20, 28) x 32 column groups of 4
        const int e = tid >> 5, cg = tid & 31;
        const int row_c = 4 + 8 * e, gy = yl + row_c;          // picture row of the "C" samples
        const int w = (int)a.L.width, h = (int)a.L.height;
        if (gy >= 8 && gy + 1 <= h - 1)                        // edge_y <= height - 2 (deblock.rs:140)
            hfilter4(s.y, POST_TW, row_c, cg * 4, strength, xl + cg * 4, (w / 8) * 8, w);
    } else if (tid < 192 && !a.luma_only) {
        // chroma: 2 planes x 2 edges (tile rows 6, 14) x 16 column groups
        const int q = tid - 128, plane = q >> 5, e = (q >> 4) & 1, cg = q & 15;
        const int row_c = 6 + 8 * e, gy = yl / 2 + row_c;
        const int w = (int)a.L.cwidth, h = (int)a.L.cheight;
        if (gy >= 8 && gy + 1 <= h - 1)
            hfilter4(s.c[plane], POST_CW, row_c, cg * 4, strength, xl / 2 + cg * 4, (w / 8) * 8, w);
    }
}

// ---- phase 2: vertical block edges (deblock_vert, deblock.rs:185-299) ----------------------
H263_DEV void post_phase_vedges(const PostArgs &a, PostSmem &s, int tid, int tile)
{
    const int tx = tile % (int)a.tiles_x, ty = tile / (int)a.tiles_x;
    const int xl = tx * POST_TW - POST_OX, yl = ty * POST_TH - POST_OY;
    const int strength = (int)a.strength;
    // luma: 32 rows x 16 edges; the quartet sits in bytes 2..5 of an aligned 8-byte window
    for (int it = 0; it < 2; it++) {
        const int item = it * POST_THREADS + tid, row = item >> 4, j = item & 15;
        const int gy = yl + row, gxa = xl + 8 * j + 2;         // picture column of the "A" sample
        const int w = (int)a.L.width, h = (int)a.L.height;
        // A..D = columns 8k-2 .. 8k+1 with k >= 1 and 8k+1 <= w-1 (chunks of row[2..], deblock.rs:281)
        if (gy >= 0 && gy < h && gxa >= 6 && gxa + 3 <= w - 1) {
            uint64_t v = *reinterpret_cast<uint64_t *>(&s.y[row * POST_TW + 8 * j]);
            int A = (v >> 16) & 0xff, B = (v >> 24) & 0xff, C = (v >> 32) & 0xff, D = (v >> 40) & 0xff;
            deblock_quartet(A, B, C, D, strength, gy < (h / 8) * 8);
            v = (v & 0xffff00000000ffffull) | ((uint64_t)A << 16) | ((uint64_t)B << 24) | ((uint64_t)C << 32) |
                ((uint64_t)D << 40);
            *reinterpret_cast<uint64_t *>(&s.y[row * POST_TW + 8 * j]) = v;
        }
    }
    if (a.luma_only) return;
    // chroma: 2 planes x 16 rows x 8 edges; the quartet is bytes 4..7 of an aligned 8-byte window
    {
        const int plane = tid >> 7, row = (tid >> 3) & 15, j = tid & 7;
        const int gy = yl / 2 + row, gxa = xl / 2 + 8 * j + 4;
        const int w = (int)a.L.cwidth, h = (int)a.L.cheight;
        if (gy >= 0 && gy < h && gxa >= 6 && gxa + 3 <= w - 1) {
            uint32_t v = *reinterpret_cast<uint32_t *>(&s.c[plane][row * POST_CW + 8 * j + 4]);
            int A = v & 0xff, B = (v >> 8) & 0xff, C = (v >> 16) & 0xff, D = (v >> 24) & 0xff;
            deblock_quartet(A, B, C, D, strength, gy < (h / 8) * 8);
            v = (uint32_t)A | ((uint32_t)B << 8) | ((uint32_t)C << 16) | ((uint32_t)D << 24);
            *reinterpret_cast<uint32_t *>(&s.c[plane][row * POST_CW + 8 * j + 4]) = v;
        }
    }
}

// ---- phase 3: BT.601 -> RGBA, optional filtered planes -------------------------------------
H263_DEV void post_phase_store(const PostArgs &a, PostSmem &s, int tid, int tile, int pic)
{
    const int tx = tile % (int)a.tiles_x, ty = tile / (int)a.tiles_x;
    const int xl = tx * POST_TW - POST_OX, yl = ty * POST_TH - POST_OY;
    const int w = (int)a.L.width, h = (int)a.L.height, cw = (int)a.L.cwidth, ch = (int)a.L.cheight;

    if (a.rgba) {
        uint8_t *rgba = a.rgba + (size_t)pic * w * h * 4;
        for (int it = 0; it < 4; it++) {
            const int item = it * POST_THREADS + tid, row = item >> 5, g = item & 31;
            const int gy = yl + row, gx = xl + 4 * g;
            if (gy < 0 || gy >= h || gx < 0 || gx >= w) continue;
            const uint32_t yv = *reinterpret_cast<const uint32_t *>(&s.y[row * POST_TW + 4 * g]);
            // nearest-neighbour chroma: pixel x uses sample x/2 of row y/2 (bt601.rs:96-98)
            const uint32_t cbv = *reinterpret_cast<const uint16_t *>(&s.c[0][(row >> 1) * POST_CW + 2 * g]);
            const uint32_t crv = *reinterpret_cast<const uint16_t *>(&s.c[1][(row >> 1) * POST_CW + 2 * g]);
            uint32_t px[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
                px[k] = bt601_pixel((yv >> (8 * k)) & 0xff, (cbv >> (8 * (k >> 1))) & 0xff, (crv >> (8 * (k >> 1))) & 0xff);
            uint8_t *dst = rgba + ((size_t)gy * w + gx) * 4;
            if (gx + 4 <= w && (w & 3) == 0) {
                *reinterpret_cast<uint4 *>(dst) = make_uint4(px[0], px[1], px[2], px[3]);
            } else {
                for (int k = 0; k < 4 && gx + k < w; k++) memcpy(dst + 4 * k, &px[k], 4);
            }
        }
    }
    if (a.planes_out) {
        // tightly packed Y | Cb | Cr, as deblock() returns them (deblock.rs:305-315)
        uint8_t *out = a.planes_out + (size_t)pic * ((size_t)w * h + 2 * (size_t)cw * ch);
        for (int it = 0; it < 4; it++) {
            const int item = it * POST_THREADS + tid, row = item >> 5, g = item & 31;
            const int gy = yl + row, gx = xl + 4 * g;
            if (gy < 0 || gy >= h || gx < 0) continue;
            for (int k = 0; k < 4 && gx + k < w; k++) out[(size_t)gy * w + gx + k] = s.y[row * POST_TW + 4 * g + k];
        }
        if (!a.luma_only) {
            for (int it = 0; it < 2; it++) {
                const int item = it * POST_THREADS + tid, plane = item >> 8, row = (item >> 4) & 15, g = item & 15;
                const int gy = yl / 2 + row, gx = xl / 2 + 4 * g;
                if (gy < 0 || gy >= ch) continue;
                uint8_t *o = out + (size_t)w * h + (size_t)plane * cw * ch;
                // the chroma tile origin is 2 mod 4: a group of 4 may straddle column 0
                for (int k = 0; k < 4; k++)
                    if (gx + k >= 0 && gx + k < cw) o[(size_t)gy * cw + gx + k] = s.c[plane][row * POST_CW + 4 * g + k];
            }
        }
    }
}

}  // namespace h263mi
