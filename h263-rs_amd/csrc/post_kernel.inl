// post_kernel.inl -- k_post: deblocking post-filter + BT.601 YUV 4:2:0 -> RGBA, fused.
//
// Replaces deblock::deblock (deblock/src/deblock.rs:305-315: deblock_horiz 136-181 then
// deblock_vert 185-299, kernel process/process_simd 29-42 / 99-127) applied to each plane
// and yuv::bt601::yuv420_to_rgba (yuv/src/bt601.rs:105-196, yuv_to_rgba_4x 12-59).
//
// Work unit = one WAVE = one strip: 128 luma columns x 8 luma rows plus the matching 64x4
// strips of Cb and Cr, with the strip origin at x = 4 (mod 128), y = 4 (mod 8).  Every
// filtered pixel group -- rows edge-2..edge+1 of a horizontal block edge, columns
// edge-2..edge+1 of a vertical one -- then lies wholly inside one strip, for luma (edges at
// multiples of 8) AND for chroma (strip origin (2,2) mod (8,4)).  So a wave needs nothing from
// any other wave: load strip -> H-edge -> V-edges -> convert -> 16-byte RGBA stores, through
// 1.5 KB of wave-private LDS, with NO workgroup barrier (DS operations of one wave execute in
// order, which is all the cross-lane hand-offs need).  A workgroup is four such waves (a
// 128x32 tile) only so that the XCD-aware work order of kernels.hip keeps neighbours together.
//
// The reference mixes two integer semantics by position: its
// SIMD lanes use arithmetic shifts (floor), its scalar tails use `/` (truncation).
//   horizontal edges: floor for columns < 8*floor(w/8), truncation right of that;
//   vertical edges  : floor for rows    < 8*floor(h/8), truncation below that.
#pragma once

#include "dev_common.h"

namespace h263mi {

#ifndef H263MI_POST_WAVES
#define H263MI_POST_WAVES 1       // measured: 1 -> 0.157 ms, 2 -> 0.162 ms, 4 -> 0.162 ms per launch (64 x 1080p)
#endif
constexpr int POST_WAVES = H263MI_POST_WAVES;       // waves per workgroup (independent: 1, 2 or 4)
constexpr int POST_THREADS = POST_WAVES * 64;
constexpr int POST_GROUP = 4;                       // vertically adjacent tiles that follow each other in the work list
constexpr int POST_TW = 128, POST_TH = 32;          // luma tile of a workgroup = 4 strips of 8 rows
constexpr int POST_SH = 8;                          // luma rows per strip (one wave)
constexpr int POST_CW = 64, POST_CSH = 4;           // chroma strip
constexpr int POST_OX = POST_TW - 4;                // strip column sx starts at sx*128 - 124
constexpr int POST_STRIPS = 4;                      // strips (vertical neighbours) per wave = one 128x32 tile

struct PostStrip {
    uint8_t y[POST_SH * POST_TW];
    uint8_t c[2][POST_CSH * POST_CW];
};

H263_HD uint32_t post_strips_y(uint32_t h) { return (h + 4 + POST_SH - 1) / POST_SH; }

// Tile columns of a picture, and the WRAP.  Tile column sx covers picture columns sx*128 - 124 .. sx*128 + 3, so column 0
// holds just the 4 leftmost picture columns (and 2 chroma columns) -- a wave for 4 x 32 pixels -- while the last tile
// usually has columns to spare beyond the right picture edge (4 of them at 1920).  The 4 left columns take part in no
// vertical-edge quartet (the first one is columns 6..9, deblock.rs:281) and horizontal edges are filtered column by
// column, so they can sit anywhere: when the last tile has >= 4 spare columns and the width is a multiple of 4, they
// ride in its strip columns w - xl .. w - xl + 3 (chroma: the 2 columns behind the last chroma column), tile column 0
// is dropped (a.wrap = 1: the first tile is sx = 1) and a 1080p picture takes 15 instead of 16 waves per 32 rows.
// In those strip columns the picture column is x - w; the quartet test of the vertical edges sees them as outside.
H263_HD uint32_t post_tile_columns(uint32_t w, uint32_t *wrap)
{
    const uint32_t tiles = (w + POST_OX + POST_TW - 1) / POST_TW;
    const uint32_t spare = tiles * POST_TW - POST_OX - w;
    *wrap = (tiles >= 2 && (w % 4) == 0 && spare >= 4) ? 1u : 0u;
    return tiles - *wrap;
}
// picture column of strip column `x` (luma), `cx` (chroma)
// (exactly the 4 / 2 columns behind the picture edge: whatever else the last tile has to spare stays outside)
H263_HD int post_wrap_x(const PostArgs &a, int x)
{
    const int w = (int)a.L.width;
    return (a.wrap && x >= w && x < w + 4) ? x - w : x;
}
H263_HD int post_wrap_cx(const PostArgs &a, int cx)
{
    const int cw = (int)a.L.cwidth;
    return (a.wrap && cx >= cw && cx < cw + 2) ? cx - cw : cx;
}

// One A,B,C,D quartet (deblock.rs:29-42 / 99-127).  The reference's SIMD lanes divide with arithmetic shifts
// (floor), its scalar tails with `/` (truncation toward zero).  Both are one shift once a bias is added to
// negative numerators: trunc(x / 2^k) = (x + ((x >> 31) & (2^k - 1))) >> k.  `tm` is 0 for the floor semantics
// and all ones for truncation, so the bias vanishes where the reference shifts.
//
// The rest is arranged for the instruction prices of gfx950 (profiles/r01_valu_rate.txt: add / sub / and / xor /
// shifts right are half the price of min / max / compare / select):
//   up_down_ramp (deblock.rs:13-15)  max(0, |d| - max(0, 2(|d| - S))) = median(0, |d|, 2S - |d|)
//   |d1 / 2|: d1 = +-mag, so it is mag >> 1, or (mag + 1) >> 1 for a negative d1 under floor division
//   clipd1 (deblock.rs:19-21) and the two saturating outputs are medians as well
H263_HD int median3_i32(int a, int b, int c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    return c < lo ? lo : (c > hi ? hi : c);
#endif
}

// (x >> 31) & m as two plain instructions (the compiler otherwise turns it into compare + select)
H263_HD int sign_and(int x, int m)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int s;
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(s) : "v"(x));
    return s & m;
#else
    return (x >> 31) & m;
#endif
}

H263_HD void deblock_quartet_tm(int &A, int &B, int &C, int &D, int strength, int tm)
{
    const int ad_ = A - D;
    const int x = ad_ + 4 * (C - B);
    const int d = (x + sign_and(x, 7 & tm)) >> 3;
    const int sd = d >> 31;                                   // 0 or -1
    const int nd = -d;
    const int ad = d > nd ? d : nd;                           // |d|
    const int mag = median3_i32(ad, 0, 2 * strength - ad);    // |d1|
    const int d1 = (mag ^ sd) - sd;
    const int lim = (mag + (sd & ~tm & 1)) >> 1;              // |d1 / 2| in the division of this position
    const int q = (ad_ + sign_and(ad_, 3 & tm)) >> 2;
    const int d2 = median3_i32(q, -lim, lim);
    A = A - d2;                                               // `as u8`: wraps, no clamp -- the caller keeps the low byte only
    B = median3_i32(B + d1, 0, 255);
    C = median3_i32(C - d1, 0, 255);
    D = D + d2;                                               // likewise
}

H263_HD void deblock_quartet(int &A, int &B, int &C, int &D, int strength, bool floor_sem)
{
    deblock_quartet_tm(A, B, C, D, strength, floor_sem ? 0 : -1);
}

// TWO quartets at once, one in each 16-bit half of a dword (v_pk_* instructions on the device): the same formula as
// deblock_quartet_tm -- every intermediate fits 12 bits.  k carries the per-position constants in both halves.
// In: byte values 0..255 per half.  Out: A and D wrapped results in the LOW BYTE of each half (the reference's
// `as u8`, deblock.rs:38,41); B and C unsaturated (-255..510): the caller packs them with a saturating pack, which
// is the clamp of deblock.rs:39-40.
struct QuartetConsts {
    uint32_t s2;               // 2 * strength
    uint32_t c7, c3;           // 7 & tm, 3 & tm: the bias that turns the shifts into truncating divisions
    uint32_t c1;               // 1 & ~tm: |d1 / 2| rounds away from zero for a negative d1 under floor division
};
H263_HD QuartetConsts quartet_consts(int strength, int tm)
{
    QuartetConsts k;
    k.s2 = (uint32_t)(2 * strength) * 0x00010001u;
    k.c7 = 0x00070007u & (uint32_t)tm;
    k.c3 = 0x00030003u & (uint32_t)tm;
    k.c1 = 0x00010001u & ~(uint32_t)tm;
    return k;
}
// FLOOR: both quartets lie where the reference divides with arithmetic shifts (every quartet of an interior tile): the
// truncation biases are zero and their two additions are left out (written through the opaque packed helpers, "+ 0"
// is not something the compiler can fold).
template <bool FLOOR = false>
H263_DEV void deblock_quartet_pk(uint32_t &A, uint32_t &B, uint32_t &C, uint32_t &D, const QuartetConsts &k)
{
    const uint32_t adm = pk_sub_u16(A, D);
    const uint32_t x = pk_mad_i16(pk_sub_u16(C, B), 0x00040004u, adm);
    const uint32_t d = pk_ashr_i16(FLOOR ? x : pk_add_u16(x, pk_ashr_i16(x, 15) & k.c7), 3);
    const uint32_t sd = pk_ashr_i16(d, 15);                                   // 0 or -1 per half
    const uint32_t ad = pk_max_i16(d, pk_sub_u16(0u, d));                     // |d|
    const uint32_t mag = pk_max_i16(pk_min_i16(ad, pk_sub_u16(k.s2, ad)), 0u);   // up_down_ramp = median(0, |d|, 2S - |d|)
    const uint32_t d1 = pk_sub_u16(mag ^ sd, sd);
    const uint32_t lim = pk_lshr_u16(pk_add_u16(mag, sd & k.c1), 1);          // |d1 / 2| in the division of this position
    const uint32_t q = pk_ashr_i16(FLOOR ? adm : pk_add_u16(adm, pk_ashr_i16(adm, 15) & k.c3), 2);
    const uint32_t d2 = pk_max_i16(pk_min_i16(q, lim), pk_sub_u16(0u, lim));   // clipd1
    A = pk_sub_u16(A, d2);
    B = pk_add_u16(B, d1);
    C = pk_sub_u16(C, d1);
    D = pk_add_u16(D, d2);
}

// byte k0 of x into bits 7:0, byte k1 into bits 23:16, zeros elsewhere: two bytes as an i16 pair
H263_DEV uint32_t bytes_to_pair(uint32_t x, int k0, int k1)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(0u, x, 0x0c000c00u | (uint32_t)k0 | ((uint32_t)k1 << 16));
#else
    return ((x >> (8 * k0)) & 0xffu) | (((x >> (8 * k1)) & 0xffu) << 16);
#endif
}
// the same with the two bytes taken from two dwords: byte k of lo_src -> bits 7:0, byte k of hi_src -> bits 23:16
H263_DEV uint32_t bytes_to_pair2(uint32_t lo_src, uint32_t hi_src, int k)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi_src, lo_src, 0x0c000c00u | (uint32_t)k | ((uint32_t)(4 + k) << 16));
#else
    return ((lo_src >> (8 * k)) & 0xffu) | (((hi_src >> (8 * k)) & 0xffu) << 16);
#endif
}
// low bytes of the two halves of an i16 pair -> bits 7:0 and 15:8
H263_DEV uint32_t pair_low_bytes(uint32_t p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(0u, p, 0x0c0c0200u);
#else
    return (p & 0xffu) | (((p >> 16) & 0xffu) << 8);
#endif
}

// tm for position `pos` against the end of the reference's SIMD region: 0 (floor) for pos < simd_end, else -1
H263_HD int trunc_mask(int pos, int simd_end) { return (simd_end - 1 - pos) >> 31; }

// byte shuffles of the filtered samples (v_perm_b32 picks byte 0 of each operand: no masks, no shifts)
// bytes 0,1 of `keep_lo` below bytes 0,1 of `put_hi`
H263_DEV uint32_t splice_lo16_hi16(uint32_t keep_lo, uint32_t put_hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(put_hi, keep_lo, 0x05040100u);
#else
    return (keep_lo & 0xffffu) | (put_hi << 16);
#endif
}
// bytes 0,1 of `put_lo` below bytes 2,3 of `keep_hi`
H263_DEV uint32_t splice_put16_keep16(uint32_t put_lo, uint32_t keep_hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(keep_hi, put_lo, 0x07060100u);
#else
    return (put_lo & 0xffffu) | (keep_hi & 0xffff0000u);
#endif
}
H263_DEV uint32_t pack2_u8(int lo, int hi)                     // (lo & 0xff) | (hi & 0xff) << 8
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x0c0c0400u);
#else
    return ((uint32_t)lo & 0xffu) | (((uint32_t)hi & 0xffu) << 8);
#endif
}
// ---- interior tiles ---------------------------------------------------------------------------
// A tile (4 strips) is INTERIOR when every byte its strips load, every sample its filters touch and every pixel it
// converts lies inside the picture, inside the region where the reference's SIMD lanes do the filtering (floor
// division: columns < 8*floor(w/8), rows < 8*floor(h/8), for luma and chroma alike), and the call asks for RGBA only.
// 82 % of the tiles of a 1080p picture are.  For such a tile every bounds test of the phases below is true: the
// INTERIOR instantiations drop them all -- no clamps, no keep-masks, no predicated stores, constant division biases --
// and, the point of the exercise, become STRAIGHT-LINE code with a fixed number of vector memory operations per strip
// (2 loads, 4 stores), so that the compiler can leave exactly the younger loads and stores in flight at each wait
// (s_waitcnt vmcnt(N)) instead of draining the queue.  Wave-uniform, decided once per wave.
H263_HD bool post_tile_is_interior(const PostArgs &a, int sx, int ty)
{
    const int xl = sx * POST_TW - POST_OX, yl = ty * POST_STRIPS * POST_SH - 4;
    const int w8 = (int)(a.L.width / 8) * 8, h8 = (int)(a.L.height / 8) * 8;
    const int cw8 = (int)(a.L.cwidth / 8) * 8, ch8 = (int)(a.L.cheight / 8) * 8;
    return xl >= 0 && yl >= 0 && xl + POST_TW <= w8 && xl / 2 + POST_CW <= cw8 && yl + POST_STRIPS * POST_SH <= h8 &&
           yl / 2 + POST_STRIPS * POST_CSH <= ch8 && a.rgba != nullptr && a.planes_out == nullptr && !a.luma_only;
}

// ---- phase 0: strip -> registers -> LDS -----------------------------------------------------
struct PostFetch {
    uint32_t y[4];
    uint32_t c[4];     // edge tiles: one 16-bit pair each, kept unpacked so that nothing touches them before the commit;
                       // interior tiles: c[0], c[1] = the lane's 8 chroma bytes as loaded
};

// lane = 16 luma bytes (row = lane/8) and 8 chroma bytes (plane = lane/32, row = (lane/8)%4).
// Every load is issued unconditionally from a clamped address (bytes outside the picture are never
// used), so the number of loads per strip is fixed: a wave queues the loads of several strips up
// front and the wait in front of each strip leaves the later ones in flight (s_waitcnt vmcnt(N)).
template <bool INTERIOR = false>
H263_DEV void post_phase_fetch(const PostArgs &a, PostFetch &r, int lane, int sx, int sy, int pic)
{
    const uint8_t *frame = a.frames + (size_t)pic * a.L.frame_bytes;
    const int xl = sx * POST_TW - POST_OX, yl = sy * POST_SH - 4;
    if (INTERIOR) {
        // two loads, nothing else: 16 luma bytes at a 4-byte aligned address, 8 chroma bytes at a 2-byte aligned one
        const uint8_t *py = frame + ((uint32_t)(yl + (lane >> 3)) * a.L.pitch_y + (uint32_t)(xl + (lane & 7) * 16));
        const uint8_t *pc = frame + ((lane >> 5 ? a.L.off_cr : a.L.off_cb) +
                                     (uint32_t)(yl / 2 + ((lane >> 3) & 3)) * a.L.pitch_c + (uint32_t)(xl / 2 + (lane & 7) * 8));
#if defined(__HIP_DEVICE_COMPILE__)
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef u32x4 __attribute__((aligned(4))) u32x4_a4;
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        typedef u32x2 __attribute__((aligned(2))) u32x2_a2;
        const u32x4 vy = *reinterpret_cast<const u32x4_a4 *>(py);
        const u32x2 vc = *reinterpret_cast<const u32x2_a2 *>(pc);
        r.y[0] = vy.x; r.y[1] = vy.y; r.y[2] = vy.z; r.y[3] = vy.w;
        r.c[0] = vc.x; r.c[1] = vc.y;
#else
        memcpy(r.y, py, 16);
        memcpy(r.c, pc, 8);
#endif
        return;
    }
    // Only the first and the last tile of a row reach beyond the allocated row (uniform test): everywhere else the
    // lane's bytes are contiguous and need no per-dword clamping -- one offset, the rest are immediate offsets.
    const bool inside = xl >= 0 && xl + POST_TW <= (int)a.L.pitch_y;
    {
        // uniform base + 32-bit per-lane offset: the loads use scalar-base addressing, no 64-bit VALU adds
        const int row = lane >> 3, col = (lane & 7) * 16;
        const uint32_t rowoff = (uint32_t)clampi(yl + row, 0, (int)a.L.rows_y - 1) * a.L.pitch_y;
        if (inside) {
            const uint8_t *p = frame + (rowoff + (uint32_t)(xl + col));
#if defined(__HIP_DEVICE_COMPILE__)
            // ONE 16-byte load at a 4-byte aligned address (the strip origin is 4 mod 16) instead of four dword loads,
            // and one 8-byte load at a 2-byte aligned address for the chroma bytes below: a strip is fetched with 2 vector
            // memory instructions instead of 8.  -5.6 % on k_post alone, -1.4 % on k_frame (A/B, three rounds).
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            typedef u32x4 __attribute__((aligned(4))) u32x4_a4;
            const u32x4 v = *reinterpret_cast<const u32x4_a4 *>(p);
            r.y[0] = v.x; r.y[1] = v.y; r.y[2] = v.z; r.y[3] = v.w;
#else
            for (int q = 0; q < 4; q++) memcpy(&r.y[q], p + 4 * q, 4);
#endif
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                // (wrap: the dword behind the last picture column is the dword of picture columns 0..3)
                const uint32_t gx = (uint32_t)clampi(post_wrap_x(a, xl + col + 4 * q), 0, (int)a.L.pitch_y - 4);
                r.y[q] = *reinterpret_cast<const uint32_t *>(frame + (rowoff + gx));
            }
        }
    }
    {
        // (a luma-only call has no chroma planes: the loads still run, from the luma plane, and are ignored)
        const int plane = lane >> 5, row = (lane >> 3) & 3, col = (lane & 7) * 8;
        const int cxl = xl / 2, cyl = yl / 2;                 // xl, yl are even (and may be negative)
        const uint32_t rowoff = (a.luma_only ? 0u : (plane ? a.L.off_cr : a.L.off_cb)) +
                                (uint32_t)clampi(cyl + row, 0, (int)a.L.rows_c - 1) * a.L.pitch_c;
        if (inside) {
            const uint8_t *p = frame + (rowoff + (uint32_t)(cxl + col));
#if defined(__HIP_DEVICE_COMPILE__)
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            typedef u32x2 __attribute__((aligned(2))) u32x2_a2;
            // (handing the two dwords to the commit as they are, with a wave-uniform switch there, was 4 % SLOWER: the
            // four halves below are what keeps the compiler's schedule of the commit the same for both kinds of tile)
            const u32x2 v = *reinterpret_cast<const u32x2_a2 *>(p);
            r.c[0] = v.x & 0xffffu; r.c[1] = v.x >> 16; r.c[2] = v.y & 0xffffu; r.c[3] = v.y >> 16;
#else
            for (int q = 0; q < 4; q++) { uint16_t t; memcpy(&t, p + 2 * q, 2); r.c[q] = t; }
#endif
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t gx = (uint32_t)clampi(post_wrap_cx(a, cxl + col + 2 * q), 0, (int)a.L.pitch_c - 2);
                r.c[q] = *reinterpret_cast<const uint16_t *>(frame + (rowoff + gx));
            }
        }
    }
}

template <bool INTERIOR = false>
H263_DEV void post_phase_commit(const PostArgs &a, PostStrip &s, const PostFetch &r, int lane)
{
    {
        const int row = lane >> 3, col = (lane & 7) * 16;
        *reinterpret_cast<uint4 *>(&s.y[row * POST_TW + col]) = make_uint4(r.y[0], r.y[1], r.y[2], r.y[3]);
    }
    if (INTERIOR) {
        const int plane = lane >> 5, row = (lane >> 3) & 3, col = (lane & 7) * 8;
        *reinterpret_cast<uint64_t *>(&s.c[plane][row * POST_CW + col]) = (uint64_t)r.c[0] | ((uint64_t)r.c[1] << 32);
        return;
    }
    if (a.luma_only) return;
    {
        const int plane = lane >> 5, row = (lane >> 3) & 3, col = (lane & 7) * 8;
        uint64_t packed = (uint64_t)(r.c[0] & 0xffff) | ((uint64_t)(r.c[1] & 0xffff) << 16) | ((uint64_t)(r.c[2] & 0xffff) << 32) | ((uint64_t)r.c[3] << 48);
        *reinterpret_cast<uint64_t *>(&s.c[plane][row * POST_CW + col]) = packed;
    }
}

// filter 2 neighbouring columns of the horizontal edge whose A row is `row_a`: one packed quartet pair.
// Columns outside the picture (the strip's 4-pixel offset, the right picture edge) keep their bytes.
// `edge_tile` (uniform): the tile reaches beyond the left or the right picture edge; only then can a column lie outside.
template <bool INTERIOR = false>
H263_DEV void hfilter2(uint8_t *t, int pitch, int row_a, int col, int strength, int gx0, int floor_cols, int w, bool edge_tile)
{
    uint32_t r[4];
#pragma unroll
    for (int q = 0; q < 4; q++) r[q] = *reinterpret_cast<uint16_t *>(t + (row_a + q) * pitch + col);
    // the two columns are 2k, 2k+1 and floor_cols is a multiple of 8: both lie on the same side of it
    // (an interior tile lies left of floor_cols altogether: floor division, constant biases)
    const QuartetConsts k = quartet_consts(strength, INTERIOR ? 0 : trunc_mask(gx0, floor_cols));
    uint32_t A = bytes_to_pair(r[0], 0, 1), B = bytes_to_pair(r[1], 0, 1), C = bytes_to_pair(r[2], 0, 1), D = bytes_to_pair(r[3], 0, 1);
    deblock_quartet_pk<INTERIOR>(A, B, C, D, k);
    uint32_t o[4] = {pair_low_bytes(A), sat_pk_u8_i16(B), sat_pk_u8_i16(C), pair_low_bytes(D)};
    if (!INTERIOR && edge_tile) {
        const uint32_t keep = (gx0 >= 0 && gx0 < w ? 0u : 0x00ffu) | (gx0 + 1 >= 0 && gx0 + 1 < w ? 0u : 0xff00u);
#pragma unroll
        for (int q = 0; q < 4; q++) o[q] = (o[q] & ~keep) | (r[q] & keep);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) *reinterpret_cast<uint16_t *>(t + (row_a + q) * pitch + col) = (uint16_t)o[q];
}

// ---- phase 1: the horizontal block edge of the strip (deblock_horiz, deblock.rs:136-181) ------
template <bool INTERIOR = false>
H263_DEV void post_phase_hedges(const PostArgs &a, PostStrip &s, int lane, int sx, int sy)
{
    const int xl = sx * POST_TW - POST_OX;
    const int strength = (int)a.strength;
    if (INTERIOR) {
        // every strip of an interior tile holds a luma edge that the reference filters (8 <= 8*sy <= h - 2), every
        // other one a chroma edge
        hfilter2<true>(s.y, POST_TW, 2, lane * 2, strength, 0, 0, 0, false);
        if ((sy & 1) == 0) hfilter2<true>(s.c[lane >> 5], POST_CW, 0, (lane & 31) * 2, strength, 0, 0, 0, false);
        return;
    }
    const bool edge_tile = xl < 0 || xl + POST_TW > (int)a.L.width;     // (the chroma strip reaches as far, in its own units)
    {
        // luma: the edge's C row is picture row 8*sy = strip row 4; every lane takes 2 columns
        const int gy = sy * POST_SH, w = (int)a.L.width, h = (int)a.L.height;
        if (gy >= 8 && gy <= h - 2)                                      // edge_y <= height - 2 (deblock.rs:140)
            hfilter2(s.y, POST_TW, 2, lane * 2, strength, post_wrap_x(a, xl + lane * 2), (w / 8) * 8, w, edge_tile);
    }
    if (!a.luma_only && (sy & 1) == 0) {
        // chroma strip rows [4*sy-2, 4*sy+2) hold an edge only when 4*sy is a multiple of 8
        const int gy = sy * POST_CSH, w = (int)a.L.cwidth, h = (int)a.L.cheight;
        const int plane = lane >> 5, col = (lane & 31) * 2;
        if (gy >= 8 && gy <= h - 2)
            hfilter2(s.c[plane], POST_CW, 0, col, strength, post_wrap_cx(a, xl / 2 + col), (w / 8) * 8, w, edge_tile);
    }
}

// ---- phase 2: vertical block edges (deblock_vert, deblock.rs:185-299) ----------------------
// A lane takes the same edge in two vertically adjacent rows 2q, 2q + 1 -- one packed quartet pair.  The strip origin
// is 4 (mod 8) in y (2 mod 4 for chroma), so a row pair never straddles a multiple of 8: both rows divide alike.
template <bool INTERIOR = false>
H263_DEV void post_phase_vedges(const PostArgs &a, PostStrip &s, int lane, int sx, int sy)
{
    const int xl = sx * POST_TW - POST_OX, yl = sy * POST_SH - 4;
    const int strength = (int)a.strength;
    // (interior tiles: every quartet of the strip lies inside the picture and in the floor-division region)
    {
        // luma: 4 row pairs x 16 edges; the quartet sits in bytes 2..5 of an aligned 8-byte window
        const int row = (lane >> 4) * 2, j = lane & 15;
        const int gy = yl + row, gxa = xl + 8 * j + 2;         // picture column of the "A" sample
        const int w = (int)a.L.width, h = (int)a.L.height;
        // A..D = columns 8k-2 .. 8k+1 with k >= 1 and 8k+1 <= w-1 (chunks of row[2..], deblock.rs:281)
        if (INTERIOR || (gy + 1 >= 0 && gy < h && gxa >= 6 && gxa + 3 <= w - 1)) {
            uint64_t *p0 = reinterpret_cast<uint64_t *>(&s.y[row * POST_TW + 8 * j]);
            uint64_t *p1 = reinterpret_cast<uint64_t *>(&s.y[(row + 1) * POST_TW + 8 * j]);
            const uint64_t v0 = *p0, v1 = *p1;
            const uint32_t l0 = (uint32_t)v0, h0 = (uint32_t)(v0 >> 32), l1 = (uint32_t)v1, h1 = (uint32_t)(v1 >> 32);
            uint32_t A = bytes_to_pair2(l0, l1, 2), B = bytes_to_pair2(l0, l1, 3), C = bytes_to_pair2(h0, h1, 0), D = bytes_to_pair2(h0, h1, 1);
            deblock_quartet_pk<INTERIOR>(A, B, C, D, quartet_consts(strength, INTERIOR ? 0 : trunc_mask(gy < 0 ? gy + 1 : gy, (h / 8) * 8)));
            // ab = [A'0, B'0, A'1, B'1], cd = [C'0, D'0, C'1, D'1] as bytes
            const uint32_t bs = sat_pk_u8_i16(B), cs = sat_pk_u8_i16(C);
#if defined(__HIP_DEVICE_COMPILE__)
            const uint32_t ab = __builtin_amdgcn_perm(bs, A, 0x05020400u), cd = __builtin_amdgcn_perm(D, cs, 0x06010400u);
            const uint32_t nl0 = __builtin_amdgcn_perm(ab, l0, 0x05040100u), nl1 = __builtin_amdgcn_perm(ab, l1, 0x07060100u);
            const uint32_t nh0 = __builtin_amdgcn_perm(cd, h0, 0x03020504u), nh1 = __builtin_amdgcn_perm(cd, h1, 0x03020706u);
#else
            const uint32_t a0 = A & 0xffu, a1 = (A >> 16) & 0xffu, b0 = bs & 0xffu, b1 = (bs >> 8) & 0xffu;
            const uint32_t c0 = cs & 0xffu, c1 = (cs >> 8) & 0xffu, d0 = D & 0xffu, d1 = (D >> 16) & 0xffu;
            const uint32_t nl0 = (l0 & 0xffffu) | (a0 << 16) | (b0 << 24), nl1 = (l1 & 0xffffu) | (a1 << 16) | (b1 << 24);
            const uint32_t nh0 = (h0 & 0xffff0000u) | c0 | (d0 << 8), nh1 = (h1 & 0xffff0000u) | c1 | (d1 << 8);
#endif
            // rows outside the picture (above the first strip, below the last row) keep their bytes
            if (INTERIOR || gy >= 0) *p0 = (uint64_t)nl0 | ((uint64_t)nh0 << 32);
            if (INTERIOR || gy + 1 < h) *p1 = (uint64_t)nl1 | ((uint64_t)nh1 << 32);
        }
    }
    if (!INTERIOR && a.luma_only) return;
    // chroma: 2 planes x 2 row pairs x 8 edges on lanes 0..31; the quartet is bytes 4..7 of an aligned 8-byte window
    if (lane < 32) {
        const int plane = lane >> 4, row = ((lane >> 3) & 1) * 2, j = lane & 7;
        const int gy = yl / 2 + row, gxa = xl / 2 + 8 * j + 4;
        const int w = (int)a.L.cwidth, h = (int)a.L.cheight;
        if (INTERIOR || (gy + 1 >= 0 && gy < h && gxa >= 6 && gxa + 3 <= w - 1)) {
            uint32_t *p0 = reinterpret_cast<uint32_t *>(&s.c[plane][row * POST_CW + 8 * j + 4]);
            uint32_t *p1 = reinterpret_cast<uint32_t *>(&s.c[plane][(row + 1) * POST_CW + 8 * j + 4]);
            const uint32_t v0 = *p0, v1 = *p1;
            uint32_t A = bytes_to_pair2(v0, v1, 0), B = bytes_to_pair2(v0, v1, 1), C = bytes_to_pair2(v0, v1, 2), D = bytes_to_pair2(v0, v1, 3);
            deblock_quartet_pk<INTERIOR>(A, B, C, D, quartet_consts(strength, INTERIOR ? 0 : trunc_mask(gy < 0 ? gy + 1 : gy, (h / 8) * 8)));
            const uint32_t bs = sat_pk_u8_i16(B), cs = sat_pk_u8_i16(C);
#if defined(__HIP_DEVICE_COMPILE__)
            const uint32_t ab = __builtin_amdgcn_perm(bs, A, 0x05020400u), cd = __builtin_amdgcn_perm(D, cs, 0x06010400u);
            const uint32_t n0 = __builtin_amdgcn_perm(cd, ab, 0x05040100u), n1 = __builtin_amdgcn_perm(cd, ab, 0x07060302u);
#else
            const uint32_t n0 = (A & 0xffu) | ((bs & 0xffu) << 8) | ((cs & 0xffu) << 16) | ((D & 0xffu) << 24);
            const uint32_t n1 = ((A >> 16) & 0xffu) | (((bs >> 8) & 0xffu) << 8) | (((cs >> 8) & 0xffu) << 16) | (((D >> 16) & 0xffu) << 24);
#endif
            if (INTERIOR || gy >= 0) *p0 = n0;
            if (INTERIOR || gy + 1 < h) *p1 = n1;
        }
    }
}

// 16 bytes to an address that is a multiple of 4 (global_store_dwordx4 asks for no more).  STREAM: a non-temporal
// store (`nt`) for output that nothing on the device reads again.
template <bool STREAM>
H263_DEV void store16_align4(uint8_t *dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef u32x4 __attribute__((aligned(4))) u32x4_a4;
    const u32x4 v = {a, b, c, d};
    if (STREAM) __builtin_nontemporal_store(v, reinterpret_cast<u32x4_a4 *>(dst));
    else *reinterpret_cast<u32x4_a4 *>(dst) = v;
#else
    const uint32_t v[4] = {a, b, c, d};
    memcpy(dst, v, 16);
#endif
}

// ---- phase 3: BT.601 -> RGBA, optional filtered planes -------------------------------------
// bt601.rs:25-58 regrouped so that everything that depends only on the chroma sample is computed
// once per 2x2 quad:  R = (Y*76309 + [Cr*104597 + K]) >> 16, etc., K = 32768 - 16*76309 - 128*coef.
// The sums are identical integers to the reference's (gray + cr2r + half), only associated differently.
struct ChromaTerms {
    int r, g, b;
};
H263_HD ChromaTerms bt601_chroma_terms(int cb, int cr)
{
    const int K = 32768 - 16 * 76309;
    ChromaTerms t;
    t.r = cr * 104597 + (K - 128 * 104597);
    t.g = cr * -53279 + cb * -25675 + (K + 128 * 53279 + 128 * 25675);
    t.b = cb * 132201 + (K - 128 * 132201);
    return t;
}

// clamp(v >> 16, 0, 255) for three channels, packed as R | G<<8 | B<<16 | 255<<24
H263_DEV uint32_t bt601_pack(int r, int g, int b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // v_ashr_pk_u8_i32 does shift + saturate + pack for two values into one half of its destination and keeps the
    // other half: bits 15:0 in the plain form (tools/probes/probe_ashr_pk.hip), bits 31:16 with op_sel:[0,0,0,1]
    // (tools/probes/probe_ashr_pk_hi.hip, profiles/r02_probe_ashr_pk_hi.txt) -- four bytes in two instructions.
    uint32_t px;
    const int alpha = 255 << 16;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 16" : "=v"(px) : "v"(r), "v"(g));
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 16 op_sel:[0,0,0,1]" : "+v"(px) : "v"(b), "v"(alpha));
    return px;
#else
    // clamp(v >> 16, 0, 255) == clamp(v, 0, 0xFFFFFF) >> 16 (the shift is monotone)
    const uint32_t R = (uint32_t)clampi(r, 0, 0xFFFFFF) >> 16;
    const uint32_t G = (uint32_t)clampi(g, 0, 0xFFFFFF) >> 16;
    const uint32_t B = (uint32_t)clampi(b, 0, 0xFFFFFF) >> 16;
    return R | (G << 8) | (B << 16) | 0xff000000u;
#endif
}

// STREAM_RGBA: the RGBA stores are non-temporal.  k_frame sets it: the 8.3 MB of RGBA per picture then no longer push
// the planes that the reconstruction half of the same launch (and the next one) reads out of the L2 / infinity cache --
// 10 % on a frame index (profiles/README.md).  k_post on its own keeps plain stores: alone, it is 10 % faster with them.
template <bool STREAM_RGBA, bool INTERIOR = false>
H263_DEV void post_phase_store(const PostArgs &a, PostStrip &s, int lane, int sx, int sy, int pic)
{
    const int xl = sx * POST_TW - POST_OX, yl = sy * POST_SH - 4;
    const int w = (int)a.L.width, h = (int)a.L.height, cw = (int)a.L.cwidth, ch = (int)a.L.cheight;

    if (INTERIOR || a.rgba) {
        // uniform 64-bit base of the picture + 32-bit lane offsets (w * h * 4 < 2^32: layout_fits)
#if defined(H263MI_TIMING_RGBA_FOLD)
        uint8_t *rgba = a.rgba;
#else
        uint8_t *rgba = a.rgba + (size_t)pic * w * h * 4;
#endif
        const int g = lane & 31, gx = INTERIOR ? xl + 4 * g : post_wrap_x(a, xl + 4 * g);
        // the lane's four pixels: all inside the picture (the only case away from the left / right picture edge),
        // or some of them (a picture whose width is not a multiple of 4), or none
        const bool col_full = INTERIOR || (gx >= 0 && gx + 4 <= w), col_some = INTERIOR || (gx >= 0 && gx < w);
        const uint32_t row_bytes = (uint32_t)w * 4u;
#pragma unroll
        for (int it = 0; it < 2; it++) {
            // a lane converts a 4x2 block: two rows that share one chroma row (nearest-neighbour
            // chroma, bt601.rs:96-98); 32 consecutive lanes write 512 contiguous bytes of a row
            const int q = (lane >> 5) + 2 * it;                  // chroma row of the strip, 0..3
            const uint32_t cbv = *reinterpret_cast<const uint16_t *>(&s.c[0][q * POST_CW + 2 * g]);
            const uint32_t crv = *reinterpret_cast<const uint16_t *>(&s.c[1][q * POST_CW + 2 * g]);
            const ChromaTerms t0 = bt601_chroma_terms(cbv & 0xff, crv & 0xff);
            const ChromaTerms t1 = bt601_chroma_terms(cbv >> 8, crv >> 8);
            const int gy0 = yl + 2 * q;
            // byte offset of pixel (gx, gy0), modulo 2^32: gy0 may be -1 with row gy0 + 1 inside the picture
            const uint32_t off0 = ((uint32_t)gy0 * (uint32_t)w + (uint32_t)gx) * 4u;
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                const int row = 2 * q + rr, gy = gy0 + rr;
                if (!INTERIOR && ((uint32_t)gy >= (uint32_t)h || !col_some)) continue;     // row or lane outside the picture
                const uint32_t yv = *reinterpret_cast<const uint32_t *>(&s.y[row * POST_TW + 4 * g]);
                uint32_t px[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    int gray = (int)((yv >> (8 * k)) & 0xff) * 76309;
#if defined(__HIP_DEVICE_COMPILE__)
                    // keep the product on its own: one 24-bit multiply + three plain adds issue faster than the three
                    // multiply-adds the compiler would make of it (profiles/r01_valu_rate.txt)
                    asm volatile("" : "+v"(gray));
#endif
                    const ChromaTerms &t = (k < 2) ? t0 : t1;
                    px[k] = bt601_pack(gray + t.r, gray + t.g, gray + t.b);
                }
#if defined(H263MI_TIMING_RGBA_FOLD)
                // TIMING EXPERIMENT ONLY (results wrong): every RGBA store of the launch lands in the first H263MI_TIMING_RGBA_FOLD
                // bytes of the surface -- the store instructions, their lines and requests stay, the DRAM traffic goes
                const uint32_t off = ((off0 + (rr ? row_bytes : 0u)) + (uint32_t)pic * 8294400u) & (uint32_t)(H263MI_TIMING_RGBA_FOLD - 1);
#else
                const uint32_t off = off0 + (rr ? row_bytes : 0u);
#endif
#if defined(H263MI_TIMING_NO_RGBA)
                if (px[0] == 0x12345678u)                                  // TIMING EXPERIMENT ONLY: (almost) no RGBA store
#endif
                if (col_full) {
                    // one 16-byte store; the address is a multiple of 4 (of 16 when the width is a multiple of 4)
                    store16_align4<STREAM_RGBA>(rgba + off, px[0], px[1], px[2], px[3]);
                } else {
                    for (int k = 0; k < 4 && gx + k < w; k++) memcpy(rgba + (off + 4u * (uint32_t)k), &px[k], 4);
                }
            }
        }
    }
    if (!INTERIOR && a.planes_out) {
        // tightly packed Y | Cb | Cr, as deblock() returns them (deblock.rs:305-315)
        uint8_t *out = a.planes_out + (size_t)pic * ((size_t)w * h + 2 * (size_t)cw * ch);
        for (int it = 0; it < 4; it++) {
            const int item = it * 64 + lane, row = item >> 5, g = item & 31;
            const int gy = yl + row, gx = post_wrap_x(a, xl + 4 * g);
            if (gy < 0 || gy >= h || gx < 0) continue;
            for (int k = 0; k < 4 && gx + k < w; k++) out[(size_t)gy * w + gx + k] = s.y[row * POST_TW + 4 * g + k];
        }
        if (!a.luma_only) {
            for (int it = 0; it < 2; it++) {
                const int item = it * 64 + lane, plane = item >> 6, row = (item >> 4) & 3, g = item & 15;
                const int gy = yl / 2 + row, gx = xl / 2 + 4 * g;
                if (gy < 0 || gy >= ch) continue;
                uint8_t *o = out + (size_t)w * h + (size_t)plane * cw * ch;
                // the chroma strip origin is 2 mod 4: a group of 4 may straddle column 0
                for (int k = 0; k < 4; k++) {
                    const int cx = post_wrap_cx(a, gx + k);            // (a group of 4 may straddle the wrap)
                    if (cx >= 0 && cx < cw) o[(size_t)gy * cw + cx] = s.c[plane][row * POST_CW + 4 * g + k];
                }
            }
        }
    }
}

}  // namespace h263mi
