// sim.cpp -- CPU logic checker for the kernel phase functions (TEST INFRASTRUCTURE).
//
// Compiles h263-rs_amd/csrc/{recon,post}_kernel.inl with g++ (-fsanitize=address,undefined,
// -ffp-contract=off) and runs the 256 threads of each workgroup in a loop, one phase at a
// time, in the order kernels.hip separates them with __syncthreads().  It catches index,
// bounds and edge-case mistakes before a GPU run; it is never linked into the product and
// measures nothing.
#include <stdlib.h>

#include "../../h263-rs_amd/csrc/post_kernel.inl"
#include "../../h263-rs_amd/csrc/recon_kernel.inl"
#include "../../h263-rs_amd/csrc/synth.inl"

using namespace h263mi;

template <bool INTERIOR>
static void sim_post_tile(const PostArgs &a, PostStrip &s, PostFetch (*pf)[64], int sx, int sy0, int pic, uint32_t strength)
{
    for (int l = 0; l < 64; l++) post_phase_fetch<INTERIOR>(a, pf[0][l], l, sx, sy0, pic);
    for (int l = 0; l < 64; l++) post_phase_fetch<INTERIOR>(a, pf[1][l], l, sx, sy0 + 1, pic);
    for (int k = 0; k < POST_STRIPS; k++) {
        const int sy = sy0 + k;      // strips past the bottom run too (nothing to store), as on the device
        memset(&s, 0xA5, sizeof s);
        for (int l = 0; l < 64; l++) post_phase_commit<INTERIOR>(a, s, pf[k & 1][l], l);
        if (k + 2 < POST_STRIPS)
            for (int l = 0; l < 64; l++) post_phase_fetch<INTERIOR>(a, pf[k & 1][l], l, sx, sy + 2, pic);
        if (strength) {
            for (int l = 0; l < 64; l++) post_phase_hedges<INTERIOR>(a, s, l, sx, sy);
            for (int l = 0; l < 64; l++) post_phase_vedges<INTERIOR>(a, s, l, sx, sy);
        }
        for (int l = 0; l < 64; l++) post_phase_store<false, INTERIOR>(a, s, l, sx, sy, pic);
    }
}

extern "C" {

// the dequantiser of the reconstruction waves on a whole table of LEVELs at one quantiser (pairs of int16 in, pairs of
// int16 out: SIXTEEN TIMES the clamped value, recon_kernel.inl: dequant_pair_i16), for tests/test_sim_kernels.py
void sim_dequant_pairs(const uint32_t *levels, uint32_t n_pairs, uint32_t quant, uint32_t *out)
{
    const uint32_t two_q2 = (2u * quant) * 0x00010001u, qmp2 = ((quant - 1u) | 1u) * 0x00010001u;
    for (uint32_t i = 0; i < n_pairs; i++) out[i] = dequant_pair_i16(levels[i], two_q2, qmp2);
}

// the wrapping dequantiser of wide rounds (recon_kernel.inl: dequant_pair_wrap): the value itself, any int16 LEVEL
void sim_dequant_pairs_wrap(const uint32_t *levels, uint32_t n_pairs, uint32_t quant, uint32_t *out)
{
    const uint32_t two_q2 = (2u * quant) * 0x00010001u, qmp2 = ((quant - 1u) | 1u) * 0x00010001u;
    for (uint32_t i = 0; i < n_pairs; i++) out[i] = dequant_pair_wrap(levels[i], two_q2, qmp2);
}

// the detector of LEVELs outside [-512, 511] (recon_kernel.inl: wide_bits_of) on n coefficient rows of 8 int16
void sim_wide_bits(const uint32_t *rows, uint32_t n_rows, uint32_t *out)
{
    for (uint32_t i = 0; i < n_rows; i++) {
        uint32_t t = 0;
        for (int j = 0; j < 4; j++) t |= wide_bits_of(rows[4 * i + j]);
        out[i] = t & WIDE_MASK_PAIR;
    }
}

// ... and on n event words (LEVEL << 16 | position)
void sim_wide_bits_events(const uint32_t *events, uint32_t n, uint32_t *out)
{
    for (uint32_t i = 0; i < n; i++) out[i] = wide_bits_of(events[i]) & WIDE_MASK_EVENT;
}

// 1: every IDCT round of sim_recon takes the wide form (which rounds do is a matter of speed only: the results must not
// change -- tests/test_sim_kernels.py)
static int force_wide_rounds = 0;
void sim_force_wide_rounds(int on) { force_wide_rounds = on; }

void sim_layout(uint32_t w, uint32_t h, FrameLayout *out) { *out = make_layout(w, h); }

// block_first_event / events: sparse coefficient transport consumed by the reconstruction wave itself (nullptr: dense)
// sparse records (ReconArgs::mb_group_index): set before a sim_recon_ex call, used by it, cleared behind it
static const uint32_t *g_group_index = nullptr;
static const uint64_t *g_mb_base = nullptr;
void sim_set_sparse_records(const uint32_t *group_index, const uint64_t *mb_base) { g_group_index = group_index; g_mb_base = mb_base; }
// the number of event words the next sim_recon_ex call tells the waves (ReconArgs::n_events: the checked mode of ABI 7's
// device-array entries); 0xffffffff = not told (the default, restored behind the call)
static uint32_t g_n_events = 0xffffffffu;
void sim_set_n_events(uint32_t n) { g_n_events = n; }

int sim_recon_ex(uint32_t w, uint32_t h, uint32_t n_pictures, const MbRecord *mbs, const int16_t *coeffs,
                 uint64_t n_blocks, const uint64_t *coeff_base, const uint8_t *ref, int has_ref, uint8_t *cur,
                 uint32_t *status, const uint32_t *block_first_event, const uint32_t *events)
{
    ReconArgs a{};
    a.mb_group_index = g_group_index;
    a.mb_base = g_mb_base;
    g_group_index = nullptr;
    g_mb_base = nullptr;
    a.L = make_layout(w, h);
    a.mbs = mbs;
    a.coeffs = coeffs;
    a.block_first_event = block_first_event;
    a.events = events;
    a.n_events = g_n_events;                     // (0xffffffff unless a test says how many: then the bounds are checked too)
    g_n_events = 0xffffffffu;
    a.coeff_base = coeff_base;
    a.ref = ref ? ref : cur;
    a.cur = cur;
    a.status = status;
    a.coeff_pool_blocks = n_blocks;
    a.coeff_checked = 1;
    a.n_pictures = n_pictures;
    a.mbs_per_picture = a.L.mbw * a.L.mbh;
    a.has_ref = has_ref;
    a.tiles_x = (a.L.mbw + TILE_MBX - 1) / TILE_MBX;
    a.tiles_y = (a.L.mbh + TILE_MBY - 1) / TILE_MBY;
    a.groups_per_picture = a.tiles_x * a.L.mbh;
    ReconWave *s = (ReconWave *)aligned_alloc(16, (sizeof(ReconWave) + 15) / 16 * 16);
    // same work units as kernels.hip::k_recon: XCD-ordered tiles, one independent wave per macroblock row of a tile (the
    // order of units does not matter)
    const uint32_t tpp = a.tiles_x * a.tiles_y, total = tpp * n_pictures, chunk = (total + 7) / 8;
    static WaveFetch f[64];
    for (uint32_t wg = 0; wg < chunk * 8; wg++) {
        const uint32_t xcd = wg & 7, t = wg >> 3, g = xcd * chunk + t;
        if (t >= chunk || g >= total) continue;
        const int tile = g % tpp;
        for (int wave = 0; wave < TILE_WAVES; wave++) {
            WavePos p;
            p.pic = g / tpp;
            p.mbx0 = (tile % (int)a.tiles_x) * TILE_MBX;
            p.mby = (tile / (int)a.tiles_x) * TILE_MBY + wave;
            if (p.mby >= (int)a.L.mbh) continue;
            p.cbase = a.coeff_base ? a.coeff_base[p.pic] : 0ull;
            memset(s, 0xA5, sizeof *s);   // LDS is not zero-initialised on the device either
            const uint32_t group_word = recon_group_word(a, p);
            if (a.mb_group_index && (group_word & 0xffu) == 0 && a.has_ref && recon_valid_mask(a, p) == 0xffu) {   // as kernels.hip::recon_wave
                for (int l = 0; l < 64; l++) recon_phase_copy(a, l, p);
                continue;
            }
            for (int l = 0; l < 64; l++) recon_phase_load(a, *s, l, p, group_word);
            WaveMasks km;
            km.valid = recon_valid_mask(a, p);
            km.act = 0;
            km.inter = 0;
            static TaskInfo ti[64];
            bool bad_index = false;
            for (int l = 0; l < 64; l++) {                                  // the device kernel does these reductions with ballots
                ti[l] = recon_phase_mark(a, *s, l, p, km.valid, recon_block_limit(a, p));
                if (ti[l].active) km.act |= 1ull << l;
                if (ti[l].inter) km.inter |= 1u << (l - MB_LANE0);
                bad_index = bad_index || ti[l].bad_index;
                // the strip origin the mark phase packs into the descriptor with literal shifts == the named-constant form
                if (l < WAVE_TASKS && desc_pix_origin(ti[l].d1) != task_pix_origin(l)) { free(s); return -2; }
            }
            for (int l = 0; l < 64; l++) recon_report(a, l, p.pic, km.inter && !a.has_ref, bad_index);
            for (int l = 0; l < 64; l++) recon_phase_compact(*s, l, ti[l], km.act);
            bool any_moving = false;
            for (int l = 0; l < 64; l++) any_moving = any_moving || ti[l].moving;
            if (recon_wave_is_static(a, km, any_moving)) {  // as kernels.hip::recon_wave
                for (int l = 0; l < 64; l++) recon_phase_copy(a, l, p);
                continue;
            }
            const bool mc = a.has_ref && km.inter;          // the dispatch of kernels.hip: recon_tail<MC>
            for (int l = 0; l < 64; l++) {
                if (mc) recon_phase_fetch<true>(a, *s, f[l], l, p, km);
                else recon_phase_fetch<false>(a, *s, f[l], l, p, km);
            }
            const int n_active = recon_n_active(km);
            // as kernels.hip::recon_tail: row pass of round 0, prediction into the strip, column pass of round 0, the
            // remaining rounds, store
            for (int round = 0; round == 0 || round * ROUND_BLOCKS < n_active; round++) {
                static RowIn ri[64];
                uint32_t wm = 0, rm = 0;
                uint64_t rows_any = 0, cols_any = 0;
                bool any_special = false;
                if (n_active > 0) {
                    // (sparse transport: the three steps of coeff_rows_from_events each over all lanes)
                    for (int stage = a.events ? 0 : 2; stage < 3; stage++)
                        for (int l = 0; l < 64; l++) recon_phase_idct_load(a, *s, f[l], l, p, round, ri[l], km, a.events ? stage : -1);
                    for (int l = 0; l < 64; l++) wm |= rowin_word_mask(ri[l]);
                    for (int l = 0; l < 64; l++) {
                        const RowClass rc = recon_row_class(ri[l], l);
                        if (rc.any) { rows_any |= 1ull << l; rm |= 1u << (l & 7); }
                        if (rc.beyond_first) cols_any |= 1ull << l;
                    }
                    bool wide = false;                                      // as kernels.hip: recon_round_rows
                    for (int l = 0; l < 64; l++) wide = wide || ri[l].wide != 0;
                    if (force_wide_rounds) wide = true;
                    for (int l = 0; l < 64; l++) {
                        if (wide) recon_phase_idct_rows<false, true>(*s, ri[l], l, cols_from_mask(wm), cols_any);
                        else recon_phase_idct_rows(*s, ri[l], l, cols_from_mask(wm), cols_any);
                    }
                    for (int l = 0; l < 64; l++) any_special = any_special || recon_block_is_special(ri[l], l, rows_any, cols_any);
                }
                if (round == 0) {
                    for (int l = 0; l < 64; l++) {
                        if (mc) recon_phase_predict<true>(a, *s, f[l], l, p, km);
                        else recon_phase_predict<false>(a, *s, f[l], l, p, km);
                    }
                }
                if (n_active > 0)
                    for (int l = 0; l < 64; l++) recon_phase_idct_cols(*s, ri[l], l, rows_from_mask(rm), rows_any, cols_any, any_special, /*strip_is_zero=*/!mc);
            }
            for (int l = 0; l < 64; l++) recon_phase_store(a, *s, l, p, km);
        }
    }
    free(s);
    return 0;
}

int sim_recon(uint32_t w, uint32_t h, uint32_t n_pictures, const MbRecord *mbs, const int16_t *coeffs,
              uint64_t n_blocks, const uint64_t *coeff_base, const uint8_t *ref, int has_ref, uint8_t *cur,
              uint32_t *status)
{
    return sim_recon_ex(w, h, n_pictures, mbs, coeffs, n_blocks, coeff_base, ref, has_ref, cur, status, nullptr, nullptr);
}

int sim_post(uint32_t w, uint32_t h, uint32_t n_pictures, const uint8_t *frames, uint32_t strength, uint8_t *rgba,
             uint8_t *planes_out, int luma_only)
{
    PostArgs a{};
    a.L = make_layout(w, h);
    a.frames = frames;
    a.rgba = rgba;
    a.planes_out = planes_out;
    a.n_pictures = n_pictures;
    a.strength = strength;
    a.tiles_x = post_tile_columns(a.L.width, &a.wrap);      // as host_common.h: set_post_tiles
    a.tiles_y = (post_strips_y(h) + POST_STRIPS - 1) / POST_STRIPS;
    a.luma_only = luma_only;
    PostStrip *s = (PostStrip *)aligned_alloc(16, (sizeof(PostStrip) + 15) / 16 * 16);
    // same work decomposition as kernels.hip::k_post: XCD-ordered workgroups of 4 waves, one tile
    // (4 strips) per wave, all loads of the tile first
    const uint32_t groups_y = (a.tiles_y + POST_GROUP - 1) / POST_GROUP;
    const uint32_t wpp = a.tiles_x * groups_y, wgs = wpp * n_pictures, chunk = (wgs + 7) / 8;
    static PostFetch pf[POST_STRIPS][64];
    for (uint32_t b = 0; b < chunk * 8; b++) {
        const uint32_t xcd = b & 7, t = b >> 3, wg = xcd * chunk + t;
        if (t >= chunk || wg >= wgs) continue;
        for (int wave = 0; wave < POST_GROUP; wave++) {
            const int pic = wg / wpp, rem = wg % wpp;
            const int sx = rem % (int)a.tiles_x + (int)a.wrap, ty = (rem / (int)a.tiles_x) * POST_GROUP + wave;
            if (ty >= (int)a.tiles_y) continue;
            const int sy0 = ty * POST_STRIPS;
            // interior tiles take the instantiations without bounds handling, as on the device (kernels.hip: post_wave)
            if (post_tile_is_interior(a, sx, ty)) sim_post_tile<true>(a, *s, pf, sx, sy0, pic, strength);
            else sim_post_tile<false>(a, *s, pf, sx, sy0, pic, strength);
        }
    }
    free(s);
    return 0;
}

// host form of the synthetic record generator (same inline code as the device kernels)
int sim_synth_picture(int kind, uint32_t w, uint32_t h, uint32_t stream_id, uint32_t frame_idx, MbRecord *mbs,
                      int16_t *coeffs, uint64_t cap_blocks, uint64_t *n_blocks)
{
    FrameLayout L = make_layout(w, h);
    uint64_t used = 0;
    for (uint32_t i = 0; i < L.mbw * L.mbh; i++) {
        MbRecord r = synth_mb_header(kind, stream_id, frame_idx, i);
        r.coeff_index = (uint32_t)used;
        for (int blk = 0; blk < 6; blk++) {
            if (!((r.cbp >> blk) & 1)) continue;
            if (used >= cap_blocks) return -1;
            synth_block_coeffs(kind, stream_id, frame_idx, i, blk, coeffs + used * 64);
            used++;
        }
        mbs[i] = r;
    }
    *n_blocks = used;
    return 0;
}

// Every (A - D, C - B) pair, each at the low and the high end of the byte range, strengths 1..12, against the
// reference-shaped quartet `ref` (the oracle's process_scalar / process_simd_lane, passed in by the test).
// Returns the number of mismatches; first[] = A,B,C,D,strength of the first one.
int sim_quartet_sweep(void (*ref)(uint8_t *, uint8_t *, uint8_t *, uint8_t *, uint8_t), int floor_sem, int *first)
{
    int bad = 0;
    for (int strength = 1; strength <= 12; strength++)
        for (int x = -255; x <= 255; x++)
            for (int px = 0; px < 2; px++)
                for (int y = -255; y <= 255; y++)
                    for (int py = 0; py < 2; py++) {
                        const int a0 = (x > 0 ? x : 0) + (px ? 255 - (x > 0 ? x : -x) : 0), d0 = a0 - x;
                        const int c0 = (y > 0 ? y : 0) + (py ? 255 - (y > 0 ? y : -y) : 0), b0 = c0 - y;
                        int A = a0, B = b0, C = c0, D = d0;
                        deblock_quartet(A, B, C, D, strength, floor_sem != 0);
                        uint8_t ra = (uint8_t)a0, rb = (uint8_t)b0, rc = (uint8_t)c0, rd = (uint8_t)d0;
                        ref(&ra, &rb, &rc, &rd, (uint8_t)strength);
                        if ((A & 0xff) != ra || B != rb || C != rc || (D & 0xff) != rd) {
                            if (!bad) { first[0] = a0; first[1] = b0; first[2] = c0; first[3] = d0; first[4] = strength; }
                            bad++;
                        }
                        // the packed form (two quartets per dword): this quartet in the low halves, its mirror image
                        // D,C,B,A in the high halves -- the mirror must come out as the mirrored result
                        // (deblock.rs:352-439 checks the same symmetry)
                        uint32_t pA = (uint32_t)a0 | ((uint32_t)d0 << 16), pB = (uint32_t)b0 | ((uint32_t)c0 << 16);
                        uint32_t pC = (uint32_t)c0 | ((uint32_t)b0 << 16), pD = (uint32_t)d0 | ((uint32_t)a0 << 16);
                        deblock_quartet_pk(pA, pB, pC, pD, quartet_consts(strength, floor_sem ? 0 : -1));
                        const uint32_t sb = sat_pk_u8_i16(pB), sc = sat_pk_u8_i16(pC);
                        uint8_t ma = (uint8_t)d0, mb = (uint8_t)c0, mc = (uint8_t)b0, md = (uint8_t)a0;
                        ref(&ma, &mb, &mc, &md, (uint8_t)strength);
                        if ((pA & 0xff) != ra || (sb & 0xff) != rb || (sc & 0xff) != rc || (pD & 0xff) != rd ||
                            ((pA >> 16) & 0xff) != ma || ((sb >> 8) & 0xff) != mb || ((sc >> 8) & 0xff) != mc || ((pD >> 16) & 0xff) != md) {
                            if (!bad) { first[0] = a0; first[1] = b0; first[2] = c0; first[3] = d0; first[4] = -strength; }
                            bad++;
                        }
                        if (floor_sem) {
                            // the instantiation of interior tiles (no truncation biases at all) must be the same function
                            uint32_t fA = (uint32_t)a0 | ((uint32_t)d0 << 16), fB = (uint32_t)b0 | ((uint32_t)c0 << 16);
                            uint32_t fC = (uint32_t)c0 | ((uint32_t)b0 << 16), fD = (uint32_t)d0 | ((uint32_t)a0 << 16);
                            deblock_quartet_pk<true>(fA, fB, fC, fD, quartet_consts(strength, 0));
                            if (fA != pA || fB != pB || fC != pC || fD != pD) {
                                if (!bad) { first[0] = a0; first[1] = b0; first[2] = c0; first[3] = d0; first[4] = -100 - strength; }
                                bad++;
                            }
                        }
                    }
    return bad;
}

}
