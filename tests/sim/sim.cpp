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

extern "C" {

void sim_layout(uint32_t w, uint32_t h, FrameLayout *out) { *out = make_layout(w, h); }

int sim_recon(uint32_t w, uint32_t h, uint32_t n_pictures, const MbRecord *mbs, const int16_t *coeffs,
              uint64_t n_blocks, const uint64_t *coeff_base, const uint8_t *ref, int has_ref, uint8_t *cur,
              uint32_t *status)
{
    ReconArgs a{};
    a.L = make_layout(w, h);
    a.mbs = mbs;
    a.coeffs = coeffs;
    a.coeff_base = coeff_base;
    a.ref = ref ? ref : cur;
    a.cur = cur;
    a.status = status;
    a.coeff_pool_blocks = n_blocks;
    a.n_pictures = n_pictures;
    a.mbs_per_picture = a.L.mbw * a.L.mbh;
    a.has_ref = has_ref;
    a.tiles_x = (a.L.mbw + TILE_MBX - 1) / TILE_MBX;
    a.tiles_y = (a.L.mbh + TILE_MBY - 1) / TILE_MBY;
    ReconSmem *s = (ReconSmem *)aligned_alloc(16, (sizeof(ReconSmem) + 15) / 16 * 16);
    for (uint32_t pic = 0; pic < n_pictures; pic++)
        for (uint32_t tile = 0; tile < a.tiles_x * a.tiles_y; tile++) {
            memset(s, 0xA5, sizeof *s);   // LDS is not zero-initialised on the device either
            for (int t = 0; t < RECON_THREADS; t++) recon_phase_load(a, *s, t, tile, pic);
            for (int t = 0; t < RECON_THREADS; t++) recon_phase_mark(a, *s, t);
            for (int t = 0; t < RECON_THREADS; t++) recon_phase_compact(a, *s, t);
            const int n_active = recon_n_active(*s);
            for (int round = 0; round * ROUND_BLOCKS < n_active; round++) {
                for (int t = 0; t < RECON_THREADS; t++) recon_phase_idct_rows(a, *s, t, pic, round);
                for (int t = 0; t < RECON_THREADS; t++) recon_phase_idct_cols(a, *s, t, round);
            }
            for (int t = 0; t < RECON_THREADS; t++) recon_phase_output(a, *s, t, tile, pic);
        }
    free(s);
    return 0;
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
    a.tiles_x = (a.L.width + POST_OX + POST_TW - 1) / POST_TW;
    a.tiles_y = (a.L.height + POST_OY + POST_TH - 1) / POST_TH;
    a.luma_only = luma_only;
    PostSmem *s = (PostSmem *)aligned_alloc(16, (sizeof(PostSmem) + 15) / 16 * 16);
    for (uint32_t pic = 0; pic < n_pictures; pic++)
        for (uint32_t tile = 0; tile < a.tiles_x * a.tiles_y; tile++) {
            memset(s, 0xA5, sizeof *s);
            for (int t = 0; t < POST_THREADS; t++) post_phase_load(a, *s, t, tile, pic);
            if (strength) {
                for (int t = 0; t < POST_THREADS; t++) post_phase_hedges(a, *s, t, tile);
                for (int t = 0; t < POST_THREADS; t++) post_phase_vedges(a, *s, t, tile);
            }
            for (int t = 0; t < POST_THREADS; t++) post_phase_store(a, *s, t, tile, pic);
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
}
