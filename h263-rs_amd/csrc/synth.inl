// synth.inl -- synthetic macroblock records for the BASELINE.json configurations
// (bench / test support; there is no counterpart in the reference, which ships no
// sample streams).  Counter-based splitmix64 so that host C++, the HIP
// generator kernels and any other implementation produce identical bytes:
//
//   seed = 0x4832363300000000 + (stream_id << 16) + frame_idx
//   rnd(seed, ctr) = mix(seed + 0x9E3779B97F4A7C15 * (ctr + 1)),  ctr = mb_index*1024 + field
//
// Field map (per macroblock): 0 quant | 1 4V draw | 2..9 mv components | 10..15 intradc |
// 16..21 cbp draws | 22..27 class draws | 64 + blk*128 + k: per-block coefficient draws.
#pragma once

#include "dev_common.h"

namespace h263mi {

H263_HD uint64_t splitmix64_at(uint64_t seed, uint64_t ctr)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (ctr + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

H263_HD uint64_t synth_seed(uint32_t stream_id, uint32_t frame_idx)
{
    return 0x4832363300000000ull + ((uint64_t)stream_id << 16) + frame_idx;
}

struct SynthRng {
    uint64_t seed, base;
    H263_HD uint32_t operator()(uint32_t field, uint32_t mod) const
    {
        return (uint32_t)(splitmix64_at(seed, base + field) % mod);
    }
};

// rle.rs:6-71 as raster index per zigzag position
H263_HD int synth_zigzag_raster(int z)
{
    constexpr uint8_t ZZ[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    return ZZ[z];
}

// symmetric non-zero level in [-amp, amp] \ {0}
H263_HD int synth_level(uint32_t draw, int amp)
{
    int l = (int)(draw % (uint32_t)(2 * amp)) - amp;
    return l >= 0 ? l + 1 : l;
}

enum { SYNTH_CLASS_DC = 0, SYNTH_CLASS_HORIZ, SYNTH_CLASS_VERT, SYNTH_CLASS_FULL_DENSE, SYNTH_CLASS_FULL_SPARSE };

// Everything of the record except coeff_index.
H263_HD MbRecord synth_mb_header(int kind, uint32_t stream_id, uint32_t frame_idx, uint32_t mb_index)
{
    SynthRng rnd{synth_seed(stream_id, frame_idx), (uint64_t)mb_index * 1024};
    MbRecord r;
    memset(&r, 0, sizeof r);
    if (kind == H263MI_SYNTH_P) {
        const bool four = rnd(1, 10) == 0;
        r.mb_type = four ? H263MI_MB_INTER4V : H263MI_MB_INTER;
        r.quant = 10;
        for (int b = 0; b < 4; b++) {
            const int src = four ? b : 0;      // one-vector macroblocks replicate mv[0] (state.rs:280-284)
            r.mv[b][0] = (int16_t)((int)rnd(2 + 2 * src, 64) - 32);
            r.mv[b][1] = (int16_t)((int)rnd(3 + 2 * src, 64) - 32);
        }
        uint32_t cbp = 0;
        for (int b = 0; b < 6; b++)
            if (rnd(16 + b, 4) == 0) cbp |= 1u << b;
        r.cbp = (uint8_t)cbp;
    } else {
        r.mb_type = H263MI_MB_INTRA;
        r.quant = (uint8_t)(1 + rnd(0, 31));
        uint32_t cbp = 0;
        for (int b = 0; b < 6; b++) {
            uint32_t c = 1 + rnd(10 + b, 254);     // 1..254, then skip the illegal 128 (types.rs:930-936)
            if (c >= 128) c += 1;
            r.intradc[b] = (uint8_t)c;
            const bool coded = kind == H263MI_SYNTH_I_DENSE || rnd(22 + b, 5) != SYNTH_CLASS_DC;
            if (coded) cbp |= 1u << b;
        }
        r.cbp = (uint8_t)cbp;
    }
    return r;
}

// The 64 raster coefficients of coded block `blk`.
H263_HD void synth_block_coeffs(int kind, uint32_t stream_id, uint32_t frame_idx, uint32_t mb_index, int blk,
                                int16_t *out /* 64 */)
{
    SynthRng rnd{synth_seed(stream_id, frame_idx), (uint64_t)mb_index * 1024};
    const uint32_t f0 = 64 + (uint32_t)blk * 128;
    for (int i = 0; i < 64; i++) out[i] = 0;
    if (kind == H263MI_SYNTH_P) {
        // 4 coefficients at zigzag positions < 16, |level| <= 31
        for (int k = 0; k < 4; k++) {
            int z = (int)rnd(f0 + 64 + k, 16);
            out[synth_zigzag_raster(z)] = (int16_t)synth_level(rnd(f0 + k, 1u << 20), 31);
        }
    } else if (kind == H263MI_SYNTH_I_DENSE) {
        for (int i = 1; i < 64; i++) out[i] = (int16_t)synth_level(rnd(f0 + i, 1u << 20), 24);
    } else {
        const uint32_t cls = rnd(22 + blk, 5);
        if (cls == SYNTH_CLASS_HORIZ) {
            for (int i = 1; i < 8; i++) out[i] = (int16_t)synth_level(rnd(f0 + i, 1u << 20), 127);
        } else if (cls == SYNTH_CLASS_VERT) {
            for (int i = 1; i < 8; i++) out[8 * i] = (int16_t)synth_level(rnd(f0 + i, 1u << 20), 127);
        } else if (cls == SYNTH_CLASS_FULL_DENSE) {
            for (int i = 1; i < 64; i++) out[i] = (int16_t)synth_level(rnd(f0 + i, 1u << 20), 127);
        } else if (cls == SYNTH_CLASS_FULL_SPARSE) {
            for (int k = 0; k < 4; k++)
                out[1 + rnd(f0 + 64 + k, 63)] = (int16_t)synth_level(rnd(f0 + k, 1u << 20), 127);
        }
    }
}

}  // namespace h263mi
