// recon_kernel.inl -- k_recon: dequantise + classify + 8x8 IDCT + half-pel motion
// compensation + residual add + clip.
//
// Replaces, per picture: the numeric part of inverse_rle (h263/src/decoder/cpu/rle.rs:
// 112-171), gather (gather.rs:140-204) and the three idct_channel calls (idct.rs:82-201)
// issued at the tail of decode_next_picture (state.rs:432-458).
//
// Work unit = one WAVE = half a row of 8 macroblocks: wave "top" owns blocks Y0, Y1 and Cb of
// the 8 macroblocks (a 128x8 luma strip + a 64x8 Cb strip), wave "bottom" owns Y2, Y3 and Cr.
// That is 24 blocks and 3 x 64 eight-pixel row segments per wave; every luma row it stores is
// one full 128-byte line.  A wave needs nothing from any other wave, so there is NO workgroup
// barrier: hand-offs between lanes go through 6 KB of wave-private LDS and rely only on the
// DS operations of one wave executing in order.  A workgroup is 4 such waves (8x2
// macroblocks) purely so that the XCD-aware work order of kernels.hip keeps neighbours on
// one L2.
//
// Wave timeline:
//   records  : 8 records (256 B) -> LDS; list of blocks that need an IDCT (coded & !kill, or
//              uncoded intra with non-zero DC) compacted in LDS; chroma vectors once per MB
//   fetch    : ALL global loads are issued now, before any arithmetic: the reference rows of
//              the lane's three segments (one unaligned 12-byte load per row, taps clamped to
//              the picture) and the coefficient row of the first IDCT round
//   idct     : rounds of 8 blocks, 8 lanes per block, lane = one coefficient row: dequant,
//              row pass T = C x B -> LDS; then lane = one pixel column: column of T (the
//              LDS transposition), column pass, rounding -> residual strip (i16) in LDS
//   output   : lane = 8 horizontal pixels: half-pel interpolation on packed bytes, + residual
//              row (packed i16 add, saturate to u8), one 8-byte store
//
// Bit-exactness rules: f32 multiply and add are separately rounded
// (translation unit built with -ffp-contract=off), accumulation order over the
// frequency index is sequential, and the Dc / Vert classes keep their own arithmetic.
#pragma once

#include "dev_common.h"

namespace h263mi {

constexpr int RECON_THREADS = 256;
constexpr int RECON_WAVES = RECON_THREADS / 64;
constexpr int TILE_MBX = 8, TILE_MBY = 2;      // macroblocks per workgroup
constexpr int WAVE_TASKS = 24;                 // 16 luma + 8 chroma blocks per wave
constexpr int ROUND_BLOCKS = 8;                // blocks per IDCT round (8 lanes each)
constexpr int TBUF_ROW = 9;                    // floats per row of a block slot: T[r][0..7] and C[r][0]
constexpr int TBUF_STRIDE = 8 * TBUF_ROW;      // 72 floats per slot: column reads are bank-conflict free
constexpr int RES_STRIDE = 192;                // residual strip row: 128 luma + 64 chroma columns

struct ReconWave {
    MbRecord rec[TILE_MBX];                    // 256 B
    int16_t  mvc[TILE_MBX][2];                 // chroma vector per macroblock (gather.rs:182)
    uint32_t valid_mask;                       // bit m: macroblock m lies inside the picture
    uint32_t act_mask;                         // bit t: block task t goes through the IDCT
    uint8_t  list[WAVE_TASKS];                 // compacted active tasks
    float    tbuf[ROUND_BLOCKS * TBUF_STRIDE]; // row pass results; column 8 of each row keeps C[r][0] for the Vert class
    uint8_t  flags[ROUND_BLOCKS * 8];
    int16_t  res[8 * RES_STRIDE];              // residual strip: 8 rows x (128 luma | 64 chroma) columns
};

// position of a wave's work: which picture, which row of macroblocks, which half
struct WavePos {
    int pic, mbx0, mby, half;
    uint64_t cbase;            // coeff_base[pic] (0 without a base array), fetched once per wave
};

H263_HD int popc32(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(v);
#else
    return __builtin_popcount(v);
#endif
}

// block task t of a wave -> (macroblock 0..7, block 0..5)
H263_HD int task_mb(int t) { return t < 16 ? (t >> 1) : (t - 16); }
H263_HD int task_blk(int t, int half) { return t < 16 ? (half * 2 + (t & 1)) : (4 + half); }

// ---- packed helpers (device: single instructions; host build: plain C for tests/sim) ----------
// per-byte (a + b + 1) >> 1
H263_DEV uint32_t avg2_u8x4(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_lerp(a, b, 0x01010101u);           // v_lerp_u8
#else
    return (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu);
#endif
}

// per-byte (a + b + c + d + 2) >> 2
H263_DEV uint32_t avg4_u8x4(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // even / odd bytes spread into 16-bit lanes by v_perm_b32, summed with v_add3_u32
    const uint32_t EV = 0x0c020c00u, OD = 0x0c030c01u;
    uint32_t e = __builtin_amdgcn_perm(0u, a, EV) + __builtin_amdgcn_perm(0u, b, EV) + __builtin_amdgcn_perm(0u, c, EV);
    e = e + __builtin_amdgcn_perm(0u, d, EV) + 0x00020002u;
    uint32_t o = __builtin_amdgcn_perm(0u, a, OD) + __builtin_amdgcn_perm(0u, b, OD) + __builtin_amdgcn_perm(0u, c, OD);
    o = o + __builtin_amdgcn_perm(0u, d, OD) + 0x00020002u;
    // bytes: e>>2 lane0, o>>2 lane0, e>>2 lane1, o>>2 lane1 (the permute drops the bits above each byte)
    return __builtin_amdgcn_perm(o >> 2, e >> 2, 0x06020400u);
#else
    const uint32_t M = 0x00ff00ffu;
    uint32_t e = (a & M) + (b & M) + (c & M) + (d & M) + 0x00020002u;
    uint32_t o = ((a >> 8) & M) + ((b >> 8) & M) + ((c >> 8) & M) + ((d >> 8) & M) + 0x00020002u;
    return ((e >> 2) & M) | (((o >> 2) & M) << 8);
#endif
}

// bytes (lo >> 8*sh) of the 64-bit pair {hi:lo}, sh in 0..3
H263_DEV uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, sh);              // v_alignbyte_b32
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * sh));
#endif
}

// four prediction bytes + four i16 residuals (two dwords) -> four clipped bytes
// (clipped_idct + mocomp_pixel).clamp(0, 255)  idct.rs:127-130, 191-194
H263_DEV uint32_t add_clip_u8x4(uint32_t pred, uint32_t r01, uint32_t r23)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short short2v __attribute__((ext_vector_type(2)));
    union U { uint32_t u; short2v v; };
    U p01, p23, a, b;
    p01.u = __builtin_amdgcn_perm(0u, pred, 0x0c010c00u);       // (p0, p1) as i16
    p23.u = __builtin_amdgcn_perm(0u, pred, 0x0c030c02u);       // (p2, p3)
    a.u = r01;
    b.u = r23;
    const short2v zero = {0, 0}, top = {255, 255};
    U s01, s23;
    s01.v = __builtin_elementwise_min(__builtin_elementwise_max(p01.v + a.v, zero), top);
    s23.v = __builtin_elementwise_min(__builtin_elementwise_max(p23.v + b.v, zero), top);
    return __builtin_amdgcn_perm(s23.u, s01.u, 0x06040200u);    // low bytes of the four lanes
#else
    uint32_t out = 0;
    const uint32_t r[2] = {r01, r23};
    for (int k = 0; k < 4; k++) {
        int rr = (int)(int16_t)(r[k >> 1] >> ((k & 1) * 16));
        int p = (int)((pred >> (8 * k)) & 0xff);
        out |= (uint32_t)clampi(p + rr, 0, 255) << (8 * k);
    }
    return out;
#endif
}

// ---- f32 pair arithmetic for the IDCT passes ----------------------------------------------------
// Two outputs of idct_1d are accumulated side by side (v_pk_mul_f32 / v_pk_add_f32 on the device);
// every product and every sum is still rounded on its own, in the reference's order.  The basis
// table lives in constant memory as 32 (x, x+1) pairs: uniform addresses, so it reaches the VALU
// through scalar registers and both passes share it.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define H263_CONST_TABLE static __device__ __constant__
#else
typedef float f32x2 __attribute__((vector_size(8)));
#define H263_CONST_TABLE static const
#endif
H263_CONST_TABLE float kBasis[8][8] = {H263MI_BASIS_ROWS};    // idct.rs:39-48

H263_DEV f32x2 splat2(float v) { f32x2 r = {v, v}; return r; }
H263_DEV f32x2 basis_pair(int f, int ip) { f32x2 r = {kBasis[f][2 * ip], kBasis[f][2 * ip + 1]}; return r; }

// idct_1d (idct.rs:52-65): out[i] = sum over f, in order, of in[f] * B[f][i].  The leading
// "0.0 +" is dropped: it can only change the sign of a zero, which never reaches the integer result.
H263_DEV void idct_1d_pairs(const float in[8], f32x2 out[4])
{
#pragma unroll
    for (int ip = 0; ip < 4; ip++) {
        f32x2 acc = splat2(in[0]) * basis_pair(0, ip);
#pragma unroll
        for (int f = 1; f < 8; f++) {
            f32x2 pr = splat2(in[f]) * basis_pair(f, ip);
            acc = acc + pr;
        }
        out[ip] = acc;
    }
}

// median of three = clamp(v, lo, hi) for lo <= hi (one v_med3_f32; no NaNs can occur here)
H263_DEV float clampf(float v, float lo, float hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(v, lo, hi);
#else
    return v < lo ? lo : (v > hi ? hi : v);
#endif
}

// rle.rs:130-133 on a float copy of the LEVEL (all values are small integers, exact in f32):
// sign(L) * (q*(2|L|+1) - (q even)) = L*2q + sign(L)*(q - parity), clamped to [-2048, 2047]; 0 stays 0
H263_DEV float dequant_f32(float level, float two_q, float q_minus_parity)
{
    const float sg = clampf(level, -1.0f, 1.0f);                            // -1, 0 or +1
    const float v = level * two_q + sg * q_minus_parity;                    // exact: integers below 2^24
    return clampf(v, -2048.0f, 2047.0f);
}

// ---- phase 0: records -> LDS -------------------------------------------------------
H263_DEV void recon_phase_load(const ReconArgs &a, ReconWave &s, int lane, const WavePos &p)
{
    if (lane < TILE_MBX * 2) {
        // 16 lanes x 16 B = the 8 records of this macroblock row segment (one 256-B run)
        const int m = lane >> 1, part = lane & 1;
        const int mbx = p.mbx0 + m;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (mbx < (int)a.L.mbw && p.mby < (int)a.L.mbh) {
            const MbRecord *r = a.mbs + (size_t)p.pic * a.mbs_per_picture + (size_t)p.mby * a.L.mbw + mbx;
            v = reinterpret_cast<const uint4 *>(r)[part];
        }
        reinterpret_cast<uint4 *>(&s.rec[m])[part] = v;
    }
    if (lane == 16) {
        uint32_t vm = 0;
        for (int m = 0; m < TILE_MBX; m++)
            if (p.mbx0 + m < (int)a.L.mbw && p.mby < (int)a.L.mbh) vm |= 1u << m;
        s.valid_mask = vm;
        s.act_mask = 0;
    }
}

// ---- phase 1: which blocks need the IDCT; chroma vectors -------------------------------
H263_DEV void recon_phase_mark(const ReconArgs &, ReconWave &s, int lane, const WavePos &p)
{
    if (lane < WAVE_TASKS) {
        const int m = task_mb(lane), blk = task_blk(lane, p.half);
        const MbRecord &r = s.rec[m];
        const bool valid = (s.valid_mask >> m) & 1;
        const bool coded = (r.cbp >> blk) & 1;
        const bool kill = (r.kill >> blk) & 1;
        const bool intra = mb_is_intra(r.mb_type);
        // coded & kill -> Zero (rle.rs:125-127); uncoded inter -> Zero; uncoded intra -> Dc(level)
        const bool active = valid && ((coded && !kill) || (!coded && intra && intradc_level(r.intradc[blk]) != 0));
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t mask = (uint32_t)__ballot(active);        // lanes 0..23 are the only ones here
        if (lane == 0) s.act_mask = mask;
#else
        if (active) s.act_mask |= 1u << lane;
#endif
    } else if (lane < WAVE_TASKS + TILE_MBX) {
        // gather.rs:182: chroma vector from the i16 sum of the four luma vectors
        const int m = lane - WAVE_TASKS;
        const MbRecord &r = s.rec[m];
        s.mvc[m][0] = (int16_t)average_sum_of_mvs(r.mv[0][0] + r.mv[1][0] + r.mv[2][0] + r.mv[3][0]);
        s.mvc[m][1] = (int16_t)average_sum_of_mvs(r.mv[0][1] + r.mv[1][1] + r.mv[2][1] + r.mv[3][1]);
    }
}

// ---- phase 2: compact the active tasks ----------------------------------------------
H263_DEV void recon_phase_compact(const ReconArgs &, ReconWave &s, int lane)
{
    if (lane >= WAVE_TASKS) return;
    const uint32_t mw = s.act_mask;
    if (!((mw >> lane) & 1)) return;
    s.list[popc32(mw & ((1u << lane) - 1u))] = (uint8_t)lane;
}

H263_DEV int recon_n_active(const ReconWave &s) { return popc32(s.act_mask); }

// ---- phase 3: issue every global load of the wave ------------------------------------------
// One 8-pixel row segment of the lane: its motion vector and the raw reference bytes (12 per
// tap row: 9 are needed at most; a dwordx3 keeps it to one load per row).
enum : uint8_t { SEG_INTER = 1, SEG_BORDER = 2 };
struct SegFetch {
    uint32_t r0[3], r1[3];     // reference row v and row v+1, bytes u .. u+11
    int16_t  mvx, mvy;
    uint8_t  flags;            // SEG_INTER: motion compensated (else prediction = 0);
                               // SEG_BORDER: some tap falls outside the picture, redone with clamping
};
struct WaveFetch {
    SegFetch seg[3];           // [0], [1]: luma rows (lane>>4) and 4 + (lane>>4); [2]: chroma row lane>>3
    uint4    coef0;            // coefficient row of the first IDCT round
};

// segment k of a lane -> geometry
struct SegGeo {
    int m, blk, task, px, py, pitch, pw, ph, resoff;   // resoff: index into the residual strip
    uint32_t plane_off;
    bool luma;
};

H263_DEV SegGeo seg_geometry(const ReconArgs &a, int lane, int k, const WavePos &p)
{
    SegGeo g;
    if (k < 2) {
        const int row = (lane >> 4) + 4 * k, sx = lane & 15;      // 16 lanes = one 128-byte luma line
        g.luma = true;
        g.m = sx >> 1;
        g.blk = p.half * 2 + (sx & 1);
        g.task = sx;
        g.px = p.mbx0 * 16 + sx * 8;
        g.py = p.mby * 16 + p.half * 8 + row;
        g.pitch = (int)a.L.pitch_y; g.pw = (int)a.L.width; g.ph = (int)a.L.height;
        g.plane_off = 0;
        g.resoff = row * RES_STRIDE + sx * 8;
    } else {
        const int row = lane >> 3, sx = lane & 7;
        g.luma = false;
        g.m = sx;
        g.blk = 4 + p.half;
        g.task = 16 + sx;
        g.px = p.mbx0 * 8 + sx * 8;
        g.py = p.mby * 8 + row;
        g.pitch = (int)a.L.pitch_c; g.pw = (int)a.L.cwidth; g.ph = (int)a.L.cheight;
        g.plane_off = p.half ? a.L.off_cr : a.L.off_cb;
        g.resoff = row * RES_STRIDE + 128 + sx * 8;
    }
    return g;
}

// 12 bytes from an arbitrarily aligned address in ONE load (global_load_dwordx3; gfx950 runs with
// unaligned access enabled)
H263_DEV void load12(const uint8_t *p, uint32_t out[3])
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
    typedef u32x3 __attribute__((aligned(1))) u32x3_unaligned;
    const u32x3 v = *reinterpret_cast<const u32x3_unaligned *>(p);
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
#else
    memcpy(out, p, 12);
#endif
}

// border path: every tap clamped to the picture on its own (gather.rs:24-25); rolled loop, rare
H263_DEV void fetch_row_clamped(const uint8_t *row, int u, int pw, uint32_t out[3])
{
    uint64_t lo8 = 0;
#pragma unroll 1
    for (int kx = 0; kx < 8; kx++) lo8 |= (uint64_t)row[clampi(u + kx, 0, pw - 1)] << (8 * kx);
    out[0] = (uint32_t)lo8;
    out[1] = (uint32_t)(lo8 >> 32);
    out[2] = row[clampi(u + 8, 0, pw - 1)];
}

// The loads are issued unconditionally and in a fixed order -- coefficient row first, then two
// 12-byte reference rows per segment -- so that the wait in front of the row pass can leave the six
// reference loads in flight (s_waitcnt vmcnt(6)): the IDCT of this wave overlaps its own motion
// compensation reads.  Lanes with nothing to fetch read a dummy line (always cache resident);
// lanes whose taps leave the picture are fixed up later, in the output phase.
H263_DEV void recon_phase_fetch(const ReconArgs &a, ReconWave &s, WaveFetch &f, int lane, const WavePos &p)
{
    const uint8_t *ref = a.ref + (size_t)p.pic * a.L.frame_bytes;
    {
        const uint8_t *src = reinterpret_cast<const uint8_t *>(a.mbs);      // dummy: any mapped address
        const int slot = lane >> 3, r = lane & 7;
        if (slot < recon_n_active(s)) {
            const int t = s.list[slot];
            const int m = task_mb(t), blk = task_blk(t, p.half);
            const MbRecord &rec = s.rec[m];
            if ((rec.cbp >> blk) & 1) {
                const uint64_t cidx = p.cbase + rec.coeff_index +
                                      (uint64_t)popc32(rec.cbp & ((1u << blk) - 1u));
                if (!(a.coeff_pool_blocks && cidx >= a.coeff_pool_blocks) && !(a.debug_flags & 4))
                    src = reinterpret_cast<const uint8_t *>(a.coeffs + cidx * 64 + (size_t)r * 8);
            }
        }
        f.coef0 = *reinterpret_cast<const uint4 *>(src);       // uncoded slots ignore it
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        SegFetch &sf = f.seg[k];
        const SegGeo g = seg_geometry(a, lane, k, p);
        const MbRecord &rec = s.rec[g.m];
        const bool valid = (s.valid_mask >> g.m) & 1;
        const bool inter = valid && mb_is_inter(rec.mb_type);    // intra: prediction stays 0 (gather.rs:136-138)
        if (inter && !a.has_ref && (lane & 15) == 0) {
            // gather.rs:149 Error::UncodedIFrameBlocks -- reported through the status word
#if defined(__HIP_DEVICE_COMPILE__)
            atomicOr(a.status, STATUS_INTER_WITHOUT_REFERENCE);
#else
            *a.status |= STATUS_INTER_WITHOUT_REFERENCE;
#endif
        }
        const bool mc = inter && a.has_ref;
        sf.mvx = g.luma ? rec.mv[g.blk][0] : s.mvc[g.m][0];
        sf.mvy = g.luma ? rec.mv[g.blk][1] : s.mvc[g.m][1];
        // HalfPel::into_lerp_parameters (types.rs:721-729): floor(mv / 2), odd -> interpolate
        const int u = g.px + (sf.mvx >> 1), v = g.py + (sf.mvy >> 1);
        const bool inside = u >= 0 && u + 8 + (sf.mvx & 1) <= g.pw;
        sf.flags = (uint8_t)((mc ? SEG_INTER : 0) | ((mc && !inside) ? SEG_BORDER : 0));
        // inside lanes read exactly at u (the 12-byte load may run past the row end: next row or padding);
        // border lanes read some mapped address, their data is replaced in the output phase
        const uint32_t uc = (uint32_t)(inside ? u : clampi(u, 0, g.pitch - 12));
        const uint32_t o0 = g.plane_off + (uint32_t)clampi(v, 0, g.ph - 1) * (uint32_t)g.pitch + uc;
        const uint32_t o1 = g.plane_off + (uint32_t)clampi(v + 1, 0, g.ph - 1) * (uint32_t)g.pitch + uc;
        const bool real = mc && !(a.debug_flags & 1);
        load12(ref + (real ? o0 : 0u), sf.r0);
        load12(ref + ((real && (sf.mvy & 1)) ? o1 : 0u), sf.r1);
    }
}

// ---- phase 4a: row pass ---------------------------------------------------------------
H263_DEV void recon_phase_idct_rows(const ReconArgs &a, ReconWave &s, const WaveFetch &f, int lane, const WavePos &p,
                                    int round)
{
    const int slot = lane >> 3, r = lane & 7;
    const int k = round * ROUND_BLOCKS + slot;
    if (k >= recon_n_active(s)) return;
    const int t = s.list[k];
    const int m = task_mb(t), blk = task_blk(t, p.half);
    const MbRecord &rec = s.rec[m];
    const bool coded = (rec.cbp >> blk) & 1;
    const bool intra = mb_is_intra(rec.mb_type);
    const int quant = rec.quant;

    float C[8];
#pragma unroll
    for (int c = 0; c < 8; c++) C[c] = 0.0f;
    if (coded) {
        const uint64_t cidx = p.cbase + rec.coeff_index +
                              (uint64_t)popc32(rec.cbp & ((1u << blk) - 1u));
        if (a.coeff_pool_blocks && cidx >= a.coeff_pool_blocks) {
            if (r == 0) {
#if defined(__HIP_DEVICE_COMPILE__)
                atomicOr(a.status, STATUS_COEFF_INDEX_OUT_OF_RANGE);
#else
                *a.status |= STATUS_COEFF_INDEX_OUT_OF_RANGE;
#endif
            }
        } else {
            // 8 lanes x 16 B = one 128-B coefficient block (raster order: lane r holds row r);
            // round 0 was loaded ahead of time by recon_phase_fetch
            uint4 raw = f.coef0;
            if (round > 0 && !(a.debug_flags & 4))
                raw = *reinterpret_cast<const uint4 *>(a.coeffs + cidx * 64 + (size_t)r * 8);
            const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
            const float two_q = (float)(2 * quant), qmp = (float)(quant - ((quant & 1) ? 0 : 1));
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const int level = (int)(int16_t)(w[c >> 1] >> ((c & 1) * 16));
                C[c] = dequant_f32((float)level, two_q, qmp);
            }
        }
    }
    // intra: the DC comes from INTRADC and TCOEFs start at zigzag 1 (rle.rs:117-121)
    if (intra && r == 0) C[0] = (float)intradc_level(rec.intradc[blk]);

    // classification inputs (rle.rs:138-149): a non-zero value with y > 0 breaks "horiz",
    // one with x > 0 breaks "vert"
    bool cols_nz = false, row_nz = (C[0] != 0.0f);
#pragma unroll
    for (int c = 1; c < 8; c++) cols_nz = cols_nz || (C[c] != 0.0f);
    row_nz = (row_nz || cols_nz) && (r > 0);
    s.flags[slot * 8 + r] = (uint8_t)((row_nz ? 1 : 0) | (cols_nz ? 2 : 0));

    // idct_1d over the coefficient row (idct.rs:52-65)
    f32x2 T[4];
    idct_1d_pairs(C, T);
    float *dst = &s.tbuf[slot * TBUF_STRIDE + r * TBUF_ROW];
#pragma unroll
    for (int i = 0; i < 8; i++) dst[i] = T[i >> 1][i & 1];
    dst[8] = C[0];
}

// ---- phase 4b: column pass, rounding, residual strip -------------------------------------
H263_DEV void recon_phase_idct_cols(const ReconArgs &, ReconWave &s, int lane, int round)
{
    const int slot = lane >> 3, i = lane & 7;
    const int k = round * ROUND_BLOCKS + slot;
    if (k >= recon_n_active(s)) return;
    const int t = s.list[k];

    uint64_t fl;
    memcpy(&fl, &s.flags[slot * 8], 8);
    const bool is_horiz = (fl & 0x0101010101010101ull) == 0;
    const bool is_vert = (fl & 0x0202020202020202ull) == 0;
    const float c00 = s.tbuf[slot * TBUF_STRIDE + 8];

    // Vert (rle.rs:162-171, idct.rs:152-169) transforms the first column directly (kept in column 8
    // of the slot); every other class reads column i of the row-pass result (the transposition of
    // idct.rs:171-177).
    const bool vert = is_vert && !is_horiz;
    const bool dc_class = is_horiz && is_vert;                     // Dc or Zero, rle.rs:151-160
    const float *src = &s.tbuf[slot * TBUF_STRIDE + (vert ? 8 : i)];
    float col[8];
#pragma unroll
    for (int r = 0; r < 8; r++) col[r] = src[r * TBUF_ROW];

    f32x2 O[4];
    idct_1d_pairs(col, O);
    // class fix-ups as one multiply and one add (both exact where they must not change the value):
    //   Vert: x * B[0][0] (idct.rs:160)          others: x * 1.0
    //   Dc  : x * 0 + dc * 0.5 (idct.rs:119: exactly 0.5, not B00*B00; dc = 0 gives the Zero class)
    const f32x2 scale = splat2(dc_class ? 0.0f : (vert ? kBasis[0][0] : 1.0f));
    const f32x2 shift = splat2(dc_class ? c00 * 0.5f : 0.0f);
    int16_t *base = &s.res[(t < 16 ? t * 8 : 128 + (t - 16) * 8) + i];
#pragma unroll
    for (int jp = 0; jp < 4; jp++) {
        const f32x2 o = O[jp] * scale + shift;
        const f32x2 q4 = o * splat2(0.25f);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            // ((v / 4.0 + signum(v) * 0.5) as i16).clamp(-256, 255)  idct.rs:189-190.  signum(+-0) only
            // decides the sign of a half that truncation removes, so copysign is enough; truncation is
            // monotone, so clamping the float to [-256.0, 255.5] first gives the same integer.
            const float v = q4[h] + __builtin_copysignf(0.5f, o[h]);
            base[(jp * 2 + h) * RES_STRIDE] = (int16_t)(int)clampf(v, -256.0f, 255.5f);
        }
    }
}

// ---- phase 5: interpolation + residual + clip + store ------------------------------------
H263_DEV void recon_phase_output(const ReconArgs &a, ReconWave &s, const WaveFetch &f, int lane, const WavePos &p)
{
    uint8_t *cur = a.cur + (size_t)p.pic * a.L.frame_bytes;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const SegGeo g = seg_geometry(a, lane, k, p);
        if (!((s.valid_mask >> g.m) & 1)) continue;
        SegFetch sf = f.seg[k];
        if (sf.flags & SEG_BORDER) {
            // some tap lies outside the picture: redo the rows with per-tap clamping (gather.rs:24-25)
            const uint8_t *plane = a.ref + (size_t)p.pic * a.L.frame_bytes + g.plane_off;
            const int u = g.px + (sf.mvx >> 1), v = g.py + (sf.mvy >> 1);
            fetch_row_clamped(plane + (size_t)clampi(v, 0, g.ph - 1) * g.pitch, u, g.pw, sf.r0);
            if (sf.mvy & 1) fetch_row_clamped(plane + (size_t)clampi(v + 1, 0, g.ph - 1) * g.pitch, u, g.pw, sf.r1);
        }

        uint32_t lo = 0, hi = 0;                  // intra macroblocks start from zeros
        if (sf.flags & SEG_INTER) {
            if (a.debug_flags & 1) {
                lo = hi = 0x01010101u * (uint32_t)(sf.mvx & 0xff);
            } else {
                const int ix = sf.mvx & 1, iy = sf.mvy & 1;
                const uint32_t a0 = sf.r0[0], a1 = sf.r0[1];
                lo = a0; hi = a1;
                if (ix | iy) {
                    const uint32_t s0 = alignbyte(sf.r0[1], sf.r0[0], 1), s1 = alignbyte(sf.r0[2], sf.r0[1], 1);
                    if (iy) {
                        const uint32_t b0 = sf.r1[0], b1 = sf.r1[1];
                        if (ix) {
                            const uint32_t t0 = alignbyte(sf.r1[1], sf.r1[0], 1), t1 = alignbyte(sf.r1[2], sf.r1[1], 1);
                            lo = avg4_u8x4(a0, s0, b0, t0);      // gather.rs:103-111
                            hi = avg4_u8x4(a1, s1, b1, t1);
                        } else {
                            lo = avg2_u8x4(a0, b0);              // gather.rs:115-121
                            hi = avg2_u8x4(a1, b1);
                        }
                    } else {
                        lo = avg2_u8x4(a0, s0);
                        hi = avg2_u8x4(a1, s1);
                    }
                }
            }
        }
        if ((s.act_mask >> g.task) & 1) {
            const uint4 rv = *reinterpret_cast<const uint4 *>(&s.res[g.resoff]);
            lo = add_clip_u8x4(lo, rv.x, rv.y);
            hi = add_clip_u8x4(hi, rv.z, rv.w);
        }
        const uint64_t out = (uint64_t)lo | ((uint64_t)hi << 32);
        if ((a.debug_flags & 2) && out != 0x123456789abcdef0ull) continue;                   // diagnosis
        *reinterpret_cast<uint64_t *>(cur + (g.plane_off + (uint32_t)g.py * (uint32_t)g.pitch + (uint32_t)g.px)) = out;
    }
}

}  // namespace h263mi
