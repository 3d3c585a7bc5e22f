// recon_kernel.inl -- k_recon: dequantise + classify + 8x8 IDCT + half-pel motion
// compensation + residual add + clip.
//
// Replaces, per picture: the numeric part of inverse_rle (h263/src/decoder/cpu/rle.rs:
// 112-171), gather (gather.rs:140-204) and the three idct_channel calls (idct.rs:82-201)
// issued at the tail of decode_next_picture (state.rs:432-458).
//
// Work unit = one WAVE = 8 whole macroblocks of one macroblock row: a 128x16 luma strip and the 64x8 strips of Cb
// and Cr -- 48 block tasks, 3 072 reconstructed bytes.  A wave needs nothing from any other wave, so there is NO
// workgroup barrier: hand-offs between lanes go through 5.9 KB of wave-private LDS and rely only on the DS
// operations of one wave executing in order.  One wave is one workgroup (its LDS and registers are released when it
// ends); the XCD-aware work order of kernels.hip keeps neighbouring waves on one L2.
//
// Round 3 redesign (profiles/README.md, r03): the wave was reshaped to execute fewer instructions -- vector, scalar and
// vector-memory alike -- per reconstructed byte (what limits the launch is the CU's memory pipeline with the vector ALUs
// close behind it: DESIGN.md section 3, "What bounds the kernels"):
//   * 8 whole macroblocks per wave instead of half of them: records, prologue and the mark phase are paid once for
//     twice the pixels, and the IDCT rounds (8 blocks each) are filled from 48 tasks instead of 24;
//   * the reconstruction is assembled in a BYTE strip in LDS: the prediction is written there first, the lanes of the
//     column pass add their residuals to it in place (read-modify-write of the 8 bytes of their column), the store
//     phase is a plain LDS -> HBM copy.  (Rounds 1-2 kept an i16 residual strip and added it to the prediction in
//     the output phase: the add / saturate / pack ran for every segment of the wave, coded or not.)
//   * a lane predicts VERTICALLY ADJACENT rows of one block (4 luma rows, 2 chroma rows): 5 + 3 reference rows
//     are loaded instead of 8 + 4, the byte alignment and the horizontal half of the half-pel filter are computed
//     once per loaded row and shared by the two output rows that use it, and vector, flags and addresses are per
//     lane, not per row.
//
// Wave timeline:
//   records  : 8 records (256 B) -> LDS
//   mark     : lane t < 48 = block task t: decides whether the block goes through the IDCT (coded & !kill, or
//              uncoded intra with non-zero DC) and builds its 8-byte DESCRIPTOR (where its coefficients are,
//              quantiser, INTRADC level, intra flag, task number); lanes 48..55 = the 8 macroblocks: chroma vector
//              (packed 16-bit arithmetic) and the inter flag.  The wave-wide masks are ballots (scalar registers);
//              the descriptors of the active tasks are compacted in LDS
//   fetch    : ALL global loads are issued now, before any arithmetic: the coefficient row of the first IDCT round,
//              then the reference rows of the lane's luma block (5) and chroma block (3): one dword-aligned 12-byte
//              load per row; rows nobody needs read offset 0; picture-edge lanes load the window that holds all
//              their clamped taps
//   rows 0   : first IDCT round, row pass (lane = one coefficient row): dequant (packed i16), T = C x B -> LDS.  Runs
//              under the reference loads.
//   predict  : lane = 4 luma rows + 2 chroma rows of 8 pixels: one branch-free half-pel form on packed bytes
//              -> byte strip in LDS (zeros where nothing is predicted)
//   cols 0.. : column pass (lane = one pixel column): column of T (the LDS transposition), rounding, residual added
//              to the strip in place with the final clamp; then the remaining rounds, row + column pass each; both
//              passes stop at the last non-zero coefficient column / row of the round; the block classes
//              (rle.rs:138-170) come out of two ballots
//   store    : strip -> frame, 8 bytes per lane and row
//
// Bit-exactness rules: f32 multiply and add are separately rounded
// (translation unit built with -ffp-contract=off), accumulation order over the
// frequency index is sequential, and the Dc / Vert classes keep their own arithmetic.
#pragma once

#include "dev_common.h"

namespace h263mi {

#ifndef H263MI_RECON_WAVES
#define H263MI_RECON_WAVES 1       // measured in round 1: 1 -> 0.237 ms, 2 -> 0.241 ms, 4 -> 0.248 ms per launch
#endif
constexpr int RECON_WAVES = H263MI_RECON_WAVES;      // waves per workgroup of k_recon (they are independent: 1 or 2)
constexpr int RECON_THREADS = RECON_WAVES * 64;
constexpr int TILE_MBX = 8, TILE_MBY = 2;      // macroblocks per tile (the unit the frame-pipelined work list is grouped by)
constexpr int TILE_WAVES = TILE_MBY;           // waves per tile: one per macroblock row
constexpr int WAVE_TASKS = 48;                 // 32 luma + 8 Cb + 8 Cr blocks per wave
constexpr int LUMA_TASKS = 32;
constexpr int ROUND_BLOCKS = 8;                // blocks per IDCT round (8 lanes each)
constexpr int TBUF_ROW = 8;                    // floats per row of a block slot: T[r][0..7], written as two 16-byte stores
constexpr int TBUF_STRIDE = 8 * TBUF_ROW + 8;  // 72 floats per slot: 8 pad floats make the column reads bank-conflict free
// Row r of a slot starts at float 8 r + (r & 4): rows 4..7 sit four floats further.  A 16-byte store is served eight
// lanes at a time; with plain 8 r the eight rows of a slot start at banks 0, 8, 16, 24, 0, 8, 16, 24 -- a two-way
// conflict on every row-pass store (most of the launch's LDS bank conflicts); shifted, they cover all 32 banks.  The
// slot still ends at float 68 < TBUF_STRIDE, and the column reads (one float per lane, eight lanes per slot) stay
// conflict-free: a constant added per row.
H263_HD int tbuf_row_offset(int r) { return r * TBUF_ROW + (r & 4); }
constexpr int PIX_STRIDE = 128;                // bytes per row of the reconstruction strip
constexpr int PIX_CHROMA = 16 * PIX_STRIDE;    // rows 0..15: luma; rows 16..23: Cb in columns 0..63, Cr in 64..127
constexpr int MB_LANE0 = WAVE_TASKS;           // lanes 48..55 of the mark phase are the 8 macroblocks

struct ReconWave {
    uint32_t rec[TILE_MBX][8];                 // the 8 records as dwords: [0] mb_type | quant << 8 | cbp << 16 | kill << 24,
                                               // [1..4] the four vectors (x | y << 16), [5] intradc 0..3, [6] intradc 4..5, [7] coeff_index
    uint32_t mvc[TILE_MBX];                    // chroma vector per macroblock, x | y << 16 (gather.rs:182)
    uint32_t desc[WAVE_TASKS][2];              // descriptors of the active tasks, compacted (see TaskInfo)
    float    tbuf[ROUND_BLOCKS * TBUF_STRIDE]; // row pass results T[r][i] of the round's 8 blocks
    uint8_t  pix[24 * PIX_STRIDE];             // the reconstruction strip: prediction, then + residuals, then stored
};
static_assert(sizeof(ReconWave) <= 6064, "27 waves per CU need <= 6 068 bytes of LDS per wave (profiles/r03_b_ab_lds_pad.txt)");
static_assert(sizeof(MbRecord) == 32 && offsetof(MbRecord, mv) == 4 && offsetof(MbRecord, intradc) == 20 &&
              offsetof(MbRecord, coeff_index) == 28, "record words used by the mark phase");

// What the mark phase knows about a block task.  d0: byte offset of the block's 64 LEVELs from the picture's first
// coefficient block, or NO_COEFFS when the block has no TCOEF (or lies outside the pool).  d1: quant | INTRADC level
// << 8 | intra << 19 | task << 20.
constexpr uint32_t NO_COEFFS = 0xffffffffu;
struct TaskInfo {
    uint32_t d0, d1;
    uint32_t active;                           // 1: the block goes through the IDCT
    uint32_t inter;                            // 1: (macroblock lanes) inside the picture and inter coded
    uint32_t bad_index;                        // 1: coded block outside the coefficient pool
    uint32_t moving;                           // 1: (macroblock lanes) some vector of the macroblock is not (0, 0)
};
H263_HD uint32_t desc_quant(uint32_t d1) { return d1 & 0xffu; }
H263_HD uint32_t desc_level(uint32_t d1) { return (d1 >> 8) & 0x7ffu; }
H263_HD bool     desc_intra(uint32_t d1) { return (d1 >> 19) & 1u; }
// (bits 20 and up: where the block's 8x8 pixels start in the reconstruction strip, in units of 8 bytes -- worked out once per
// task in the mark phase; rounds 1-3 carried the task number and every lane of every round derived the origin from it)
H263_HD int      desc_pix_origin(uint32_t d1) { return (int)((d1 >> 20) << 3); }
// (recon_phase_mark builds that field with literal shifts: a 128-byte strip row, the chroma rows behind 16 luma rows, and
// twelve bits for origin / 8)
static_assert(PIX_STRIDE == 128 && PIX_CHROMA == 16 * PIX_STRIDE && (PIX_CHROMA / 8 + 15) < 4096,
              "the strip geometry recon_phase_mark's descriptor field (org8) is written for");

// Wave-wide bit masks.  On the device they come out of ballots and live in scalar registers; the CPU logic checker
// (tests/sim) runs the lanes one after the other and ORs the lanes' bits together.
struct WaveMasks {
    uint32_t valid;            // bit m: macroblock m lies inside the picture
    uint64_t act;              // bit t: block task t goes through the IDCT
    uint32_t inter;            // bit m: macroblock m is inside the picture and inter coded
};

// position of a wave's work: which picture, which 8 macroblocks
struct WavePos {
    int pic, mbx0, mby;
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
H263_HD int popc64(uint64_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(v);
#else
    return __builtin_popcountll(v);
#endif
}
// set bits of `mask` below bit `lane` (v_mbcnt_lo / v_mbcnt_hi on the device)
H263_DEV int popc_below(uint64_t mask, int lane)
{
#if defined(__HIP_DEVICE_COMPILE__)
    (void)lane;
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
#else
    return __builtin_popcountll(mask & ((1ull << lane) - 1ull));
#endif
}

H263_HD int recon_n_active(const WaveMasks &k) { return popc64(k.act); }

// Block task t of a wave: t < 32 luma, block row t >> 4 (the upper / lower blocks of the macroblocks), block column
// t & 15; then the 8 Cb blocks, then the 8 Cr blocks.
H263_HD int task_mb(int t) { return t < LUMA_TASKS ? ((t >> 1) & 7) : (t & 7); }
H263_HD int task_blk(int t) { return t < LUMA_TASKS ? (((t >> 4) << 1) | (t & 1)) : 4 + ((t >> 3) & 1); }
// where the block's 8x8 pixels start in the reconstruction strip
H263_HD int task_pix_origin(int t)
{
    return t < LUMA_TASKS ? (t >> 4) * 8 * PIX_STRIDE + (t & 15) * 8 : PIX_CHROMA + ((t >> 3) & 1) * 64 + (t & 7) * 8;
}

// ---- packed helpers (device: single instructions; host build: plain C for tests/sim) ----------
// v_lerp_u8: per byte (a + b + (c & 1)) >> 1
H263_DEV uint32_t lerp_u8x4(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_lerp(a, b, c);
#else
    uint32_t out = 0;
    for (int k = 0; k < 4; k++) {
        const uint32_t x = (a >> (8 * k)) & 0xff, y = (b >> (8 * k)) & 0xff, r = (c >> (8 * k)) & 1;
        out |= ((x + y + r) >> 1) << (8 * k);
    }
    return out;
#endif
}

// clamp(v, lo, hi) for lo <= hi as one v_med3_i32
H263_DEV int med3i(int v, int lo, int hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
#else
    return clampi(v, lo, hi);
#endif
}

// row * pitch + x with 24-bit operands (one v_mad_u32_u24; rows and pitches are far below 2^24)
H263_DEV uint32_t mad24(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b) + c;
#else
    return a * b + c;
#endif
}

// a value every lane of the wave agrees on, moved to a scalar register
H263_DEV uint32_t wave_uniform(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
#else
    return v;
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
// The same table times 0.25, for the column pass.  idct.rs:189 divides the result of the second pass by 4.0 before it
// rounds; a power of two commutes with every IEEE rounding on the way (no product or partial sum comes anywhere near
// the subnormal range: the smallest non-zero one is ~0.19 * 0.19 / 4), so sum(in[f] * (B[f][i] / 4)) IS sum(in[f] *
// B[f][i]) / 4, bit for bit, and the multiplication by 0.25 costs nothing.
template <int DIVISOR>
struct BasisScaled {
    float v[8][8];
    constexpr BasisScaled() : v{}
    {
        constexpr float b[8][8] = {H263MI_BASIS_ROWS};
        for (int f = 0; f < 8; f++)
            for (int i = 0; i < 8; i++) v[f][i] = b[f][i] * (1.0f / DIVISOR);
    }
};
typedef BasisScaled<4> BasisQuarter;
H263_CONST_TABLE BasisQuarter kBasisQuarter = BasisQuarter();
// ... and times 1/16, for the row pass: the dequantiser hands the coefficients on times 16 (dequant_pair_i16), the same
// argument makes sum((16 c[f]) * (B[f][i] / 16)) = sum(c[f] * B[f][i]) bit for bit.
typedef BasisScaled<16> BasisSixteenth;
H263_CONST_TABLE BasisSixteenth kBasisSixteenth = BasisSixteenth();

H263_DEV f32x2 splat2(float v) { f32x2 r = {v, v}; return r; }
// four consecutive floats (16-byte aligned) in one store
H263_DEV void float4_store(float *dst, f32x2 a, f32x2 b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 v = {a[0], a[1], b[0], b[1]};
    *reinterpret_cast<f32x4 *>(dst) = v;
#else
    dst[0] = a[0]; dst[1] = a[1]; dst[2] = b[0]; dst[3] = b[1];
#endif
}
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) float (*BasisPtr)[8];     // constant address space: scalar loads
#else
typedef const float (*BasisPtr)[8];
#endif
// The table pointer behind an opaque scalar asm: the 64 scalar registers the table occupies are then
// (re)loaded where a pass starts instead of being held -- and spilled around -- for the whole kernel.
H263_DEV BasisPtr basis_table()
{
#if defined(__HIP_DEVICE_COMPILE__)
    BasisPtr b = (BasisPtr)kBasis;
    asm volatile("" : "+s"(b));
    return b;
#else
    return kBasis;
#endif
}
H263_DEV BasisPtr basis_table_quarter()
{
#if defined(__HIP_DEVICE_COMPILE__)
    BasisPtr b = (BasisPtr)kBasisQuarter.v;
    asm volatile("" : "+s"(b));
    return b;
#else
    return kBasisQuarter.v;
#endif
}
H263_DEV BasisPtr basis_table_sixteenth()
{
#if defined(__HIP_DEVICE_COMPILE__)
    BasisPtr b = (BasisPtr)kBasisSixteenth.v;
    asm volatile("" : "+s"(b));
    return b;
#else
    return kBasisSixteenth.v;
#endif
}
H263_DEV f32x2 basis_pair(BasisPtr B, int f, int ip) { f32x2 r = {B[f][2 * ip], B[f][2 * ip + 1]}; return r; }

}  // namespace h263mi
#include "mutants.h"         // the arithmetic mutants of the tests (compile-time off in the product)
namespace h263mi {

// idct_1d (idct.rs:52-65): out[i] = sum over f, in order, of in[f] * B[f][i].  The leading
// "0.0 +" is dropped: it can only change the sign of a zero, which never reaches the integer result.
// Only the terms f < n are accumulated: the caller guarantees in[f] == 0 for f >= n, and adding a
// zero product changes nothing but (again) the sign of a zero.  n is uniform over the wave, so the
// early exits are scalar branches.
H263_DEV void idct_1d_pairs(BasisPtr B, const float in[8], f32x2 out[4], int n, bool first_term_raw = false, float raw_scale = 1.0f)
{
    // first_term_raw: the row pass of a Vert / Dc block hands in[0] on unscaled (see recon_phase_idct_rows); row 0
    // of the basis is one constant
    // (idct.rs:40: BASIS_TABLE[0][i] = 0.70710677 for every i)
    // (raw_scale: what `in` is scaled by relative to the table -- a power of two; 1 / raw_scale takes it out again)
    const f32x2 first = splat2(in[0]) * splat2(first_term_raw ? raw_scale : B[0][0]);
    if (mutants::kPairwise) {                        // (compile-time false in the product: mutants.h)
        mutants::idct_1d_pairwise(B, in, out, first);
        return;
    }
    if (n <= 1) {                                    // uniform
#pragma unroll
        for (int ip = 0; ip < 4; ip++) out[ip] = first;
        return;
    }
    // (the second term is added to the one register pair that holds the first: written as `out = first` followed by the
    // loop from f = 1, the compiler copies the first term into all eight result registers ahead of the branch)
#pragma unroll
    for (int ip = 0; ip < 4; ip++) out[ip] = first + splat2(in[1]) * basis_pair(B, 1, ip);
#pragma unroll
    for (int f = 2; f < 8; f++) {
        if (f >= n) break;
#pragma unroll
        for (int ip = 0; ip < 4; ip++) {
            const f32x2 pr = splat2(in[f]) * basis_pair(B, f, ip);
            out[ip] = out[ip] + pr;
        }
    }
}

// Sparsity of one IDCT round, shared by the 8 blocks of the wave (what rle.rs:138-171 does per block
// with its Horiz / Vert / Dc classes, generalised): number of leading coefficient columns (in pairs)
// and rows that can be non-zero.
H263_HD int cols_from_mask(uint32_t word_mask)      // bit j: some lane has a non-zero LEVEL in columns 2j, 2j+1
{
    return (word_mask & 8) ? 8 : (word_mask & 4) ? 6 : (word_mask & 2) ? 4 : 2;
}
H263_HD int rows_from_mask(uint32_t row_mask)       // bit r: some block has a non-zero coefficient in row r
{
    int n = 1;
    for (int r = 1; r < 8; r++)
        if ((row_mask >> r) & 1) n = r + 1;
    return n;
}

// rle.rs:130-133 for two LEVELs at once, as they arrive (a pair of int16 in one dword):
// sign(L) * (q*(2|L|+1) - (q even)) = L*2q + sign(L)*(q - parity), clamped to [-2048, 2047]; 0 stays 0.
// Returns SIXTEEN TIMES that value in each half: computed as L * 32q + sign(L) * 16 (q - parity) with a saturating
// multiply-add, the clamp to 12 bits IS the saturation to 16 bits -- a value beyond it lands on 32767 or -32768, and
// with the four low bits cleared those are 16 * 2047 and 16 * -2048 (every other result is a multiple of 16 already).
// One multiply-add and one AND instead of a multiply-add and two clamps; the row pass takes the factor out again
// through its table (kBasisSixteenth).
// VALID ONLY WHERE THE REFERENCE'S i16 ARITHMETIC DOES NOT OVERFLOW: rle.rs:130-133 multiplies in i16, and a release build
// (what Ruffle ships) WRAPS where q * (2|L| + 1) exceeds 32767 -- reachable with Sorenson's 11-bit escape LEVELs from q = 16
// up (parser/block.rs:694-708) -- so the wrapped product, not the mathematical one, is what it clamps.  No |L| <= 511
// overflows at any quantiser (31 * 1023 = 31713); a round that holds a wider LEVEL (RowIn::wide, one ballot) takes
// dequant_pair_wrap below instead.
// `two_q2`, `qmp2`: 2q and q - parity in both halves of a dword (2q <= 62, q - parity <= 31: times 16 they fit 10 bits).
constexpr float DEQUANT_SCALE = 16.0f;
H263_DEV uint32_t dequant_pair_i16(uint32_t levels, uint32_t two_q2, uint32_t qmp2)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // all in inline asm: written with vector min/max the compiler turns the sign into four compares and selects
    uint32_t sg, t, v;
    asm("v_pk_min_i16 %0, %1, 1 op_sel_hi:[1,0]\n\tv_pk_max_i16 %0, %0, -1 op_sel_hi:[1,0]" : "=&v"(sg) : "v"(levels));   // -1, 0 or +1
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(t) : "v"(sg), "v"(qmp2 << 4));
    asm("v_pk_mad_i16 %0, %1, %2, %3 clamp" : "=v"(v) : "v"(levels), "v"(two_q2 << 4), "v"(t));
    return mutants::kDequantSaturation ? v : (v & 0xfff0fff0u);       // (mutants.h: compile-time false in the product)
#else
    uint32_t out = 0;
    for (int h = 0; h < 2; h++) {
        const int level = (int)(int16_t)(levels >> (16 * h));
        const int two_q = (int)(two_q2 & 0xffffu), qmp = (int)(int16_t)(qmp2 & 0xffffu);
        const int sg = level > 0 ? 1 : (level < 0 ? -1 : 0);
        const int v = clampi(16 * (level * two_q + sg * qmp), -32768, 32767) & ~15;
        out |= ((uint32_t)v & 0xffffu) << (16 * h);
    }
    return out;
#endif
}

// rle.rs:130-133 exactly as a release build of the reference executes it, for ANY int16 LEVEL: every step is an i16 that
// wraps.  sign(L) * (q * (2|L| + 1) + parity) = L * 2q + sign(L) * (q + parity) modulo 2^16 (|L| itself wraps for -32768, and
// so does this form), then the clamp to [-2048, 2047]: a multiply-add WITHOUT saturation, a max and a min.  Returns the
// value itself (not 16 x): the row pass of such a round uses the unscaled table.  Examples: q = 31, L = -1024: 31 * 2049 =
// 63519 = -2017 (mod 2^16), times -1 -> +2017; q = 31, L = 529: 32829 = -32707 -> -2048 (where the mathematical product
// would clamp to +2047).  Only rounds that hold a LEVEL outside [-512, 511] come here (RowIn::wide).
H263_DEV uint32_t dequant_pair_wrap(uint32_t levels, uint32_t two_q2, uint32_t qmp2)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t sg, t, v;
    asm("v_pk_min_i16 %0, %1, 1 op_sel_hi:[1,0]\n\tv_pk_max_i16 %0, %0, -1 op_sel_hi:[1,0]" : "=&v"(sg) : "v"(levels));   // -1, 0 or +1
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(t) : "v"(sg), "v"(qmp2));
    if (mutants::kDequantWrap) asm("v_pk_mad_i16 %0, %1, %2, %3 clamp" : "=v"(v) : "v"(levels), "v"(two_q2), "v"(t));   // (mutants.h)
    else asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(v) : "v"(levels), "v"(two_q2), "v"(t));
    return pk_min_i16(pk_max_i16(v, 0xf800f800u), 0x07ff07ffu);
#else
    uint32_t out = 0;
    for (int h = 0; h < 2; h++) {
        const int level = (int)(int16_t)(levels >> (16 * h));
        const int two_q = (int)(two_q2 & 0xffffu), qmp = (int)(int16_t)(qmp2 & 0xffffu);
        const int sg = level > 0 ? 1 : (level < 0 ? -1 : 0);
        const int v = clampi((int)(int16_t)(uint16_t)((uint32_t)(level * two_q + sg * qmp) & 0xffffu), -2048, 2047);
        out |= ((uint32_t)v & 0xffffu) << (16 * h);
    }
    return out;
#endif
}

// ---- phase 0: records -> LDS -------------------------------------------------------
H263_DEV uint32_t recon_valid_mask(const ReconArgs &a, const WavePos &p)
{
    const int n = (int)a.L.mbw - p.mbx0;                        // macroblocks of the wave inside the picture
    return (p.mby < (int)a.L.mbh && n > 0) ? (n >= TILE_MBX ? 0xffu : (1u << n) - 1u) : 0u;
}

// Coded blocks this wave may address: the pool holds a.coeff_pool_blocks blocks (when the caller told us), the
// picture's first block is p.cbase, and block offsets are 32-bit byte offsets (2^25 blocks of 128 bytes).
H263_DEV uint32_t recon_block_limit(const ReconArgs &a, const WavePos &p)
{
    if (!a.coeff_checked) return 1u << 25;
    // min(pool - cbase, 2^25), 0 when the picture's base lies at or beyond the end of the pool; in 32-bit pieces so that it
    // stays on the scalar unit.  The base comes out of device memory nobody has validated (h263mi_batch_decode[_events]): it
    // is compared as the unsigned 64-bit number it is -- round 6's GPU fuzzer found that the sign of `pool - cbase` alone lets
    // a base of 2^64 - 2^20 through (the difference wraps to a small positive number, the block address to 128 MB in FRONT
    // of the pool: a memory access fault).
    const uint32_t ch = (uint32_t)(p.cbase >> 32), cl = (uint32_t)p.cbase;
    const uint32_t ph = (uint32_t)(a.coeff_pool_blocks >> 32), pl = (uint32_t)a.coeff_pool_blocks;
    if (ch > ph || (ch == ph && cl >= pl)) return 0u;
    const uint64_t left = a.coeff_pool_blocks - p.cbase;        // 1 .. pool
    const uint32_t hi = (uint32_t)(left >> 32), lo = (uint32_t)left;
    return (hi || lo > (1u << 25)) ? (1u << 25) : lo;
}

// the wave's word of the sparse record index (ReconArgs::mb_group_index), 0 when the records are dense: wave-uniform
H263_DEV uint32_t recon_group_word(const ReconArgs &a, const WavePos &p)
{
    if (!a.mb_group_index) return 0u;
    return a.mb_group_index[(size_t)p.pic * a.groups_per_picture + (size_t)p.mby * a.tiles_x + (size_t)(p.mbx0 >> 3)];
}

H263_DEV void recon_phase_load(const ReconArgs &a, ReconWave &s, int lane, const WavePos &p, uint32_t group_word = 0)
{
    if (a.mb_group_index) {
        // sparse records: macroblock m of the wave has a record when bit m of the group word is set -- the (number of set
        // bits below m)-th behind the group's first --, else it is not coded: INTER, quantiser 1 (never used), all else zero
        if (lane < TILE_MBX * 2) {
            const int m = lane >> 1, part = lane & 1;
            const uint32_t mask = group_word & 0xffu;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (p.mbx0 + m < (int)a.L.mbw && p.mby < (int)a.L.mbh) {
                if ((mask >> m) & 1u) {
                    const size_t k = (size_t)(group_word >> 8) + (size_t)popc32(mask & ((1u << m) - 1u));
                    v = reinterpret_cast<const uint4 *>(a.mbs + a.mb_base[p.pic] + k)[part];
                } else if (part == 0) {
                    v.x = (uint32_t)H263MI_MB_INTER | (1u << 8);
                }
            }
            reinterpret_cast<uint4 *>(&s.rec[m][0])[part] = v;
        }
        return;
    }
    if (lane < TILE_MBX * 2) {
        // 16 lanes x 16 B = the 8 records of this macroblock row segment (one 256-B run)
        const int m = lane >> 1, part = lane & 1;
        const int mbx = p.mbx0 + m;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (mbx < (int)a.L.mbw && p.mby < (int)a.L.mbh) {
#if defined(H263MI_TIMING_REC_WRAP)
            const MbRecord *r = a.mbs + (((size_t)p.mby * a.L.mbw + mbx) & 2047u);   // TIMING EXPERIMENT ONLY: 64 KB of records
#else
            const MbRecord *r = a.mbs + (size_t)p.pic * a.mbs_per_picture + (size_t)p.mby * a.L.mbw + mbx;
#endif
            v = reinterpret_cast<const uint4 *>(r)[part];
        }
        reinterpret_cast<uint4 *>(&s.rec[m][0])[part] = v;
    }
}

// ---- phase 1: which blocks need the IDCT, their descriptors; chroma vectors -------------------------------
// Straight-line and branch-free: every lane reads the eight words of "its" record (four two-dword LDS reads from one
// address register), the decisions are plain arithmetic.  Lanes 0..47 are block task `lane`, lanes 48..55 macroblock
// lane - 48; the other lanes compute along and are ignored.
H263_DEV TaskInfo recon_phase_mark(const ReconArgs &, ReconWave &s, int lane, const WavePos &, uint32_t valid_mask,
                                   uint32_t block_limit)
{
    // Decisions are 0 / 1 integers combined with shifts, ands and ors -- the 2-cycle VALU instructions; written with
    // bool / ?: the same logic compiles to compares and selects at twice the price each (profiles/README.md).
    const uint32_t ln = (uint32_t)lane;
    const uint32_t is_task = (ln - (uint32_t)WAVE_TASKS) >> 31;                // lane < 48
    const uint32_t is_mb = ((ln - (uint32_t)MB_LANE0) < (uint32_t)TILE_MBX) ? 1u : 0u;
    const uint32_t luma_task = (ln - (uint32_t)LUMA_TASKS) >> 31;              // lane < 32
    const uint32_t m = (ln >> luma_task) & 7u;                                 // lane >> 1 for luma tasks, lane & 7 else
    const uint32_t blk = luma_task ? (((ln >> 4) << 1) | (ln & 1u)) : 4u + ((ln >> 3) & 1u);
    const uint32_t *r = s.rec[m];
    const uint32_t w0 = r[0], w7 = r[7], w5 = r[5], w6 = r[6], w1 = r[1], w2 = r[2], w3 = r[3], w4 = r[4];

    const uint32_t mb_type = w0 & 0xffu;
    const uint32_t valid = (valid_mask >> m) & 1u;
    // MacroblockType (types.rs:631-649, 653-663) as bit tables: intra = {3, 4}, inter = {0, 1, 2, 5}.  (A type above 5
    // never comes out of a parser and is refused by every host entry point.)
    const uint32_t intra = (0x18u >> (mb_type & 31u)) & 1u, inter_type = (0x27u >> (mb_type & 31u)) & 1u;
    const uint32_t ck = (w0 >> 16) >> blk;                                    // bit 0: coded, bit 8: killed
    const uint32_t coded = ck & 1u, killed = (ck >> 8) & 1u;
    const uint32_t dcb = (uint32_t)((((uint64_t)w6 << 32) | w5) >> (8u * blk)) & 0xffu;
    const uint32_t dc_nonzero = (dcb + 255u) >> 8, dc_is_ff = (dcb + 1u) >> 8;
    // number of this block among the macroblock's coded blocks -> its place in the pool
    const uint32_t idx = w7 + (uint32_t)popc32((w0 >> 16) & ((1u << blk) - 1u) & 0x3fu);
    const uint32_t worst = w7 > idx ? w7 : idx;                               // (idx may have wrapped)
    const uint32_t in_range = worst < block_limit ? 1u : 0u;

    TaskInfo t;
    // coded & kill -> Zero (rle.rs:125-127); uncoded inter -> Zero; uncoded intra -> Dc(level)
    t.active = is_task & valid & ((coded & (killed ^ 1u)) | ((coded ^ 1u) & intra & dc_nonzero));
    t.bad_index = t.active & coded & (in_range ^ 1u);
    t.d0 = (idx << 7) | (0u - ((coded & in_range) ^ 1u));                    // NO_COEFFS unless coded and in range
    // IntraDc::into_level (types.rs:955-961): code << 3, 0xff -> 1024; here already shifted into its descriptor field
    const uint32_t level8 = (dcb << 11) ^ ((0u - dc_is_ff) & ((2040u ^ 1024u) << 8));
    // task_pix_origin(lane) / 8: luma (ln >> 4) * 128 + (ln & 15), chroma 256 + ((ln >> 3) & 1) * 8 + (ln & 7)
    const uint32_t org_l = ((ln >> 4) << 7) + (ln & 15u), org_c = 256u + (((ln >> 3) & 1u) << 3) + (ln & 7u);
    const uint32_t org8 = org_c ^ ((org_c ^ org_l) & (0u - luma_task));
    t.d1 = ((w0 >> 8) & 0xffu) | level8 | ((0x18u << 19 >> (mb_type & 31u)) & (1u << 19)) | (org8 << 20);
    (void)intra;
    // which macroblocks take a prediction from the reference picture (gather.rs:136-149)
    t.inter = is_mb & valid & inter_type;
    const uint32_t any_mv = w1 | w2 | w3 | w4;
    t.moving = is_mb & ((any_mv | (0u - any_mv)) >> 31);                      // some vector component is not zero

    // gather.rs:182 / types.rs:759-768: chroma vector from the i16 sum of the four luma vectors, both components at
    // once in the two halves of a dword.  s = sum, whole = (s >> 4) << 1 = (s >> 3) & ~1, frac = s & 15,
    // result = whole + (frac > 2) + (frac >= 14) = whole + ((frac + 13) >> 4) + ((frac + 2) >> 4)
    {
        const uint32_t sum = pk_add_u16(pk_add_u16(w1, w2), pk_add_u16(w3, w4));
        const uint32_t whole = pk_ashr_i16(sum, 3) & 0xfffefffeu;
        const uint32_t frac = sum & 0x000f000fu;
        const uint32_t up = (((frac + 0x000d000du) >> 4) & 0x00010001u) + (((frac + 0x00020002u) >> 4) & 0x00010001u);
        if (is_mb) s.mvc[m] = pk_add_u16(whole, up);
    }
    return t;
}

// ---- phase 2: compact the descriptors of the active tasks ----------------------------------------------
H263_DEV void recon_phase_compact(ReconWave &s, int lane, const TaskInfo &t, uint64_t act_mask)
{
    if (!t.active) return;
    uint32_t *d = s.desc[popc_below(act_mask, lane)];
    d[0] = t.d0;
    d[1] = t.d1;
}

// status bits of the wave (wave-uniform on the device: the caller passes ballots)
H263_DEV void recon_report(const ReconArgs &a, int lane, int pic, bool inter_without_reference, bool bad_index)
{
    const uint32_t bits = (inter_without_reference ? STATUS_INTER_WITHOUT_REFERENCE : 0u) |    // Error::UncodedIFrameBlocks
                          (bad_index ? STATUS_COEFF_INDEX_OUT_OF_RANGE : 0u);
    if (!bits || lane != 0) return;
#if defined(__HIP_DEVICE_COMPILE__)
    atomicOr(a.status + pic, bits);
#else
    a.status[pic] |= bits;
#endif
}

// ---- phase 3: issue every global load of the wave ------------------------------------------
// A lane predicts two pieces of the strip: rows 4g .. 4g+3 of one luma block column (g = lane >> 4, column lane & 15)
// and rows 2g' .. 2g'+1 of one chroma block (plane lane >> 5, g' = (lane >> 3) & 3, macroblock lane & 7).  All rows
// of a piece lie in ONE block: one vector, one set of flags, and output row j needs reference rows v + j and
// v + j + (mvy & 1) -- 5 (3) consecutive rows for 4 (2) output rows.
enum : uint8_t { SEG_INTER = 1, SEG_BORDER = 2 };
constexpr int LUMA_ROWS = 4, CHROMA_ROWS = 2;
struct WaveFetch {
    uint32_t ly[LUMA_ROWS + 1][3];     // luma: reference rows v .. v+4, bytes ua .. ua+11 (the last one only where mvy is odd)
    uint32_t ch[CHROMA_ROWS + 1][3];   // chroma: rows v .. v+2
    uint32_t mvw[2];                   // [0] luma, [1] chroma: mvx | mvy << 16; (0, 0) when the macroblock takes no prediction
    uint32_t flags;                    // bits 0..1 luma, 2..3 chroma: SEG_INTER: motion compensated (else prediction = 0);
                                       // SEG_BORDER: some tap falls outside the picture, redone with clamping
    uint32_t d0, d1;                   // descriptor of the lane's block in the first IDCT round
    uint4    coef0;                    // ... and its coefficient row
};

// piece k of a lane (0 luma, 1 chroma) -> geometry
struct PieceGeo {
    int m, blk, px, py, pitch, pw, ph, pixoff;
    uint32_t plane_off;                // luma: 0; chroma: the lane's plane (lanes 0..31 Cb, 32..63 Cr)
};

H263_DEV PieceGeo piece_geometry(const ReconArgs &a, int lane, int k, const WavePos &p)
{
    PieceGeo g;
    if (k == 0) {
        const int sx = lane & 15, rg = lane >> 4;                 // 16 lanes = one 128-byte luma line
        g.m = sx >> 1;
        g.blk = ((rg >> 1) << 1) | (sx & 1);
        g.px = p.mbx0 * 16 + sx * 8;
        g.py = p.mby * 16 + rg * LUMA_ROWS;
        g.pitch = (int)a.L.pitch_y; g.pw = (int)a.L.width; g.ph = (int)a.L.height;
        g.plane_off = 0;
        g.pixoff = rg * LUMA_ROWS * PIX_STRIDE + sx * 8;
    } else {
        const int plane = lane >> 5, sx = lane & 7, rg = (lane >> 3) & 3;
        g.m = sx;
        g.blk = 4 + plane;
        g.px = p.mbx0 * 8 + sx * 8;
        g.py = p.mby * 8 + rg * CHROMA_ROWS;
        g.pitch = (int)a.L.pitch_c; g.pw = (int)a.L.cwidth; g.ph = (int)a.L.cheight;
        g.plane_off = plane ? a.L.off_cr : a.L.off_cb;
        g.pixoff = PIX_CHROMA + rg * CHROMA_ROWS * PIX_STRIDE + plane * 64 + sx * 8;
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

// Border path (gather.rs:24-25: every tap is clamped to the picture on its own).  Clamping only ever repeats the
// first or the last pixel of the row, so all nine taps of a row lie in one 12-byte window of that row:
// column 0.. when the piece starts left of the picture, else the last aligned window that still reaches the
// final pixel.  The fetch phase loads that window; the predict phase picks tap k = window[clamp(u + k) - ub].
H263_HD int border_window(int u, int pw)
{
    const int last = (pw - 9) & ~3;                    // smallest multiple of 4 that is >= pw - 12
    return (u < 0 || last < 0) ? 0 : last;
}

H263_DEV void gather_row_clamped(uint32_t w[3], int u, int ub, int pw)
{
    uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
#if defined(__HIP_DEVICE_COMPILE__)
    // pin the window in registers: left to itself the compiler turns the byte picks below into indexed loads from
    // a scratch copy of the array
    asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2));
#endif
    const uint64_t lo64 = (uint64_t)w0 | ((uint64_t)w1 << 32), hi64 = (uint64_t)w1 | ((uint64_t)w2 << 32);
    uint32_t o0 = 0, o1 = 0, o2 = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int idx = med3i(u + k, 0, pw - 1) - ub;                    // 0 .. 11
        // byte idx of the 12-byte window: from bytes 0..7 when idx < 4, else from bytes 4..11
        const uint32_t b = (uint32_t)(idx < 4 ? lo64 >> (8 * idx) : hi64 >> (8 * (idx - 4))) & 0xffu;
        if (k < 4) o0 |= b << (8 * k);
        else if (k < 8) o1 |= b << (8 * (k - 4));
        else o2 = b;
    }
    w[0] = o0; w[1] = o1; w[2] = o2;
}

// 16 bytes of read-once data (coefficients)
H263_DEV uint4 load16_stream(const uint8_t *p)
{
#if defined(__HIP_DEVICE_COMPILE__) && defined(H263MI_EXP_NT_COEFS)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const uint4 *>(p);
#endif
}

// coefficient row `r` of the block descriptor d0 describes (any mapped address when there is none): a wave-uniform base
// plus a 32-bit lane offset -- the picture's first coefficient block (or, for a call without a pool, the records) and the
// block's byte offset or 0 -- so that the load takes its base from scalar registers and the lane needs one add and one
// AND for its address (round 3 selected between two 64-bit addresses per lane: two 64-bit adds, two selects)
H263_DEV const uint8_t *coeff_row_address(const ReconArgs &a, const WavePos &p, uint32_t d0, bool wanted, int r)
{
    const bool wants = wanted && d0 != NO_COEFFS;
    uint32_t want = wants ? 0xffffffffu : 0u;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(want));                  // (a mask in a register: ANDs below, not selects)
#endif
    // The base is the picture's first coefficient block when SOME lane of the wave reads a block -- then that block exists,
    // and the lanes without one read its first row -- and the records otherwise (a picture, or a pool, without any coded
    // block has no address of its own that is safe to touch).
#if defined(__HIP_DEVICE_COMPILE__)
    const bool some = __ballot(wants) != 0;
#else
    const bool some = wants;                        // (the CPU checker runs the lanes one by one)
#endif
    const uint8_t *pic = some ? reinterpret_cast<const uint8_t *>(a.coeffs) + p.cbase * 128u
                              : reinterpret_cast<const uint8_t *>(a.mbs);                   // uniform
#if defined(H263MI_TIMING_COEF_WRAP)
    // TIMING EXPERIMENT ONLY (results wrong): every coefficient block comes out of one 64 KB region (cache resident)
    return reinterpret_cast<const uint8_t *>(a.coeffs) + (((d0 & 0xff80u) + (uint32_t)r * 16u) & want);
#endif
    return pic + ((d0 + (uint32_t)r * 16u) & want);
}

// The loads are issued unconditionally and in a fixed order -- coefficient row first, then the 5 + 3 reference rows --
// so that the wait in front of the first row pass can leave the eight reference loads in flight (s_waitcnt vmcnt(8)):
// the first IDCT round overlaps the wave's own motion compensation reads.  A lane whose macroblock takes no prediction
// reads offset 0 (its bytes are dropped in the predict phase); lanes whose taps leave the picture are fixed up there
// too.  Every address is a wave-uniform frame base plus a 32-bit lane offset.
// MC = false: no macroblock of the wave takes a prediction (every wave of an I picture): only the coefficient row is
// requested -- no reference rows, no addresses for them -- and the predict phase writes zeros.
template <bool MC = true>
H263_DEV void recon_phase_fetch(const ReconArgs &a, ReconWave &s, WaveFetch &f, int lane, const WavePos &p, const WaveMasks &km)
{
    const uint8_t *ref = a.ref + (size_t)p.pic * a.L.frame_bytes;
    {
        const int slot = lane >> 3, r = lane & 7;
        f.d0 = s.desc[slot][0];                                // (garbage beyond the active tasks: never used)
        f.d1 = s.desc[slot][1];
        if (a.events) {
            // sparse transport: where the events of the lane's block start and end (the two words are requested here, with
            // everything else the wave loads; the events themselves follow in the first round's load phase)
            const bool has = slot < recon_n_active(km) && f.d0 != NO_COEFFS;
            const uint32_t *fe = has ? a.block_first_event + (p.cbase + (f.d0 >> 7)) : reinterpret_cast<const uint32_t *>(a.mbs);
            f.coef0 = make_uint4(fe[0], fe[1], 0u, 0u);
        } else {
            f.coef0 = load16_stream(coeff_row_address(a, p, f.d0, slot < recon_n_active(km), r));
        }
    }
    if (!MC) {
        f.flags = f.mvw[0] = f.mvw[1] = 0;
        return;
    }
    // gather.rs:149: without a reference picture nothing is motion compensated (the error is already
    // in the status word)
    const uint32_t mc_mask = a.has_ref ? km.inter : 0u;
    f.flags = 0;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const PieceGeo g = piece_geometry(a, lane, k, p);
        const uint32_t mc = (mc_mask >> g.m) & 1u;
        uint32_t mvw = k == 0 ? s.rec[g.m][1 + g.blk] : s.mvc[g.m];     // (mvx, mvy) as one LDS word
        mvw &= 0u - mc;
#if defined(H263MI_TIMING_ZERO_MV)
        mvw = 0;                                                     // TIMING EXPERIMENT ONLY (results wrong): every vector (0, 0)
#elif defined(H263MI_TIMING_SMALL_MV)
        mvw &= 0x00030003u;                                          // TIMING EXPERIMENT ONLY: vectors 0..1.5 pixels, half-pel phases kept
#endif
        const int mvx = (int16_t)(mvw & 0xffffu), mvy = (int16_t)(mvw >> 16);
        // HalfPel::into_lerp_parameters (types.rs:721-729): floor(mv / 2), odd -> interpolate
        const int ix = mvx & 1, iy = mvy & 1;
        const int u = g.px + (mvx >> 1), v = g.py + (mvy >> 1);
        const bool inside = u >= 0 && u <= g.pw - 8 - ix;
        f.mvw[k] = mvw;
        f.flags |= (mc | ((mc && !inside) ? (uint32_t)SEG_BORDER : 0u)) << (2 * k);
        // inside lanes read at u (the 12-byte load may run past the row end: next row or padding); border lanes
        // read the window that holds all their clamped taps (border_window).  The loads themselves are dword aligned
        // (bytes (uc & ~3) .. +11 still cover the 9 taps); the predict phase shifts the window by uc & 3.
        const uint32_t uc = (uint32_t)(inside ? u : border_window(u, g.pw));
        const uint32_t ua = g.plane_off + (uc & ~3u);
        const int n_rows = k == 0 ? LUMA_ROWS : CHROMA_ROWS;
#pragma unroll
        for (int j = 0; j <= n_rows; j++) {
            // rows nobody needs (no prediction; the extra row of an integer vertical vector) read offset 0: one cache
            // line for the whole wave instead of one per lane
            const bool need = mc && (j < n_rows || iy);
            const uint32_t row = (uint32_t)med3i(v + j, 0, g.ph - 1);
            load12(ref + (need ? mad24(row, (uint32_t)g.pitch, ua) : 0u), k == 0 ? f.ly[j] : f.ch[j]);
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    // keep the per-lane state as three vector registers: without this the compiler carries the lane
    // predicates behind `flags` across the IDCT as scalar masks and spills other scalars around them
    asm volatile("" : "+v"(f.flags), "+v"(f.mvw[0]), "+v"(f.mvw[1]));
#endif
}

// ---- phase 4a: coefficient row of the lane, row pass -----------------------------------------
struct RowIn {
    uint32_t bad_events;       // sparse transport: 1 once a block of this lane had unusable event bounds (kept across the rounds;
                               // the caller zeroes it in front of the first round and reports it behind the last)
    uint32_t w[4];             // the 8 LEVELs of the lane's coefficient row (zeros when the block has no TCOEF)
    uint32_t d1;               // descriptor word 1 of the lane's block (quantiser, INTRADC level, intra, task)
    uint32_t wide;             // non-zero: some LEVEL the lane handled in this round lies outside [-512, 511] (see wide_bits_of)
    bool     active;           // the lane's slot holds a block in this round
};

// A LEVEL outside [-512, 511] -- the only kind that can overflow the reference's i16 product (dequant_pair_i16) -- has
// bits 15..9 that are not all equal.  Bit k of w ^ (w + w) is bit k ^ bit k - 1 of w: for a pair of LEVELs in a dword the
// bits 15..10 and 31..26 answer for the two halves (the bit the low half's doubling pushes into the high half lands on bit
// 16, outside the mask); for an event word (LEVEL in the high half) bits 31..26.  Two instructions of the 2-cycle class per
// word; the caller ORs the words of a round together and masks once.
constexpr uint32_t WIDE_MASK_PAIR = 0xfc00fc00u, WIDE_MASK_EVENT = 0xfc000000u;
H263_DEV uint32_t wide_bits_of(uint32_t w)
{
    uint32_t twice;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_add_u32 %0, %1, %1" : "=v"(twice) : "v"(w));       // (written as w << 1 it becomes the 4-cycle v_lshlrev)
#else
    twice = w + w;
#endif
    return w ^ twice;
}

// Sparse transport: the coefficient rows of a round straight from the EVENTS of its 8 blocks (one 32-bit word per
// non-zero LEVEL, the form the host parser emits and the PCIe link carries) -- what k_expand used to turn into a dense
// pool in HBM first.  The round's 8 x 64 LEVELs are rebuilt in LDS (1 KB, in the space the row pass writes its results
// to afterwards: every lane has read its row before any lane of the wave gets there): the buffer is zeroed, the 8 lanes
// of a block take the block's events eight at a time (lane r the events r, r + 8, ...: one coalesced 32-byte read per
// block and trip) and drop each LEVEL at its position, then every lane reads back its row.  The trip count is that of
// the longest list among the 8 blocks (uniform): one trip for the blocks of a P picture, eight for a dense block.
// The positions of a block's events must be distinct (every entry point that takes events from a caller checks it; the
// parser emits each position at most once).
// `stage`: -1 = the three steps one after the other (the device: the wave's lanes run in lock step and the fences order
// the steps); 0, 1, 2 = one step only (the CPU logic checker runs the lanes one after the other and therefore each step
// over all lanes before the next).
// `wide`: receives wide_bits_of over the events this lane placed (masked: non-zero = a LEVEL outside [-512, 511]) -- two
// instructions per event here instead of twelve per coefficient row behind it.
H263_DEV void coeff_rows_from_events(const ReconArgs &a, ReconWave &s, const WavePos &p, uint32_t d0, bool has, int lane,
                                     uint32_t w[4], uint32_t *bad_events, uint32_t *wide, int stage = -1,
                                     bool bounds_known = false, uint32_t known_first = 0, uint32_t known_next = 0)
{
    const int slot = lane >> 3, r = lane & 7;
    int16_t *dense = reinterpret_cast<int16_t *>(s.tbuf);       // [8 slots][64 positions]
    if (stage < 0 || stage == 0) *reinterpret_cast<uint4 *>(dense + lane * 8) = make_uint4(0, 0, 0, 0);
    if (stage < 0 || stage == 1) {
        uint32_t at = 0, end = 0;
        uint32_t bad = 0, wd = 0;
        const bool told = a.n_events != 0xffffffffu;            // uniform
        if (has) {
            const uint32_t *fe = a.block_first_event + (p.cbase + (d0 >> 7));
            // (the first round's bounds were requested by the fetch phase)
            const uint32_t first = bounds_known ? known_first : fe[0], next = bounds_known ? known_next : fe[1];
            // The bounds come out of device memory nobody may have validated (h263mi_batch_decode_events).  When the caller
            // has said how many events there are (a.n_events != 0xffffffff; the host entry points always do), a pair that is
            // not ascending or that reaches beyond them reads NOTHING and rejects the picture; a caller that did not say
            // vouches for its arrays, and the wave spends nothing on them (checked and reported inside the round, the test
            // cost the 64-stream launch 3.5 %: profiles/README.md r04_i -- the verdict is collected in `bad_events` and
            // reported once per wave).
            if (told) bad = (first > next || next > a.n_events) ? 1u : 0u;
            const uint32_t count = bad ? 0u : next - first;
            // (a rejected block starts at 0 with no events: `first + r` would WRAP for a hostile `first` within 8 of 2^32 --
            // `at` small, `end` huge, and the loop below walks the whole array and beyond.  Round 6's GPU fuzzer found it as a
            // memory access fault; tests/sim replays it under AddressSanitizer.  A block that passes has first <= next <=
            // n_events <= 0xffffff00: nothing wraps.)
            const uint32_t start = bad ? 0u : first;
            at = start + (uint32_t)r;
            end = start + (count > 64u ? 64u : count);          // (a block has 64 positions)
        }
        if (told) *bad_events |= bad;
        wave_fence();                                           // zeroed before the first LEVEL lands
#if defined(__HIP_DEVICE_COMPILE__)
        // Blocks with more than 8 events (intra pictures, mostly) take all their trips at once: the (up to) eight reads are in
        // flight together instead of one memory round trip after the other -- a block of 64 events was eight of them.
        while (__ballot(at + 8u < end) != 0) {
            uint32_t ev[8];
#pragma unroll
            for (int j = 0; j < 8; j++) ev[j] = at + 8u * (uint32_t)j < end ? a.events[at + 8u * (uint32_t)j] : 0u;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (at + 8u * (uint32_t)j < end) dense[slot * 64 + (int)(ev[j] & 63u)] = (int16_t)(ev[j] >> 16);
                wd |= wide_bits_of(ev[j]);                      // (a word that was not read is 0)
            }
            at += 64u;
        }
        while (__ballot(at < end) != 0) {
#else
        while (at < end) {
#endif
            if (at < end) {
                const uint32_t ev = a.events[at];
                dense[slot * 64 + (int)(ev & 63u)] = (int16_t)(ev >> 16);
                wd |= wide_bits_of(ev);
            }
            at += 8u;
        }
        *wide = wd & WIDE_MASK_EVENT;
        wave_fence();                                           // all LEVELs of the round are in place
    }
    if (stage < 0 || stage == 2) {
        const uint4 row = *reinterpret_cast<const uint4 *>(dense + slot * 64 + r * 8);
        w[0] = row.x; w[1] = row.y; w[2] = row.z; w[3] = row.w;
        wave_fence();                                           // (the row pass overwrites this space: reads first)
    }
}

// (Requesting the NEXT round's coefficient row a round early -- so that it arrives under this round's arithmetic
// instead of being waited for at the top of its own round -- was measured and dropped: dense I pictures +0.9...+1.4 %,
// 65 instead of 62 vector registers; profiles/README.md r03_v.)
H263_DEV void recon_phase_idct_load(const ReconArgs &a, ReconWave &s, const WaveFetch &f, int lane, const WavePos &p,
                                    int round, RowIn &ri, const WaveMasks &km, int events_stage = -1)
{
    const int slot = lane >> 3, r = lane & 7;
    const int k = round * ROUND_BLOCKS + slot;
    ri.active = k < recon_n_active(km);
    uint32_t d0 = f.d0;
    uint4 raw = f.coef0;                                        // round 0 was loaded ahead of time by recon_phase_fetch
    ri.d1 = f.d1;
    if (round > 0) {                                            // uniform
        // (k < WAVE_TASKS: a round starts below the number of active tasks, which are at most WAVE_TASKS, a multiple
        // of ROUND_BLOCKS; the minimum is one instruction where `k % WAVE_TASKS` was eight)
        static_assert(WAVE_TASKS % ROUND_BLOCKS == 0, "the last round must end inside the descriptor array");
        const int kk = k < WAVE_TASKS - 1 ? k : WAVE_TASKS - 1;
        d0 = s.desc[kk][0];
        ri.d1 = s.desc[kk][1];
    }
    const bool has = ri.active && d0 != NO_COEFFS;
    if (a.events) {                                             // uniform
        coeff_rows_from_events(a, s, p, d0, has, lane, ri.w, &ri.bad_events, &ri.wide, events_stage, round == 0, raw.x, raw.y);
        return;
    }
    if (round > 0) {
        // 8 lanes x 16 B = one 128-B coefficient block (raster order: lane r holds row r)
        raw = load16_stream(coeff_row_address(a, p, d0, ri.active, r));
    }
    // (a lane without a block has loaded some other block's row: masked away with four ANDs -- cheaper than a ballot, a
    // branch and the register copies the all-lanes-have-a-block shortcut of round 3 came out as)
    uint32_t keep = has ? 0xffffffffu : 0u;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(keep));
#endif
    ri.w[0] = raw.x & keep; ri.w[1] = raw.y & keep; ri.w[2] = raw.z & keep; ri.w[3] = raw.w & keep;
    ri.wide = (wide_bits_of(ri.w[0]) | wide_bits_of(ri.w[1]) | wide_bits_of(ri.w[2]) | wide_bits_of(ri.w[3])) & WIDE_MASK_PAIR;
}

// lane's contribution to cols_from_mask
H263_DEV uint32_t rowin_word_mask(const RowIn &ri)
{
    return (ri.w[1] ? 2u : 0u) | (ri.w[2] ? 4u : 0u) | (ri.w[3] ? 8u : 0u);
}

// What a lane's coefficient row contributes to the classification of its block (rle.rs:138-149): a non-zero value
// with y > 0 breaks "horiz", one with x > 0 breaks "vert".  A LEVEL is non-zero exactly when its dequantised value
// is (|v| >= 3q - 1), so this is decided on the raw words -- before the row pass, which needs to know the class.
struct RowClass {
    bool any;                  // the row holds a non-zero coefficient
    bool beyond_first;         // ... in a column x > 0
};

H263_DEV RowClass recon_row_class(const RowIn &ri, int lane)
{
    RowClass rc = {false, false};
    if (!ri.active) return rc;
    const bool use_dc = desc_intra(ri.d1) && (lane & 7) == 0;   // intra block: the DC comes from INTRADC (rle.rs:117-121)
    rc.beyond_first = ((ri.w[0] >> 16) | ri.w[1] | ri.w[2] | ri.w[3]) != 0;
    rc.any = rc.beyond_first || (use_dc ? desc_level(ri.d1) != 0 : (ri.w[0] & 0xffffu) != 0);
    return rc;
}

// cols_any: bit slot*8 + r set when coefficient row r of the slot's block holds a non-zero value in a column x > 0
// (the ballot of RowClass::beyond_first on the device).  A block none of whose rows does is Vert, Dc or Zero
// (rle.rs:151-171): the reference does not run its first column through the row pass at all -- it transforms that
// column directly (idct.rs:152-169) or uses the DC as it stands (idct.rs:119).  Here such a block runs the row
// pass with 1.0 in the place of B[0][i]: T[r][i] = C[r][0] exactly, for every i, so that the column pass finds the
// untouched first column in whichever column of T it reads.
// DENSE (round 4): the round is known to be "all Full" -- eight blocks, every coefficient row of every one of them with a
// non-zero LEVEL in its last pair (kernels.hip: recon_round_rows decides it with one ballot).  Nothing of what the general
// form spends on being general is left: no activity mask, no class, no column count -- dequantise 4 pairs, 8 terms, store.
// The arithmetic is the general form's with n_cols = 8 and no first-column-only block: bit for bit the same results.
// WIDE (round 5): some LEVEL of the round lies outside [-512, 511] (RowIn::wide): the dequantiser is the wrapping
// one (dequant_pair_wrap: the reference's i16 arithmetic as a release build executes it), the coefficients are not scaled
// and the table is the basis itself.  Where nothing overflows both forms give the same bits (powers of two commute with
// every rounding on the way), so WHICH rounds take this form is a matter of speed only.
template <bool DENSE = false, bool WIDE = false>
H263_DEV void recon_phase_idct_rows(ReconWave &s, const RowIn &ri, int lane, int n_cols, uint64_t cols_any)
{
    static_assert(!(DENSE && WIDE), "a wide round takes the general form");
    if (!DENSE && !ri.active) return;
    if (DENSE) n_cols = 8;
    const int slot = lane >> 3, r = lane & 7;
    const uint32_t quant = desc_quant(ri.d1);
    // 2q and q - (q even) = (q - 1) | 1 in both halves of a dword
    const uint32_t two_q2 = (2u * quant) * 0x00010001u, qmp2 = ((quant - 1u) | 1u) * 0x00010001u;
    const bool use_dc = desc_intra(ri.d1) && r == 0;
    const bool first_column_only = !DENSE && ((uint32_t)(cols_any >> (8 * slot)) & 0xffu) == 0;

    float C[8];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        C[2 * j] = C[2 * j + 1] = 0.0f;
        if (2 * j < n_cols) {                               // uniform
            const uint32_t v = WIDE ? dequant_pair_wrap(ri.w[j], two_q2, qmp2) : dequant_pair_i16(ri.w[j], two_q2, qmp2);
            C[2 * j] = (float)(int)(int16_t)(v & 0xffffu);
            C[2 * j + 1] = (float)((int)v >> 16);
        }
    }
    if (use_dc) C[0] = (float)(int)(desc_level(ri.d1) * (WIDE ? 1u : (uint32_t)DEQUANT_SCALE));     // (at most 2032 * 16)

    // idct_1d over the coefficient row (idct.rs:52-65); C holds 16 x the coefficients, the table 1/16 of the basis
    // (WIDE: the coefficients and the basis as they are)
    f32x2 T[4];
    if (WIDE) idct_1d_pairs(basis_table(), C, T, n_cols, first_column_only, 1.0f);
    else idct_1d_pairs(basis_table_sixteenth(), C, T, n_cols, first_column_only, 1.0f / DEQUANT_SCALE);
    float4_store(&s.tbuf[slot * TBUF_STRIDE + tbuf_row_offset(r)], T[0], T[1]);
    float4_store(&s.tbuf[slot * TBUF_STRIDE + tbuf_row_offset(r) + 4], T[2], T[3]);
}

// ---- phase 4b: column pass, rounding, residual into the strip -----------------------------
// rows_any / cols_any: bit slot*8 + r set when coefficient row r of the slot's block holds a non-zero value /
// one in a column x > 0 (ballots of RowClass on the device).  A block is Horiz when no row r > 0 holds anything,
// Vert when no row holds anything beyond column 0, Dc (or Zero) when both (rle.rs:138-171).
// strip_is_zero: no macroblock of the wave takes a prediction (an all-intra wave): the strip holds zeros wherever a
// block of the wave is about to be written, so the residual is the pixel and the read + add of the strip are left out.
template <bool DENSE = false>
H263_DEV void recon_phase_idct_cols(ReconWave &s, const RowIn &ri, int lane, int n_rows, uint64_t rows_any, uint64_t cols_any,
                                    bool any_special, bool strip_is_zero = false)
{
    if (!DENSE && !ri.active) return;
    if (DENSE) { n_rows = 8; any_special = false; }
    const int slot = lane >> 3, i = lane & 7;
    const uint32_t slot_rows = (uint32_t)(rows_any >> (8 * slot)) & 0xfeu, slot_cols = (uint32_t)(cols_any >> (8 * slot)) & 0xffu;
    const bool is_horiz = slot_rows == 0, is_vert = slot_cols == 0;

    // Every class reads column i of the row-pass result (the transposition of idct.rs:171-177); for the Vert class
    // (rle.rs:162-171, idct.rs:152-169) and the Dc class that is the block's first coefficient column itself (see
    // recon_phase_idct_rows).
    const bool vert = is_vert && !is_horiz;
    const bool dc_class = is_horiz && is_vert;                     // Dc or Zero, rle.rs:151-160
    const float *src = &s.tbuf[slot * TBUF_STRIDE + i];
    float col[8];
#pragma unroll
    for (int r = 0; r < 8; r++) col[r] = r < n_rows ? src[tbuf_row_offset(r)] : 0.0f;   // uniform: rows >= n_rows are zero

    // O = a QUARTER of the second pass's result (basis table times 0.25, see kBasisQuarter): what idct.rs:189 rounds
    f32x2 O[4];
    const BasisPtr B4 = basis_table_quarter();
    idct_1d_pairs(B4, col, O, n_rows);
    if (any_special) {                                             // uniform: some block of the round is Vert, Dc or Zero
        // class fix-ups as one multiply and one add (both exact where they must not change the value):
        //   Vert: x * B[0][0] (idct.rs:160)          others: x * 1.0
        //   Dc  : x * 0 + dc * 0.5 (idct.rs:119: exactly 0.5, not B00*B00; dc = 0 gives the Zero class)
        // on quarters: (x / 4) * B00 = (x * B00) / 4 and dc * 0.125 = (dc * 0.5) / 4, exactly
        const float c00 = col[0];
        const f32x2 scale = splat2(dc_class ? 0.0f : (vert ? B4[0][0] * 4.0f : 1.0f));
        const f32x2 shift = splat2(dc_class ? c00 * 0.125f : 0.0f);
#pragma unroll
        for (int jp = 0; jp < 4; jp++) O[jp] = O[jp] * scale + shift;
    }
    // The lane's column of the block in the strip already holds the prediction (or zeros): add the residual in place.
    //   r = ((v / 4.0 + signum(v) * 0.5) as i16).clamp(-256, 255)   idct.rs:189-190: signum(+-0) only decides the sign
    //       of a half that truncation removes, so copysign is enough; `as i16` truncates toward zero
    //   pixel = (r + prediction).clamp(0, 255)                      idct.rs:127-130, 191-194
    // The clamp of r to [-256, 255] is implied by the final one (the prediction is 0..255: any r >= 255 ends at 255,
    // any r <= -255 at 0), and |v| stays far below 2^31, so the conversion cannot saturate.
    uint8_t *base = &s.pix[desc_pix_origin(ri.d1) + i];
#pragma unroll
    for (int jp = 0; jp < 4; jp++) {
        const f32x2 q4 = O[jp];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            // (strip_is_zero: the pixel is clamp(r, 0, 255).  A negative quarter gives r <= 0 with either sign of the
            // half -- trunc(q - 0.5) <= 0 and trunc(q + 0.5) <= 0 for q < 0 -- so the half is added as it is)
            const float v = q4[h] + (strip_is_zero ? 0.5f : __builtin_copysignf(0.5f, q4[h]));
            uint8_t *px = base + (jp * 2 + h) * PIX_STRIDE;
            *px = (uint8_t)med3i(strip_is_zero ? (int)v : (int)v + (int)*px, 0, 255);
        }
    }
}

// whether a lane's block needs the class fix-ups of the column pass (the caller ORs this over the wave)
H263_DEV bool recon_block_is_special(const RowIn &ri, int lane, uint64_t rows_any, uint64_t cols_any)
{
    const int slot = lane >> 3;
    return ri.active && ((((uint32_t)(rows_any >> (8 * slot)) & 0xfeu) == 0) || (((uint32_t)(cols_any >> (8 * slot)) & 0xffu) == 0));
}

// ---- phase 5: prediction -> strip -----------------------------------------------------------
// What a reference row contributes to the half-pel filter, per 4 output pixels: with a = the taps at u.., s = the taps
// at u + (mvx & 1)..:  H = (a + s) >> 1 per byte, X = a ^ s (its low bit is the bit the floor average dropped).
struct RowTerms {
    uint32_t h[2], x[2];       // [0] pixels 0..3, [1] pixels 4..7
};
H263_DEV RowTerms row_terms(const uint32_t r[3], uint32_t sh, uint32_t ix)
{
    const uint32_t a0 = alignbyte(r[1], r[0], sh), a1 = alignbyte(r[2], r[1], sh), a2 = alignbyte(0u, r[2], sh);
    const uint32_t s0 = alignbyte(a1, a0, ix), s1 = alignbyte(a2, a1, ix);
    RowTerms t;
    t.h[0] = lerp_u8x4(a0, s0, 0u); t.x[0] = a0 ^ s0;
    t.h[1] = lerp_u8x4(a1, s1, 0u); t.x[1] = a1 ^ s1;
    return t;
}
// One form for the four half-pel cases of gather.rs:84-132.  The pixel is (a + s + b + t + 2) >> 2 with a, s from row
// v + j and b, t from row v + j + (mvy & 1) (repeated taps make it the two-tap average or the tap itself).  With
// a + s = 2 Ha + la and b + t = 2 Hb + lb that is (Ha + Hb + 1 + (la & lb)) >> 1: the rounding average of the two floor
// averages, plus one exactly where both dropped bits are set and Ha + Hb is even (an odd sum absorbs the extra one).
// `my` = all ones for an odd vertical vector (the lower row is row j + 1), 0 else (it is row j again).
H263_DEV uint32_t blend_rows(uint32_t ha, uint32_t xa, uint32_t hn, uint32_t xn, uint32_t my)
{
    const uint32_t dm = (ha ^ hn) & my;            // Ha ^ Hb
    const uint32_t hb = ha ^ dm;
    const uint32_t both = xa & (xn | ~my) & 0x01010101u;
    return lerp_u8x4(ha, hb, 0x01010101u) + (both ^ (both & dm));
}

template <int N_ROWS>
H263_DEV void predict_piece(ReconWave &s, uint32_t (*rows)[3], uint32_t mvw, uint32_t flags, const PieceGeo &g, bool all_integer,
                            bool all_inter)
{
    const int mvx = (int16_t)(mvw & 0xffffu);
    const uint32_t ix = (uint32_t)mvx & 1u, my = 0u - ((mvw >> 16) & 1u);
    const int u = g.px + (mvx >> 1);
    uint32_t sh = (uint32_t)u & 3u;          // the rows were loaded from the dword at or below u
    if (flags & SEG_BORDER) {
        // some tap lies outside the picture: rebuild the rows tap by tap from the loaded window
        const int ub = border_window(u, g.pw);
#pragma unroll
        for (int j = 0; j <= N_ROWS; j++) gather_row_clamped(rows[j], u, ub, g.pw);
        sh = 0;
    }
    const uint32_t keep = all_inter ? 0xffffffffu : 0u - (flags & (uint32_t)SEG_INTER);   // intra macroblocks start from zeros (gather.rs:136-138)
    uint8_t *dst = &s.pix[g.pixoff];
    if (all_integer) {                       // whole wave on integer vectors: the bytes are the prediction
#pragma unroll
        for (int j = 0; j < N_ROWS; j++) {
            const uint32_t a0 = alignbyte(rows[j][1], rows[j][0], sh), a1 = alignbyte(rows[j][2], rows[j][1], sh);
            *reinterpret_cast<uint64_t *>(dst + j * PIX_STRIDE) = (uint64_t)(a0 & keep) | ((uint64_t)(a1 & keep) << 32);
        }
        return;
    }
    RowTerms cur = row_terms(rows[0], sh, ix);
#pragma unroll
    for (int j = 0; j < N_ROWS; j++) {
        const RowTerms nxt = row_terms(rows[j + 1], sh, ix);
        const uint32_t lo = blend_rows(cur.h[0], cur.x[0], nxt.h[0], nxt.x[0], my);
        const uint32_t hi = blend_rows(cur.h[1], cur.x[1], nxt.h[1], nxt.x[1], my);
        *reinterpret_cast<uint64_t *>(dst + j * PIX_STRIDE) = (uint64_t)(lo & keep) | ((uint64_t)(hi & keep) << 32);
        cur = nxt;
    }
}

// MC = false: nothing is predicted anywhere in the wave: the strip starts from zeros (gather.rs:136-138).
template <bool MC = true>
H263_DEV void recon_phase_predict(const ReconArgs &a, ReconWave &s, WaveFetch &f, int lane, const WavePos &p, const WaveMasks &km)
{
    const PieceGeo gl = piece_geometry(a, lane, 0, p), gc = piece_geometry(a, lane, 1, p);
    if (!MC) {
#pragma unroll
        for (int j = 0; j < LUMA_ROWS; j++) *reinterpret_cast<uint64_t *>(&s.pix[gl.pixoff + j * PIX_STRIDE]) = 0ull;
#pragma unroll
        for (int j = 0; j < CHROMA_ROWS; j++) *reinterpret_cast<uint64_t *>(&s.pix[gc.pixoff + j * PIX_STRIDE]) = 0ull;
        return;
    }
    const bool all_inter = a.has_ref && km.inter == km.valid && km.valid == 0xffu;     // uniform: no lane needs zeros
#if defined(__HIP_DEVICE_COMPILE__)
    const bool luma_integer = __ballot(((f.mvw[0] | (f.mvw[0] >> 16)) & 1u) != 0) == 0;
    const bool chroma_integer = __ballot(((f.mvw[1] | (f.mvw[1] >> 16)) & 1u) != 0) == 0;
#else
    const bool luma_integer = false, chroma_integer = false;   // (the general form covers integer vectors: checked by the CPU suite)
#endif
    predict_piece<LUMA_ROWS>(s, f.ly, f.mvw[0], f.flags, gl, luma_integer, all_inter);
    predict_piece<CHROMA_ROWS>(s, f.ch, f.mvw[1], f.flags >> 2, gc, chroma_integer, all_inter);
}

// ---- the short cut: eight macroblocks that are not coded and do not move ---------------------------------------
// Most of a real P picture is COD = 1 (state.rs:207-216: Inter, zero vector, nothing coded) -- backgrounds, in runs.  A
// wave whose eight macroblocks are all like that (uniform: nothing goes through the IDCT, every macroblock inter, every
// vector zero) copies its 128x16 luma strip and the two 64x8 chroma strips from the reference frame to the new one, 16
// bytes per lane and access: three loads, three stores, no prediction arithmetic (the reference's analogue is the
// copy path of gather_block, gather.rs:63-79).
H263_HD bool recon_wave_is_static(const ReconArgs &a, const WaveMasks &km, bool any_moving)
{
    return a.has_ref && km.act == 0 && km.valid == 0xffu && km.inter == 0xffu && !any_moving;
}

H263_DEV void recon_phase_copy(const ReconArgs &a, int lane, const WavePos &p)
{
    const uint8_t *ref = a.ref + (size_t)p.pic * a.L.frame_bytes;
    uint8_t *cur = a.cur + (size_t)p.pic * a.L.frame_bytes;
    const uint32_t ly = mad24((uint32_t)(p.mby * 16 + (lane >> 3)), a.L.pitch_y, (uint32_t)(p.mbx0 * 16 + (lane & 7) * 16));
    const uint32_t ly2 = ly + 8u * a.L.pitch_y;
    const uint32_t lc = (lane >> 5 ? a.L.off_cr : a.L.off_cb) +
                        mad24((uint32_t)(p.mby * 8 + ((lane >> 2) & 7)), a.L.pitch_c, (uint32_t)(p.mbx0 * 8 + (lane & 3) * 16));
    const uint4 v0 = *reinterpret_cast<const uint4 *>(ref + ly), v1 = *reinterpret_cast<const uint4 *>(ref + ly2),
                v2 = *reinterpret_cast<const uint4 *>(ref + lc);
    *reinterpret_cast<uint4 *>(cur + ly) = v0;
    *reinterpret_cast<uint4 *>(cur + ly2) = v1;
    *reinterpret_cast<uint4 *>(cur + lc) = v2;
}

// ---- phase 6: strip -> frame ---------------------------------------------------------------
H263_DEV void recon_phase_store(const ReconArgs &a, ReconWave &s, int lane, const WavePos &p, const WaveMasks &km)
{
    uint8_t *cur = a.cur + (size_t)p.pic * a.L.frame_bytes;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const PieceGeo g = piece_geometry(a, lane, k, p);
        if (!((km.valid >> g.m) & 1)) continue;
        const uint32_t off = g.plane_off + mad24((uint32_t)g.py, (uint32_t)g.pitch, (uint32_t)g.px);
        const int n_rows = k == 0 ? LUMA_ROWS : CHROMA_ROWS;
#pragma unroll
        for (int j = 0; j < n_rows; j++)
            *reinterpret_cast<uint64_t *>(cur + (off + (uint32_t)(j * g.pitch))) =
                *reinterpret_cast<const uint64_t *>(&s.pix[g.pixoff + j * PIX_STRIDE]);
    }
}

}  // namespace h263mi
