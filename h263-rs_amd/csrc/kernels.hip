// kernels.hip -- __global__ wrappers and launchers of the gfx950 kernels.
// Built with: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no fast-math).
#include "kernels.h"

#include "post_kernel.inl"
#include "recon_kernel.inl"
#include "synth.inl"

namespace h263mi {

// Issue priority (s_setprio, 0..3) of the post-processing waves of k_frame; the reconstruction waves stay at 0.  A
// post wave is a chain of loads and 16-byte stores with little arithmetic in between; let it issue ahead of the
// reconstruction waves it shares a SIMD with and its memory operations are under way while those fill the vector
// ALUs.  A/B on two boxes (tools/ab.sh, 2 and 3 rounds): 0 -> 3 is -1.3 % / -1.5 % per frame index, dense I pictures
// -1 % / +-0; priority 1 -1.1 %.  Raising the reconstruction waves instead (until their loads are in flight, or in
// the output phase) gains nothing and costs dense I pictures 3 % (profiles/README.md).
#ifndef H263MI_PRIO_POST
#define H263MI_PRIO_POST 3
#endif

// n / d for a wave-uniform n with n * d < 2^32, r = ceil(2^32 / d) made by the launcher (d = 1 has no such r)
__device__ __forceinline__ uint32_t div_tiles_x(uint32_t n, uint32_t d, uint32_t r) { return d == 1 ? n : __umulhi(n, r); }

#if defined(H263MI_PROFILE_PHASES)
// diagnosis build: wall-clock cycles (s_memtime) a wave spends in each phase of k_recon, summed over waves
__device__ unsigned long long g_phase_cycles[8];
#define PHASE_MARK(i)                                                                  \
    do {                                                                               \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                  \
        if (lane == 0) atomicAdd(&g_phase_cycles[i], now_ - t_prev_);                  \
        t_prev_ = now_;                                                                \
    } while (0)
extern "C" int h263mi_debug_read_phases(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_cycles), sizeof(g_phase_cycles)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define PHASE_MARK(i) do { } while (0)
#endif
// H263MI_ISA_MARKERS: comments in the generated assembly at the phase boundaries, for tools/isa_mix.py (a static
// per-phase instruction-mix table of the disassembly); emits no instruction
#if defined(H263MI_ISA_MARKERS)
#define ISA_MARK(name) asm volatile("; ISA_MARK " name)
#else
#define ISA_MARK(name) do { } while (0)
#endif
// the same inside code that exists in two instantiations: the marker says which one
#define ISA_MARK2(first, prefix_a, prefix_b, name) do { if (first) ISA_MARK(prefix_a name); else ISA_MARK(prefix_b name); } while (0)
// COUNTING BUILDS ONLY (results wrong by construction; tools/phase_insts.sh): the waves of one kind stop behind a phase,
// so that the launch's instruction counters (SQ_INSTS_VALU ...) of a series of such builds, differenced, say what every
// phase EXECUTES -- the dynamic counterpart of tools/isa_mix.py's static table.  What a phase leaves behind is in LDS or
// named in an empty asm, so that the compiler keeps everything in front of the stop.
//   (9 = the waves of that kind return at once)
//   H263MI_STOP_RECON  1 records + mark + compact   2 ... + every load issued and arrived   3 ... + row pass of round 0
//                      4 ... + prediction into the strip   5 ... + all IDCT rounds (no store)
//   H263MI_STOP_POST   1 strips fetched and in LDS   2 ... + horizontal edges   3 ... + vertical edges (no conversion, no store)
#ifndef H263MI_STOP_RECON
#define H263MI_STOP_RECON 0
#endif
#ifndef H263MI_STOP_POST
#define H263MI_STOP_POST 0
#endif

// L2 prefetch through the SCALAR cache.  Every reconstruction wave starts with a load nothing can hide: its eight
// records, read once, always an HBM miss (a timing experiment with the records of all pictures folded into 64 KB ran
// 10 % faster: profiles/r03_e_timing_wrap.txt).  The XCD-aware work order makes the future predictable: this XCD starts
// the wave `q` (the caller names it: a few macroblock rows further down the same band, or the top of the band in the
// picture the XCD takes next) two or three microseconds from now.  Three scalar loads touch the 128-byte lines of THAT
// wave's records: scalar loads go SQC -> L2 (they occupy no slot of the CU's vector memory pipeline, which is what
// the launch is short of) and leave the lines in this XCD's L2, where the later wave's vector load finds them.  The
// values are never used; they are kept live until the wave's first LDS wait so that the registers are not reused while
// the loads are in flight.
// EXPERIMENT, compiled out (H263MI_EXP_COEF_PREFETCH): the same trick for the COEFFICIENTS (also read once, also
// always an HBM miss, the first thing the IDCT waits for: folded into 64 KB they ran another 10 % faster) -- but one
// scalar load per 128-byte block is up to 48 of them per wave, and the launch got SLOWER (P pictures +1.3 %, dense I
// pictures +12 %, in-process A/B profiles/r03_f_ab_inproc_prefetch.txt).  Where they are is written in the records of the wave that
// will read them, so it takes two steps: this wave prefetches the records of the wave 2D positions ahead (`far`), and
// reads -- with scalar loads, out of the L2 where the wave D positions back has put them -- the coefficient range in
// the records of the wave D positions ahead (`near`), whose 128-byte coefficient blocks it then touches one scalar
// load each.  Those go out right behind the wave's own vector loads and are only waited for where the wave would wait
// for its own HBM accesses anyway (scalar loads share the LDS counter: an earlier LDS wait would expose them).
struct PrefetchPlan { WavePos far, near; };
struct PrefetchTokens {
    uint32_t t[3];                       // far: the three lines of its records
    uint32_t c_first, c_last, w0_last;   // near: coeff_index of its first record, of its last record, and that record's word 0 (cbp)
    uint64_t cbase;                      // near: coeff_base of its picture
    bool near_ok;
};
typedef const __attribute__((address_space(4))) uint32_t *ScalarPtr32;
typedef const __attribute__((address_space(4))) uint64_t *ScalarPtr64;
__device__ __forceinline__ bool wave_exists(const ReconArgs &a, const WavePos &q)
{
    return q.pic >= 0 && q.pic < (int)a.n_pictures && q.mby >= 0 && q.mby < (int)a.L.mbh && q.mbx0 < (int)a.L.mbw;
}
__device__ __forceinline__ PrefetchTokens recon_prefetch(const ReconArgs &a, const PrefetchPlan &plan)
{
    PrefetchTokens k = {{0u, 0u, 0u}, 0u, 0u, 0u, 0ull, false};
#if !defined(H263MI_NO_PREFETCH)
    // (sparse records: where a later wave's records are is written in ITS index word -- two dependent accesses; not prefetched)
    if (!a.mb_group_index && wave_exists(a, plan.far)) {        // uniform
        const WavePos &q = plan.far;
        const int n = (int)a.L.mbw - q.mbx0;                    // records of that wave: min(n, 8)
        const ScalarPtr32 r = (ScalarPtr32)(uintptr_t)(a.mbs + (size_t)q.pic * a.mbs_per_picture + (size_t)q.mby * a.L.mbw + q.mbx0);
        const int last = (n < TILE_MBX ? n : TILE_MBX) * 8 - 1; // last dword of the wave's records
        k.t[0] = r[0];
        k.t[1] = r[last < 32 ? last : 32];
        k.t[2] = r[last];
    }
#if defined(H263MI_EXP_COEF_PREFETCH)       // measured: +1.3 % on P pictures, +12 % on dense I pictures (profiles/r03_f_*): off
    if (wave_exists(a, plan.near)) {
        const WavePos &q = plan.near;
        const int n = (int)a.L.mbw - q.mbx0;
        const ScalarPtr32 r = (ScalarPtr32)(uintptr_t)(a.mbs + (size_t)q.pic * a.mbs_per_picture + (size_t)q.mby * a.L.mbw + q.mbx0);
        const int last = (n < TILE_MBX ? n : TILE_MBX) * 8 - 1;
        k.c_first = r[7];
        k.w0_last = r[last - 7];
        k.c_last = r[last];
        k.cbase = a.coeff_base ? ((ScalarPtr64)(uintptr_t)a.coeff_base)[q.pic] : 0ull;
        k.near_ok = true;
    }
#endif
#endif
    return k;
}
// the coefficient blocks of the `near` wave: [first block, number of blocks) in a.coeffs; all of it from records nobody
// has validated yet, so the range is bounded before anything is touched
struct CoefRange { const uint8_t *base; uint32_t lines; };
__device__ __forceinline__ CoefRange recon_prefetch_retire(const ReconArgs &a, const PrefetchTokens &k)
{
    asm volatile("" :: "s"(k.t[0]), "s"(k.t[1]), "s"(k.t[2]));
    CoefRange cr = {nullptr, 0u};
    if (k.near_ok) {
        const uint32_t end = k.c_last + (uint32_t)__popc((k.w0_last >> 16) & 0x3fu);
        const uint32_t lines = end - k.c_first;                 // (wraps for nonsense records: caught by the bound)
        const uint64_t first = k.cbase + k.c_first;
        const bool inside = !a.coeff_checked || (first <= a.coeff_pool_blocks && lines <= a.coeff_pool_blocks - first);
        if (lines <= (uint32_t)WAVE_TASKS && end >= k.c_first && inside) {
            cr.base = reinterpret_cast<const uint8_t *>(a.coeffs) + first * 128u;
            cr.lines = lines;
        }
    }
    return cr;
}
// one scalar load per 128-byte coefficient block; the caller keeps the token alive until it has waited (lgkmcnt(0))
__device__ __forceinline__ uint32_t recon_prefetch_lines(const CoefRange &cr)
{
    uint32_t tok = 0;
    const uint8_t *p = cr.base;
    for (uint32_t i = 0; i < cr.lines; i++, p += 128)           // uniform: a scalar loop
        asm volatile("s_load_dword %0, %1, 0x0" : "=s"(tok) : "s"(p) : "memory");
    return tok;
}
__device__ __forceinline__ void recon_prefetch_lines_retire(uint32_t tok)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("" :: "s"(tok));
}

// ---------------------------------------------------------------------------------------
// k_recon: one wave = 8 whole macroblocks (recon_kernel.inl), no workgroup barrier, work taken in XCD-aware order
// (see k_post below).
// ---------------------------------------------------------------------------------------
// One IDCT round of a wave: 8 blocks, 8 lanes each.  The row pass and the column pass are separate calls because the
// first round runs its row pass BEFORE the prediction is written to the strip (under the reference loads) and its
// column pass after it.
struct RoundState {
    RowIn ri;
    uint64_t rows_any, cols_any;        // ballots of the row classes (recon_row_class)
    uint32_t rows_mask;                 // bit r: some block of the round has something in coefficient row r
    bool any_special;                   // some block of the round is Vert, Dc or Zero
    bool dense;                         // every row of every block of the round reaches its last pair: eight Full blocks
};

template <bool FIRST, bool MC>
__device__ __forceinline__ void recon_round_rows(const ReconArgs &a, ReconWave &s, const WaveFetch &f, int lane, const WavePos &p,
                                                 int round, const WaveMasks &km, RoundState &rs)
{
    int ln = lane;
    asm volatile("" : "+v"(ln));
    ISA_MARK2(MC, "mc_", "intra_", "round_begin");
    recon_phase_idct_load(a, s, f, ln, p, FIRST ? 0 : round, rs.ri, km);
    ISA_MARK2(MC, "mc_", "intra_", "idct_load_end");
    // The dense round -- every block of a dense I picture's waves (BASELINE configs[1]) -- is recognised by ONE ballot: a
    // non-zero LEVEL in columns 6..7 of every lane's coefficient row means eight active blocks (an empty slot holds zeros)
    // that all have something beyond their first column in every row: eight Full blocks, eight columns, eight rows
    // (rle.rs:138-149).  They take the instantiation without any of the general form's bookkeeping.
    // A LEVEL outside [-512, 511] anywhere in the round (one ballot): the reference's i16 dequantiser may have overflowed,
    // and the round takes the wrapping form of the row pass (recon_kernel.inl: dequant_pair_wrap) -- hostile or broken
    // streams only, but bit for bit what a release build of the reference makes of them.
    const bool wide = __ballot(rs.ri.wide != 0) != 0;
    rs.dense = !wide && __ballot(rs.ri.w[3] != 0) == ~0ull;
    if (rs.dense) {
        wave_fence();                       // the column pass of the previous round has read tbuf
        recon_phase_idct_rows<true>(s, rs.ri, ln, 8, ~0ull);
        wave_fence();                       // the row pass results are in LDS
        ISA_MARK2(MC, "mc_", "intra_", "idct_rows_end");
        return;
    }
    // which coefficient columns / rows the 8 blocks of this round use at all: the passes stop there
    const uint32_t wm = (__ballot(rs.ri.w[1] != 0) ? 2u : 0u) | (__ballot(rs.ri.w[2] != 0) ? 4u : 0u) |
                        (__ballot(rs.ri.w[3] != 0) ? 8u : 0u);
    const RowClass rc = recon_row_class(rs.ri, ln);
    rs.rows_any = __ballot(rc.any);
    rs.cols_any = __ballot(rc.beyond_first);                          // bit slot*8 + row
    wave_fence();                           // the column pass of the previous round has read tbuf
    if (wide) recon_phase_idct_rows<false, true>(s, rs.ri, ln, cols_from_mask(wm), rs.cols_any);
    else recon_phase_idct_rows(s, rs.ri, ln, cols_from_mask(wm), rs.cols_any);
    uint32_t rows_mask = (uint32_t)rs.rows_any | (uint32_t)(rs.rows_any >> 32);
    rows_mask |= rows_mask >> 16;
    rows_mask |= rows_mask >> 8;
    rs.rows_mask = rows_mask & 0xffu;
    rs.any_special = __ballot(recon_block_is_special(rs.ri, ln, rs.rows_any, rs.cols_any)) != 0;
    wave_fence();                           // the row pass results are in LDS
    ISA_MARK2(MC, "mc_", "intra_", "idct_rows_end");
}

template <bool MC>
__device__ __forceinline__ void recon_round_cols(ReconWave &s, int lane, const RoundState &rs)
{
    int ln = lane;
    asm volatile("" : "+v"(ln));
    if (rs.dense) recon_phase_idct_cols<true>(s, rs.ri, ln, 8, ~0ull, ~0ull, false, /*strip_is_zero=*/!MC);
    else recon_phase_idct_cols(s, rs.ri, ln, rows_from_mask(rs.rows_mask), rs.rows_any, rs.cols_any, rs.any_special, /*strip_is_zero=*/!MC);
    ISA_MARK2(MC, "mc_", "intra_", "idct_cols_end");
}

// fetch -> row pass of the first round -> prediction into the strip -> column pass -> remaining rounds -> store.
// MC: some macroblock of the wave takes a prediction.
template <bool MC>
__device__ __forceinline__ void recon_tail(const ReconArgs &a, ReconWave &s, int lane, const WavePos &p, const WaveMasks &km,
                                           const CoefRange &ahead_coefs, unsigned long long &t_prev_)
{
    (void)t_prev_;                                  // (only the diagnosis build H263MI_PROFILE_PHASES reads the clock)
    ISA_MARK2(MC, "mc_", "intra_", "tail_begin");
    int ln = lane;
    WaveFetch f;
    recon_phase_fetch<MC>(a, s, f, ln, p, km);      // every global load of this wave is in flight from here
    const uint32_t pf_tok = recon_prefetch_lines(ahead_coefs);      // ... and the L2 prefetch for a later wave behind them
    const int n_active = recon_n_active(km);
#if H263MI_STOP_RECON == 2
    {
        for (int j = 0; j <= LUMA_ROWS; j++) asm volatile("" :: "v"(f.ly[j][0]), "v"(f.ly[j][1]), "v"(f.ly[j][2]));
        for (int j = 0; j <= CHROMA_ROWS; j++) asm volatile("" :: "v"(f.ch[j][0]), "v"(f.ch[j][1]), "v"(f.ch[j][2]));
        asm volatile("" :: "v"(f.coef0.x), "v"(f.coef0.y), "v"(f.coef0.z), "v"(f.coef0.w), "v"(f.flags), "v"(f.mvw[0]), "v"(f.mvw[1]));
        recon_prefetch_lines_retire(pf_tok);
        return;
    }
#endif
    ISA_MARK2(MC, "mc_", "intra_", "fetch_end");
    PHASE_MARK(2);
    // The first round is peeled off the loop: its coefficient row was requested by the fetch phase, ahead of the
    // reference rows, and straight-line code is what lets the compiler wait for exactly that load
    // (s_waitcnt vmcnt(8)) and leave the eight reference loads in flight under the row pass.
    RoundState rs;
    rs.ri.bad_events = 0;
    if (n_active > 0) recon_round_rows<true, MC>(a, s, f, ln, p, 0, km, rs);
    PHASE_MARK(3);
#if H263MI_STOP_RECON == 3
    {
        for (int j = 0; j <= LUMA_ROWS; j++) asm volatile("" :: "v"(f.ly[j][0]), "v"(f.ly[j][1]), "v"(f.ly[j][2]));
        for (int j = 0; j <= CHROMA_ROWS; j++) asm volatile("" :: "v"(f.ch[j][0]), "v"(f.ch[j][1]), "v"(f.ch[j][2]));
        recon_prefetch_lines_retire(pf_tok);
        return;
    }
#endif
    asm volatile("" : "+v"(ln));
    ISA_MARK2(MC, "mc_", "intra_", "predict_begin");
    recon_phase_predict<MC>(a, s, f, ln, p, km);    // waits for the reference rows
    ISA_MARK2(MC, "mc_", "intra_", "predict_end");
    PHASE_MARK(4);
    wave_fence();                                   // the prediction is in the strip
    recon_prefetch_lines_retire(pf_tok);            // (the wave's own loads have arrived: so have these)
#if H263MI_STOP_RECON == 4
    return;
#endif
    if (n_active > 0) recon_round_cols<MC>(s, ln, rs);
#pragma unroll 1
    for (int round = 1; round * ROUND_BLOCKS < n_active; round++) {
        recon_round_rows<false, MC>(a, s, f, ln, p, round, km, rs);
        recon_round_cols<MC>(s, ln, rs);
    }
    PHASE_MARK(5);
#if H263MI_STOP_RECON == 5
    return;
#endif
    asm volatile("" : "+v"(ln));
    // event bounds that could not be used (only looked at when the caller said how many events there are): one report per wave
    if (a.events && a.n_events != 0xffffffffu) recon_report(a, ln, p.pic, false, __ballot(rs.ri.bad_events != 0) != 0);
    wave_fence();                                   // the strip is complete
    ISA_MARK2(MC, "mc_", "intra_", "store_begin");
    recon_phase_store(a, s, ln, p, km);
    ISA_MARK2(MC, "mc_", "intra_", "store_end");
}

// One wave's share of the reconstruction: the 8 macroblocks at `p`.
__device__ __forceinline__ void recon_wave(const ReconArgs &a0, ReconWave &s, int lane, WavePos p, const PrefetchPlan &plan,
                                           ScalarPtr32 kernarg_words)
{
    if (p.mby >= (int)a0.L.mbh) return;
#if H263MI_STOP_RECON == 9
    return;                                         // (counting build: the wave does nothing at all)
#endif
    // A batch whose streams have drifted apart (dev_common.h: STREAM_*) says per stream where its reference lives, whether
    // it has one, and whether it takes part in this call at all (uniform: one scalar load per wave).
    ReconArgs a = a0;
    if (a0.stream_state || a0.words_inline) {
        const uint32_t st = a0.words_inline ? kernarg_words[p.pic] : a0.stream_state[p.pic];      // (kernel arguments, or device memory)
        if (st & STREAM_RECON_SKIP) return;
        const uint32_t set = st & STREAM_REF_SET1;
        a.ref = a0.frame_set[set];
        a.cur = a0.frame_set[set ^ 1u];
        a.has_ref = (st & STREAM_HAS_REF) ? 1u : 0u;
    }
    p.cbase = a.coeff_base ? a.coeff_base[p.pic] : 0ull;      // uniform: a scalar load
    const PrefetchTokens pf = recon_prefetch(a, plan);

#if defined(H263MI_PROFILE_PHASES)
    unsigned long long t_prev_ = __builtin_amdgcn_s_memtime();
#else
    unsigned long long t_prev_ = 0;
#endif
    ISA_MARK("prologue_end");
    const uint32_t group_word = recon_group_word(a, p);       // (sparse records: which macroblocks of the wave have one)
    if (a.mb_group_index && (group_word & 0xffu) == 0 && a.has_ref && recon_valid_mask(a, p) == 0xffu) {
        // sparse records: none of the eight macroblocks has a record = none is coded: the copy path, without ever touching a
        // record (the dense form reads 256 bytes of records and runs the mark phase to find that out)
        recon_phase_copy(a, lane, p);
        return;
    }
    recon_phase_load(a, s, lane, p, group_word);
    ISA_MARK("load_end");
    PHASE_MARK(0);                                  // records requested and in LDS
    // `ln`: the lane index behind an opaque asm, re-derived per phase so that lane-only expressions are
    // recomputed where they are used instead of being kept in registers across the whole kernel
    int ln = lane;
    asm volatile("" : "+v"(ln));
    WaveMasks km;
    km.valid = recon_valid_mask(a, p);
    wave_fence();                                   // the records are in LDS
    const TaskInfo ti = recon_phase_mark(a, s, ln, p, km.valid, recon_block_limit(a, p));
    const uint64_t act64 = __ballot(ti.active != 0), inter64 = __ballot(ti.inter != 0);
    km.act = act64;                                 // bits 0..47
    km.inter = (uint32_t)(inter64 >> MB_LANE0) & 0xffu;
    recon_report(a, ln, p.pic, km.inter && !a.has_ref, __ballot(ti.bad_index != 0) != 0);
    recon_phase_compact(s, ln, ti, km.act);
    wave_fence();                                   // descriptors and chroma vectors are in LDS
    const CoefRange ahead_coefs = recon_prefetch_retire(a, pf);
    ISA_MARK("mark_end");
#if H263MI_STOP_RECON == 1
    return;
#endif
    if (recon_wave_is_static(a, km, __ballot(ti.moving != 0) != 0)) {     // eight macroblocks that are not coded and do not move
        recon_phase_copy(a, ln, p);
        return;
    }
    PHASE_MARK(1);
    // Two copies of the rest, chosen per wave (uniform): waves with a prediction to fetch, and waves without one --
    // every wave of an I picture -- which issue no reference loads, compute no addresses for them and skip the
    // interpolation.  Two copies rather than a switch inside one: the wait in front of the first IDCT round must
    // know how many loads were issued behind the coefficient row.
    if (a.has_ref && km.inter) recon_tail<true>(a, s, ln, p, km, ahead_coefs, t_prev_);
    else recon_tail<false>(a, s, ln, p, km, ahead_coefs, t_prev_);
    PHASE_MARK(6);
}

// The launch's StreamWords argument comes FIRST in every kernel below: it then sits at offset 0 of the kernarg segment, and a
// wave reads its picture's word from there with ONE SCALAR load (indexing the by-value argument itself became a per-lane
// flat load, and every pointer and strength selected from the word a vector value: +174 vector instructions in k_frame).
__device__ __forceinline__ ScalarPtr32 kernarg_stream_words()
{
    return (ScalarPtr32)__builtin_amdgcn_kernarg_segment_ptr();
}

__global__ __launch_bounds__(RECON_THREADS) void k_recon(StreamWords, ReconArgs a)
{
    __shared__ __attribute__((aligned(16))) ReconWave waves[RECON_WAVES];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));    // wave index: scalar
    // grid.y = picture; grid.x walks the picture's (tile, macroblock row) list, RECON_WAVES entries per workgroup, in
    // XCD-aware order (gridDim.x is a multiple of 8, so blockIdx.x & 7 names the XCD for every picture).
    // No integer division on the device: the only one left, by tiles_x, is a multiply-high with a host-made reciprocal.
    constexpr uint32_t kUnitsPerTile = TILE_WAVES / RECON_WAVES;
    const uint32_t upp = a.tiles_x * a.tiles_y * kUnitsPerTile;           // work items of one picture
    // a picture's list is dealt to a.bands XCDs in contiguous chunks, 8 / a.bands pictures side by side (see k_frame)
    // (bands is a power of two: shifts -- as divisions by a run-time value they were 60 scalar instructions per wave)
    const uint32_t bands = a.bands, bsh = (uint32_t)__builtin_ctz(bands), xcd = blockIdx.x & 7;
    const uint32_t chunk = (upp + bands - 1) >> bsh, t = blockIdx.x >> 3, g = (xcd & (bands - 1)) * chunk + t;
    const uint32_t pic = blockIdx.y * (8u >> bsh) + (xcd >> bsh);
    if (t >= chunk || g >= upp || pic >= a.n_pictures) return;
    const uint32_t tile = g / kUnitsPerTile;                               // power of two
    const int tw = (int)(g % kUnitsPerTile) * RECON_WAVES + wave;          // macroblock row of the tile
    const uint32_t tile_y = div_tiles_x(tile, a.tiles_x, a.inv_tiles_x), tile_x = tile - tile_y * a.tiles_x;
    WavePos p;
    p.pic = (int)pic;
    p.mbx0 = (int)tile_x * TILE_MBX;
    p.mby = (int)tile_y * TILE_MBY + tw;
    p.cbase = 0;
    PrefetchPlan plan = {p, p};
    plan.far.pic = plan.near.pic = -1;              // (no prefetch in the stand-alone kernel)
    recon_wave(a, waves[wave], lane, p, plan, kernarg_stream_words());
}

// Bands per picture: the XCDs that share one picture's work list (k_recon, k_frame).  Measured on the 64-stream bench
// with k_frame (rocprofv3 FETCH_SIZE, A/Bs of six rounds): 8 bands fetch 372 MB per launch, 4 bands 341 MB and run
// 1.5-3 % faster (fewer band borders, whose reference rows two L2s fetch), 2 bands 326 MB (the algorithmic reads are
// 317 MB) but run no faster than 8, 1 band (a picture per XCD) is 4 % slower.  Small batches keep 8 bands: with fewer,
// 8 / bands pictures are needed to occupy every XCD.
// Prefetch distance of k_frame's reconstruction waves, in groups of the work list (recon_prefetch): far enough for an HBM
// access to complete before the target wave starts, near enough for the lines to survive in the XCD's 4 MB of L2.
#ifndef H263MI_PF_GROUPS
#define H263MI_PF_GROUPS 3
#endif
#ifndef H263MI_FRAME_BANDS
#define H263MI_FRAME_BANDS 4
#endif
static uint32_t frame_bands(uint32_t n_pictures) { return n_pictures >= 16 ? H263MI_FRAME_BANDS : 8u; }

// ceil(2^32 / d): n / d == mul_hi(n, r) for n * d < 2^32
static uint32_t reciprocal_u32(uint32_t d) { return d <= 1 ? 0u : (uint32_t)((0x100000000ull + d - 1) / d); }

// the streams' words of a launch -> its StreamWords argument; false (and nothing set) when there are none or too many pictures
static bool inline_words(const uint32_t *words, uint32_t n_pictures, StreamWords &sw)
{
    if (!words || n_pictures > STREAM_WORDS_INLINE) return false;
    for (uint32_t i = 0; i < STREAM_WORDS_INLINE; i++) sw.w[i] = i < n_pictures ? words[i] : 0u;
    return true;
}

hipError_t launch_recon(const ReconArgs &args, hipStream_t stream, const uint32_t *words)
{
    if (!args.n_pictures) return hipSuccess;
    if (args.n_pictures > 65535 || args.tiles_x * args.tiles_y >= (1u << 20)) return hipErrorInvalidValue;
    if (words && args.n_pictures > STREAM_WORDS_INLINE) return hipErrorInvalidValue;      // (the caller's job: device words)
    ReconArgs a = args;
    StreamWords sw{};
    a.words_inline = inline_words(words, args.n_pictures, sw) ? 1u : 0u;
    a.inv_tiles_x = reciprocal_u32(args.tiles_x);
    const uint32_t upp = args.tiles_x * args.tiles_y * (TILE_WAVES / RECON_WAVES);
    a.bands = frame_bands(args.n_pictures);
    const uint32_t chunk = (upp + a.bands - 1) / a.bands, side_by_side = 8 / a.bands;
    hipLaunchKernelGGL(k_recon, dim3(chunk * 8, (args.n_pictures + side_by_side - 1) / side_by_side), dim3(RECON_THREADS), 0,
                       stream, sw, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// k_post: one wave per workgroup = one 128x32 tile; grid = 8 x (tiles per XCD).
//
// Work order is XCD-aware: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 names
// the group that shares an L2 -- a speed assumption only, never correctness), so XCD k walks the
// contiguous range [k*chunk, (k+1)*chunk) of the (picture, tile) list and its workgroups sit on
// neighbouring tiles at any moment: the cache lines that the 4-pixel tile offset makes two tiles
// share are then fetched once per L2 instead of once per XCD.
// ---------------------------------------------------------------------------------------
template <bool FETCH_AHEAD, bool STREAM_RGBA, bool INTERIOR>
__device__ __forceinline__ void post_strip(const PostArgs &a, PostStrip &s, PostFetch &pf, int lane, int sx, int sy, int pic)
{
    // `ln`: the lane index behind an opaque asm, re-derived per strip so that lane-only expressions (LDS
    // offsets, column indices, ...) are recomputed where used instead of being kept in registers across
    // the whole kernel -- the hoisted form cost half of the occupancy.
    int ln = lane;
    asm volatile("" : "+v"(ln));
    ISA_MARK2(INTERIOR, "interior_", "edge_", "strip_begin");
    wave_fence();                                   // the previous strip has been read out of LDS
    post_phase_commit<INTERIOR>(a, s, pf, ln);
    ISA_MARK2(INTERIOR, "interior_", "edge_", "commit_end");
    if (FETCH_AHEAD) post_phase_fetch<INTERIOR>(a, pf, ln, sx, sy + 2, pic);
    ISA_MARK2(INTERIOR, "interior_", "edge_", "fetch_end");
    wave_fence();                                   // the strip is in LDS
#if H263MI_STOP_POST == 1
    return;
#endif
    if (a.strength) {
        post_phase_hedges<INTERIOR>(a, s, ln, sx, sy);
        wave_fence();
        ISA_MARK2(INTERIOR, "interior_", "edge_", "hedges_end");
#if H263MI_STOP_POST == 2
        return;
#endif
        post_phase_vedges<INTERIOR>(a, s, ln, sx, sy);
        wave_fence();
        ISA_MARK2(INTERIOR, "interior_", "edge_", "vedges_end");
    }
#if H263MI_STOP_POST == 3
    return;
#endif
    post_phase_store<STREAM_RGBA, INTERIOR>(a, s, ln, sx, sy, pic);
    ISA_MARK2(INTERIOR, "interior_", "edge_", "store_end");
}

// Two strips of a tile.  FETCH_AHEAD: queue the loads of the strips two further down right after each
// commit, so that they are in flight while this pair is filtered and stored.
template <bool FETCH_AHEAD, bool STREAM_RGBA, bool INTERIOR>
__device__ __forceinline__ void post_strip_pair(const PostArgs &a, PostStrip &s, PostFetch &pf0, PostFetch &pf1, int lane,
                                                int sx, int sy, int pic)
{
    post_strip<FETCH_AHEAD, STREAM_RGBA, INTERIOR>(a, s, pf0, lane, sx, sy, pic);
    post_strip<FETCH_AHEAD, STREAM_RGBA, INTERIOR>(a, s, pf1, lane, sx, sy + 1, pic);
}

// One wave's share of the post-processing: the 128x32 tile (sx, ty) of picture `pic` = 4 strips.
// No workgroup barrier anywhere: the wave owns its strips from load to store.  Two strips are always in flight ahead
// of the one being filtered.  Strips past the bottom of the picture are processed like any other (clamped loads, no
// rows to store), which keeps the code straight-line.  INTERIOR (post_tile_is_interior, 82 % of the tiles of a 1080p
// picture): no bounds handling at all and a fixed number of vector memory operations per strip, so that every wait is
// an exact s_waitcnt vmcnt(N) that leaves the younger loads and stores in flight.
template <bool STREAM_RGBA, bool INTERIOR>
__device__ __forceinline__ void post_tile(const PostArgs &a, PostStrip &s, int lane, int sx, int ty, int pic)
{
    const int sy0 = ty * POST_STRIPS;
    ISA_MARK2(INTERIOR, "interior_", "edge_", "tile_begin");
    // (The edge-tile instantiation is 5 700 instructions, half of k_frame, for 18 % of the tiles.  As a loop over the
    // strips with one copy of the strip code k_frame shrinks from 12 100 to 7 700 lines of assembly and runs exactly as
    // fast, dense I pictures 1 % slower: the instruction cache is not what limits it; the unrolled form stays.)
    PostFetch pf0, pf1;
    post_phase_fetch<INTERIOR>(a, pf0, lane, sx, sy0, pic);
    post_phase_fetch<INTERIOR>(a, pf1, lane, sx, sy0 + 1, pic);
    post_strip_pair<true, STREAM_RGBA, INTERIOR>(a, s, pf0, pf1, lane, sx, sy0, pic);          // strips 0,1; queues the loads of 2,3
    post_strip_pair<false, STREAM_RGBA, INTERIOR>(a, s, pf0, pf1, lane, sx, sy0 + 2, pic);     // strips 2,3
}

template <bool STREAM_RGBA>
__device__ __forceinline__ void post_wave(const PostArgs &a0, PostStrip &s, int lane, int sx, int ty, int pic, ScalarPtr32 kernarg_words)
{
    if (ty >= (int)a0.tiles_y) return;
#if H263MI_STOP_POST == 9
    return;                                         // (counting build: the wave does nothing at all)
#endif
    PostArgs a = a0;
    if (a0.stream_state || a0.words_inline) {       // streams that differ (dev_common.h: STREAM_*): uniform
        const uint32_t st = a0.words_inline ? kernarg_words[pic] : a0.stream_state[pic];
        if (st & STREAM_POST_SKIP) return;
        a.frames = a0.frame_set[(st & STREAM_POST_SET1) ? 1 : 0];
        a.strength = (st >> STREAM_STRENGTH_SHIFT) & STREAM_STRENGTH_MASK;      // this picture's own (deblock.rs:5-8)
        // per-stream output buffers: the phases address picture `pic` at a.rgba + pic * w*h*4 -- hand them the base that
        // puts it at its own pointer (uniform: two scalar loads)
        if (a0.rgba_ptrs) a.rgba = a0.rgba_ptrs[pic] - (size_t)pic * a0.L.width * a0.L.height * 4u;
    }
    if (post_tile_is_interior(a, sx, ty)) post_tile<STREAM_RGBA, true>(a, s, lane, sx, ty, pic);       // wave-uniform
    else post_tile<STREAM_RGBA, false>(a, s, lane, sx, ty, pic);
}

__global__ __launch_bounds__(POST_THREADS) void k_post(StreamWords, PostArgs a)
{
    __shared__ __attribute__((aligned(16))) PostStrip strips[POST_WAVES];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));    // wave index: scalar
    // one wave = one 128x32 tile = 4 strips; a workgroup = 4 vertically adjacent tiles (a 128x128 block), and
    // workgroups follow each other along x in the XCD-ordered list.  Horizontal neighbours -- which share the
    // cache lines the 4-pixel tile offset straddles -- are thus in different workgroups, start at slightly
    // different times, and the second one finds the shared lines in L2 (4 lock-stepped waves of one
    // workgroup missing on the same line at the same moment fetched it twice: +50 % FETCH_SIZE).
    // grid.y = picture; grid.x walks the picture's list of (column sx, group of POST_GROUP vertical tiles, tile of the
    // group) in XCD-aware order; divisions by tiles_x are multiply-highs with a host-made reciprocal
    const uint32_t groups_y = (a.tiles_y + POST_GROUP - 1) / POST_GROUP;
    constexpr uint32_t kUnitsPerGroup = POST_GROUP / POST_WAVES;
    const uint32_t upp = a.tiles_x * groups_y * kUnitsPerGroup;
    const uint32_t chunk = (upp + 7) / 8, xcd = blockIdx.x & 7;
    const uint32_t t = blockIdx.x >> 3, unit = xcd * chunk + t;
    if (t >= chunk || unit >= upp) return;
    const uint32_t wg = unit / kUnitsPerGroup;
    const int gw = (int)(unit % kUnitsPerGroup) * POST_WAVES + wave;       // tile of the group, 0..3
    const int pic = (int)blockIdx.y;
    const uint32_t gy = div_tiles_x(wg, a.tiles_x, a.inv_tiles_x);
    const int sx = (int)(wg - gy * a.tiles_x + a.wrap), ty = (int)gy * POST_GROUP + gw;      // (wrap: the first tile column is 1)
    post_wave<false>(a, strips[wave], lane, sx, ty, pic, kernarg_stream_words());
}

// ---------------------------------------------------------------------------------------
// k_frame: reconstruction of picture f and post-processing of picture f - 1 of every stream in ONE launch
// (the frame-pipelined form of h263mi_batch_decode).  Both halves read the same frame set -- k_recon as its
// reference picture, k_post as the picture to filter and convert -- and neither writes it, so they need no ordering
// between them.  The work list of a picture is cut into GROUPS of 32 luma rows: the 2 * rtx reconstruction
// waves of two macroblock rows (8 macroblocks each), then the ptx post tiles over (almost) the same rows; the list is dealt to the 8
// XCDs in contiguous chunks as in the two kernels above.  Waves that read the same rows of the same picture thus run
// at the same time on the same XCD: the planes are fetched from HBM once per frame instead of twice, and waves
// bound by arithmetic and address work (reconstruction) share every CU with waves bound by stores (RGBA).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_frame(StreamWords, ReconArgs ra, PostArgs pa, FrameGeom fg)
{
#if defined(H263MI_LDS_PAD)
    // experiment: what a larger LDS footprint per wave (fewer resident waves) costs
    __shared__ __attribute__((aligned(16))) union { ReconWave r; PostStrip p; uint8_t pad[H263MI_LDS_PAD]; } lds;
#else
    __shared__ __attribute__((aligned(16))) union { ReconWave r; PostStrip p; } lds;
#endif
    const int lane = threadIdx.x & 63;
    const uint32_t per_group = fg.recon_per_group + fg.post_per_group;
    const uint32_t upp = fg.groups * per_group;
    // A picture's work list is dealt to `bands` XCDs in contiguous chunks, 8 / bands pictures side by side (fg.bands:
    // 8, 4 or 2; blockIdx.y counts sets of 8 / bands pictures).  Fewer, taller bands: fewer band borders, whose
    // reference rows two L2s fetch.
    // (bands is a power of two: shifts -- as divisions by a run-time value they were 60 scalar instructions per wave)
    const uint32_t bands = fg.bands, bsh = (uint32_t)__builtin_ctz(bands), xcd = blockIdx.x & 7;
    const uint32_t chunk = (upp + bands - 1) >> bsh, band = xcd & (bands - 1), side = xcd >> bsh;
    const uint32_t t = blockIdx.x >> 3, g = band * chunk + t;
    const uint32_t pic_y = blockIdx.y * (8u >> bsh) + side;
    if (t >= chunk || g >= upp || pic_y >= ra.n_pictures) return;
    const uint32_t group = div_tiles_x(g, per_group, fg.inv_per_group);
    const uint32_t r = g - group * per_group;      // (post tiles in front of the reconstruction waves, or the two kinds
                                                   // alternating 2 : 1 through the group: no difference, profiles/README.md r04_a)
    const int pic = fg.flip ? (int)(ra.n_pictures - 1 - pic_y) : (int)pic_y;
    if (r < fg.recon_per_group) {
        WavePos p;
        p.pic = pic;
        p.mbx0 = (int)(r >> 1) * TILE_MBX;
        p.mby = (int)group * TILE_MBY + (int)(r & 1);
        p.cbase = 0;
        // Which waves to prefetch for (recon_prefetch): `near` = the wave H263MI_PF_GROUPS groups further down this band's
        // list (same tile column, same macroblock row parity), `far` = twice as far; past the end of the band, the item
        // that far into the band of the picture this XCD takes next, when that item is a reconstruction wave.
        PrefetchPlan plan = {p, p};
#pragma unroll
        for (int k = 0; k < 2; k++) {
            WavePos &q = k ? plan.far : plan.near;
            const uint32_t dist = (k ? 2u : 1u) * H263MI_PF_GROUPS;
            const uint32_t g2 = g + dist * per_group, band_end = (band + 1) * chunk < upp ? (band + 1) * chunk : upp;
            if (g2 < band_end) {
                q.mby = p.mby + (int)dist * TILE_MBY;
            } else {
                const uint32_t g3 = g2 - band_end + band * chunk;          // counted from the start of the band
                const uint32_t group3 = div_tiles_x(g3, per_group, fg.inv_per_group), r3 = g3 - group3 * per_group;
                const int pic3 = ((int)blockIdx.y + 1) * (int)(8u >> bsh) + (int)side;
                q.pic = r3 < fg.recon_per_group && pic3 < (int)ra.n_pictures
                            ? (fg.flip ? (int)ra.n_pictures - 1 - pic3 : pic3) : -1;
                q.mbx0 = (int)(r3 >> 1) * TILE_MBX;
                q.mby = (int)group3 * TILE_MBY + (int)(r3 & 1);
            }
        }
        recon_wave(ra, lds.r, lane, p, plan, kernarg_stream_words());
    } else {
        if (H263MI_PRIO_POST) __builtin_amdgcn_s_setprio(H263MI_PRIO_POST);
#if defined(H263MI_EXP_PLAIN_RGBA)
        post_wave<false>(pa, lds.p, lane, (int)(r - fg.recon_per_group + pa.wrap), (int)group, pic, kernarg_stream_words());
#else
        post_wave<true>(pa, lds.p, lane, (int)(r - fg.recon_per_group + pa.wrap), (int)group, pic, kernarg_stream_words());
#endif
    }
}

hipError_t launch_frame(const ReconArgs &rargs, const PostArgs &pargs, hipStream_t stream, bool descending, const uint32_t *words)
{
    if (!rargs.n_pictures) return hipSuccess;
    if (rargs.n_pictures != pargs.n_pictures || rargs.n_pictures > 65535) return hipErrorInvalidValue;
    if (words && rargs.n_pictures > STREAM_WORDS_INLINE) return hipErrorInvalidValue;
    static_assert(RECON_WAVES == 1 && TILE_WAVES == 2 && POST_WAVES == 1, "k_frame is written for single-wave workgroups");
    FrameGeom fg;
    fg.recon_per_group = rargs.tiles_x * TILE_WAVES;
    fg.post_per_group = pargs.tiles_x;
    fg.groups = rargs.tiles_y > pargs.tiles_y ? rargs.tiles_y : pargs.tiles_y;
    const uint32_t per_group = fg.recon_per_group + fg.post_per_group;
    if ((uint64_t)fg.groups * per_group >= (1u << 24)) return hipErrorInvalidValue;
    fg.inv_per_group = reciprocal_u32(per_group);
    fg.flip = descending ? 1u : 0u;
    fg.bands = frame_bands(rargs.n_pictures);
    ReconArgs ra = rargs;
    PostArgs pa = pargs;
    ra.inv_tiles_x = reciprocal_u32(rargs.tiles_x);
    pa.inv_tiles_x = reciprocal_u32(pargs.tiles_x);
    StreamWords sw{};
    ra.words_inline = pa.words_inline = inline_words(words, rargs.n_pictures, sw) ? 1u : 0u;
    const uint32_t chunk = (fg.groups * per_group + fg.bands - 1) / fg.bands, side_by_side = 8 / fg.bands;
    hipLaunchKernelGGL(k_frame, dim3(chunk * 8, (rargs.n_pictures + side_by_side - 1) / side_by_side), dim3(64), 0, stream, sw,
                       ra, pa, fg);
    return hipGetLastError();
}

hipError_t launch_post(const PostArgs &args, hipStream_t stream, const uint32_t *words)
{
    if (!args.n_pictures) return hipSuccess;
    if (args.n_pictures > 65535 || args.tiles_x * args.tiles_y >= (1u << 20)) return hipErrorInvalidValue;
    if (words && args.n_pictures > STREAM_WORDS_INLINE) return hipErrorInvalidValue;
    PostArgs a = args;
    StreamWords sw{};
    a.words_inline = inline_words(words, args.n_pictures, sw) ? 1u : 0u;
    a.inv_tiles_x = reciprocal_u32(args.tiles_x);
    const uint32_t groups_y = (args.tiles_y + POST_GROUP - 1) / POST_GROUP;
    const uint32_t upp = args.tiles_x * groups_y * (POST_GROUP / POST_WAVES), chunk = (upp + 7) / 8;
    hipLaunchKernelGGL(k_post, dim3(chunk * 8, args.n_pictures), dim3(POST_THREADS), 0, stream, sw, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// synthetic record generators (bench / test support)
// ---------------------------------------------------------------------------------------
__global__ void k_synth_headers(SynthArgs a)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_streams * a.mbs_per_picture) return;
    const uint32_t p = g / a.mbs_per_picture, mb = g % a.mbs_per_picture;
    MbRecord r = synth_mb_header(a.kind, a.first_stream_id + p * a.stream_stride, a.frame_idx, mb);
    a.mbs[g] = r;
    a.counts[g] = (uint32_t)__popc(r.cbp);
}

// exclusive scan of the per-macroblock coded-block counts of one picture -> coeff_index
__global__ __launch_bounds__(1024) void k_synth_scan(SynthArgs a)
{
    __shared__ uint32_t sums[1024];
    const uint32_t p = blockIdx.x, tid = threadIdx.x;
    const uint32_t n = a.mbs_per_picture, chunk = (n + 1023) / 1024;
    const uint32_t lo = tid * chunk, hi = lo + chunk < n ? lo + chunk : n;
    uint32_t local = 0;
    for (uint32_t i = lo; i < hi; i++) local += a.counts[p * n + i];
    sums[tid] = local;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t v = tid >= off ? sums[tid - off] : 0;
        __syncthreads();
        sums[tid] += v;
        __syncthreads();
    }
    uint32_t run = sums[tid] - local;     // exclusive prefix of this thread's chunk
    for (uint32_t i = lo; i < hi; i++) {
        a.mbs[p * n + i].coeff_index = run;
        run += a.counts[p * n + i];
    }
    if (tid == 1023) a.totals[p] = sums[1023];
}

__global__ void k_synth_coeffs(SynthArgs a)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_streams * a.mbs_per_picture * 6) return;
    const uint32_t blk = g % 6, gm = g / 6, p = gm / a.mbs_per_picture, mb = gm % a.mbs_per_picture;
    const MbRecord r = a.mbs[gm];
    if (!((r.cbp >> blk) & 1)) return;
    int16_t c[64];
    synth_block_coeffs(a.kind, a.first_stream_id + p * a.stream_stride, a.frame_idx, mb, (int)blk, c);
    const uint64_t idx = a.coeff_base[p] + r.coeff_index + (uint64_t)__popc(r.cbp & ((1u << blk) - 1u));
    uint4 *dst = reinterpret_cast<uint4 *>(a.coeffs + idx * 64);
#pragma unroll
    for (int q = 0; q < 8; q++) {
        uint32_t w[4];
        for (int k = 0; k < 4; k++) w[k] = (uint16_t)c[q * 8 + 2 * k] | ((uint32_t)(uint16_t)c[q * 8 + 2 * k + 1] << 16);
        dst[q] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

hipError_t launch_synth_headers(const SynthArgs &a, hipStream_t stream)
{
    const uint32_t n = a.n_streams * a.mbs_per_picture;
    hipLaunchKernelGGL(k_synth_headers, dim3((n + 255) / 256), dim3(256), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_synth_scan, dim3(a.n_streams), dim3(1024), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_synth_coeffs(const SynthArgs &a, hipStream_t stream)
{
    const uint32_t n = a.n_streams * a.mbs_per_picture * 6;
    hipLaunchKernelGGL(k_synth_coeffs, dim3((n + 255) / 256), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// on-box memory ceilings (bench support: roofline.peak_measured).  Streaming kernels with 16-byte accesses: the rate the
// HBM system of THIS box sustains for a copy, a pure read and a pure write.  Round 2's plain grid-stride probes
// (2 048 workgroups, one access per iteration) reported 4.8 TB/s for a copy where the MI355X guide measures 6.29:
// tools/probes/ceiling.hip swept the launch shape (profiles/r03_a_ceiling_probe.txt) -- what reaches 6.2 TB/s is
// non-temporal accesses, four of them in flight per lane (copy); a read wants non-temporal loads, a write is fastest
// with few (256) workgroups of plain stores.  `shape` selects among the winners; the C entry point reports the best.
// ---------------------------------------------------------------------------------------
typedef uint32_t probe_u32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int U, bool NT>
__global__ __launch_bounds__(1024) void k_probe(const probe_u32x4 *__restrict__ in, probe_u32x4 *__restrict__ out, size_t n)
{
    // a workgroup walks the buffer in steps of gridDim * blockDim * U elements; its U accesses of a step are blockDim apart
    const size_t step = (size_t)gridDim.x * blockDim.x * U;
    probe_u32x4 acc = {0, 0, 0, 0};
    for (size_t base = (size_t)blockIdx.x * blockDim.x * U + threadIdx.x; base < n; base += step) {
        probe_u32x4 v[U];
        if (MODE != 2) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const size_t i = base + (size_t)u * blockDim.x;
                v[u] = i < n ? (NT ? __builtin_nontemporal_load(in + i) : in[i]) : acc;
            }
        }
        if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < U; u++) acc ^= v[u];
        } else {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const size_t i = base + (size_t)u * blockDim.x;
                if (MODE == 2) v[u] = probe_u32x4{(uint32_t)i, 1, 2, 3};
                if (i < n) {
                    if (NT) __builtin_nontemporal_store(v[u], out + i);
                    else out[i] = v[u];
                }
            }
        }
    }
    if (MODE == 1 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = acc;      // never true for the probe's fill pattern
}

int probe_shapes(int mode) { return mode == 0 ? 2 : mode == 1 ? 2 : 3; }
const char *probe_shape_name(int mode, int shape)
{
    static const char *names[3][3] = {
        {"non-temporal, 4 accesses in flight, 4096 x 256 threads", "non-temporal, 4 accesses in flight, 256 x 512 threads", ""},
        {"non-temporal loads, 4 in flight, 256 x 512 threads", "non-temporal loads, 512 x 1024 threads", ""},
        {"plain stores, 256 x 256 threads", "plain stores, 8 per iteration, 256 x 1024 threads", "non-temporal stores, 256 x 256 threads"}};
    return names[mode][shape];
}

hipError_t launch_probe(int mode, int shape, const void *in, void *out, size_t bytes, hipStream_t stream)
{
    const size_t n = bytes / 16;
    const probe_u32x4 *pi = (const probe_u32x4 *)in;
    probe_u32x4 *po = (probe_u32x4 *)out;
    if (mode == 0 && shape == 0) hipLaunchKernelGGL((k_probe<0, 4, true>), dim3(4096), dim3(256), 0, stream, pi, po, n);
    else if (mode == 0) hipLaunchKernelGGL((k_probe<0, 4, true>), dim3(256), dim3(512), 0, stream, pi, po, n);
    else if (mode == 1 && shape == 0) hipLaunchKernelGGL((k_probe<1, 4, true>), dim3(256), dim3(512), 0, stream, pi, po, n);
    else if (mode == 1) hipLaunchKernelGGL((k_probe<1, 1, true>), dim3(512), dim3(1024), 0, stream, pi, po, n);
    else if (shape == 0) hipLaunchKernelGGL((k_probe<2, 1, false>), dim3(256), dim3(256), 0, stream, pi, po, n);
    else if (shape == 1) hipLaunchKernelGGL((k_probe<2, 8, false>), dim3(256), dim3(1024), 0, stream, pi, po, n);
    else hipLaunchKernelGGL((k_probe<2, 1, true>), dim3(256), dim3(256), 0, stream, pi, po, n);
    return hipGetLastError();
}

}  // namespace h263mi
