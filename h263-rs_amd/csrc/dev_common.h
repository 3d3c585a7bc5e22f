// dev_common.h -- types and helpers shared by the HIP kernels and their host launchers.
//
// Kernel bodies are written as per-phase inline functions taking an explicit lane index
// and wave position (recon_kernel.inl, post_kernel.inl); kernels.hip calls them from the
// __global__ functions, one wave per workgroup, no barriers.  tests/sim/ compiles the
// same phase functions with g++ (ASan/UBSan) and runs the 64 lanes of a wave in a loop,
// one phase at a time -- a logic checker for index math and edge handling, used by the
// CPU test-suite only; it is not a product path.
#pragma once

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "../../include/h263mi.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define H263_DEV __device__ __forceinline__
#define H263_HD __host__ __device__ __forceinline__
#else
#define H263_DEV inline
#define H263_HD inline
// g++ build (tests/sim only): minimal stand-ins for the HIP vector types the phases use
struct uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
#endif

namespace h263mi {

// Hand-offs between the lanes of ONE wave go through LDS without a workgroup barrier: the DS operations of a wave
// execute in order.  What still has to be said is that the COMPILER may not move a lane's LDS load above an earlier
// store of another lane's data (or a store below a load): a release / acquire fence pair at wavefront scope with a
// wave barrier in between orders them and emits no instruction.
H263_DEV void wave_fence()
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// ---------------------------------------------------------------------------
// Frame layout in HBM.  One frame = Y plane, then Cb, then Cr, each PITCHED and
// padded to whole macroblocks so kernels can write whole 8x8 blocks; only the
// w x h (cw x ch) window is ever read back or used as reference samples.
// ---------------------------------------------------------------------------
struct FrameLayout {
    uint32_t width, height;    // luma picture size
    uint32_t cwidth, cheight;  // chroma picture size: ceil(w/2), ceil(h/2) (picture.rs:45-46)
    uint32_t mbw, mbh;         // ceil(w/16), ceil(h/16) (state.rs:173-174)
    uint32_t pitch_y, pitch_c; // row pitch in bytes (pitch_y = 2 * pitch_c, multiple of 128 / 64)
    uint32_t rows_y, rows_c;   // allocated rows: mbh*16, mbh*8
    uint32_t off_cb, off_cr;   // byte offset of the chroma planes inside a frame
    uint32_t frame_bytes;      // bytes per frame incl. tail padding (multiple of 256)
    uint32_t pad;
};

H263_HD FrameLayout make_layout(uint32_t w, uint32_t h)
{
    FrameLayout L;
    L.width = w;
    L.height = h;
    L.cwidth = (w + 1) / 2;
    L.cheight = (h + 1) / 2;
    L.mbw = (w + 15) / 16;
    L.mbh = (h + 15) / 16;
    L.pitch_c = ((L.mbw * 8 + 63) / 64) * 64;
    L.pitch_y = L.pitch_c * 2;
    L.rows_y = L.mbh * 16;
    L.rows_c = L.mbh * 8;
    L.off_cb = L.pitch_y * L.rows_y;
    L.off_cr = L.off_cb + L.pitch_c * L.rows_c;
    // +256: unaligned 8/16-byte motion-compensation reads may run a few bytes past the last row
    L.frame_bytes = ((L.off_cr + L.pitch_c * L.rows_c + 256 + 255) / 256) * 256;
    L.pad = 0;
    return L;
}

// Every offset inside a frame is a 32-bit quantity on the device.  Pictures whose frame store would not fit
// kMaxFrameBytes (1 GiB: e.g. 26 000 x 26 000) are rejected by the host entry points BEFORE anything is allocated
// or launched: the 16-bit width / height of a Sorenson custom format come straight from an untrusted bitstream.
constexpr uint64_t kMaxFrameBytes = 1ull << 30;
inline bool layout_fits(uint64_t w, uint64_t h)
{
    if (!w || !h || w > 65535 || h > 65535) return false;
    const uint64_t mbw = (w + 15) / 16, mbh = (h + 15) / 16;
    const uint64_t pitch_c = ((mbw * 8 + 63) / 64) * 64;
    const uint64_t bytes = 2 * pitch_c * mbh * 16 + 2 * pitch_c * mbh * 8 + 512;
    return bytes <= kMaxFrameBytes;
}

typedef h263mi_mb_record MbRecord;
static_assert(sizeof(MbRecord) == 32, "record layout is part of the ABI");

// status bits written by kernels into the device word of the picture's stream, read back at sync time
enum : uint32_t {
    STATUS_INTER_WITHOUT_REFERENCE = 1u,   // gather.rs:149 Error::UncodedIFrameBlocks
    STATUS_COEFF_INDEX_OUT_OF_RANGE = 2u,
};

// Per-stream state of a batch whose streams no longer agree (one of them was reset, skipped a call, or had a picture
// rejected: every stream is its own H263State, state.rs:16-50).  One word per stream, read by the waves of its picture:
enum : uint32_t {
    STREAM_REF_SET1 = 1u,     // the stream's last picture (= its reference, state.rs:72-78) lives in frame set 1; the new one goes to the other set
    STREAM_HAS_REF = 2u,      // reference_picture.is_some() (state.rs:29-31)
    STREAM_RECON_SKIP = 4u,   // no picture for this stream in this call: its frames are not touched
    STREAM_POST_SET1 = 8u,    // the picture to post-process lives in frame set 1
    STREAM_POST_SKIP = 16u,   // nothing to post-process for this stream
    // bits 8..11: the post-filter strength of the stream's picture, 0 (no deblocking) .. 12 -- the consumer picks it per picture
    // (QUANT_TO_STRENGTH of the picture's quantiser, deblock.rs:5-8): with per-stream words in use every post-processing wave
    // takes its strength from here, whether or not the streams' strengths differ
    STREAM_STRENGTH_SHIFT = 8u,
    STREAM_STRENGTH_MASK = 15u,
};

// A launch of up to STREAM_WORDS_INLINE pictures carries the streams' words IN ITS KERNEL ARGUMENTS (round 6): nothing is
// copied to the device in front of the launch -- the small H2D copy between two launches cost the 64-stream loop 2 % (5 us
// of bubble per frame index, tools/probes/strength_ab.py) -- and a wave reads its picture's word with one scalar load from
// the kernarg segment.  Larger launches read them from a device array (ReconArgs / PostArgs::stream_state).
constexpr uint32_t STREAM_WORDS_INLINE = 64;
struct StreamWords {
    uint32_t w[STREAM_WORDS_INLINE];
};

// ---------------------------------------------------------------------------
// reconstruction kernel arguments
// ---------------------------------------------------------------------------
struct ReconArgs {
    FrameLayout L;
    const MbRecord *mbs;         // n_pictures * mbs_per_picture records
    const int16_t *coeffs;       // coefficient pool (dense transport)
    const uint32_t *block_first_event;  // sparse transport (events != nullptr): events of coded block b of the pool are
    const uint32_t *events;             // [block_first_event[b], block_first_event[b + 1]); level << 16 | x + 8 * y
    const uint64_t *coeff_base;  // per picture base (blocks) or nullptr
    const uint8_t *ref;          // reference frames (picture p at + p*frame_bytes); never null
    uint8_t *cur;                // output frames
    uint32_t *status;            // device status words, one per picture (stream)
    const uint32_t *stream_state;// per-stream STREAM_* words (device array), or nullptr: with words_inline 0 too, every stream
                                 // reads `ref`, writes `cur`, has `has_ref`
    uint8_t *frame_set[2];       // the two frame sets (used with stream_state)
    uint64_t coeff_pool_blocks;  // size of the pool (blocks), used when coeff_checked is set
    uint32_t coeff_checked;      // 1: a coded block whose index is >= coeff_pool_blocks is an error (and is not read)
    uint32_t n_pictures;
    uint32_t mbs_per_picture;
    uint32_t has_ref;            // 0: inter macroblocks are an error
    uint32_t tiles_x, tiles_y;
    uint32_t inv_tiles_x;        // ceil(2^32 / tiles_x): tile / tiles_x == mul_hi(tile, inv_tiles_x) (set by the launcher)
    uint32_t bands;              // k_recon: XCDs that share one picture (8 or 4; set by the launcher, see launch_frame)
    uint32_t n_events;           // sparse transport: words in `events` (0xffffffff: the caller did not say) -- a block whose
                                 // bounds are not ascending or reach beyond it is not read and the picture is rejected
    // SPARSE RECORDS (round 5; the batch entry that parses bitstreams on the host): `mbs` holds records for the macroblocks that
    // are coded only.  mb_group_index[pic * groups_per_picture + mby * tiles_x + mbx0 / 8] = first << 8 | mask for the 8
    // macroblocks a reconstruction wave works on: bit k of mask = macroblock k has a record, `first` = the number of its first
    // record counted from the picture's first (mb_base[pic], in records).  A macroblock without a record is not coded (INTER,
    // zero vectors, nothing coded: state.rs:207-216).  nullptr: `mbs` is the dense raster array.
    const uint32_t *mb_group_index;
    const uint64_t *mb_base;
    uint32_t groups_per_picture;
    uint32_t words_inline;       // 1: the per-stream words are the launch's StreamWords argument (set by the launcher), not stream_state
};

// ---------------------------------------------------------------------------
// post kernel (deblock + BT.601) arguments
// ---------------------------------------------------------------------------
struct PostArgs {
    FrameLayout L;
    const uint8_t *frames;       // picture p at + p*frame_bytes
    const uint32_t *stream_state;// per-stream STREAM_* words, or nullptr: every stream is read from `frames`
    const uint8_t *frame_set[2]; // the two frame sets (used with stream_state)
    uint8_t *rgba;               // n_pictures * w*h*4, tightly packed, or nullptr
    uint8_t *planes_out;         // n_pictures * (w*h + 2*cw*ch) tightly packed deblocked planes, or nullptr
    uint32_t n_pictures;
    uint32_t strength;           // 0 = no deblocking (with stream_state: unused, the strength is in the stream's word)
    uint32_t tiles_x, tiles_y;
    uint32_t luma_only;          // standalone deblock() of a single plane
    uint32_t inv_tiles_x;        // ceil(2^32 / tiles_x) (set by the launcher)
    uint32_t wrap;               // 1: tile column 0 does not exist, its 4 picture columns ride in the last tile (post_kernel.inl)
    uint32_t words_inline;       // 1: the per-stream words are the launch's StreamWords argument (set by the launcher)
    uint8_t *const *rgba_ptrs;   // with stream_state only (or nullptr): picture p's RGBA goes to rgba_ptrs[p] instead of
                                 // rgba + p * w*h*4 -- streams whose outputs are separate buffers (a batch of mixed sizes)
};

// ---------------------------------------------------------------------------
// small integer helpers
// ---------------------------------------------------------------------------
H263_HD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---------------------------------------------------------------------------
// packed 16-bit integer arithmetic: both halves of a dword at once (one v_pk_* instruction each on the device,
// plain C on the two halves in the CPU logic checker).  Wrapping, like i16 arithmetic in a release build of the
// reference.  Shift counts are compile-time constants.
// ---------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
#define H263_PK_ASM2(insn, x, y) uint32_t r_; asm(insn " %0, %1, %2" : "=v"(r_) : "v"(x), "v"(y)); return r_
#define H263_PK_SHIFT(insn, x, n) uint32_t r_; asm(insn " %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r_) : "n"(n), "v"(x)); return r_
#endif
#define H263_PK_HALVES(expr)                                                                     \
    uint32_t out_ = 0;                                                                           \
    for (int h_ = 0; h_ < 2; h_++) {                                                             \
        const int a = (int16_t)(x >> (16 * h_)), b = (int16_t)(y >> (16 * h_));                  \
        out_ |= ((uint32_t)(expr) & 0xffffu) << (16 * h_);                                       \
    }                                                                                            \
    return out_
H263_DEV uint32_t pk_add_u16(uint32_t x, uint32_t y)
{
#if defined(__HIP_DEVICE_COMPILE__)
    H263_PK_ASM2("v_pk_add_u16", x, y);
#else
    H263_PK_HALVES(a + b);
#endif
}
H263_DEV uint32_t pk_sub_u16(uint32_t x, uint32_t y)
{
#if defined(__HIP_DEVICE_COMPILE__)
    H263_PK_ASM2("v_pk_sub_u16", x, y);
#else
    H263_PK_HALVES(a - b);
#endif
}
H263_DEV uint32_t pk_max_i16(uint32_t x, uint32_t y)
{
#if defined(__HIP_DEVICE_COMPILE__)
    H263_PK_ASM2("v_pk_max_i16", x, y);
#else
    H263_PK_HALVES(a > b ? a : b);
#endif
}
H263_DEV uint32_t pk_min_i16(uint32_t x, uint32_t y)
{
#if defined(__HIP_DEVICE_COMPILE__)
    H263_PK_ASM2("v_pk_min_i16", x, y);
#else
    H263_PK_HALVES(a < b ? a : b);
#endif
}
H263_DEV uint32_t pk_mad_i16(uint32_t x, uint32_t y, uint32_t z)        // x * y + z per half
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
    return r;
#else
    uint32_t out = 0;
    for (int h = 0; h < 2; h++)
        out |= ((uint32_t)((int16_t)(x >> (16 * h)) * (int16_t)(y >> (16 * h)) + (int16_t)(z >> (16 * h))) & 0xffffu) << (16 * h);
    return out;
#endif
}
H263_DEV uint32_t pk_ashr_i16(uint32_t x, uint32_t n)           // arithmetic shift right of both halves by the constant n
{
#if defined(__HIP_DEVICE_COMPILE__)
    H263_PK_SHIFT("v_pk_ashrrev_i16", x, n);
#else
    const uint32_t lo = (uint32_t)((int16_t)(x & 0xffffu) >> n) & 0xffffu, hi = (uint32_t)((int16_t)(x >> 16) >> n) & 0xffffu;
    return lo | (hi << 16);
#endif
}
H263_DEV uint32_t pk_lshr_u16(uint32_t x, uint32_t n)
{
#if defined(__HIP_DEVICE_COMPILE__)
    H263_PK_SHIFT("v_pk_lshrrev_b16", x, n);
#else
    return ((x & 0xffffu) >> n) | (((x >> 16) >> n) << 16);
#endif
}
H263_DEV uint32_t pk_lshl_u16(uint32_t x, uint32_t n)
{
#if defined(__HIP_DEVICE_COMPILE__)
    H263_PK_SHIFT("v_pk_lshlrev_b16", x, n);
#else
    return ((x << n) & 0xffffu) | ((((x >> 16) << n) & 0xffffu) << 16);
#endif
}
// both halves saturated to 0..255; the two bytes arrive in bits 15:0 (bits 31:16 are not to be relied upon)
H263_DEV uint32_t sat_pk_u8_i16(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(x));
    return r;
#else
    const int a = (int16_t)(x & 0xffffu), b = (int16_t)(x >> 16);
    return (uint32_t)(a < 0 ? 0 : (a > 255 ? 255 : a)) | ((uint32_t)(b < 0 ? 0 : (b > 255 ? 255 : b)) << 8);
#endif
}

// types.rs:955-961 IntraDc::into_level
H263_HD int intradc_level(uint32_t code) { return code == 0xFFu ? 1024 : (int)(code << 3); }

// types.rs:653-658 / 661-663
H263_HD bool mb_is_inter(uint32_t t) { return t == 0 || t == 1 || t == 2 || t == 5; }
H263_HD bool mb_is_intra(uint32_t t) { return t == 3 || t == 4; }

// types.rs:759-768 HalfPel::average_sum_of_mvs on the i16 sum of four vectors
H263_HD int average_sum_of_mvs(int sum)
{
    int s = (int)(int16_t)sum;
    int whole = (s >> 4) * 2;   // (s >> 4) << 1 in the reference; written without shifting a negative
    int frac = s & 15;
    return frac <= 2 ? whole : (frac >= 14 ? whole + 2 : whole + 1);
}

// idct.rs:39-48 BASIS_TABLE[freq][x] -- the reference's literals (not exact cosines).
#define H263MI_BASIS_ROWS                                                                                          \
    {0.70710677f, 0.70710677f, 0.70710677f, 0.70710677f, 0.70710677f, 0.70710677f, 0.70710677f, 0.70710677f},       \
    {0.98078525f, 0.8314696f, 0.5555702f, 0.19509023f, -0.19509032f, -0.55557036f, -0.83146966f, -0.9807853f},      \
    {0.9238795f, 0.38268343f, -0.38268352f, -0.9238796f, -0.9238795f, -0.38268313f, 0.3826836f, 0.92387956f},       \
    {0.8314696f, -0.19509032f, -0.9807853f, -0.55557f, 0.55557007f, 0.98078525f, 0.19509007f, -0.8314698f},         \
    {0.70710677f, -0.70710677f, -0.70710665f, 0.707107f, 0.70710677f, -0.70710725f, -0.70710653f, 0.7071068f},      \
    {0.5555702f, -0.9807853f, 0.19509041f, 0.83146936f, -0.8314698f, -0.19508928f, 0.9807853f, -0.55557007f},       \
    {0.38268343f, -0.9238795f, 0.92387974f, -0.3826839f, -0.38268384f, 0.9238793f, -0.92387974f, 0.3826839f},       \
    {0.19509023f, -0.55557f, 0.83146936f, -0.9807852f, 0.98078525f, -0.83147013f, 0.55557114f, -0.19508967f}

}  // namespace h263mi
