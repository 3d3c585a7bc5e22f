/*
 * h263mi.h -- C ABI of the MI355X-native macroblock back-end for h263-rs.
 *
 * This is the drop-in boundary: every entry point replaces one interface of the
 * reference (ruffle-rs/h263-rs, cited as file:line relative to the reference root).
 * The serial bitstream/VLC parse stays on the host; per picture it emits a flat array
 * of macroblock records which crosses this ABI into hand-written HIP kernels (gfx950)
 * that do dequantisation + 8x8 inverse DCT, half-pel motion compensation, residual add
 * with clipping, the deblocking post-filter and the BT.601 YUV->RGBA conversion.
 *
 * All functions return H263MI_OK (0) or a negative error code.  There is NO CPU
 * fallback: without a HIP device every compute entry point returns
 * H263MI_ERR_NO_DEVICE.
 *
 * Threading (mirrors `&mut self` on H263State, state.rs:16-38): one h263mi_state /
 * h263mi_batch per stream (or per batch of streams), not thread-safe per object;
 * distinct objects may be driven from different host threads.
 */
#ifndef H263MI_H
#define H263MI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define H263MI_ABI_VERSION 7

/* ---- error codes: h263/src/error.rs:6-58, one per `Error` variant, in order ---- */
#define H263MI_OK                                  0
#define H263MI_ERR_INTERNAL_DECODER_ERROR        (-1)
#define H263MI_ERR_MIDDLE_OF_BITSTREAM           (-2)
#define H263MI_ERR_INVALID_MACROBLOCK_HEADER     (-3)
#define H263MI_ERR_INVALID_MACROBLOCK_CODED_BITS (-4)
#define H263MI_ERR_INVALID_INTRA_DC              (-5)
#define H263MI_ERR_INVALID_SHORT_COEFFICIENT     (-6)
#define H263MI_ERR_INVALID_LONG_COEFFICIENT      (-7)
#define H263MI_ERR_INVALID_MVD                   (-8)
#define H263MI_ERR_INVALID_PTYPE                 (-9)
#define H263MI_ERR_INVALID_PLUS_PTYPE            (-10)
#define H263MI_ERR_INVALID_GOB_HEADER            (-11)
#define H263MI_ERR_INVALID_BITSTREAM             (-12)
#define H263MI_ERR_PICTURE_FORMAT_MISSING        (-13)
#define H263MI_ERR_PICTURE_FORMAT_INVALID        (-14)
#define H263MI_ERR_UNCODED_IFRAME_BLOCKS         (-15)  /* gather.rs:149, state.rs:211-213 */
#define H263MI_ERR_UNHANDLED_IO_ERROR            (-16)  /* incl. UnexpectedEof */
#define H263MI_ERR_UNIMPLEMENTED_DECODING        (-17)
/* back-end errors (no counterpart in the reference) */
#define H263MI_ERR_INVALID_ARGUMENT              (-100)
#define H263MI_ERR_NO_DEVICE                     (-101)
#define H263MI_ERR_HIP                           (-102)
#define H263MI_ERR_OUT_OF_MEMORY                 (-103)
#define H263MI_ERR_NO_PICTURE                    (-104) /* get_last_picture() == None, state.rs:61-67 */

const char *h263mi_strerror(int code);          /* messages of error.rs:7-57 */
int  h263mi_abi_version(void);

/* ---- decoder options: h263/src/decoder/types.rs:3-17 (DecoderOption bitflags) ---- */
#define H263MI_SORENSON_SPARK_BITSTREAM 0x1u
#define H263MI_USE_SCALABILITY_MODE     0x2u

/* ---- macroblock types: h263/src/types.rs:631-649 (MacroblockType) ---- */
#define H263MI_MB_INTER     0
#define H263MI_MB_INTER_Q   1
#define H263MI_MB_INTER4V   2
#define H263MI_MB_INTRA     3
#define H263MI_MB_INTRA_Q   4
#define H263MI_MB_INTER4V_Q 5

/*
 * One record per macroblock, raster order (state.rs:199-202).  It carries exactly the
 * state the reference holds at its cut line (state.rs:429-438) before it calls
 * gather() + idct_channel():
 *
 *   mb_type      MacroblockType; an uncoded or padded macroblock is Inter with cbp 0 and
 *                mv 0 (state.rs:207-216, 421-427).
 *   quant        quantiser in force, 1..31, after the DQUANT clamp (state.rs:226-227).
 *   cbp          bit b set <=> block b (Y0 Y1 Y2 Y3 Cb Cr) carries TCOEFs
 *                (CodedBlockPattern, types.rs:683-687).
 *   kill         bit b set <=> a run of block b walked past zigzag 63: the block
 *                contributes nothing, its INTRADC included (rle.rs:125-127).
 *   mv           the four absolute half-pel luma motion vectors {x, y} after prediction
 *                (state.rs:238-284); one-vector macroblocks replicate mv[0].
 *   intradc      raw INTRADC code of each block for intra macroblocks (types.rs:923-961:
 *                level = code << 3, 0xFF -> 1024; 0 and 128 never occur), else 0.
 *   coeff_index  index, in 64-coefficient blocks, of this macroblock's first coded
 *                block in the coefficient array; its coded blocks follow each other in
 *                block order.
 *
 * Coefficient array: int16_t[64] per coded block, RASTER order (x + 8*y) -- i.e. already
 * run-length expanded and de-zigzagged (rle.rs:6-71, 122-136) -- holding the quantised
 * LEVEL (0 = absent).  Dequantisation (rle.rs:130-133), the Zero/Dc/Horiz/Vert/Full
 * classification (rle.rs:94-109, 138-170) and everything after it run on the GPU.
 * For intra blocks coefficient 0 is ignored (the DC comes from `intradc`).
 * Contract (as in the reference, rle.rs:130): quant * (2*|LEVEL| + 1) <= 32767.
 */
typedef struct h263mi_mb_record {
    uint8_t  mb_type;
    uint8_t  quant;
    uint8_t  cbp;
    uint8_t  kill;
    int16_t  mv[4][2];
    uint8_t  intradc[6];
    uint8_t  reserved[2];
    uint32_t coeff_index;
} h263mi_mb_record;                                  /* 32 bytes */

/* Picture-level fields the back-end needs from `Picture` (types.rs:20-90). */
#define H263MI_PICTURE_I 0   /* PictureTypeCode::IFrame: clears the reference (state.rs:464-470) */
#define H263MI_PICTURE_P 1   /* PFrame */
#define H263MI_PICTURE_DISPOSABLE_P 2 /* does not become the reference (state.rs:474-480) */
/* The remaining PictureTypeCode values (types.rs:251-288) only come out of picture headers: 3 = the reserved code of
 * the Sorenson 2-bit field, the others exist in ITU-T H.263 PTYPE / MPPTYPE only.  The macroblock layer of the
 * reference decodes I and P pictures; a coded macroblock in any other type is H263MI_ERR_UNIMPLEMENTED_DECODING
 * (macroblock.rs:461-465).  For h263mi_submit_picture every type but I predicts from the reference picture. */
#define H263MI_PICTURE_RESERVED_SORENSON 3
#define H263MI_PICTURE_PB 4
#define H263MI_PICTURE_IMPROVED_PB 5
#define H263MI_PICTURE_B 6
#define H263MI_PICTURE_EI 7
#define H263MI_PICTURE_EP 8
#define H263MI_PICTURE_RESERVED 9
typedef struct h263mi_picture_desc {
    uint16_t width, height;        /* SourceFormat::into_width_and_height (state.rs:169-171) */
    uint8_t  picture_type;
    uint8_t  pquant;               /* Picture::quantizer */
    uint8_t  use_deblocker;        /* PictureOption::USE_DEBLOCKER (types.rs:213-216), advisory */
    uint8_t  reserved0;
    uint16_t temporal_reference;   /* Picture::temporal_reference: key of the reference store */
    uint16_t reserved1;
} h263mi_picture_desc;                               /* 12 bytes */

/* Where the back-end runs.  `stream` is a hipStream_t (NULL = the device's null stream).
 * flags: H263MI_CFG_OVERLAP_POST (batches only): h263mi_batch_render_rgba runs on a second, internal stream so
 * that post-processing picture i overlaps reconstructing picture i+1; h263mi_batch_sync waits for both. */
#define H263MI_CFG_OVERLAP_POST 0x1u
/* H263MI_CFG_PIPELINE_POST (batches only): h263mi_batch_decode defers the deblock + BT.601 of a picture to the launch
 * that reconstructs the NEXT picture of the batch: one launch (k_frame) then reads the frame set once for both
 * purposes -- as the reference picture of the new pictures and as the pictures to filter and convert.  The output
 * buffers handed to h263mi_batch_decode are complete after the next h263mi_batch_sync (or after the next call that
 * renders or decodes on this batch has been followed by a sync); a consumer that reads them in stream order must
 * call h263mi_batch_sync first.  Results are identical to the immediate mode. */
#define H263MI_CFG_PIPELINE_POST 0x2u
/* H263MI_CFG_TRUSTED_ARRAYS (batches only, ABI 7): the caller vouches for the DEVICE arrays it hands to h263mi_batch_submit /
 * _decode / _decode_events.  Without it -- the default -- nothing in those arrays is believed: the sizes the caller gives
 * (coeff_pool_blocks, n_events) bound what the waves read, a size it does not give (0) is taken from the allocation the
 * pointer lies in (hipMemGetAddressRange; a pointer the runtime does not know is H263MI_ERR_INVALID_ARGUMENT), and the record
 * and base arrays must fit theirs.  A coded block outside the pool or an event list whose bounds do not ascend or reach beyond
 * the events is not read and rejects its stream's picture at the next sync.  The checks cost the 64-stream launch nothing
 * measurable (bench.py: roofline.trusted_mode); the flag is for callers that cannot afford even the look-up.  The OUTPUT
 * buffers of those entry points (d_rgba, d_deblocked) are held to their allocations likewise: one the runtime knows and that
 * cannot hold the n_streams pictures the launch writes is H263MI_ERR_INVALID_ARGUMENT before anything is queued. */
#define H263MI_CFG_TRUSTED_ARRAYS 0x4u
typedef struct h263mi_backend_cfg {
    int32_t  device_id;
    uint32_t flags;
    void    *stream;
} h263mi_backend_cfg;

/* ======================================================================= */
/* H263State  (h263/src/decoder/state.rs)                                   */
/* ======================================================================= */
typedef struct h263mi_state h263mi_state;

/* H263State::new(decoder_options)  state.rs:42-50.  cfg may be NULL (device 0, null stream). */
int  h263mi_state_new(uint32_t decoder_options, const h263mi_backend_cfg *cfg, h263mi_state **out);
void h263mi_state_free(h263mi_state *s);
/* H263State::is_sorenson  state.rs:53-56 */
int  h263mi_state_is_sorenson(const h263mi_state *s);
/* Seeking rule of state.rs:134-137: discard all decoder state. */
int  h263mi_state_reset(h263mi_state *s);
/* H263State::cleanup_buffers  state.rs:81-98 (drops every picture but last/reference). */
int  h263mi_state_cleanup_buffers(h263mi_state *s);

/*
 * The record-level form of H263State::decode_next_picture (state.rs:138-489): the caller
 * has run the serial parse (state.rs:193-417) and hands over the macroblock records; this
 * call performs state.rs:421-483 -- pad missing macroblocks as Inter/mv 0, gather(),
 * idct_channel() x3 on the GPU, then the reference bookkeeping.  `mbs` and `coeffs` are
 * HOST pointers; n_mbs <= ceil(w/16)*ceil(h/16).  On error the state is unchanged
 * (state.rs:142, 464-487).  Returns H263MI_ERR_UNCODED_IFRAME_BLOCKS when an inter
 * macroblock has no reference picture (gather.rs:149).
 */
int h263mi_submit_picture(h263mi_state *s, const h263mi_picture_desc *desc,
                          const h263mi_mb_record *mbs, size_t n_mbs,
                          const int16_t *coeffs, size_t n_coeff_blocks);

/*
 * The same with sparse coefficient transport: instead of a dense 128-byte block per coded block, one 32-bit
 * event per non-zero LEVEL, `(uint16_t)level << 16 | position` with position = x + 8*y (the de-zigzagged place,
 * DEZIGZAG_MAPPING rle.rs:6-71; an intra block's DC is not an event, it travels in the record).  Coded block k
 * (the numbering of coeff_index) owns events [block_first_event[k], block_first_event[k+1]); the array has
 * n_coeff_blocks + 1 entries, starts at 0 and ends at n_events; a block has at most 64 events and names every position
 * at most once (checked: H263MI_ERR_INVALID_ARGUMENT).  A typical P picture needs a quarter of the bytes of the dense
 * form over PCIe, and the reconstruction waves read the events as they are: no dense block is ever built in device
 * memory.
 */
#define H263MI_EVENT(position, level) (((uint32_t)(uint16_t)(int16_t)(level) << 16) | ((uint32_t)(position) & 63u))
int h263mi_submit_picture_events(h263mi_state *s, const h263mi_picture_desc *desc,
                                 const h263mi_mb_record *mbs, size_t n_mbs,
                                 const uint32_t *block_first_event, size_t n_coeff_blocks,
                                 const uint32_t *events, size_t n_events);

/*
 * H263State::decode_next_picture(reader)  state.rs:138-141, over a byte buffer holding
 * one coded picture (Ruffle hands one FLV video tag per reader).  `*consumed` receives the
 * bytes used.  The serial parse runs on the host (h263-rs_amd/host/bitstream.cpp: Sorenson Spark and ITU-T H.263
 * PTYPE / PLUSPTYPE headers, MCBPC/CBPY/MVD/TCOEF tables, Annex D vectors, motion vector prediction, the
 * start-code resynchronisation of state.rs:387-408) and feeds h263mi_submit_picture_events.  What the reference leaves
 * unimplemented (GOB headers, PB / B / EI / EP macroblocks, Annex T, reference picture resampling, back-channel
 * messages) returns H263MI_ERR_UNIMPLEMENTED_DECODING here too.  On any error the state -- including what it
 * remembers of the last picture header -- is unchanged.
 */
int h263mi_decode_next_picture(h263mi_state *s, const uint8_t *data, size_t len, size_t *consumed);
/* H263State::parse_picture(reader, previous_picture)  state.rs:102-111: header peek only (frame dependency
 * queries); fills the fields of h263mi_picture_desc from the picture header (width = height = 0 when the header
 * restates no format).  `previous_picture` is the state's last decoded picture.  H263MI_ERR_MIDDLE_OF_BITSTREAM when
 * the data does not start with a picture start code. */
int h263mi_parse_picture_header(const h263mi_state *s, const uint8_t *data, size_t len, h263mi_picture_desc *out);

/* DecodedPicture accessors (picture.rs:61-142) of get_last_picture() (state.rs:61-67). */
typedef struct h263mi_frame_view {
    uint16_t width, height;              /* luma_samples_per_row, rows */
    uint16_t chroma_width, chroma_height;/* chroma_samples_per_row (picture.rs:45-46) */
    uint16_t temporal_reference;
    uint8_t  picture_type;
    uint8_t  pquant;
    uint8_t  use_deblocker;
    uint8_t  reserved[3];
    const uint8_t *dev_y, *dev_cb, *dev_cr; /* DEVICE pointers, valid until the next decode */
    uint32_t dev_pitch_y, dev_pitch_c;      /* device row pitches in bytes */
} h263mi_frame_view;
int h263mi_get_last_picture(const h263mi_state *s, h263mi_frame_view *out);
/* get_reference_picture (state.rs:72-78): mirrors the reference, which returns the LAST
 * picture whenever a reference exists. */
int h263mi_get_reference_picture(const h263mi_state *s, h263mi_frame_view *out);
/* DecodedPicture::as_yuv (picture.rs:140-142): tightly packed planes (stride = width),
 * w*h, cw*ch, cw*ch bytes, copied to HOST memory. */
int h263mi_copy_yuv(const h263mi_state *s, uint8_t *y, uint8_t *cb, uint8_t *cr);
/*
 * What the consumer does after decoding: optional deblock() of each plane
 * with `strength` (0 = no deblocking, else 1..12) followed by yuv420_to_rgba(), fused on
 * the device for the last picture; w*h*4 bytes to HOST memory.  The reference planes are
 * not modified (post-filter, deblock.rs:1-2).
 *
 * The strength is the consumer's choice PER PICTURE: the reference exports QUANT_TO_STRENGTH for it (deblock.rs:5-8) and
 * hands out the picture's quantiser and its USE_DEBLOCKER flag through DecodedPicture::as_header (picture.rs:61-64,
 * types.rs:94-96, 216; set at parser/picture.rs:322).  (ABI 7) Wherever this library has parsed the picture header itself,
 * `strength` may be H263MI_STRENGTH_FROM_HEADER: the picture is filtered with
 *     use_deblocker ? h263mi_quant_to_strength[pquant] : 0
 * of ITS OWN header -- every stream of a batch call with its own value.  Entry points over records (no header in sight)
 * answer H263MI_ERR_INVALID_ARGUMENT to it; their *_ps forms take one value per stream from the caller instead.
 */
#define H263MI_STRENGTH_FROM_HEADER 0xFFu
int h263mi_render_rgba(const h263mi_state *s, uint8_t strength, uint8_t *rgba);
/*
 * The same into PINNED host memory (ABI 4), for a caller that renders every picture: `rgba` must lie in memory from
 * h263mi_host_alloc or registered with h263mi_host_register (else H263MI_ERR_INVALID_ARGUMENT).  The kernel's RGBA stores
 * go straight over the link into that memory -- no device buffer, no second copy, no pageable staging inside the
 * runtime -- and the call returns when they have landed.  What yuv420_to_rgba returns as a fresh Vec<u8> per frame
 * (bt601.rs:128) is here a buffer the caller owns and reuses.
 */
int h263mi_render_rgba_pinned(const h263mi_state *s, uint8_t strength, uint8_t *rgba_pinned);
/* page-locked, device-visible host memory for h263mi_render_rgba_pinned (and faster h263mi_copy_yuv / h263mi_render_rgba
 * targets); h263mi_host_register pins memory the caller already owns (page granularity is the runtime's business). */
int h263mi_host_alloc(size_t bytes, void **out);
int h263mi_host_free(void *p);
int h263mi_host_register(void *p, size_t bytes);
int h263mi_host_unregister(void *p);

/* ======================================================================= */
/* deblock crate  (deblock/src/deblock.rs)                                  */
/* ======================================================================= */
/* pub const QUANT_TO_STRENGTH: [u8; 32]  deblock.rs:5-8 */
extern const uint8_t h263mi_quant_to_strength[32];
/* pub fn deblock(data, width, strength) -> Vec<u8>  deblock.rs:305-315.  HOST buffers,
 * `out` holds `len` bytes; len % width == 0, 1 <= strength <= 12. */
int h263mi_deblock(const uint8_t *data, size_t len, size_t width, uint8_t strength, uint8_t *out);
/* ... on the device and stream of `cfg` (NULL = device 0, null stream).  Both forms keep their device scratch per host
 * thread and device between calls: a caller that invokes them once per frame allocates nothing after the first call. */
int h263mi_deblock_on(const h263mi_backend_cfg *cfg, const uint8_t *data, size_t len, size_t width, uint8_t strength,
                      uint8_t *out);

/* ======================================================================= */
/* yuv crate  (yuv/src/bt601.rs)                                            */
/* ======================================================================= */
/* pub fn yuv420_to_rgba(y, chroma_b, chroma_r, y_width) -> Vec<u8>  bt601.rs:105-196.
 * HOST buffers; rgba_out holds 4*y_len bytes.  Preconditions of bt601.rs:100-104 are
 * checked and reported as H263MI_ERR_INVALID_ARGUMENT. */
int h263mi_bt601_yuv420_to_rgba(const uint8_t *y, size_t y_len,
                                const uint8_t *chroma_b, const uint8_t *chroma_r, size_t c_len,
                                size_t y_width, uint8_t *rgba_out);
int h263mi_bt601_yuv420_to_rgba_on(const h263mi_backend_cfg *cfg, const uint8_t *y, size_t y_len,
                                   const uint8_t *chroma_b, const uint8_t *chroma_r, size_t c_len,
                                   size_t y_width, uint8_t *rgba_out);

/* ======================================================================= */
/* Batch of independent streams on one GPU (no counterpart in the reference:*/
/* it is the data-parallel form of N H263States advancing in lock step).    */
/* ======================================================================= */
typedef struct h263mi_batch h263mi_batch;

int  h263mi_batch_create(uint32_t n_streams, uint16_t width, uint16_t height,
                         const h263mi_backend_cfg *cfg, h263mi_batch **out);
void h263mi_batch_destroy(h263mi_batch *b);
uint32_t h263mi_batch_mbs_per_picture(const h263mi_batch *b);
/*
 * One picture per stream.  DEVICE pointers: d_mbs holds n_streams * mbs_per_picture
 * records (stream s at s * mbs_per_picture, all macroblocks present); d_coeffs is the
 * coefficient pool; d_coeff_base[s] (may be NULL = all 0) is the block index in the pool
 * that stream s's coeff_index values are relative to.  Asynchronous on the batch stream.
 * picture_type as in h263mi_picture_desc.  An inter macroblock without a reference is
 * detected on the device and reported by the next h263mi_batch_sync().
 */
int h263mi_batch_submit(h263mi_batch *b, uint8_t picture_type,
                        const h263mi_mb_record *d_mbs, const int16_t *d_coeffs,
                        const uint64_t *d_coeff_base);
/*
 * Decode + post-process in ONE call: h263mi_batch_submit followed by h263mi_batch_render_rgba (d_rgba / d_deblocked as
 * there; both NULL = reconstruction only).  Knowing both halves up front lets the library order the work so that
 * the reconstructed planes are still on chip when they are filtered and converted.  coeff_pool_blocks is the size of
 * d_coeffs in 64-coefficient blocks: a coded block that would lie outside is not read and makes the next
 * h263mi_batch_sync fail with H263MI_ERR_INVALID_ARGUMENT.  0 = not told: the pool then ends where the allocation d_coeffs
 * lies in ends (ABI 7; on a H263MI_CFG_TRUSTED_ARRAYS batch: nothing is checked).
 *
 * Errors the device detects (this one; an inter macroblock without a reference picture) surface at the next
 * h263mi_batch_sync, per stream (h263mi_batch_sync_streams says which).  A stream whose picture was rejected forgets
 * it, as the reference leaves its state unchanged on any error (state.rs:142, 464-487): if one picture was submitted
 * for it since the last successful sync, the picture before it is its last picture again; if several were, none
 * survives (its frame sets have been reused) and the stream is as after h263mi_batch_reset_stream.  The other streams
 * keep their pictures.
 * Limit: a stream's coeff_index values (relative to d_coeff_base[s]) address at most 2^25 blocks (4 GiB of
 * coefficients); a larger index is reported like one outside the pool.
 */
int h263mi_batch_decode(h263mi_batch *b, uint8_t picture_type,
                        const h263mi_mb_record *d_mbs, const int16_t *d_coeffs, const uint64_t *d_coeff_base,
                        uint64_t coeff_pool_blocks, uint8_t strength, uint8_t *d_rgba, uint8_t *d_deblocked);
/* (ABI 7) ... with one post-filter strength PER STREAM: strengths (HOST array of n_streams values 0..12, or NULL = `strength`
 * for every stream, as above).  The streams of a batch are independent: each picture has its own quantiser, so each has its
 * own strength (deblock.rs:5-8).  The post-processing waves of a launch read their picture's value with one scalar load;
 * 0 = that picture is converted without deblocking. */
int h263mi_batch_decode_ps(h263mi_batch *b, uint8_t picture_type,
                           const h263mi_mb_record *d_mbs, const int16_t *d_coeffs, const uint64_t *d_coeff_base,
                           uint64_t coeff_pool_blocks, uint8_t strength, const uint8_t *strengths, uint8_t *d_rgba,
                           uint8_t *d_deblocked);
/* h263mi_batch_decode with sparse coefficient transport (see h263mi_submit_picture_events), everything in DEVICE memory:
 * d_block_first_event[k], [k + 1] bound the events of coded block k of the pool (k = d_coeff_base[s] + the stream's own
 * block number; the array has one entry more than the pool has blocks), d_events holds `level << 16 | x + 8 * y` per
 * non-zero LEVEL (an intra block's DC travels in the record), at most 64 per block, every position at most once.  This
 * is the form the host parser emits and h263mi_batch_decode_next_pictures copies to the device: the reconstruction
 * waves read it as it is.
 * n_events (ABI 4): the number of words d_events holds, or 0 = not told.  The device arrays are the caller's and nobody
 * has validated them: a block whose bounds are not ascending or reach beyond n_events is NOT read and the stream's picture
 * is rejected at the next sync (H263MI_ERR_INVALID_ARGUMENT, like a coded block outside the pool).  With 0 (ABI 7) the
 * events end where the allocation d_events lies in ends, and the pool has as many blocks as the allocation of
 * d_block_first_event has entries, less one: whatever the arrays hold, no wave reads outside memory the caller owns.  Only
 * on a H263MI_CFG_TRUSTED_ARRAYS batch does 0 mean "the caller vouches": the bounds are then used as they are.
 * At most 0xffffff00 words (event indices are 32-bit on the device; more is H263MI_ERR_INVALID_ARGUMENT). */
int h263mi_batch_decode_events(h263mi_batch *b, uint8_t picture_type, const h263mi_mb_record *d_mbs,
                               const uint32_t *d_block_first_event, const uint32_t *d_events, const uint64_t *d_coeff_base,
                               uint64_t coeff_pool_blocks, uint64_t n_events, uint8_t strength, uint8_t *d_rgba,
                               uint8_t *d_deblocked);
/* (ABI 7) ... with one strength per stream (see h263mi_batch_decode_ps) */
int h263mi_batch_decode_events_ps(h263mi_batch *b, uint8_t picture_type, const h263mi_mb_record *d_mbs,
                                  const uint32_t *d_block_first_event, const uint32_t *d_events, const uint64_t *d_coeff_base,
                                  uint64_t coeff_pool_blocks, uint64_t n_events, uint8_t strength, const uint8_t *strengths,
                                  uint8_t *d_rgba, uint8_t *d_deblocked);
/* deblock (strength 0 = off) + BT.601 of every stream's last picture into d_rgba
 * (DEVICE, n_streams * w*h*4 bytes, stream-major); d_deblocked (DEVICE, may be NULL)
 * additionally receives the filtered planes, n_streams * (w*h + 2*cw*ch) bytes as
 * Y,Cb,Cr per stream, tightly packed. */
/*
 * The same from HOST records, one picture per stream: the batch counterpart of h263mi_submit_picture for a server
 * whose parser threads fill one record array per stream.  mbs[s] / coeffs[s] hold stream s' macroblocks
 * (n_mbs[s] <= mbs_per_picture; missing ones are padded as Inter / mv 0, state.rs:421-427) and coded blocks
 * (coeff_index counts from the stream's own first block).  The arrays are packed into pinned staging -- two slots,
 * used alternately, so packing picture i+1 overlaps the copy and the kernel of picture i -- copied with one
 * asynchronous H2D per array and decoded by one launch (k_recon; k_frame on a H263MI_CFG_PIPELINE_POST batch with a
 * deferred post-processing pending).  The host arrays may be reused on return.
 * Records are validated like h263mi_submit_picture validates them (types, quantiser, coded block indices against
 * n_coeff_blocks[s]; H263MI_ERR_INVALID_ARGUMENT before anything is queued); an inter macroblock without a
 * reference picture surfaces at h263mi_batch_sync like for h263mi_batch_submit.
 */
int h263mi_batch_submit_host(h263mi_batch *b, uint8_t picture_type,
                             const h263mi_mb_record *const *mbs, const uint32_t *n_mbs,
                             const int16_t *const *coeffs, const uint32_t *n_coeff_blocks);
/* ... and with sparse coefficient transport (see h263mi_submit_picture_events): block_first_event[s] has
 * n_coeff_blocks[s] + 1 entries counting from 0, events[s] has n_events[s] entries. */
int h263mi_batch_submit_host_events(h263mi_batch *b, uint8_t picture_type,
                                    const h263mi_mb_record *const *mbs, const uint32_t *n_mbs,
                                    const uint32_t *const *block_first_event, const uint32_t *n_coeff_blocks,
                                    const uint32_t *const *events, const uint32_t *n_events);
/*
 * N x H263State::decode_next_picture(reader) (state.rs:138-141) in one call: data[s] / len[s] hold one coded picture
 * of stream s (what Ruffle hands one reader per FLV video tag).  The serial parse of each stream (state.rs:143-427)
 * runs on `n_threads` host threads (0 = h263mi_default_parser_threads(n_streams, ...), below), one stream per task; the records of all streams
 * then cross to the device as events (h263mi_batch_submit_host_events) and ONE launch decodes them: k_recon, or -- on a
 * H263MI_CFG_PIPELINE_POST batch through the _ex form -- k_frame, which also post-processes the previous picture.
 * consumed[s] (may be NULL) receives the bytes used.  decoder_options as for h263mi_state_new.  If any stream fails
 * to parse, the error of the first such stream is returned and NOTHING changes for any stream.  Every picture must
 * have the batch's width and height.  The host parser is the bit-at-a-time reader of the reference (reader.rs:94-134,
 * 272-290) redesigned around a 64-bit window and table lookups (h263-rs_amd/host/bitstream.cpp).
 * The host threads belong to the batch (created at the first call, parked between calls); the parser writes a stream's
 * records, block offsets and events straight into the pinned staging memory the copies to the device read (per-stream parts
 * sized for the worst case of the call's pictures -- 8 * len[s] / 3 events: hand in ONE coded picture per stream, not the
 * rest of the file, or the call falls back to packing the arrays in a second pass; two slots, about 2 x 120 MB of pinned
 * memory for 64 streams of 1080p).  A batch is driven from one thread at a time.  With H263MI_TRACE_E2E set in the environment the batch prints, when it is destroyed, where the host time of
 * these calls went (parser threads / waiting for a staging slot / packing / enqueueing).
 * Every stream decodes in this form: data[s] == NULL is accepted only with len[s] == 0 and is an EMPTY reader (the stream
 * gets the parser's end-of-stream error and, all or nothing, the call fails with it); "no picture for this stream" exists
 * in the _ex form only.
 */
/*
 * (ABI 6) The parser threads a call with n_threads = 0 uses for `n_streams` streams.  Without a CPU-time quota: the CPUs the process
 * may run on (hardware threads, affinity mask).  Under a container's quota (cgroup cpu.max of Q CPUs on a host with more):
 * the fewest threads that give the rounds of Q + Q/2 threads -- 22 for 64 streams on 16 CPUs -- and the workers PARK as soon
 * as they run out of work instead of spinning for the next call: a quota limits CPU time, not threads, and a spinning worker
 * spends it like a parsing one (h263-rs_amd/csrc/worker_pool.h: HostThreadPlan; H263MI_QUOTA_OVERSUBSCRIBE=0: Q threads).  A
 * caller that passes more threads than the quota has CPUs gets the parking workers too.  Both limits are divided by the
 * number of processes that share the node: h263mi_set_ranks_per_node, else H263MI_RANKS_PER_NODE, else the launcher's
 * LOCAL_WORLD_SIZE.  The host's limits are read once per process.  *cpu_quota (may be NULL) receives Q, 0 = no quota.
 */
uint32_t h263mi_default_parser_threads(uint32_t n_streams, uint32_t *cpu_quota);
/* (ABI 7) How many processes share this node's CPUs (and its CPU-time quota) with this one: both limits above are divided by
 * it.  A job's launcher knows (one rank per GPU); a lone service rank started under the same launcher says 1.  0 = back to
 * the environment: H263MI_RANKS_PER_NODE, else LOCAL_WORLD_SIZE.  Process-wide; takes effect with the next decode call and
 * for batches made afterwards (their host threads' CPU slice, below). */
void h263mi_set_ranks_per_node(uint32_t ranks);
/* (ABI 7) NUMA placement of a batch's host side.  On a multi-socket node the parser threads and the pinned staging memory of
 * a batch belong on the socket its GPU hangs off: the batch looks up the NUMA node of its device (sysfs numa_node of the PCI
 * function), confines its host threads to that node's CPUs -- with several ranks per node: to this rank's slice of them, the
 * node's cores dealt to the devices 0 .. ranks - 1 that hang off it in device order, so that no two ranks parse on the same
 * cores -- and allocates its staging memory under a preferred-node policy for it.  H263MI_NUMA=0 switches it off,
 * H263MI_NUMA_NODE=k forces node k (experiments).  This call reports what was done: the device's node (-1 = unknown / off),
 * the node the staging memory really lies on (-1 = none yet / unknown) and the CPUs the host threads are confined to
 * (*n_pool_cpus = how many, 0 = not confined / no threads yet; cpus, may be NULL, receives up to cpus_cap of their numbers). */
int h263mi_batch_host_placement(const h263mi_batch *b, int *device_numa_node, int *staging_numa_node, uint32_t *n_pool_cpus,
                                uint16_t *cpus, uint32_t cpus_cap);
/* TEST HOOK (makes no HIP call): the placement a batch on device `device` would get on a host whose devices have the PCI
 * addresses pci_ids[0 .. n_devices) when `ranks` processes share the node, read from the sysfs tree under sysfs_root (NULL =
 * H263MI_SYSFS_ROOT, else /sys).  *node: the NUMA node, -1 = none. */
int h263mi_debug_host_placement(const char *const *pci_ids, uint32_t n_devices, int device, uint32_t ranks, const char *sysfs_root,
                                int *node, uint16_t *cpus, uint32_t cpus_cap, uint32_t *n_cpus);
int h263mi_batch_decode_next_pictures(h263mi_batch *b, uint32_t decoder_options, const uint8_t *const *data,
                                      const size_t *len, size_t *consumed, uint32_t n_threads);
/*
 * The same with every stream treated as the H263State it is (state.rs:16-50: its own last / reference picture,
 * its own format, its own errors):
 *   stream_rc[s]  receives the outcome of stream s: H263MI_OK, or the error of ITS decode_next_picture -- a parse error,
 *                 H263MI_ERR_PICTURE_FORMAT_INVALID (not the batch's size), H263MI_ERR_UNCODED_IFRAME_BLOCKS (inter
 *                 macroblocks and no reference picture yet, gather.rs:149: found on the host, before anything is
 *                 queued).  A stream that fails keeps its state, parser state included (state.rs:142); the others advance.
 *   data[s] NULL  stream s has no picture in this call and is left alone (also: h263mi_batch_set_active).  A non-NULL
 *                 pointer with len[s] == 0 is an empty reader, not "no picture": the stream gets its parse error.
 *   d_rgba / d_deblocked (may be NULL)  deblock(strength) + BT.601 of the pictures just decoded, as h263mi_batch_decode
 *                 does it: on a H263MI_CFG_PIPELINE_POST batch deferred to the next call's launch (k_frame), else a launch
 *                 of its own behind the reconstruction.  Streams that did not decode a picture in this call are not
 *                 rendered (their part of the buffers is not written).
 * Returns H263MI_OK when the call itself went through (look at stream_rc), else a back-end error.  An error of the
 * RENDERING half (the launch or an allocation behind it) is returned after the pictures were decoded: stream_rc and
 * consumed are valid then, the streams have advanced (a failed rendering does not un-decode anything); every error in
 * front of the reconstruction launch leaves all streams as they were.
 */
int h263mi_batch_decode_next_pictures_ex(h263mi_batch *b, uint32_t decoder_options, const uint8_t *const *data,
                                         const size_t *len, size_t *consumed, uint32_t n_threads, int *stream_rc,
                                         uint8_t strength, uint8_t *d_rgba, uint8_t *d_deblocked);
/* (ABI 7) The post-filter strength per picture.  `strength` of the _ex form may be H263MI_STRENGTH_FROM_HEADER: every stream's
 * picture is filtered with use_deblocker ? h263mi_quant_to_strength[pquant] : 0 of the header this call has just parsed for
 * it (deblock.rs:5-8, picture.rs:61-64, types.rs:94-96, 216) -- 64 streams with 64 quantisers render drop-in, in the same
 * launch (k_frame) as before.  The _ps form also takes the caller's own choice per stream: strengths (HOST array of n_streams
 * values 0..12; NULL = `strength`, which may be H263MI_STRENGTH_FROM_HEADER). */
int h263mi_batch_decode_next_pictures_ps(h263mi_batch *b, uint32_t decoder_options, const uint8_t *const *data,
                                         const size_t *len, size_t *consumed, uint32_t n_threads, int *stream_rc,
                                         uint8_t strength, const uint8_t *strengths, uint8_t *d_rgba, uint8_t *d_deblocked);
int h263mi_batch_render_rgba(h263mi_batch *b, uint8_t strength, uint8_t *d_rgba, uint8_t *d_deblocked);
/* (ABI 7) ... every stream's last picture with its own strength (strengths: HOST array of n_streams values, NULL = `strength`) */
int h263mi_batch_render_rgba_ps(h263mi_batch *b, uint8_t strength, const uint8_t *strengths, uint8_t *d_rgba, uint8_t *d_deblocked);
int h263mi_batch_sync(h263mi_batch *b);
/* h263mi_batch_sync with the verdict of the device per stream: stream_rc[s] = H263MI_OK,
 * H263MI_ERR_UNCODED_IFRAME_BLOCKS or H263MI_ERR_INVALID_ARGUMENT (a coded block outside the pool).  Only the streams
 * whose picture was rejected go back to their previous picture (or to none, when several of their pictures were in
 * flight); the others keep theirs.  Returns the first stream's error, like h263mi_batch_sync. */
int h263mi_batch_sync_streams(h263mi_batch *b, int *stream_rc);
/* H263State::new for every stream again: all pictures and parser states are forgotten (a deferred post-processing is
 * still delivered). */
int h263mi_batch_reset(h263mi_batch *b);
/* ... for ONE stream: the seeking rule of state.rs:134-137 applied to stream `stream` only; it then needs an I picture
 * (its next inter macroblock is H263MI_ERR_UNCODED_IFRAME_BLOCKS) while the other streams go on predicting. */
int h263mi_batch_reset_stream(h263mi_batch *b, uint32_t stream);
/* Which streams take part in the following submits / decodes: active[s] != 0, or NULL = all (the default).  A stream
 * that does not take part keeps its pictures; its records are ignored and nothing is rendered for it. */
int h263mi_batch_set_active(h263mi_batch *b, const uint8_t *active);
/* get_last_picture().is_some() of stream `stream` (1 / 0) */
int h263mi_batch_stream_has_picture(const h263mi_batch *b, uint32_t stream);
/* as_yuv of stream `stream`'s last picture -> HOST, tightly packed. */
int h263mi_batch_copy_yuv(h263mi_batch *b, uint32_t stream, uint8_t *y, uint8_t *cb, uint8_t *cr);

/* Per-kernel device time between begin/end, measured with hipEvents on the batch stream.  Back-to-back launches of the
 * same kernel are bracketed by one pair of events (begin in front of the first, end behind the last) and share the
 * elapsed time: *_ms / *_launches is the average time from one launch to the next, gaps between launches included --
 * an upper bound of the kernel's own duration (a pair of events around every launch cost 2 % of the throughput). */
typedef struct h263mi_kernel_times {
    double   recon_ms;  uint32_t recon_launches;  uint32_t pad0;   /* k_recon on its own */
    double   post_ms;   uint32_t post_launches;   uint32_t pad1;   /* k_post on its own */
    double   frame_ms;  uint32_t frame_launches;  uint32_t pad2;   /* k_frame: both in one launch (H263MI_CFG_PIPELINE_POST) */
} h263mi_kernel_times;
int h263mi_batch_timing_begin(h263mi_batch *b);
int h263mi_batch_timing_end(h263mi_batch *b, h263mi_kernel_times *out);

/* ======================================================================= */
/* Streams of DIFFERENT picture sizes behind one call  (ABI 4)              */
/* ======================================================================= */
/*
 * The reference resolves the picture format per H263State and per picture (state.rs:157-176): a server holds QCIF, CIF and
 * 1080p streams side by side and a stream may change its size at an I picture.  h263mi_batch takes one size; h263mi_mixed
 * keeps one such batch per size CLASS (created when the first stream of that size shows up) and decodes, per call, every
 * class that has pictures with ONE launch, back to back on cfg's HIP stream: QCIF + CIF + 1080p streams cost three
 * launches per call, not one per stream.  cfg->flags: H263MI_CFG_PIPELINE_POST makes every class frame-pipelined (the
 * RGBA of a picture is written by the launch that decodes the class's NEXT pictures, or -- when the class has none in
 * the next call -- at the end of that call; h263mi_mixed_sync delivers everything).
 *
 * h263mi_mixed_decode_next_pictures = N x H263State::decode_next_picture (state.rs:138-141), every stream its own state:
 *   data[s] / len[s]   one coded picture of stream s, or NULL: no picture for it in this call;
 *   stream_rc[s]       H263MI_OK or the stream's own error: a parse error, H263MI_ERR_UNCODED_IFRAME_BLOCKS (inter
 *                      macroblocks without a reference picture), H263MI_ERR_PICTURE_FORMAT_INVALID (a picture of another
 *                      size than the stream's last one that is not all intra: the reference indexes the new planes with the old
 *                      strides there, gather.rs:150,183), H263MI_ERR_INVALID_ARGUMENT (rgba_capacity[s] too small), a
 *                      back-end error of its class's launch.  A stream that fails keeps its state, parser state included
 *                      (state.rs:142); the others advance.  A picture of another size that is all intra MOVES the stream
 *                      to that size (its old picture is given up when the new one's launch is queued, not before);
 *   d_rgba[s]          (d_rgba may be NULL: no rendering) DEVICE buffer of rgba_capacity[s] >= w*h*4 bytes of the picture
 *                      being decoded, or NULL for that stream: deblock(strength) + BT.601 of the picture, tightly packed;
 *   descs[s]           (may be NULL) receives the header fields of the picture stream s decoded, its size included.
 * Returns H263MI_OK when the call itself went through (look at stream_rc), else a back-end error.  A back-end error of
 * the RENDERING of one class (deblock + BT.601 into d_rgba) does not stop the call: every class that has pictures is still
 * decoded -- stream_rc[s] == H263MI_OK always means "stream s decoded its picture and advanced", consumed[s] says so too --
 * and the first such error is returned at the end; the contents of d_rgba of that call are then unspecified, the pictures
 * themselves are intact (h263mi_mixed_copy_yuv, the next call's prediction).
 */
typedef struct h263mi_mixed h263mi_mixed;
int h263mi_mixed_create(uint32_t n_streams, const h263mi_backend_cfg *cfg, h263mi_mixed **out);
void h263mi_mixed_destroy(h263mi_mixed *m);
int h263mi_mixed_decode_next_pictures(h263mi_mixed *m, uint32_t decoder_options, const uint8_t *const *data,
                                      const size_t *len, size_t *consumed, uint32_t n_threads, int *stream_rc,
                                      uint8_t strength, uint8_t *const *d_rgba, const size_t *rgba_capacity,
                                      h263mi_picture_desc *descs);
/* (ABI 7) `strength` above may be H263MI_STRENGTH_FROM_HEADER (each picture with what its own header asks for, see
 * h263mi_batch_decode_next_pictures_ps); the _ps form also takes one value per stream of the SET (strengths[s], 0..12; NULL =
 * `strength`). */
int h263mi_mixed_decode_next_pictures_ps(h263mi_mixed *m, uint32_t decoder_options, const uint8_t *const *data,
                                         const size_t *len, size_t *consumed, uint32_t n_threads, int *stream_rc,
                                         uint8_t strength, const uint8_t *strengths, uint8_t *const *d_rgba,
                                         const size_t *rgba_capacity, h263mi_picture_desc *descs);
/* waits for everything queued; stream_rc (may be NULL): the device's verdict per stream, as h263mi_batch_sync_streams */
int h263mi_mixed_sync(h263mi_mixed *m, int *stream_rc);
/* size of stream `stream`'s last picture (H263MI_ERR_NO_PICTURE and 0 x 0 when it has none) */
int h263mi_mixed_stream_size(const h263mi_mixed *m, uint32_t stream, uint16_t *width, uint16_t *height);
/* number of size classes (fixed-geometry batches) alive.  A class no stream belongs to any more is given up when the next
 * class is made, so a stream that changes its size with every key frame does not make the set grow. */
uint32_t h263mi_mixed_size_classes(const h263mi_mixed *m);
/* Device memory of the set's frame stores.  A class holds two frames of its size per SLOT, and it has as many slots as it has
 * members, rounded up to a power of two (doubled when a stream joins a full class -- the members' frames move into the new
 * store --, halved when three quarters stand empty): 63 QCIF streams and one 1080p stream hold 64 x 2 QCIF frames and
 * 1 x 2 1080p frames.  Picture sizes come out of untrusted bitstreams: a picture whose class would take the frame stores of
 * the set beyond the limit (while a class grows, its old and its new store count both) is refused for its stream with
 * H263MI_ERR_OUT_OF_MEMORY (the stream keeps its state; classes nobody belongs to are given up first).  The default limit is
 * half of the device's memory; 0 = no limit. */
int h263mi_mixed_set_memory_limit(h263mi_mixed *m, uint64_t bytes);
uint64_t h263mi_mixed_frame_store_bytes(const h263mi_mixed *m);
/* DecodedPicture::as_yuv of stream `stream`'s last picture: tightly packed planes to HOST memory */
int h263mi_mixed_copy_yuv(h263mi_mixed *m, uint32_t stream, uint8_t *y, uint8_t *cb, uint8_t *cr);
/* H263State::new for one stream: it forgets its pictures and its size */
int h263mi_mixed_reset_stream(h263mi_mixed *m, uint32_t stream);
/* Create the events for `n_launches` timed kernel launches ahead of time, so that none is created inside a
 * timed region. */
int h263mi_batch_timing_reserve(h263mi_batch *b, uint32_t n_launches);

/* ======================================================================= */
/* Device memory + synthetic macroblock records (bench / test support)      */
/* ======================================================================= */
int h263mi_device_count(int *count);
int h263mi_device_malloc(int device_id, size_t bytes, void **out);
int h263mi_device_free(int device_id, void *p);
int h263mi_device_memcpy_h2d(int device_id, void *dst, const void *src, size_t bytes);
int h263mi_device_memcpy_d2h(int device_id, void *dst, const void *src, size_t bytes);
int h263mi_device_synchronize(int device_id);
/* TEST HOOK (tests/test_gpu_round4.py): the n-th HIP runtime call the host entry points make from now on (allocations,
 * copies, event operations, launches; n >= 1) is not executed and fails as out-of-memory; 0 or negative switches the hook
 * off.  Returns the calls still to go before the injected failure (<= 0: it has fired, or the hook is off).  Sweeping n
 * over one call proves that an error at any point leaves the state as it was (state.rs:142). */
int h263mi_debug_fail_nth_hip_call(int n);

/* On-box memory ceiling of the device (bench support, BASELINE.md section 4 "measure an on-box copy-kernel
 * ceiling"): streams `bytes` through a streaming kernel `reps` times on cfg's stream, in each of a few launch shapes
 * (non-temporal / plain 16-byte accesses, 1 to 8 in flight per lane, 256 to 4096 workgroups: the winners of the sweep
 * in tools/probes/ceiling.hip), and reports the rate of the fastest.  mode 0: copy (bytes read + bytes written are both
 * counted), 1: read only, 2: write only.  h263mi_probe_bandwidth_shape also names the shape that won. */
#define H263MI_PROBE_COPY  0
#define H263MI_PROBE_READ  1
#define H263MI_PROBE_WRITE 2
int h263mi_probe_bandwidth(const h263mi_backend_cfg *cfg, int mode, size_t bytes, int reps, double *gb_per_s);
int h263mi_probe_bandwidth_shape(const h263mi_backend_cfg *cfg, int mode, size_t bytes, int reps, double *gb_per_s,
                                 const char **shape_name);

#define H263MI_SYNTH_I_DENSE 0  /* BASELINE config 2 "dense": every block Full */
#define H263MI_SYNTH_I_MIXED 1  /* config 2 "mixed": Dc / Horiz / Vert / Full-dense / Full-sparse */
#define H263MI_SYNTH_P       2  /* config 3: half-pel MVs in [-32,31], 25 % coded blocks, quant 10 */
/* Counter-based splitmix64 records; host and device versions are bit-identical. */
int h263mi_synth_picture_host(int kind, uint16_t width, uint16_t height,
                              uint32_t stream_id, uint32_t frame_idx,
                              h263mi_mb_record *mbs, int16_t *coeffs,
                              size_t coeff_capacity_blocks, size_t *n_coeff_blocks);
/* n_streams pictures (stream ids first_stream_id ..) straight into DEVICE memory.
 * d_coeff_base receives each picture's base (blocks) in d_coeffs; *total_blocks the pool use. */
int h263mi_synth_batch_device(const h263mi_backend_cfg *cfg, int kind, uint16_t width, uint16_t height,
                              uint32_t n_streams, uint32_t first_stream_id, uint32_t frame_idx,
                              h263mi_mb_record *d_mbs, int16_t *d_coeffs, size_t coeff_capacity_blocks,
                              uint64_t *d_coeff_base, size_t *total_blocks);
/* (ABI 5) ... stream ids first_stream_id, first_stream_id + stream_stride, ...: the streams ONE rank of a job owns when
 * stream s of T is pinned to GPU s mod n_gpu (bench.py --total-streams). */
int h263mi_synth_batch_device_strided(const h263mi_backend_cfg *cfg, int kind, uint16_t width, uint16_t height,
                                      uint32_t n_streams, uint32_t first_stream_id, uint32_t stream_stride,
                                      uint32_t frame_idx, h263mi_mb_record *d_mbs, int16_t *d_coeffs,
                                      size_t coeff_capacity_blocks, uint64_t *d_coeff_base, size_t *total_blocks);

#ifdef __cplusplus
}
#endif
#endif /* H263MI_H */
