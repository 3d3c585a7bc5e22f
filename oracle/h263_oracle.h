/*
 * h263_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the hot path of ruffle-rs/h263-rs (reference mounted
 * at /root/reference when this was written).  Only tests/, bench.py's
 * `cpu_baseline` leg and __graft_entry__.smoke() may link or call this file;
 * the shipped library (h263-rs_amd/) never does.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - orc_deblock            : PINNED by the reference's own known-answer tests
 *                              (deblock/src/deblock.rs:320-558, tests/golden/deblock_*.json)
 *   - orc_yuv420_to_rgba     : PINNED by the reference's own known-answer tests
 *                              (yuv/src/bt601.rs:198-483, tests/golden/bt601_*.json)
 *   - orc_inverse_rle / orc_idct_channel / orc_gather / orc_decode_picture :
 *                              PARITY UNPINNED -- the reference holds no test for
 *                              these functions and its Rust toolchain is absent here;
 *                              cross-checked against an independent numpy restatement
 *                              (oracle/np_restatement.py), an FPU-free soft-float model, hand-
 *                              derivable identities, and -- from the other side -- ITU-T H.263
 *                              itself (tests/test_oracle_vs_standard.py: the IDCT accuracy
 *                              specification of Annex A, the prediction formulas of 6.1.2).
 */
#ifndef H263_ORACLE_H
#define H263_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- h263/src/types.rs:902-916  DecodedDctBlock ------------------------- */
enum { ORC_ZERO = 0, ORC_DC = 1, ORC_HORIZ = 2, ORC_VERT = 3, ORC_FULL = 4 };

typedef struct {
    int32_t tag;       /* ORC_* */
    float   v[64];     /* Dc: v[0]; Horiz/Vert: v[0..7]; Full: v[y*8+x] */
} orc_dct_block;

/* ---- h263/src/types.rs:887-893,971-986  Block / TCoefficient ------------ */
typedef struct {
    int32_t  has_intradc;   /* Option<IntraDc> discriminant */
    uint8_t  intradc;       /* raw FLC code (types.rs:923-961) */
    int32_t  n_tcoef;
    uint8_t  run[80];
    int16_t  level[80];
} orc_block;

/* ---- macroblock record crossing the C ABI (include/h263mi.h) ------------ */
typedef struct {
    uint8_t  mb_type;       /* 0 Inter 1 InterQ 2 Inter4V 3 Intra 4 IntraQ 5 Inter4Vq */
    uint8_t  quant;
    uint8_t  cbp;
    uint8_t  kill;
    int16_t  mv[4][2];
    uint8_t  intradc[6];
    uint8_t  reserved[2];
    uint32_t coeff_index;
} orc_mb_record;

/* error codes shared with include/h263mi.h */
#define ORC_OK                        0
#define ORC_ERR_UNCODED_IFRAME_BLOCKS (-15)
#define ORC_ERR_INVALID_ARGUMENT      (-100)

int16_t orc_intradc_into_level(uint8_t code);                 /* types.rs:955-961 */
void    orc_lerp_parameters(int16_t halfpel, int16_t *delta, int *interp); /* types.rs:721-729 */
int16_t orc_average_sum_of_mvs(int16_t sum);                  /* types.rs:759-768 */

/* h263/src/decoder/cpu/rle.rs:82-172 */
void orc_inverse_rle(const orc_block *blk, orc_dct_block *levels,
                     size_t pos_x, size_t pos_y, size_t blk_per_line, uint8_t quant);

/* h263/src/decoder/cpu/idct.rs:82-201 */
void orc_idct_channel(const orc_dct_block *levels, size_t n_levels,
                      uint8_t *output, size_t output_len,
                      size_t blk_per_line, size_t samples_per_line);

/* h263/src/decoder/cpu/gather.rs:140-204; ref_* may be NULL (no reference picture) */
int orc_gather(const uint8_t *mb_types, const int16_t (*mvs)[4][2], size_t n_mbs,
               const uint8_t *ref_y, const uint8_t *ref_cb, const uint8_t *ref_cr,
               size_t width, size_t height, size_t mb_per_line,
               uint8_t *new_y, uint8_t *new_cb, uint8_t *new_cr);

/* Record-level picture reconstruction = the tail of decode_next_picture
 * (h263/src/decoder/state.rs:173-191, 421-458), fed by macroblock records. */
int orc_decode_picture(uint16_t width, uint16_t height,
                       const orc_mb_record *mbs, size_t n_mbs,
                       const int16_t *coeffs, size_t n_coeff_blocks,
                       const uint8_t *ref_y, const uint8_t *ref_cb, const uint8_t *ref_cr,
                       uint8_t *out_y, uint8_t *out_cb, uint8_t *out_cr);

/* deblock/src/deblock.rs:305-315 (+ 136-299, 29-42, 99-127) */
extern const uint8_t orc_quant_to_strength[32];               /* deblock.rs:5-8 */
void orc_deblock_process_scalar(uint8_t *a, uint8_t *b, uint8_t *c, uint8_t *d, uint8_t strength);
void orc_deblock_process_simd_lane(uint8_t *a, uint8_t *b, uint8_t *c, uint8_t *d, uint8_t strength);
int  orc_deblock(const uint8_t *data, size_t len, size_t width, uint8_t strength, uint8_t *out);

/* yuv/src/bt601.rs:105-196 (+ 12-59) */
int orc_yuv420_to_rgba(const uint8_t *y, size_t y_len,
                       const uint8_t *cb, const uint8_t *cr, size_t c_len,
                       size_t y_width, uint8_t *rgba_out);

#ifdef __cplusplus
}
#endif
#endif
