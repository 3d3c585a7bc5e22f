// bitstream.hpp -- host-side Sorenson Spark / H.263 bitstream parser that emits macroblock records
// (SURVEY section 8 row f-1).  This is the serial half of H263State::decode_next_picture
// (h263/src/decoder/state.rs:138-427): bit reader (parser/reader.rs), picture layer (parser/picture.rs:
// 611-661, 271-327, 577-596), macroblock layer (parser/macroblock.rs:445-549), block layer
// (parser/block.rs:670-755) and motion vector prediction (decoder/cpu/mvd_pred.rs:27-134).  Instead
// of DecodedDctBlock enums it writes the h263mi_mb_record array + dense coefficient blocks that cross
// the C ABI; no pixel arithmetic happens here.
//
// Differences from the reference implementation (not from its behaviour): a 64-bit bit buffer over a
// byte span instead of a VecDeque fed one byte at a time, and table-driven multi-bit VLC decoding
// instead of a bit-at-a-time tree walk.  Error kinds (EOF vs invalid code) are preserved because the
// state machine branches on them (state.rs:387-412).
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/h263mi.h"

namespace h263mi {
namespace bits {

// ---- reader (parser/reader.rs) ----------------------------------------------------------------
class BitReader {
public:
    BitReader(const uint8_t *data, size_t len) : p_(data), nbits_(len * 8), pos_(0) {}
    size_t position() const { return pos_; }          // checkpoint()  reader.rs:361-363
    void rollback(size_t pos) { pos_ = pos; }         // rollback()    reader.rs:374-382
    size_t remaining() const { return nbits_ - pos_; }

    // up to 32 bits starting at the cursor, zero-padded past the end of the data; does not advance
    uint32_t peek_padded(uint32_t n) const;
    // peek_bits / read_bits / read_signed_bits (reader.rs:94-213): H263MI_ERR_UNHANDLED_IO_ERROR (EOF)
    // when fewer than n bits remain, cursor unchanged
    int peek_bits(uint32_t n, uint32_t &out) const;
    int read_bits(uint32_t n, uint32_t &out);
    int read_signed_bits(uint32_t n, int32_t &out);
    int skip_bits(uint32_t n);
    int read_u8(uint32_t &out) { return read_bits(8, out); }
    // recognize_start_code (reader.rs:244-262): *skipped = bits in front of the start code, or -1 for None
    int recognize_start_code(bool in_error, int &skipped) const;

private:
    const uint8_t *p_;
    size_t nbits_, pos_;
};

// ---- variable length codes (parser/vlc.rs, tables of ITU-T H.263 (01/2005)) -----------------------
struct VlcCode {
    const char *bits;      // code word, MSB first
    int16_t v0, v1, v2;    // table specific payload
};
struct VlcHit {
    bool valid;            // false: the bits read so far cannot start any code word
    int16_t v0, v1, v2;
};
class VlcTable {
public:
    VlcTable(const VlcCode *codes, size_t n);
    // read_vlc (reader.rs:272-290): consumes exactly the bits a bit-by-bit walk would have consumed;
    // EOF if the data ends inside a code word
    int decode(BitReader &r, VlcHit &hit) const;
    int max_len() const { return max_len_; }

private:
    struct Slot { uint8_t len; uint8_t valid; int16_t v0, v1, v2; };
    std::vector<Slot> lut_;
    int max_len_;
};

const VlcTable &tcoef_table();     // Table 16/H.263: v0 = last, v1 = run, v2 = level; escape: v0 = -1
const VlcTable &mcbpc_i_table();   // Table 7: v0 = MacroblockType, v1 = codes Cb, v2 = codes Cr; stuffing: v0 = -1
const VlcTable &mcbpc_p_table();   // Table 8
const VlcTable &cbpy_table();      // Table 13 (intra sense): v0 = 4-bit pattern, bit 3 = first luma block
const VlcTable &mvd_table();       // Table 14: v0 = vector in half-pel units

// ---- one coded block (parser/block.rs:670-755) -----------------------------------------------------
struct ParsedBlock {
    bool has_intradc = false;
    uint8_t intradc = 0;                   // raw FLC code
    int n_tcoef = 0;
    struct { uint8_t is_short; uint8_t run; int16_t level; } tcoef[64 + 8];
};
int decode_block(BitReader &r, bool sorenson, int version, bool intra, bool tcoef_present, ParsedBlock &out);

// ---- picture header (parser/picture.rs:611-661, Sorenson branch) -------------------------------------
struct PictureHeader {
    int version = 0;                       // Sorenson keeps it where H.263 has the GOB number
    uint16_t temporal_reference = 0;
    uint16_t width = 0, height = 0;
    bool format_valid = false;             // SourceFormat::Reserved has no dimensions
    uint8_t picture_type = 0;              // 0 I, 1 P, 2 disposable P, 3 reserved
    bool use_deblocker = false;
    uint8_t quantizer = 0;
    std::vector<uint8_t> extra;            // PEI / PSUPP bytes
};
// decode_picture(reader, options, prev): H263MI_OK with *is_picture = false when a GOB start was found
int decode_picture_header(BitReader &r, uint32_t decoder_options, PictureHeader &out, bool &is_picture);

// ---- whole picture -> records (state.rs:138-427) ------------------------------------------------------
struct ParsedPicture {
    h263mi_picture_desc desc{};
    std::vector<h263mi_mb_record> mbs;     // the macroblocks present in the bitstream (<= mbw*mbh)
    std::vector<int16_t> coeffs;           // 64 per coded block, raster order
    size_t bits_consumed = 0;
};
// Returns H263MI_OK or the error the reference's decode_next_picture would return before touching any
// pixel.  `have_last_format`: the last decoded picture's size, used when a header carries no format.
int parse_picture(const uint8_t *data, size_t len, uint32_t decoder_options, ParsedPicture &out);

}  // namespace bits
}  // namespace h263mi
