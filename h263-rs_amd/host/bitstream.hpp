// bitstream.hpp -- host-side Sorenson Spark / H.263 bitstream parser that emits macroblock records
// (SURVEY section 8 rows f-1 and f-4).  This is the serial half of H263State::decode_next_picture
// (h263/src/decoder/state.rs:138-427): bit reader (parser/reader.rs), picture layer (parser/picture.rs:
// 21-661), GOB resynchronisation stub (parser/gob.rs:20-41), macroblock layer (parser/macroblock.rs:445-549), block layer
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
    // the bits at the cursor in the top of a 64-bit word, zero-padded past the end; at least the leading
    // 57 bits (64 - cursor % 8) belong to the stream
    uint64_t peek_window() const { return window_at(pos_); }
    // the same at any bit position (hot loops keep their own cursor in a register and store it back with rollback())
    size_t size_bits() const { return nbits_; }
    const uint8_t *data() const { return p_; }
    uint64_t window_at(size_t pos) const
    {
        const size_t byte = pos >> 3, nbytes = nbits_ >> 3;
        uint64_t window;
        if (byte + 8 <= nbytes) {
            __builtin_memcpy(&window, p_ + byte, 8);          // one unaligned load + byte swap
            window = __builtin_bswap64(window);
        } else {
            window = 0;
            for (size_t k = 0; k < 8; k++) window = (window << 8) | (byte + k < nbytes ? p_[byte + k] : 0);
        }
        return window << (pos & 7);
    }
    // peek_bits / read_bits / read_signed_bits (reader.rs:94-213): H263MI_ERR_UNHANDLED_IO_ERROR (EOF)
    // when fewer than n bits remain, cursor unchanged
    int peek_bits(uint32_t n, uint32_t &out) const;
    int read_bits(uint32_t n, uint32_t &out);
    int read_signed_bits(uint32_t n, int32_t &out);
    int skip_bits(uint32_t n);
    int read_u8(uint32_t &out) { return read_bits(8, out); }
    // the caller has checked remaining() >= n
    void advance(uint32_t n) { pos_ += n; }
    // recognize_start_code (reader.rs:244-262): *skipped = bits in front of the start code, or -1 for None
    int recognize_start_code(bool in_error, int &skipped) const;
    // read_umv (reader.rs:298-324): unrestricted motion vector component, half-pel units
    int read_umv(int &out);

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
    struct Slot { uint8_t len; uint8_t valid; int8_t v0, v1, v2; };
    // the slot of the code that starts a 32-bit window of the bitstream (the bits beyond the end of the data, if
    // any, are zeros: the caller compares Slot::len with what remains).  Two levels: the first is indexed by the
    // leading 8 bits (256 entries, always in L1) and resolves every code word of up to 8 bits -- the frequent ones;
    // longer code words continue in a small per-prefix table indexed by the remaining max_len - 8 bits.  (One flat
    // table of 2^13 entries for TCOEF was 40 KB with accesses spread all over it.)
    const Slot &lookup32(uint32_t window) const
    {
        const First &f = first_[window >> (32 - first_bits_)];
        if (f.direct) return f.slot;
        return second_[f.sub + ((window >> (32 - max_len_)) & sub_mask_)];
    }

private:
    struct First { Slot slot; uint8_t direct; uint16_t sub; };
    std::vector<First> first_;
    std::vector<Slot> second_;
    int max_len_, first_bits_;
    uint32_t sub_mask_;
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
    struct { uint8_t is_short; uint8_t run; int16_t level; } tcoef[64 + 8];   // [0, n_tcoef) valid, the rest is not initialised
};
int decode_block(BitReader &r, bool sorenson, int version, bool intra, bool tcoef_present, ParsedBlock &out);

// ---- picture header (parser/picture.rs:21-661: PTYPE, PLUSPTYPE and the Sorenson variant) ----------------
// PictureOption (types.rs:195-216)
enum : uint32_t {
    OPT_USE_SPLIT_SCREEN = 1u << 0,
    OPT_USE_DOCUMENT_CAMERA = 1u << 1,
    OPT_RELEASE_FULL_PICTURE_FREEZE = 1u << 2,
    OPT_UNRESTRICTED_MOTION_VECTORS = 1u << 3,
    OPT_SYNTAX_BASED_ARITHMETIC_CODING = 1u << 4,
    OPT_ADVANCED_PREDICTION = 1u << 5,
    OPT_ADVANCED_INTRA_CODING = 1u << 6,
    OPT_DEBLOCKING_FILTER = 1u << 7,
    OPT_SLICE_STRUCTURED = 1u << 8,
    OPT_REFERENCE_PICTURE_SELECTION = 1u << 9,
    OPT_INDEPENDENT_SEGMENT_DECODING = 1u << 10,
    OPT_ALTERNATIVE_INTER_VLC = 1u << 11,
    OPT_MODIFIED_QUANTIZATION = 1u << 12,
    OPT_REFERENCE_PICTURE_RESAMPLING = 1u << 13,
    OPT_REDUCED_RESOLUTION_UPDATE = 1u << 14,
    OPT_ROUNDING_TYPE_ONE = 1u << 15,
    OPT_USE_DEBLOCKER = 1u << 16,            // Sorenson only
};
// types.rs:223-240
constexpr uint32_t OPPTYPE_OPTIONS = OPT_UNRESTRICTED_MOTION_VECTORS | OPT_SYNTAX_BASED_ARITHMETIC_CODING |
                                     OPT_ADVANCED_PREDICTION | OPT_ADVANCED_INTRA_CODING | OPT_DEBLOCKING_FILTER |
                                     OPT_SLICE_STRUCTURED | OPT_REFERENCE_PICTURE_SELECTION |
                                     OPT_INDEPENDENT_SEGMENT_DECODING | OPT_ALTERNATIVE_INTER_VLC |
                                     OPT_MODIFIED_QUANTIZATION;
constexpr uint32_t MPPTYPE_OPTIONS = OPT_REFERENCE_PICTURE_RESAMPLING | OPT_REDUCED_RESOLUTION_UPDATE | OPT_ROUNDING_TYPE_ONE;

// SourceFormat (types.rs:120-181) with Option<> folded in; equality is the derived PartialEq of the reference
// (Extended compares aspect ratio and both indications, SubQcif != Extended(128x96))
struct SourceFormat {
    enum Kind : uint8_t { NONE = 0, SUB_QCIF, QCIF, CIF, FOUR_CIF, SIXTEEN_CIF, RESERVED, EXTENDED };
    uint8_t kind = NONE;
    uint8_t par = 0;                       // PixelAspectRatio: 1 square .. 5 40:33, 15 extended, other = Reserved(par)
    uint8_t par_width = 0, par_height = 0; // EPAR
    uint16_t width = 0, height = 0;        // picture_width_indication / picture_height_indication
    bool operator==(const SourceFormat &o) const
    {
        if (kind != o.kind) return false;
        if (kind != EXTENDED) return true;
        if (par != o.par || width != o.width || height != o.height) return false;
        return par != 15 || (par_width == o.par_width && par_height == o.par_height);
    }
    bool operator!=(const SourceFormat &o) const { return !(*this == o); }
    // into_width_and_height (types.rs:168-180); false for Reserved (and for NONE)
    bool dimensions(uint16_t &w, uint16_t &h) const;
};

// PictureTypeCode (types.rs:251-288) as carried in h263mi_picture_desc.picture_type: 0..3 are the values of
// the Sorenson 2-bit field, the rest only exist in standard H.263 headers
enum : uint8_t { PT_PB = H263MI_PICTURE_PB, PT_IMPROVED_PB = H263MI_PICTURE_IMPROVED_PB, PT_B = H263MI_PICTURE_B,
                 PT_EI = H263MI_PICTURE_EI, PT_EP = H263MI_PICTURE_EP, PT_RESERVED = H263MI_PICTURE_RESERVED };

struct PictureHeader {
    int version = -1;                      // Sorenson keeps it where H.263 has the GOB number; -1 = None
    uint16_t temporal_reference = 0;
    SourceFormat format;                   // Option<SourceFormat>: kind NONE when the header restates none
    uint16_t width = 0, height = 0;        // of `format`, when it has dimensions
    bool format_valid = false;
    uint32_t options = 0;                  // PictureOption bits
    bool has_plusptype = false, has_opptype = false;
    uint8_t picture_type = 0;              // 0 I, 1 P, 2 disposable P, 3 reserved (Sorenson), PT_* above
    bool use_deblocker = false;            // OPT_USE_DEBLOCKER
    uint8_t mv_range = 0;                  // MotionVectorRange: 0 None, 1 Extended, 2 Unlimited
    uint8_t quantizer = 0;
    std::vector<uint8_t> extra;            // PEI / PSUPP bytes
};

// What decode_next_picture takes from the state about the last decoded picture (state.rs:143-167):
// its header (`previous_picture`) and its resolved format.
struct ParserContext {
    bool have_last = false;
    SourceFormat last_header_format;       // last_picture.as_header().format
    uint32_t last_header_options = 0;      // last_picture.as_header().options
    SourceFormat last_format;              // last_picture.format()
};

// decode_picture(reader, options, previous_picture): H263MI_OK with *is_picture = false when a GOB start
// was found.  `prev` may be null (no previous picture).
int decode_picture_header(BitReader &r, uint32_t decoder_options, const ParserContext *prev, PictureHeader &out,
                          bool &is_picture);

// A std::vector whose resize() does not zero-fill: the parser's word arrays are sized for the worst case of a picture up
// front (200 KB for the block offsets of a 1080p picture) and every element that counts is written before it is read.
template <class T>
struct DefaultInitAlloc : std::allocator<T> {
    template <class U> struct rebind { typedef DefaultInitAlloc<U> other; };
    DefaultInitAlloc() = default;
    template <class U> DefaultInitAlloc(const DefaultInitAlloc<U> &) {}
    template <class U> void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
typedef std::vector<uint32_t, DefaultInitAlloc<uint32_t>> WordBuffer;

// ---- whole picture -> records (state.rs:138-427) ------------------------------------------------------
typedef std::vector<h263mi_mb_record, DefaultInitAlloc<h263mi_mb_record>> RecordBuffer;

struct ParsedPicture {
    h263mi_picture_desc desc{};
    RecordBuffer mbs;                      // the macroblocks present in the bitstream (<= mbw*mbh); see sparse_records
    // Optional destination of the records (set before parsing; e.g. a slot of pinned staging memory): when the picture
    // has at most mbs_ext_cap macroblocks the records are written there instead of into `mbs` (which stays empty), and
    // n_mbs_ext says how many.  A picture with more macroblocks than that uses `mbs` as usual (mbs_ext_used = false).
    h263mi_mb_record *mbs_ext = nullptr;
    size_t mbs_ext_cap = 0, n_mbs_ext = 0;
    bool mbs_ext_used = false;
    const h263mi_mb_record *records() const { return mbs_ext_used ? mbs_ext : mbs.data(); }
    size_t n_records() const { return mbs_ext_used ? n_mbs_ext : mbs.size(); }
    std::vector<int16_t> coeffs;           // 64 per coded block, raster order (only with want_dense)
    // the same coefficients as events, level << 16 | raster position, block k = [block_first_event[k], [k+1])
    WordBuffer block_first_event, events;
    bool want_dense = true;                // set to false before parsing to skip the dense blocks
    // SPARSE RECORDS (round 5; set before parsing): two thirds of the macroblocks of a real P
    // picture are not coded (COD = 1: INTER, zero vectors, nothing coded, state.rs:207-216) and their 32-byte records say
    // nothing -- 177 of the 261 KB a 1080p picture sends over the link.  With this set, records() holds the records of the OTHER
    // macroblocks only, in raster order, and `group_index` one word per group of 8 macroblocks of a row (the unit a
    // reconstruction wave works on; groups per row = ceil(mbw / 8)): first << 8 | mask -- bit k of `mask`: macroblock k of the
    // group has a record, `first`: the number of its first record in records().  A macroblock without a record is not coded.
    // (Macroblocks the bitstream does not reach have none either: they are padded as not coded, state.rs:421-427.)
    bool sparse_records = false;
    WordBuffer group_index;
    // Optional destinations of the WORD arrays (set before parsing; e.g. a stream's part of a pinned staging slot): when all
    // of them are given and large enough for the worst case of this picture -- events: event_words_bound(len, macroblocks);
    // block offsets: block_offset_words_bound(macroblocks); index: one word per group -- the events, the block offsets and the
    // group index are written there and the vectors above stay empty (words_ext_used; n_events_ext says how many events).
    // The block offsets then count from `event_base` (block k = [first[k], first[k+1]) - event_base of events_ext): a caller
    // that lays the streams' parts out one behind the other needs no second pass over any of it.
    uint32_t *events_ext = nullptr, *first_event_ext = nullptr, *group_index_ext = nullptr;
    size_t events_ext_cap = 0, first_event_ext_cap = 0, group_index_ext_cap = 0, n_events_ext = 0;
    uint32_t event_base = 0;
    bool words_ext_used = false;
    const uint32_t *event_words() const { return words_ext_used ? events_ext : events.data(); }
    size_t n_event_words() const { return words_ext_used ? n_events_ext : events.size(); }
    const uint32_t *first_event_words() const { return words_ext_used ? first_event_ext : block_first_event.data(); }
    const uint32_t *group_index_words() const { return words_ext_used ? group_index_ext : group_index.data(); }
    size_t n_macroblocks = 0;              // macroblocks the bitstream held (records + the ones without one)
    // some macroblock takes a prediction (inter type, not coded, or not reached by the bitstream): the picture needs a
    // reference picture (gather.rs:149).  Set by the parser in every mode.
    bool any_inter = false;
    // The caller's limit on the picture size (null = none), asked right behind the header -- before any of the arrays below is
    // sized for the picture: a Sorenson custom format carries 16-bit dimensions out of an untrusted bitstream, and 65 535 x
    // 65 535 is 16.7 M macroblocks (1.2 GB of records, vectors and block offsets) whether or not any data follows.  A picture
    // beyond the limit is H263MI_ERR_PICTURE_FORMAT_INVALID.
    bool (*size_fits)(uint32_t width, uint32_t height) = nullptr;
    // Test switch: read every field on its own, with its own end-of-data check -- the transcription of the reference's
    // parser that defines the behaviour -- instead of the windowed fast paths.  tests/test_parser_paths.py holds the two
    // against each other on valid, truncated and corrupted streams.
    bool field_by_field = false;
    size_t n_coded_blocks = 0;
    size_t bits_consumed = 0;
    WordBuffer scratch;                    // parser-internal (the vectors of the macroblocks decoded so far)
    ParserContext next;                    // the context once this picture has been decoded successfully
};
// The most event words a picture of `len` bytes and `macroblocks` macroblocks can make the parser write: an event is a
// code word of two bits at least and a sign (block.rs:689-724), a block places 64 at most, and the parser wants room for one
// whole macroblock (6 x 64) in front of its cursor.
inline size_t event_words_bound(size_t len, size_t macroblocks)
{
    const size_t by_bits = len * 8 / 3 + 1, by_blocks = macroblocks * 6 * 64;
    return (by_bits < by_blocks ? by_bits : by_blocks) + 6 * 64 + 8;
}
inline size_t block_offset_words_bound(size_t macroblocks) { return (macroblocks + 1) * 6 + 1; }
// Returns H263MI_OK or the error the reference's decode_next_picture would return before touching any
// pixel.  `ctx`: the last decoded picture (null = none), needed by standard H.263 headers that carry no format.
int parse_picture(const uint8_t *data, size_t len, uint32_t decoder_options, const ParserContext *ctx, ParsedPicture &out);

}  // namespace bits
}  // namespace h263mi
