// bitstream.cpp -- see bitstream.hpp.  Plain C++17 host code (no HIP): it is linked into libh263mi.so
// and, separately, into the CPU-only parser test library (tests/parser).
#include "bitstream.hpp"

#include <cstdlib>
#include <cstring>
// Records are written with non-temporal stores where the target has them (see store_record in parse_picture);
// -DH263MI_NO_NT_RECORDS: plain stores
#if !defined(H263MI_NO_NT_RECORDS) && defined(__SSE2__)
#include <emmintrin.h>
#define H263MI_STREAM_RECORDS 1
#else
#define H263MI_STREAM_RECORDS 0
#endif

namespace h263mi {
namespace bits {

namespace {
constexpr int kEof = H263MI_ERR_UNHANDLED_IO_ERROR;

// rle.rs:6-71 DEZIGZAG_MAPPING as raster index x + 8*y per zigzag position
const uint8_t kZigzagRaster[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
}  // namespace

// ---------------------------------------------------------------------------------------------------
// BitReader
// ---------------------------------------------------------------------------------------------------
uint32_t BitReader::peek_padded(uint32_t n) const
{
    if (!n) return 0u;
    return (uint32_t)(peek_window() >> (64 - n));               // n <= 32 <= 57
}

int BitReader::peek_bits(uint32_t n, uint32_t &out) const
{
    if (n > 32) return H263MI_ERR_INTERNAL_DECODER_ERROR;      // reader.rs:95-97
    if (n > remaining()) return kEof;
    out = peek_padded(n);
    return H263MI_OK;
}

int BitReader::read_bits(uint32_t n, uint32_t &out)
{
    int rc = peek_bits(n, out);
    if (rc == H263MI_OK) pos_ += n;
    return rc;
}

int BitReader::read_signed_bits(uint32_t n, int32_t &out)
{
    uint32_t v;
    int rc = read_bits(n, v);
    if (rc != H263MI_OK) return rc;
    out = (n < 32 && (v >> (n - 1))) ? (int32_t)(v | (~0u << n)) : (int32_t)v;   // reader.rs:176-187
    return H263MI_OK;
}

int BitReader::skip_bits(uint32_t n)
{
    if (n > remaining()) return kEof;
    pos_ += n;
    return H263MI_OK;
}

int BitReader::recognize_start_code(bool in_error, int &skipped) const
{
    // reader.rs:244-262.  A start code is 16 zero bits and a one; up to (bits to the next byte boundary)
    // stuffing bits may precede it -- and, as in the reference's loop, one more than that.
    const uint32_t max_skip = (8 - (uint32_t)(pos_ & 7)) % 8;
    BitReader look = *this;
    uint32_t skip = 0, code;
    int rc = look.peek_bits(17, code);
    if (rc != H263MI_OK) return rc;
    while (code != 1) {
        if (!in_error && skip > max_skip) {
            skipped = -1;
            return H263MI_OK;
        }
        if ((rc = look.skip_bits(1)) != H263MI_OK) return rc;
        skip++;
        if ((rc = look.peek_bits(17, code)) != H263MI_OK) return rc;
    }
    skipped = (int)skip;
    return H263MI_OK;
}

int BitReader::read_umv(int &out)
{
    // reader.rs:298-324 (Table D.3/H.263): "1" = 0; otherwise pairs of bits: x0 continues with mantissa bit x,
    // 00 ends positive, 10 ends negative; magnitudes of 4096 and above are an error
    uint32_t v;
    int rc = read_bits(1, v);
    if (rc != H263MI_OK) return rc;
    if (v == 1) { out = 0; return H263MI_OK; }
    int mantissa = 0, bulk = 1;
    while (bulk < 4096) {
        if ((rc = read_bits(2, v)) != H263MI_OK) return rc;
        switch (v) {
        case 0: out = mantissa + bulk; return H263MI_OK;
        case 2: out = -(mantissa + bulk); return H263MI_OK;
        case 1: mantissa <<= 1; break;
        default: mantissa = (mantissa << 1) | 1; break;
        }
        bulk <<= 1;
    }
    return H263MI_ERR_INVALID_MVD;
}

// ---------------------------------------------------------------------------------------------------
// VLC tables
// ---------------------------------------------------------------------------------------------------
VlcTable::VlcTable(const VlcCode *codes, size_t n) : max_len_(0), first_bits_(0), sub_mask_(0)
{
    constexpr int kFirstBits = 8;
    for (size_t i = 0; i < n; i++) {
        int l = (int)strlen(codes[i].bits);
        if (l > max_len_) max_len_ = l;
    }
    // (payloads fit the 8-bit slot fields: checked at compile time, see H263MI_TABLE)
    // Flat table first.  For every max_len_-bit pattern: the length at which a bit-by-bit walk of the code tree
    // stops -- either on a code word, or on the shortest prefix that no code word starts with (the tree's
    // "invalid" leaves).
    std::vector<Slot> flat((size_t)1 << max_len_, Slot{0, 0, 0, 0, 0});
    for (uint32_t pat = 0; pat < flat.size(); pat++) {
        for (int l = 1; l <= max_len_; l++) {
            const uint32_t prefix = pat >> (max_len_ - l);
            bool is_code = false, extendable = false;
            size_t which = 0;
            for (size_t i = 0; i < n && !is_code; i++) {
                const int cl = (int)strlen(codes[i].bits);
                if (cl < l) continue;
                uint32_t cv = 0;
                for (int k = 0; k < cl; k++) cv = (cv << 1) | (uint32_t)(codes[i].bits[k] - '0');
                if ((cv >> (cl - l)) == prefix) {
                    extendable = true;
                    if (cl == l) { is_code = true; which = i; }
                }
            }
            if (is_code) {
                flat[pat] = Slot{(uint8_t)l, 1, (int8_t)codes[which].v0, (int8_t)codes[which].v1, (int8_t)codes[which].v2};
                break;
            }
            if (!extendable) {
                flat[pat] = Slot{(uint8_t)l, 0, 0, 0, 0};
                break;
            }
        }
    }
    // Two levels out of it: a prefix of first_bits_ bits whose walks all stop within the prefix gets its slot in the
    // first level; the others get a sub-table of the remaining bits.
    first_bits_ = max_len_ < kFirstBits ? max_len_ : kFirstBits;
    const int rest = max_len_ - first_bits_;
    sub_mask_ = (1u << rest) - 1u;
    first_.resize((size_t)1 << first_bits_);
    for (uint32_t p = 0; p < first_.size(); p++) {
        const Slot *run = &flat[(size_t)p << rest];
        bool direct = true;
        for (uint32_t k = 0; k <= sub_mask_; k++) direct = direct && run[k].len <= first_bits_;
        first_[p].direct = direct ? 1 : 0;
        first_[p].slot = run[0];
        first_[p].sub = 0;
        if (!direct) {
            first_[p].sub = (uint16_t)second_.size();
            second_.insert(second_.end(), run, run + sub_mask_ + 1);
        }
    }
}

int VlcTable::decode(BitReader &r, VlcHit &hit) const
{
    const Slot &s = lookup32(r.peek_padded(32));
    if (s.len > r.remaining()) {
        // the data ends inside the code word: the reference reads bit by bit and fails on the missing bit
        (void)r.skip_bits((uint32_t)r.remaining());
        return kEof;
    }
    r.advance(s.len);
    hit.valid = s.valid != 0;
    hit.v0 = s.v0; hit.v1 = s.v1; hit.v2 = s.v2;
    return H263MI_OK;
}

#include "vlc_tables.inc"

// the LUT slots keep the payloads in 8 bits: types, runs <= 40, levels <= 12, vectors +-32 all fit (checked here,
// at compile time, for every generated table)
template <size_t N> static constexpr bool payloads_fit_i8(const VlcCode (&codes)[N])
{
    for (size_t i = 0; i < N; i++)
        if (codes[i].v0 < -128 || codes[i].v0 > 127 || codes[i].v1 < -128 || codes[i].v1 > 127 || codes[i].v2 < -128 ||
            codes[i].v2 > 127)
            return false;
    return true;
}

#define H263MI_TABLE(fn, arr)                                                   \
    static_assert(payloads_fit_i8(arr), "VLC payload does not fit a LUT slot"); \
    const VlcTable &fn()                                                        \
    {                                                                           \
        static const VlcTable t(arr, sizeof(arr) / sizeof(arr[0]));             \
        return t;                                                               \
    }
H263MI_TABLE(tcoef_table, kTcoefCodes)
H263MI_TABLE(mcbpc_i_table, kMcbpcICodes)
H263MI_TABLE(mcbpc_p_table, kMcbpcPCodes)
H263MI_TABLE(cbpy_table, kCbpyCodes)
H263MI_TABLE(mvd_table, kMvdCodes)

// ---- combined look-ups for the macroblock header windows (round 5) -----------------------------------------------
// The header of a coded macroblock is a chain of code words whose positions depend on each other: MCBPC, then CBPY, then
// (DQUANT and) the vector difference pairs -- one dependent table access after the other (25 ns per inter macroblock, half
// of the parse time of a real P picture: profiles/r04_q_parser_stage_costs.txt).  Two tables made of the code tables above
// take two links out of the chain, for the short code words that make up nearly all of a real stream; everything they do
// not resolve (entry 0) goes the way of the single code tables, which is the definition of the behaviour
// (macroblock.rs:445-549):
//   header12[12 bits behind COD]   MCBPC + CBPY when both lie inside 12 bits: len | type << 4 | coded6 << 7 (coded6: bit
//                                  5 - b = block b, CBPY already in the sense of the macroblock's kind, macroblock.rs:479-489)
//   mvd_pair10[10 bits]            a horizontal and a vertical MVD code word (Table 14) that lie inside 10 bits together:
//                                  len | (dx & 63) << 4 | (dy & 63) << 10 -- the small differences of slow motion
namespace {
constexpr uint32_t kEventLastBit = 5, kEventEscape = 1u << 6, kEventInvalid = 1u << 7;     // (bits 0..4: bits the event takes)
struct HeaderTables {
    uint16_t p12[4096], i12[4096], mvd10[1024];
    // tcoef13[flavour][13 bits]: EVERY TCOEF code word (12 bits at most, Table 16/H.263) together with its sign bit, as the
    // event it stands for: used (code + sign, 3..13) | last << 5 | run << 8 | level (signed) << 16.  ESCAPE ("0000 011") is
    // kEventEscape WITH the bits the whole escape takes in `used` (22; Sorenson v1: 22 or 26 by the flag bit behind the code
    // word, which is part of the index -- hence a table per flavour): how far the cursor moves comes straight out of the table
    // for every kind of event, and the cursor is what the next event waits for.  A prefix no code word starts with is
    // kEventInvalid (used = 0).  2 x 32 KB; what a stream touches are the blocks of the frequent short code words of ONE.
    uint32_t tcoef13[2][8192];
    HeaderTables()
    {
        for (int v1 = 0; v1 < 2; v1++)
            for (uint32_t idx = 0; idx < 8192; idx++) {
                const VlcTable::Slot &t = tcoef_table().lookup32(idx << 19);
                if (!t.valid) { tcoef13[v1][idx] = kEventInvalid; continue; }
                if (t.v0 < 0) {
                    // ESCAPE (7 bits) [Sorenson v1: 1 bit selects an 11- or 7-bit LEVEL] LAST (1) RUN (6) LEVEL
                    const uint32_t flag = (idx >> 5) & 1u;
                    tcoef13[v1][idx] = kEventEscape | (v1 ? 8u + 7u + (flag ? 11u : 7u) : 7u + 7u + 8u);
                    continue;
                }
                const uint32_t sign = (idx >> (12 - t.len)) & 1u;
                const int level = sign ? -(int)t.v2 : (int)t.v2;
                tcoef13[v1][idx] = (uint32_t)(t.len + 1) | ((uint32_t)(t.v0 != 0) << 5) | ((uint32_t)t.v1 << 8) | ((uint32_t)(uint16_t)(int16_t)level << 16);
            }
        for (int pic = 0; pic < 2; pic++) {
            const VlcTable &mc = pic ? mcbpc_i_table() : mcbpc_p_table();
            uint16_t *out = pic ? i12 : p12;
            for (uint32_t idx = 0; idx < 4096; idx++) {
                out[idx] = 0;
                const uint32_t w = idx << 20;
                const VlcTable::Slot &m = mc.lookup32(w);
                if (!m.valid || m.v0 < 0 || m.len > 12) continue;           // invalid or stuffing: the ordinary way
                const VlcTable::Slot &c = cbpy_table().lookup32(w << m.len);
                if (!c.valid || m.len + c.len > 12) continue;
                const bool intra = m.v0 == H263MI_MB_INTRA || m.v0 == H263MI_MB_INTRA_Q;
                const uint32_t luma = intra ? (uint32_t)c.v0 : (~(uint32_t)c.v0 & 0xfu);
                const uint32_t coded6 = (luma << 2) | ((uint32_t)m.v1 << 1) | (uint32_t)m.v2;
                out[idx] = (uint16_t)((uint32_t)(m.len + c.len) | ((uint32_t)m.v0 << 4) | (coded6 << 7));
            }
        }
        for (uint32_t idx = 0; idx < 1024; idx++) {
            mvd10[idx] = 0;
            const uint32_t w = idx << 22;
            const VlcTable::Slot &x = mvd_table().lookup32(w);
            if (!x.valid || x.len > 10) continue;
            const VlcTable::Slot &y = mvd_table().lookup32(w << x.len);
            if (!y.valid || x.len + y.len > 10) continue;
            mvd10[idx] = (uint16_t)((uint32_t)(x.len + y.len) | (((uint32_t)x.v0 & 63u) << 4) | (((uint32_t)y.v0 & 63u) << 10));
        }
    }
};
const HeaderTables &header_tables()
{
    static const HeaderTables t;
    return t;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------
// block layer: parser/block.rs:670-755
// ---------------------------------------------------------------------------------------------------
// `put(is_short, run, level)` receives the TCOEF events in bitstream order; `intradc` the INTRADC code of an intra
// block.  On an error the reader is back where the block started (with_transaction, block.rs:682).
namespace {
template <class Sink>
inline int decode_block_to(BitReader &r, bool sorenson, int version, bool intra, bool tcoef_present, uint8_t &intradc, Sink &&put,
                           bool field_by_field = false)
{
    const size_t checkpoint = r.position();
    int rc = H263MI_OK;
    do {
        const VlcTable &table = tcoef_table();
        const bool sorenson_v1 = sorenson && version == 1;   // Sorenson v1: 1 bit selects an 11- or 7-bit LEVEL
        // (the cursor of the fast path lives in a local: the reader object is shared with code the compiler cannot see
        // through, and a load + store of its position per event was a quarter of the time of a dense picture)
        size_t pos = checkpoint;
        const size_t end32 = r.size_bits() >= 32 ? r.size_bits() - 32 : 0;
        const bool have32 = r.size_bits() >= 32 && !field_by_field;
        uint32_t esc_streak = 0;
        if (intra) {
            uint32_t code;
            if ((rc = r.read_u8(code)) != H263MI_OK) break;
            if (code == 0 || code == 128) { rc = H263MI_ERR_INVALID_INTRA_DC; break; }   // IntraDc::from_u8, types.rs:930-936
            intradc = (uint8_t)code;
            pos = r.position();
        }
        while (tcoef_present) {
            if (have32 && pos <= end32) {
                // Fast path: the longest TCOEF event -- escape (7) + Sorenson width flag + LAST + RUN (6) + LEVEL (11)
                // = 26 bits -- lies inside one 32-bit window of data that is all there: one peek, no per-field
                // end-of-data checks.  Same bits consumed and same errors as the field-by-field path below.
                const uint32_t w = (uint32_t)(r.window_at(pos) >> 32);
                if (esc_streak >= 2 && (w >> 25) == 3u) {
                    // A run of ESCAPEs ("0000 011", Table 16/H.263) -- the blocks of an intra picture with large
                    // levels are little else: the fields sit at fixed places behind the 7-bit code word, no table
                    // access, a third of the arithmetic of the general form below.  Tried only after two ESCAPEs in a
                    // row, so that the test is a well-predicted branch where it is taken at all (block.rs:689-724).
                    // (Reading a second ESCAPE out of the same 64-bit window was tried: the dense I picture gains
                    // nothing and the general form below comes out of the compiler 50 % slower.)
                    const uint32_t flagbit = (w >> 24) & 1u;                          // Sorenson v1: selects an 11- or 7-bit LEVEL
                    const uint32_t width = sorenson_v1 ? 7u + 4u * flagbit : 8u;
                    const uint32_t e0 = sorenson_v1 ? 8u : 7u;                        // position of LAST
                    const uint32_t raw = (w >> (25 - e0 - width)) & ((1u << width) - 1u);
                    const int level = (int)((raw ^ (1u << (width - 1))) - (1u << (width - 1)));
                    pos += e0 + 7u + width;
                    if (level == 0) { rc = H263MI_ERR_INVALID_LONG_COEFFICIENT; break; }
                    put(false, (int)((w >> (25 - e0)) & 63u), level);
                    tcoef_present = ((w >> (31 - e0)) & 1u) == 0;
                    continue;
                }
                const VlcTable::Slot &sl = table.lookup32(w);
                const uint32_t len = sl.len;
                if (!sl.valid) { rc = H263MI_ERR_INVALID_SHORT_COEFFICIENT; break; }
                // Both readings of the window are computed and one is selected without a branch (short code + sign
                // against ESCAPE, block.rs:689-724): which of the two an event is has no pattern a predictor could learn.
                // (written as mask arithmetic: the compiler turned `?:` on these into branches again)
                const uint32_t esc = sl.v0 < 0 ? ~0u : 0u;
                // short code: sign bit behind the code word
                const uint32_t sign = (w >> (31 - len)) & 1u;
                const uint32_t short_level = ((uint32_t)sl.v2 ^ (0u - sign)) + sign;             // sign ? -v2 : v2
                // ESCAPE: [Sorenson v1: 1 bit selects an 11- or 7-bit LEVEL] LAST, RUN (6), LEVEL (two's complement)
                const uint32_t flag = sorenson_v1 ? 1u : 0u;
                const uint32_t width = sorenson_v1 ? 7u + 4u * sign : 8u;                         // (the flag sits where a short code has its sign)
                const uint32_t e0 = len + flag;                         // position of LAST
                const uint32_t esc_last = (w >> (31 - e0)) & 1u;
                const uint32_t esc_run = (w >> (25 - e0)) & 63u;
                const uint32_t raw = (w >> (25 - e0 - width)) & ((1u << width) - 1u);
                const uint32_t esc_level = (raw ^ (1u << (width - 1))) - (1u << (width - 1));    // reader.rs:176-187
                const uint32_t used = ((e0 + 7u + width) & esc) | ((len + 1u) & ~esc);
                const int level = (int)((esc_level & esc) | (short_level & ~esc));
                const int run = (int)((esc_run & esc) | ((uint32_t)sl.v1 & ~esc));
                const bool last = ((esc_last & esc) | ((uint32_t)(sl.v0 != 0) & ~esc)) != 0;
                pos += used;
                if (level == 0) { rc = H263MI_ERR_INVALID_LONG_COEFFICIENT; break; }     // only an ESCAPE can say 0
                put(esc == 0, run, level);
                tcoef_present = !last;
                esc_streak = (esc_streak + 1u) & esc;            // ESCAPEs in a row
                continue;
            }
            // near the end of the data: field by field, every read checked
            r.rollback(pos);
            VlcHit h;
            if ((rc = table.decode(r, h)) != H263MI_OK) break;
            if (!h.valid) { rc = H263MI_ERR_INVALID_SHORT_COEFFICIENT; break; }
            bool last;
            if (h.v0 < 0) {                              // ESCAPE (block.rs:689-724)
                uint32_t width = 8, v;
                if (sorenson && version == 1) {
                    if ((rc = r.read_bits(1, v)) != H263MI_OK) break;
                    width = v ? 11 : 7;
                }
                if ((rc = r.read_bits(1, v)) != H263MI_OK) break;
                last = v == 1;
                if ((rc = r.read_bits(6, v)) != H263MI_OK) break;
                const int run = (int)v;
                int32_t lv;
                if ((rc = r.read_signed_bits(width, lv)) != H263MI_OK) break;
                if (lv == 0) { rc = H263MI_ERR_INVALID_LONG_COEFFICIENT; break; }
                // (the reference's second check, `level == i16::MAX << level_width`, can never hold for a
                // sign-extended LEVEL of that width: block.rs:708-715)
                put(false, run, (int)lv);
            } else {
                uint32_t sign;
                if ((rc = r.read_bits(1, sign)) != H263MI_OK) break;
                last = h.v0 != 0;
                put(true, (int)h.v1, sign ? -(int)h.v2 : (int)h.v2);
            }
            tcoef_present = !last;
            pos = r.position();
        }
        if (rc == H263MI_OK) r.rollback(pos);        // hand the cursor back
    } while (0);
    if (rc != H263MI_OK) r.rollback(checkpoint);
    return rc;
}
}  // namespace

// ---- the TCOEF events of one block, as a function of its own (round 5) ----------------------------------------------------
// parse_picture is one very large function: inlined into it, the event loop above keeps its own state (zigzag position, event
// count, the output pointer, the overrun flag) in stack slots -- a load, an add and a store of the zigzag position per event,
// a reload of the output pointer per event -- because sixteen registers do not hold the loop's values and the macroblock
// loop's around it.  Here the loop has the registers to itself: it takes what it needs by value and hands back where it
// stopped.  It runs while a whole 64-bit window of data lies behind the cursor (pos <= end64: every event, 26 bits at most,
// is wholly inside the data, and the 8-byte load is inside the buffer); the last bytes of a picture, and everything
// irregular, are left to decode_block_to, which continues from the returned state.  Same bits, same events, same errors
// (tests/test_parser_paths.py and the fuzzer hold the two forms against each other).
namespace {
struct BlockRun {
    size_t pos;            // bit position behind the last event taken
    uint32_t zz;           // next zigzag index
    uint32_t n_ev;         // events stored
    int rc;                // H263MI_OK, or the error of the event at `pos`
    bool more;             // no LAST yet: the caller goes on from pos
    bool overrun;          // a run walked past zigzag 63 (rle.rs:125-127): the block is void, the rest was parsed and dropped
};

template <bool SORENSON_V1>
__attribute__((noinline)) BlockRun block_events_fast(const uint8_t *base, size_t pos, size_t end64, uint32_t zz, uint32_t *ev,
                                                     const VlcTable &table)
{
    uint32_t n_ev = 0, esc_streak = 0;
    bool overrun = false, more = true;
    int rc = H263MI_OK;
    while (more && pos <= end64) {
        uint64_t v;
        __builtin_memcpy(&v, base + (pos >> 3), 8);
        const uint32_t w = (uint32_t)((__builtin_bswap64(v) << (pos & 7)) >> 32);
        uint32_t used, run;
        int level;
        bool last;
        if (esc_streak >= 2 && (w >> 25) == 3u) {            // a run of ESCAPEs: fixed field positions, no table access
            const uint32_t flagbit = (w >> 24) & 1u;
            const uint32_t width = SORENSON_V1 ? 7u + 4u * flagbit : 8u;
            const uint32_t e0 = SORENSON_V1 ? 8u : 7u;
            const uint32_t raw = (w >> (25 - e0 - width)) & ((1u << width) - 1u);
            level = (int)((raw ^ (1u << (width - 1))) - (1u << (width - 1)));
            used = e0 + 7u + width;
            run = (w >> (25 - e0)) & 63u;
            last = ((w >> (31 - e0)) & 1u) != 0;
        } else {
            const VlcTable::Slot &sl = table.lookup32(w);
            const uint32_t len = sl.len;
            if (!sl.valid) { rc = H263MI_ERR_INVALID_SHORT_COEFFICIENT; break; }
            // both readings of the window, one selected by mask (short code + sign against ESCAPE, block.rs:689-724)
            const uint32_t esc = sl.v0 < 0 ? ~0u : 0u;
            const uint32_t sign = (w >> (31 - len)) & 1u;
            const uint32_t short_level = ((uint32_t)sl.v2 ^ (0u - sign)) + sign;
            const uint32_t flag = SORENSON_V1 ? 1u : 0u;
            const uint32_t width = SORENSON_V1 ? 7u + 4u * sign : 8u;
            const uint32_t e0 = len + flag;
            const uint32_t esc_last = (w >> (31 - e0)) & 1u;
            const uint32_t esc_run = (w >> (25 - e0)) & 63u;
            const uint32_t raw = (w >> (25 - e0 - width)) & ((1u << width) - 1u);
            const uint32_t esc_level = (raw ^ (1u << (width - 1))) - (1u << (width - 1));
            used = ((e0 + 7u + width) & esc) | ((len + 1u) & ~esc);
            level = (int)((esc_level & esc) | (short_level & ~esc));
            run = (esc_run & esc) | ((uint32_t)sl.v1 & ~esc);
            last = ((esc_last & esc) | ((uint32_t)(sl.v0 != 0) & ~esc)) != 0;
            esc_streak = (esc_streak + 1u) & esc;
        }
        if (level == 0) { rc = H263MI_ERR_INVALID_LONG_COEFFICIENT; break; }     // only an ESCAPE can say 0
        pos += used;
        more = !last;
        if (overrun) continue;
        zz += run;
        if (zz >= 64) { overrun = true; continue; }
        ev[n_ev++] = ((uint32_t)(uint16_t)(int16_t)level << 16) | kZigzagRaster[zz++];
    }
    return BlockRun{pos, zz, n_ev, rc, more, overrun};
}

// ---- one event out of the bits at the cursor, without a branch on what kind of event it is --------------------------------
// `bits`: the unread bits at the top of a 64-bit word (30 of them, at least, are data).  A short code word + sign is one
// access to the 13-bit table of ready-made events (HeaderTables::tcoef13).  An ESCAPE (block.rs:689-724) has its fields at
// FIXED places behind the 7-bit code word -- [Sorenson v1: 1 bit that selects an 11- or 7-bit LEVEL] LAST, RUN (6), LEVEL --
// so that reading is computed from the bits beside the table access, whatever the table says, and a mask picks one of the
// two.  (Real key frames are a quarter ESCAPEs, a third code words of nine bits and more: a branch on either kind is a coin
// toss, and round 4's form -- both readings computed from the table's LENGTH -- spent seventy instructions per event.)
// Returns the event in the table's format (kEventInvalid set: no code word starts here) and the bits it takes in `used`.
template <bool SORENSON_V1>
inline __attribute__((always_inline)) uint32_t event_at(uint64_t bits, const uint32_t *tcoef13, uint32_t &used)
{
    const uint32_t e = tcoef13[bits >> 51];
    used = e & 31u;                                                              // (the one thing the next event waits for)
    const uint32_t w = (uint32_t)(bits >> 32);
    const uint32_t e0 = SORENSON_V1 ? 8u : 7u;                                   // position of LAST
    const uint32_t width = SORENSON_V1 ? 7u + 4u * ((w >> 24) & 1u) : 8u;
    const uint32_t raw = (w >> (25u - e0 - width)) & ((1u << width) - 1u);
    const uint32_t level = (raw ^ (1u << (width - 1u))) - (1u << (width - 1u));   // two's complement of `width` bits (reader.rs:176-187)
    const uint32_t esc_event = (((w >> (31u - e0)) & 1u) << kEventLastBit) | (((w >> (25u - e0)) & 63u) << 8) | (level << 16) | kEventEscape;
    const uint32_t m = 0u - ((e >> 6) & 1u);                                     // all ones for an ESCAPE
    return (esc_event & m) | (e & ~m);
}

// ALL coded blocks of one INTER macroblock in one loop.  An inter block is nothing but its events (no INTRADC in front,
// block.rs:682-687), so the blocks of a macroblock follow each other in the bitstream like one long list with LAST marks in
// it.  Taken block by block, every block ends in a loop exit no predictor can learn (a block has two or three events, or one,
// or seven) and the macroblock in another (it has one coded block, or none, or three): about two mispredicted branches per
// macroblock, a third of the parse time of a real P picture.  Here a LAST event ends a block WITHOUT a branch -- the block's
// offset word is written behind every event and the write position moves on by `last`, the zigzag position is reset by a
// conditional move -- and the one branch that depends on the data is "was that the macroblock's last block".
// The bits come out of a register that is refilled beside the decode (an unaligned 8-byte load OR-ed in below the valid
// bits; stream position of the next unread bit = 8 * (ptr - base) - cnt), so that the loop-carried chain is table -> length
// -> shift.
// Hands back `done` = false (nothing of what it wrote counts) for everything out of the ordinary: the data ends inside the
// macroblock, a run walks past zigzag 63; the caller then takes the macroblock block by block from where it began.
struct InterRun {
    size_t pos;
    uint32_t n_ev;
    int rc;
    bool done;
};
inline uint64_t be64_at(const uint8_t *q)
{
    uint64_t x;
    __builtin_memcpy(&x, q, 8);
    return __builtin_bswap64(x);
}
template <bool SORENSON_V1>
__attribute__((noinline)) InterRun inter_macroblock_events(const uint8_t *base, size_t pos, size_t end64, uint32_t remaining, uint32_t *ev,
                                                           uint32_t *fe, uint32_t events_before, const uint32_t *tcoef13)
{
    if (pos > end64) return InterRun{pos, 0, H263MI_OK, false};
    // (pos <= end64: 8 bytes at pos >> 3 are inside the data; `safe` = the last place an 8-byte load may start)
    const uint8_t *const safe = base + (end64 >> 3);
    const uint8_t *ptr = base + (pos >> 3) + 7;
    uint64_t bits = be64_at(ptr - 7) << (pos & 7);
    uint32_t cnt = 56u - (uint32_t)(pos & 7);                // (the byte at `ptr` is in `bits` already, unaccounted)
    uint32_t *out = ev;
    uint32_t zz = 0;
    for (;;) {
        if (__builtin_expect(ptr > safe, 0)) return InterRun{pos, 0, H263MI_OK, false};
        uint32_t used;
        const uint32_t e = event_at<SORENSON_V1>(bits, tcoef13, used);          // at least 30 bits of `bits` are accounted for here
        bits |= be64_at(ptr) >> cnt; ptr += (63u - cnt) >> 3; cnt |= 56u;       // refill
        if (__builtin_expect((e & kEventInvalid) != 0, 0)) return InterRun{pos, 0, H263MI_ERR_INVALID_SHORT_COEFFICIENT, false};
        if (__builtin_expect((e >> 16) == 0, 0)) return InterRun{pos, 0, H263MI_ERR_INVALID_LONG_COEFFICIENT, false};   // only an ESCAPE can say 0
        bits <<= used; cnt -= used;
        zz += (e >> 8) & 63u;
        if (__builtin_expect(zz >= 64, 0)) return InterRun{pos, 0, H263MI_OK, false};          // rle.rs:125-127: block by block
        *out++ = (e & 0xffff0000u) | kZigzagRaster[zz];
        const uint32_t last = (e >> kEventLastBit) & 1u;
        fe[1] = events_before + (uint32_t)(out - ev);       // the block's end (rewritten until its LAST event)
        fe += last;
        zz = last ? 0u : zz + 1u;
        remaining -= last;
        if (remaining == 0) break;
    }
    return InterRun{(size_t)(ptr - base) * 8 - cnt, (uint32_t)(out - ev), H263MI_OK, true};
}

// The six blocks of one INTRA macroblock as ONE list of items.  A block is an INTRADC code (8 bits, block.rs:682-686) and,
// when it is coded, events up to a LAST mark; block after block that is "DC, event, event, DC, DC, event, DC, ..." -- and taken
// block by block every "is it coded" and every "was that the last event" is a branch on fresh data (a key frame: 49 000
// blocks of one or two events).  Here an iteration takes ONE item, whichever kind it is: both readings of the bits at the
// cursor are at hand (the top byte as an INTRADC code, the event of event_at) and the iteration's bookkeeping -- bits
// consumed, where the output goes, whether the block ends, the zigzag position -- is selected by the item's kind with
// conditional moves.  The one branch that depends on the data is the end of the macroblock.  `done` = false as above.
template <bool SORENSON_V1>
__attribute__((noinline)) InterRun intra_macroblock_events(const uint8_t *base, size_t pos, size_t end64, uint32_t coded6, uint32_t *ev,
                                                           uint32_t *fe, uint32_t events_before, const uint32_t *tcoef13, uint8_t *dc_out)
{
    if (pos > end64) return InterRun{pos, 0, H263MI_OK, false};
    const uint8_t *const safe = base + (end64 >> 3);
    const uint8_t *ptr = base + (pos >> 3) + 7;
    uint64_t bits = be64_at(ptr - 7) << (pos & 7);
    uint32_t cnt = 56u - (uint32_t)(pos & 7);
    uint32_t *out = ev;
    uint32_t b = 0, is_dc = 1, zz = 1;
    uint8_t dcs[16];                                         // [0..5] the codes, [8..13] where the write of an event item lands
    for (;;) {
        if (__builtin_expect(ptr > safe, 0)) return InterRun{pos, 0, H263MI_OK, false};
        const uint32_t top = (uint32_t)(bits >> 56);
        uint32_t ev_used;
        const uint32_t e = event_at<SORENSON_V1>(bits, tcoef13, ev_used);
        bits |= be64_at(ptr) >> cnt; ptr += (63u - cnt) >> 3; cnt |= 56u;       // refill
        const uint32_t is_ev = is_dc ^ 1u;
        if (__builtin_expect(is_dc & (uint32_t)(top == 0 || top == 128), 0)) return InterRun{pos, 0, H263MI_ERR_INVALID_INTRA_DC, false};   // types.rs:930-936
        if (__builtin_expect(is_ev & (uint32_t)((e & kEventInvalid) != 0), 0)) return InterRun{pos, 0, H263MI_ERR_INVALID_SHORT_COEFFICIENT, false};
        if (__builtin_expect(is_ev & (uint32_t)((e >> 16) == 0), 0)) return InterRun{pos, 0, H263MI_ERR_INVALID_LONG_COEFFICIENT, false};
        const uint32_t used = is_dc ? 8u : ev_used;
        bits <<= used; cnt -= used;
        dcs[b + 8u * is_ev] = (uint8_t)top;
        const uint32_t zz_next = zz + ((e >> 8) & 63u);
        if (__builtin_expect(is_ev & (uint32_t)(zz_next >= 64), 0)) return InterRun{pos, 0, H263MI_OK, false};             // rle.rs:125-127
        *out = (e & 0xffff0000u) | kZigzagRaster[zz_next & 63u];
        out += is_ev;
        const uint32_t ev_last = is_ev & (e >> kEventLastBit);              // this event ends its block
        const uint32_t coded_b = (coded6 >> (5u - b)) & 1u;
        const uint32_t fin = is_dc ? (coded_b ^ 1u) : ev_last;             // this item ends its block
        fe[1] = events_before + (uint32_t)(out - ev);                        // the block's end (rewritten until its LAST event)
        fe += ev_last;
        zz = is_dc ? 1u : zz_next + 1u;                                      // the TCOEFs of an intra block start at zigzag 1 (rle.rs:117-121)
        b += fin;
        is_dc = fin;
        if (b == 6) break;
    }
    for (int k = 0; k < 6; k++) dc_out[k] = dcs[k];
    return InterRun{(size_t)(ptr - base) * 8 - cnt, (uint32_t)(out - ev), H263MI_OK, true};
}
}  // namespace

int decode_block(BitReader &r, bool sorenson, int version, bool intra, bool tcoef_present, ParsedBlock &out)
{
    out.has_intradc = false;
    out.intradc = 0;
    out.n_tcoef = 0;
    const int capacity = (int)(sizeof(out.tcoef) / sizeof(out.tcoef[0]));
    const int rc = decode_block_to(r, sorenson, version, intra, tcoef_present, out.intradc, [&](bool is_short, int run, int level) {
        // more events than a block can place: every further one lands beyond zigzag 63 anyway; the parse goes on
        // (the bitstream position matters) but nothing more is stored
        if (out.n_tcoef >= capacity) return;
        out.tcoef[out.n_tcoef].is_short = is_short;
        out.tcoef[out.n_tcoef].run = (uint8_t)run;
        out.tcoef[out.n_tcoef].level = (int16_t)level;
        out.n_tcoef++;
    });
    out.has_intradc = intra && rc == H263MI_OK;
    if (rc != H263MI_OK) out.intradc = 0;
    return rc;
}

// ---------------------------------------------------------------------------------------------------
// picture layer: parser/picture.rs
// ---------------------------------------------------------------------------------------------------
bool SourceFormat::dimensions(uint16_t &w, uint16_t &h) const
{
    switch (kind) {                                              // types.rs:168-180
    case SUB_QCIF: w = 128; h = 96; return true;
    case QCIF: w = 176; h = 144; return true;
    case CIF: w = 352; h = 288; return true;
    case FOUR_CIF: w = 704; h = 576; return true;
    case SIXTEEN_CIF: w = 1408; h = 1152; return true;
    case EXTENDED: w = width; h = height; return true;
    default: return false;                                       // Reserved (and a missing format)
    }
}

namespace {
#define RD(n, var) do { if ((rc = r.read_bits((n), var)) != H263MI_OK) return rc; } while (0)

SourceFormat extended_format(uint16_t w, uint16_t h)
{
    SourceFormat f;
    f.kind = SourceFormat::EXTENDED; f.par = 1; f.width = w; f.height = h;      // PixelAspectRatio::Square
    return f;
}
SourceFormat plain_format(uint8_t kind)
{
    SourceFormat f;
    f.kind = kind;
    return f;
}

// decode_pei (picture.rs:577-596)
int decode_pei(BitReader &r, std::vector<uint8_t> &extra)
{
    int rc;
    uint32_t v;
    for (;;) {
        RD(1, v);
        if (!v) return H263MI_OK;
        RD(8, v);
        extra.push_back((uint8_t)v);
    }
}

// Sorenson header after the version field (picture.rs:626-659, decode_sorenson_ptype 271-327)
int decode_sorenson_header(BitReader &r, PictureHeader &out)
{
    int rc;
    uint32_t v, fmt;
    RD(8, v);
    out.temporal_reference = (uint16_t)v;
    RD(3, fmt);
    switch (fmt) {
    case 0:
    case 1: {
        const uint32_t n = fmt == 0 ? 8 : 16;
        uint32_t w, h;
        RD(n, w);
        RD(n, h);
        out.format = extended_format((uint16_t)w, (uint16_t)h);
        break;
    }
    case 2: out.format = plain_format(SourceFormat::CIF); break;
    case 3: out.format = plain_format(SourceFormat::QCIF); break;
    case 4: out.format = plain_format(SourceFormat::SUB_QCIF); break;
    case 5: out.format = extended_format(320, 240); break;
    case 6: out.format = extended_format(160, 120); break;
    default: out.format = plain_format(SourceFormat::RESERVED); break;
    }
    RD(2, v);
    out.picture_type = (uint8_t)v;               // 0 I, 1 P, 2 disposable P, 3 Reserved(3)
    RD(1, v);
    out.use_deblocker = v == 1;
    if (v) out.options |= OPT_USE_DEBLOCKER;
    RD(5, v);
    out.quantizer = (uint8_t)v;
    if ((rc = decode_pei(r, out.extra)) != H263MI_OK) return rc;
    out.mv_range = 2;                            // "Sorenson is always unlimited" (picture.rs:644)
    return H263MI_OK;
}

// decode_ptype (picture.rs:21-79); *plus = true: PLUSPTYPE follows and nothing else was read
int decode_ptype(BitReader &r, PictureHeader &out, bool &plus)
{
    int rc;
    uint32_t hi, lo;
    plus = false;
    RD(8, hi);
    if ((hi & 0xC0) != 0x80) return H263MI_ERR_INVALID_PTYPE;
    if (hi & 0x20) out.options |= OPT_USE_SPLIT_SCREEN;
    if (hi & 0x10) out.options |= OPT_USE_DOCUMENT_CAMERA;
    if (hi & 0x08) out.options |= OPT_RELEASE_FULL_PICTURE_FREEZE;
    switch (hi & 7) {
    case 0: return H263MI_ERR_INVALID_PTYPE;
    case 1: out.format = plain_format(SourceFormat::SUB_QCIF); break;
    case 2: out.format = plain_format(SourceFormat::QCIF); break;
    case 3: out.format = plain_format(SourceFormat::CIF); break;
    case 4: out.format = plain_format(SourceFormat::FOUR_CIF); break;
    case 5: out.format = plain_format(SourceFormat::SIXTEEN_CIF); break;
    case 6: out.format = plain_format(SourceFormat::RESERVED); break;
    default: plus = true; return H263MI_OK;
    }
    RD(5, lo);
    out.picture_type = (lo & 0x10) ? H263MI_PICTURE_I : H263MI_PICTURE_P;       // as the reference reads the bit (picture.rs:55-59)
    if (lo & 0x08) out.options |= OPT_UNRESTRICTED_MOTION_VECTORS;
    if (lo & 0x04) out.options |= OPT_SYNTAX_BASED_ARITHMETIC_CODING;
    if (lo & 0x02) out.options |= OPT_ADVANCED_PREDICTION;
    if (lo & 0x01) out.picture_type = PT_PB;
    return H263MI_OK;
}

enum : uint32_t {                                                // PlusPTypeFollower (picture.rs:88-97)
    FOLLOW_CUSTOM_FORMAT = 1, FOLLOW_CUSTOM_CLOCK = 2, FOLLOW_MV_RANGE = 4, FOLLOW_SLICE_SUBMODE = 8,
    FOLLOW_REFERENCE_LAYER = 16, FOLLOW_RPS_MODE = 32,
};

// decode_plusptype (picture.rs:135-268)
int decode_plusptype(BitReader &r, uint32_t decoder_options, uint32_t previous_options, PictureHeader &out,
                     uint32_t &followers)
{
    int rc;
    uint32_t ufep, v;
    followers = 0;
    RD(3, ufep);
    if (ufep > 1) return H263MI_ERR_INVALID_PLUS_PTYPE;
    out.has_opptype = ufep == 1;
    uint32_t options = 0;
    out.format = SourceFormat();                                 // None unless OPPTYPE restates it
    if (out.has_opptype) {
        RD(18, v);
        if ((v & 0xF) != 0x8) return H263MI_ERR_INVALID_PLUS_PTYPE;      // H.263 5.1.4.2
        switch ((v & 0x38000) >> 15) {
        case 1: out.format = plain_format(SourceFormat::SUB_QCIF); break;
        case 2: out.format = plain_format(SourceFormat::QCIF); break;
        case 3: out.format = plain_format(SourceFormat::CIF); break;
        case 4: out.format = plain_format(SourceFormat::FOUR_CIF); break;
        case 5: out.format = plain_format(SourceFormat::SIXTEEN_CIF); break;
        case 6: followers |= FOLLOW_CUSTOM_FORMAT; break;        // format stays None until CPFMT
        default: out.format = plain_format(SourceFormat::RESERVED); break;      // 0 and 7
        }
        if (v & 0x04000) followers |= FOLLOW_CUSTOM_CLOCK;
        if (v & 0x02000) { options |= OPT_UNRESTRICTED_MOTION_VECTORS; followers |= FOLLOW_MV_RANGE; }
        if (v & 0x01000) options |= OPT_SYNTAX_BASED_ARITHMETIC_CODING;
        if (v & 0x00800) options |= OPT_ADVANCED_PREDICTION;
        if (v & 0x00400) options |= OPT_ADVANCED_INTRA_CODING;
        if (v & 0x00200) options |= OPT_DEBLOCKING_FILTER;
        if (v & 0x00100) { options |= OPT_SLICE_STRUCTURED; followers |= FOLLOW_SLICE_SUBMODE; }
        if (v & 0x00080) { options |= OPT_REFERENCE_PICTURE_SELECTION; followers |= FOLLOW_RPS_MODE; }
        if (v & 0x00040) options |= OPT_INDEPENDENT_SEGMENT_DECODING;
        if (v & 0x00020) options |= OPT_ALTERNATIVE_INTER_VLC;
        if (v & 0x00010) options |= OPT_MODIFIED_QUANTIZATION;
        if (decoder_options & H263MI_USE_SCALABILITY_MODE) followers |= FOLLOW_REFERENCE_LAYER;
    } else {
        options |= previous_options & OPPTYPE_OPTIONS;           // carried forward (picture.rs:233)
    }
    RD(9, v);
    if ((v & 7) != 1) return H263MI_ERR_INVALID_PLUS_PTYPE;      // H.263 5.1.4.3
    switch ((v & 0x1C0) >> 6) {
    case 0: out.picture_type = H263MI_PICTURE_I; break;
    case 1: out.picture_type = H263MI_PICTURE_P; break;
    case 2: out.picture_type = PT_IMPROVED_PB; break;
    case 3: out.picture_type = PT_B; break;
    case 4: out.picture_type = PT_EI; break;
    case 5: out.picture_type = PT_EP; break;
    default: out.picture_type = PT_RESERVED; break;
    }
    if (v & 0x020) options |= OPT_REFERENCE_PICTURE_RESAMPLING;
    if (v & 0x010) options |= OPT_REDUCED_RESOLUTION_UPDATE;
    if (v & 0x008) options |= OPT_ROUNDING_TYPE_ONE;
    out.options |= options;
    return H263MI_OK;
}

// decode_cpm_and_psbi (picture.rs:335-346)
int decode_cpm_and_psbi(BitReader &r)
{
    int rc;
    uint32_t v;
    RD(1, v);
    if (v) RD(2, v);
    return H263MI_OK;
}

// decode_cpfmt (picture.rs:349-395)
int decode_cpfmt(BitReader &r, SourceFormat &f)
{
    int rc;
    uint32_t v;
    RD(23, v);
    if (!(v & 0x000200)) return H263MI_ERR_PICTURE_FORMAT_INVALID;
    f = SourceFormat();
    f.kind = SourceFormat::EXTENDED;
    f.par = (uint8_t)((v & 0x780000) >> 19);
    if (f.par == 0) return H263MI_ERR_PICTURE_FORMAT_INVALID;
    if (f.par == 15) {
        uint32_t pw, ph;
        RD(8, pw);
        RD(8, ph);
        if (!pw || !ph) return H263MI_ERR_PICTURE_FORMAT_INVALID;
        f.par_width = (uint8_t)pw; f.par_height = (uint8_t)ph;
    }
    f.width = (uint16_t)((((v & 0x07FC00) >> 10) + 1) * 4);
    f.height = (uint16_t)((v & 0x0000FF) * 4);                   // the reference masks 8 of the 9 PHI bits
    return H263MI_OK;
}

// the standard H.263 header after the (zero) GOB number: picture.rs:662-808
int decode_standard_header(BitReader &r, uint32_t decoder_options, const ParserContext *prev, PictureHeader &out)
{
    int rc;
    uint32_t v, low_tr;
    RD(8, low_tr);
    bool plus = false;
    if ((rc = decode_ptype(r, out, plus)) != H263MI_OK) return rc;
    uint32_t followers = 0;
    bool have_cpm = false;
    if (plus) {
        out.has_plusptype = true;
        if ((rc = decode_plusptype(r, decoder_options, prev && prev->have_last ? prev->last_header_options : 0u, out,
                                   followers)) != H263MI_OK)
            return rc;
        if ((rc = decode_cpm_and_psbi(r)) != H263MI_OK) return rc;
        have_cpm = true;
    }
    if (followers & FOLLOW_CUSTOM_FORMAT)
        if ((rc = decode_cpfmt(r, out.format)) != H263MI_OK) return rc;
    bool custom_clock = false;
    if (followers & FOLLOW_CUSTOM_CLOCK) {                       // decode_cpcfc (picture.rs:398-410)
        RD(8, v);
        custom_clock = true;
    }
    out.temporal_reference = (uint16_t)low_tr;
    if (custom_clock) {                                          // ETR (picture.rs:717-723)
        RD(2, v);
        out.temporal_reference = (uint16_t)((v << 8) | low_tr);
    }
    if (followers & FOLLOW_MV_RANGE) {                           // decode_uui (picture.rs:413-428)
        RD(1, v);
        if (v) out.mv_range = 1;
        else {
            RD(1, v);
            if (!v) return H263MI_ERR_INVALID_BITSTREAM;
            out.mv_range = 2;
        }
    }
    if (followers & FOLLOW_SLICE_SUBMODE) RD(2, v);              // decode_sss
    if (decoder_options & H263MI_USE_SCALABILITY_MODE) {         // decode_elnum_rlnum (picture.rs:452-471)
        RD(4, v);
        if (followers & FOLLOW_REFERENCE_LAYER) RD(4, v);
    }
    if (followers & FOLLOW_RPS_MODE) RD(3, v);                   // decode_rpsmf
    if (out.options & OPT_REFERENCE_PICTURE_SELECTION) {
        RD(1, v);                                                // decode_trpi (picture.rs:498-513)
        if (v) RD(10, v);
        RD(1, v);                                                // decode_bcm (picture.rs:516-537)
        if (v) return H263MI_ERR_UNIMPLEMENTED_DECODING;
        RD(1, v);
        if (!v) return H263MI_ERR_INVALID_BITSTREAM;
    }
    // decode_rprp is a stub (picture.rs:540-545); it is reached when the option is set or when the previous
    // header's Option<SourceFormat> differs from this one's (picture.rs:760-769)
    if ((out.options & OPT_REFERENCE_PICTURE_RESAMPLING) ||
        (prev && prev->have_last && prev->last_header_format != out.format))
        return H263MI_ERR_UNIMPLEMENTED_DECODING;
    RD(5, v);
    out.quantizer = (uint8_t)v;
    if (!have_cpm && (rc = decode_cpm_and_psbi(r)) != H263MI_OK) return rc;
    if (out.picture_type == PT_PB || out.picture_type == PT_IMPROVED_PB) {
        RD(custom_clock ? 5u : 3u, v);                           // decode_trb (picture.rs:548-559)
        RD(2, v);                                                // decode_dbquant
    }
    return decode_pei(r, out.extra);
}
#undef RD
}  // namespace

int decode_picture_header(BitReader &r, uint32_t decoder_options, const ParserContext *prev, PictureHeader &out,
                          bool &is_picture)
{
    const size_t checkpoint = r.position();          // with_transaction_union (picture.rs:619)
    out = PictureHeader();
    is_picture = false;
    int rc;
    do {
        int skipped;
        if ((rc = r.recognize_start_code(false, skipped)) != H263MI_OK) break;
        if (skipped < 0) { rc = H263MI_ERR_MIDDLE_OF_BITSTREAM; break; }
        if ((rc = r.skip_bits(17 + (uint32_t)skipped)) != H263MI_OK) break;
        uint32_t gob_id;
        if ((rc = r.read_bits(5, gob_id)) != H263MI_OK) break;
        if (decoder_options & H263MI_SORENSON_SPARK_BITSTREAM) {
            out.version = (int)gob_id;               // "Sorenson abuses the GOB ID as a version field"
            rc = decode_sorenson_header(r, out);
        } else if (gob_id != 0) {
            break;                                   // a GOB header: Ok(None), position restored below
        } else {
            rc = decode_standard_header(r, decoder_options, prev, out);
        }
        if (rc != H263MI_OK) break;
        out.format_valid = out.format.dimensions(out.width, out.height);
        is_picture = true;
    } while (0);
    if (rc != H263MI_OK || !is_picture) r.rollback(checkpoint);
    return rc;
}

// ---------------------------------------------------------------------------------------------------
// motion vector prediction: decoder/cpu/mvd_pred.rs
// ---------------------------------------------------------------------------------------------------
namespace {
struct Mv { int16_t x, y; };
inline bool mb_type_is_inter(uint32_t t) { return t == H263MI_MB_INTER || t == H263MI_MB_INTER_Q || t == H263MI_MB_INTER4V || t == H263MI_MB_INTER4V_Q; }

// HalfPel::median_of (types.rs:772-800): the reference's comparison chain returns the median of the three; here as
// min / max (no data-dependent branches: the vectors of a P picture are as good as random to a branch predictor)
inline int16_t median3(int16_t a, int16_t b, int16_t c)
{
    const int16_t lo = a < b ? a : b, hi = a < b ? b : a;
    const int16_t m = hi < c ? hi : c;
    return lo > m ? lo : m;
}

// predict_candidate (mvd_pred.rs:27-67).  The candidates are the macroblock to the left, the one above and the one above to
// the right: only the current and the previous macroblock ROW are ever looked at, so the vectors of the macroblocks decoded so
// far live in a ring of two rows (`row_cur`, `row_prev`: 4 vectors per macroblock; round 5 -- the array over the whole
// picture was 130 KB written and read per 1080p picture beside the 261 KB of records).  `col`, `line`: position of the
// current macroblock, kept by the caller.  (The reference's `last_line_mb < current_mb` tests hold whenever line > 0.)
inline Mv predict_candidate(const Mv *row_cur, const Mv *row_prev, const Mv cur[4], size_t mb_per_line, int index, size_t col, size_t line)
{
    const Mv zero{0, 0};
    Mv mv1;
    if (index == 0 || index == 2) mv1 = col == 0 ? zero : row_cur[(col - 1) * 4 + (size_t)index + 1];
    else mv1 = cur[index - 1];

    Mv mv2;
    if (index <= 1) mv2 = line == 0 ? mv1 : row_prev[col * 4 + (size_t)index + 2];
    else mv2 = cur[0];
    const bool end_of_line = col == (mb_per_line ? mb_per_line - 1 : 0);
    Mv mv3;
    if (index <= 1) {
        if (end_of_line) mv3 = zero;
        else if (line == 0) mv3 = mv1;
        else mv3 = row_prev[(col + 1) * 4 + 2];
    } else {
        mv3 = cur[1];
    }
    return Mv{median3(mv1.x, mv2.x, mv3.x), median3(mv1.y, mv2.y, mv3.y)};
}

// halfpel_decode (mvd_pred.rs:70-117).  Ranges are HalfPel::STANDARD_RANGE .. EXTENDED_RANGE_BEYONDCIF
// (types.rs:700-704); `dim` is the picture width for x and the height for y.
int16_t halfpel_decode(uint32_t running_options, const PictureHeader &hdr, int dim, int16_t predictor, int16_t mvd, bool is_x)
{
    int range = 32;
    int out = mvd + predictor;
    const bool umv = (running_options & OPT_UNRESTRICTED_MOTION_VECTORS) != 0;
    if (umv && !hdr.has_plusptype) {
        if (-32 <= predictor && predictor < 32) return (int16_t)out;
        range = 64;
    } else if (umv && hdr.mv_range == 1) {
        if (is_x) range = dim <= 352 ? 64 : (356 <= dim && dim <= 704) ? 128 : (708 <= dim && dim <= 1408) ? 256 : dim >= 1412 ? 512 : 64;
        else range = dim <= 288 ? 64 : (292 <= dim && dim <= 576) ? 128 : dim >= 580 ? 256 : 64;
    }
    if (!(-range <= out && out < range)) {
        const int inv = mvd > 0 ? mvd - 64 : (mvd < 0 ? mvd + 64 : mvd);     // HalfPel::invert (types.rs:736-742)
        out = inv + predictor;
    }
    return (int16_t)out;
}

// decode_gob (gob.rs:20-41), as used by the resynchronisation of state.rs:387-408.  Returns true when the
// macroblock loop should stop (end of picture); false with rc set when the decode fails.
bool resync_ends_picture(BitReader &r, int &rc)
{
    int skipped;
    rc = r.recognize_start_code(false, skipped);
    if (rc == kEof) { rc = H263MI_OK; return true; }                 // "Treat EOF/GOB errors as end of picture"
    if (rc != H263MI_OK) return false;
    if (skipped < 0) return true;                                    // InvalidGobHeader: a GOB error, ends the picture
    BitReader look = r;                                              // with_transaction_union: nothing is consumed
    uint32_t gob_id = 0;
    if (look.skip_bits(17 + (uint32_t)skipped) != H263MI_OK || look.read_bits(5, gob_id) != H263MI_OK) return true;
    if (gob_id == 0 || gob_id == 15) return true;                    // picture start code (or 15, as the reference has it)
    rc = H263MI_ERR_UNIMPLEMENTED_DECODING;                          // real GOB headers are a stub in the reference
    return false;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------
// whole picture: state.rs:138-427 up to the cut line
// ---------------------------------------------------------------------------------------------------
int parse_picture(const uint8_t *data, size_t len, uint32_t decoder_options, const ParserContext *ctx, ParsedPicture &out)
{
    const bool want_dense = out.want_dense;
    // (the vectors keep their capacity from one picture of a stream to the next: a 1080p I picture has a million
    // events, and growing that vector from nothing costs as much as parsing a P picture)
    out.desc = h263mi_picture_desc{};
    out.mbs.clear();
    out.n_mbs_ext = 0;
    out.mbs_ext_used = false;
    out.n_events_ext = 0;
    out.words_ext_used = false;
    out.coeffs.clear();
    out.block_first_event.clear();
    // (out.events keeps its size as the room to write into -- its elements are plain words -- and is cut to the events
    // of this picture at the end; after an error its contents mean nothing)
    out.n_coded_blocks = 0;
    out.bits_consumed = 0;
    out.next = ParserContext();
    out.block_first_event.push_back(0);
    BitReader r(data, len);
    PictureHeader hdr;
    bool is_picture = false;
    int rc = decode_picture_header(r, decoder_options, ctx, hdr, is_picture);
    if (rc != H263MI_OK) return rc;
    if (!is_picture) return H263MI_ERR_MIDDLE_OF_BITSTREAM;                    // state.rs:143-145
    const bool sorenson = (decoder_options & H263MI_SORENSON_SPARK_BITSTREAM) != 0;

    // state.rs:147-155.  `self.running_options` is never written by the reference, so it is the empty set:
    // OPPTYPE options only take effect in a picture that carries OPPTYPE itself, and the UMV / SAC / AP bits of
    // a plain PTYPE never do.
    const uint32_t state_running_options = 0;
    uint32_t running_options;
    if (hdr.has_plusptype && hdr.has_opptype) running_options = hdr.options;
    else if (hdr.has_plusptype) running_options = (hdr.options & ~OPPTYPE_OPTIONS) | (state_running_options & OPPTYPE_OPTIONS);
    else running_options = (hdr.options & ~OPPTYPE_OPTIONS & ~MPPTYPE_OPTIONS) | (state_running_options & (OPPTYPE_OPTIONS | MPPTYPE_OPTIONS));

    // state.rs:157-171: the format of this picture
    SourceFormat format = hdr.format;
    if (format.kind == SourceFormat::NONE) {
        if (hdr.picture_type == H263MI_PICTURE_I) return H263MI_ERR_PICTURE_FORMAT_MISSING;
        if (!ctx || !ctx->have_last) return H263MI_ERR_PICTURE_FORMAT_MISSING;
        format = ctx->last_format;
    }
    if (!format.dimensions(hdr.width, hdr.height)) return H263MI_ERR_PICTURE_FORMAT_INVALID;
    if (!hdr.width || !hdr.height) return H263MI_ERR_PICTURE_FORMAT_INVALID;   // no picture to hold (back-end limit)
    if (out.size_fits && !out.size_fits(hdr.width, hdr.height)) return H263MI_ERR_PICTURE_FORMAT_INVALID;   // (see ParsedPicture::size_fits)

    out.desc.width = hdr.width;
    out.desc.height = hdr.height;
    out.desc.picture_type = hdr.picture_type;
    out.desc.pquant = hdr.quantizer;
    out.desc.use_deblocker = hdr.use_deblocker ? 1 : 0;
    out.desc.temporal_reference = hdr.temporal_reference;
    out.next.have_last = true;
    out.next.last_header_format = hdr.format;
    out.next.last_header_options = hdr.options;
    out.next.last_format = format;

    const size_t mb_per_line = (hdr.width + 15u) / 16u, mb_height = (hdr.height + 15u) / 16u;   // state.rs:173-174
    const size_t total = mb_per_line * mb_height;
    int in_force_quantizer = hdr.quantizer;

    // Output arrays at their largest size for this picture, written in place and cut to what was used at the end (no
    // per-element push_back; the event array grows by doubling, a macroblock at a time).
    static_assert(sizeof(Mv) == sizeof(uint32_t), "Mv is stored in ParsedPicture::scratch");
    out.scratch.resize(2 * mb_per_line * 4);         // vectors of the current and the previous macroblock row, 4 per macroblock
    Mv *const pv_rows = reinterpret_cast<Mv *>(out.scratch.data());
    // the vectors of the macroblock at (mb_line, col)
    const auto pv_at = [&](size_t line, size_t col) { return pv_rows + ((line & 1) * mb_per_line + col) * 4; };
    // the records go to the caller's array when it can hold the picture, else into out.mbs
    const bool ext = out.mbs_ext != nullptr && total <= out.mbs_ext_cap;
    out.mbs_ext_used = ext;
    // sparse records (ParsedPicture::sparse_records): no record for a macroblock that is not coded, a word per group of 8
    const bool sparse_rec = out.sparse_records;
    const size_t groups_per_line = (mb_per_line + 7) / 8;
    // the word arrays -- events, block offsets, group index -- go to the caller's memory when all of it can hold this picture's
    // worst case (ParsedPicture::events_ext): nothing below ever checks for room there
    const bool wext = !want_dense && out.events_ext && out.first_event_ext && out.events_ext_cap >= event_words_bound(len, total) &&
                      out.first_event_ext_cap >= block_offset_words_bound(total) &&
                      (!sparse_rec || (out.group_index_ext && out.group_index_ext_cap >= groups_per_line * mb_height));
    out.words_ext_used = wext;
    const uint32_t event_base = wext ? out.event_base : 0u;
    out.group_index.clear();
    if (sparse_rec && wext) memset(out.group_index_ext, 0, groups_per_line * mb_height * sizeof(uint32_t));
    else if (sparse_rec) out.group_index.assign(groups_per_line * mb_height, 0u);
    uint32_t *const group_index = wext ? out.group_index_ext : out.group_index.data();
    size_t n_rec = 0;                                // records written (== n_mbs unless sparse)
    bool any_inter = false;
    // (no zero-fill of the record array: a record is assembled in registers and stored whole, 32 bytes, when its macroblock
    // is done -- round 3 cleared 261 KB per 1080p picture first and then wrote most of it again)
    if (!ext) out.mbs.resize(total);
    h263mi_mb_record *const recs = ext ? out.mbs_ext : out.mbs.data();
    if (!wext) out.block_first_event.resize(block_offset_words_bound(total));
    uint32_t *const first_event = wext ? out.first_event_ext : out.block_first_event.data();
    first_event[0] = event_base;
    size_t n_mbs = 0, n_events = 0, n_blocks = 0;
    // A record is written once and never read again by this thread -- in the product it lies in pinned staging memory that
    // only the copy engine reads: non-temporal stores (no read-for-ownership of the line, no place taken in the caches the
    // parser's tables and the bitstream live in; 261 KB of records per 1080p picture), with a fence in front of the return.
    // On the GPU boxes' EPYC: realistic P pictures 10.2 k -> 12.2 k pictures/s on one thread, 137 k -> 171 k on 16 (records
    // into a 16.7 MB ring, the shape of the batch staging); key frames +-0 (profiles/r05_f_nt_records.txt).
    const auto store_record = [](h263mi_mb_record *dst, const h263mi_mb_record &src) {
#if H263MI_STREAM_RECORDS
        if (((uintptr_t)dst & 15u) == 0) {
            __m128i lo, hi;
            __builtin_memcpy(&lo, &src, 16);
            __builtin_memcpy(&hi, reinterpret_cast<const uint8_t *>(&src) + 16, 16);
            _mm_stream_si128(reinterpret_cast<__m128i *>(dst), lo);
            _mm_stream_si128(reinterpret_cast<__m128i *>(dst) + 1, hi);
            return;
        }
#endif
        *dst = src;
    };
    const auto finish = [&](int code) {
#if H263MI_STREAM_RECORDS
        _mm_sfence();
#endif
        out.n_macroblocks = n_mbs < total ? n_mbs : total;
        out.any_inter = any_inter || n_mbs < total;      // (macroblocks the bitstream does not reach are padded as Inter, state.rs:421-427)
        if (ext) out.n_mbs_ext = sparse_rec ? n_rec : (n_mbs < total ? n_mbs : total);
        else out.mbs.resize(sparse_rec ? n_rec : n_mbs);
        if (wext) {
            out.block_first_event.clear();
            out.events.clear();
            out.n_events_ext = n_events;
        } else {
            out.block_first_event.resize(n_blocks + 1);
            out.events.resize(n_events);
        }
        out.n_coded_blocks = n_blocks;
        return code;
    };

    size_t mb_col = 0, mb_line = 0;                  // position of the macroblock being decoded
    const VlcTable &t_mcbpc_i = mcbpc_i_table(), &t_mcbpc_p = mcbpc_p_table(), &t_cbpy = cbpy_table(), &t_mvd = mvd_table();
    const bool is_i = hdr.picture_type == H263MI_PICTURE_I, is_p = hdr.picture_type == H263MI_PICTURE_P;
    const bool umv_vectors = (running_options & OPT_UNRESTRICTED_MOTION_VECTORS) && hdr.has_plusptype;
    const bool umv = (running_options & OPT_UNRESTRICTED_MOTION_VECTORS) != 0;
    // The macroblock header of an I or P picture can be read out of 64-bit windows (COD + MCBPC + CBPY + DQUANT are 22
    // bits at most, a vector 26) unless it uses Annex D vectors or Annex T.  Anything irregular -- an invalid code, the
    // end of the data -- is left to the field-by-field path below, which is the definition of the behaviour.
    const bool window_header = (is_i || is_p) && !umv_vectors && !(running_options & OPT_MODIFIED_QUANTIZATION) && !out.field_by_field;
    const VlcTable &t_mcbpc = is_i ? t_mcbpc_i : t_mcbpc_p;
    static const int kDquant[4] = {-1, -2, 1, 2};
    const HeaderTables &ht = header_tables();
    // blocks out of 64-bit windows (block_events_fast): the product's configuration -- events only, no dense blocks
    const bool fast_blocks = !want_dense && !out.field_by_field && r.size_bits() >= 64;
    const size_t end64 = r.size_bits() >= 64 ? r.size_bits() - 64 : 0;
    const bool sorenson_v1 = sorenson && hdr.version == 1;
    const VlcTable &t_tcoef = tcoef_table();
    const uint16_t *const header12 = is_i ? ht.i12 : ht.p12;
    for (;;) {                                       // state.rs:193-417
        size_t mb_checkpoint = r.position();         // decode_macroblock runs in a transaction (macroblock.rs:454)
        int mrc = H263MI_OK;
        bool stuffing = false, uncoded = false;
        int mb_type = 0, cb = 0, cr = 0, luma = 0, dquant = 0;
        bool has_dquant = false;
        Mv mvd[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
        bool have_header = false, run_taken = false;
        if (window_header && r.remaining() >= 64) do {
            size_t avail = r.remaining();
            uint64_t w = r.peek_window();
            uint32_t used = 0, total_used = 0;       // bits taken from this window / from earlier windows
            uint32_t shifted = 0;                    // bits the window has been shifted by already (a run of COD = 1 in front)
            const auto top32 = [&]() { return (uint32_t)((w << used) >> 32); };
            if (is_p) {
                used = 1;
                if (w >> 63) {
                    // COD = 1.  Real P pictures are mostly runs of these (60-70 % of the macroblocks): the whole run of
                    // one bits at the top of the window is taken at once -- as many as the picture still has room for; a
                    // macroblock beyond that goes the ordinary way and is reported there.  Each is Macroblock::Uncoded
                    // (state.rs:207-216): an INTER record with zero vectors in the zero-initialised array.
                    const size_t room = total > n_mbs ? total - n_mbs : 0;
                    size_t run = (size_t)__builtin_clzll(~w | 1ull);           // leading ones, 1..63
                    if (run > room) run = room;
                    if (run == 0) { uncoded = true; r.advance(1); have_header = true; break; }   // (no room: reported below)
                    h263mi_mb_record skipped{};
                    skipped.mb_type = H263MI_MB_INTER;
                    skipped.quant = (uint8_t)(in_force_quantizer < 1 ? 1 : in_force_quantizer);
                    if (!sparse_rec)
                        for (size_t k = 0; k < run; k++) store_record(recs + n_mbs + k, skipped);
                    any_inter = true;
                    n_mbs += run;
                    // zero vectors for the run, row segment by row segment of the two-row ring
                    for (size_t left = run; left;) {
                        const size_t seg = left < mb_per_line - mb_col ? left : mb_per_line - mb_col;
                        memset(pv_at(mb_line, mb_col), 0, seg * 4 * sizeof(Mv));
                        left -= seg;
                        mb_col += seg;
                        if (mb_col == mb_per_line) { mb_col = 0; mb_line++; }
                    }
                    r.advance((uint32_t)run);
                    // The run ends in front of a coded macroblock (a zero bit) unless the window or the picture cut it short:
                    // that macroblock's header is taken out of the same window right away -- one trip through the loop, and one
                    // data-dependent branch, for "some macroblocks that are not coded, then one that is" (round 4 took two).
                    // Enough of the window must be left for COD + MCBPC + CBPY + DQUANT (15 bits) and a vector pair (26).
                    if (run > 57 - 41 || n_mbs >= total || r.remaining() < 64) { run_taken = true; break; }
                    w <<= run;
                    if (w >> 63) { run_taken = true; break; }
                    shifted = (uint32_t)run;
                    avail -= run;
                    mb_checkpoint = r.position();                              // the coded macroblock's own transaction
                }
            }
            bool intra;
            if (const uint32_t e = header12[(uint32_t)((w << used) >> 52)]) {
                // MCBPC and CBPY in one access (the short code words: nearly every macroblock of a real stream)
                used += e & 15u;
                mb_type = (int)((e >> 4) & 7u);
                const uint32_t coded6 = e >> 7;
                luma = (int)(coded6 >> 2); cb = (int)((coded6 >> 1) & 1u); cr = (int)(coded6 & 1u);
                intra = mb_type == H263MI_MB_INTRA || mb_type == H263MI_MB_INTRA_Q;
            } else {
                const VlcTable::Slot &m = t_mcbpc.lookup32(top32());
                if (!m.valid) break;
                used += m.len;
                if (m.v0 < 0) { stuffing = true; r.advance(used); have_header = true; break; }
                mb_type = m.v0; cb = m.v1; cr = m.v2;
                intra = mb_type == H263MI_MB_INTRA || mb_type == H263MI_MB_INTRA_Q;
                const VlcTable::Slot &c = t_cbpy.lookup32(top32());
                if (!c.valid) break;
                used += c.len;
                luma = intra ? c.v0 : (~c.v0 & 0xf);                     // macroblock.rs:479-489
            }
            if (mb_type == H263MI_MB_INTER_Q || mb_type == H263MI_MB_INTRA_Q || mb_type == H263MI_MB_INTER4V_Q) {
                dquant = kDquant[(w << used) >> 62];                     // decode_dquant (macroblock.rs:257-271)
                used += 2;
                has_dquant = true;
            }
            bool ok = true;
            if (!intra) {
                const int n_mv = (mb_type == H263MI_MB_INTER4V || mb_type == H263MI_MB_INTER4V_Q) ? 4 : 1;
                for (int k = 0; k < n_mv; k++) {
                    if (used + shifted > 57 - 26) {  // not enough of the window left for two more code words
                        if (total_used + used > avail) { ok = false; break; }
                        r.advance(used);
                        total_used += used;
                        used = 0;
                        shifted = 0;
                        w = r.peek_window();
                    }
                    if (const uint32_t e = ht.mvd10[(uint32_t)((w << used) >> 54)]) {
                        // both code words of the pair in one access (small differences: slow motion, good prediction)
                        used += e & 15u;
                        mvd[k] = Mv{(int16_t)((int32_t)(e << 22) >> 26), (int16_t)((int32_t)(e << 16) >> 26)};
                        continue;
                    }
                    const VlcTable::Slot &sx = t_mvd.lookup32(top32());  // decode_motion_vector (macroblock.rs:414-438)
                    used += sx.len;
                    const VlcTable::Slot &sy = t_mvd.lookup32(top32());
                    used += sy.len;
                    if (!sx.valid || !sy.valid) { ok = false; break; }
                    mvd[k] = Mv{sx.v0, sy.v0};
                }
            }
            if (!ok || total_used + used > avail) { r.rollback(mb_checkpoint); break; }   // irregular: field by field
            r.advance(used);
            have_header = true;
        } while (0);
        if (run_taken) continue;
        if (!have_header) {
            stuffing = uncoded = has_dquant = false;
            do {                                     // decode_macroblock (macroblock.rs:445-549)
                uint32_t v;
                uint32_t cod = 0;
                if (!is_i && (mrc = r.read_bits(1, cod)) != H263MI_OK) break;
                if (cod) { uncoded = true; break; }
                VlcHit h;
                if (is_i) mrc = t_mcbpc_i.decode(r, h);
                else if (is_p) mrc = t_mcbpc_p.decode(r, h);
                else mrc = H263MI_ERR_UNIMPLEMENTED_DECODING;            // macroblock.rs:461-465
                if (mrc != H263MI_OK) break;
                if (!h.valid) { mrc = H263MI_ERR_INVALID_MACROBLOCK_HEADER; break; }
                if (h.v0 < 0) { stuffing = true; break; }
                mb_type = h.v0; cb = h.v1; cr = h.v2;
                const bool intra = mb_type == H263MI_MB_INTRA || mb_type == H263MI_MB_INTRA_Q;
                if ((mrc = t_cbpy.decode(r, h)) != H263MI_OK) break;
                if (!h.valid) { mrc = H263MI_ERR_INVALID_MACROBLOCK_CODED_BITS; break; }
                luma = intra ? h.v0 : (~h.v0 & 0xf);                     // macroblock.rs:479-489
                if (running_options & OPT_MODIFIED_QUANTIZATION) { mrc = H263MI_ERR_UNIMPLEMENTED_DECODING; break; }   // macroblock.rs:497-498
                if (mb_type == H263MI_MB_INTER_Q || mb_type == H263MI_MB_INTRA_Q || mb_type == H263MI_MB_INTER4V_Q) {
                    if ((mrc = r.read_bits(2, v)) != H263MI_OK) break;   // decode_dquant (macroblock.rs:257-271)
                    dquant = kDquant[v];
                    has_dquant = true;
                }
                if (!intra) {
                    const int n_mv = (mb_type == H263MI_MB_INTER4V || mb_type == H263MI_MB_INTER4V_Q) ? 4 : 1;
                    for (int k = 0; k < n_mv && mrc == H263MI_OK; k++) {
                        if (umv_vectors) {
                            int ux, uy;                                  // Annex D vectors (macroblock.rs:424-430)
                            if ((mrc = r.read_umv(ux)) != H263MI_OK) break;
                            if ((mrc = r.read_umv(uy)) != H263MI_OK) break;
                            mvd[k] = Mv{(int16_t)ux, (int16_t)uy};
                            continue;
                        }
                        VlcHit hx, hy;                                   // decode_motion_vector (macroblock.rs:414-438)
                        if ((mrc = t_mvd.decode(r, hx)) != H263MI_OK) break;
                        if (!hx.valid) { mrc = H263MI_ERR_INVALID_MVD; break; }
                        if ((mrc = t_mvd.decode(r, hy)) != H263MI_OK) break;
                        if (!hy.valid) { mrc = H263MI_ERR_INVALID_MVD; break; }
                        mvd[k] = Mv{hx.v0, hy.v0};
                    }
                }
            } while (0);
        }
        if (mrc != H263MI_OK) {
            r.rollback(mb_checkpoint);
            // state.rs:387-412: in standard H.263 a macroblock header error looks for the next GOB or picture
            // start code (never in Sorenson mode); EOF ends the picture; anything else fails the decode
            if (!sorenson && (mrc == H263MI_ERR_INVALID_MACROBLOCK_HEADER || mrc == H263MI_ERR_INVALID_MACROBLOCK_CODED_BITS)) {
                int grc = H263MI_OK;
                if (resync_ends_picture(r, grc)) break;
                return finish(grc);
            }
            if (mrc == kEof) break;
            return finish(mrc);
        }
        if (stuffing) continue;                      // Macroblock::Stuffing (state.rs:206)

        h263mi_mb_record rec{};                      // assembled here, stored whole below
        Mv motion_vectors[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
        if (uncoded) {
            // Macroblock::Uncoded: an I picture has no COD bit, so this is always a P picture (state.rs:207-216)
            // (the quantiser of a macroblock without coefficients is never used; a PQUANT of 0, which the header syntax
            // can express, must not make the record invalid)
            rec.mb_type = H263MI_MB_INTER;
            rec.quant = (uint8_t)(in_force_quantizer < 1 ? 1 : in_force_quantizer);
        } else {
            const int q = in_force_quantizer + (has_dquant ? dquant : 0);      // state.rs:226-227
            in_force_quantizer = q < 1 ? 1 : (q > 31 ? 31 : q);
            const bool intra = mb_type == H263MI_MB_INTRA || mb_type == H263MI_MB_INTRA_Q;
            if (!intra) {                                                      // state.rs:229-285
                const bool four = mb_type == H263MI_MB_INTER4V || mb_type == H263MI_MB_INTER4V_Q;
                for (int k = 0; k < (four ? 4 : 1); k++) {
                    const Mv pred = predict_candidate(pv_at(mb_line, 0), pv_at(mb_line + 1, 0), motion_vectors, mb_per_line, k, mb_col, mb_line);
                    if (!umv) {
                        // halfpel_decode with the standard range: a sum outside [-32, 32) takes the other
                        // representative of the difference (mvd_pred.rs:70-117, HalfPel::invert types.rs:736-742)
                        int x = mvd[k].x + pred.x, y = mvd[k].y + pred.y;
                        if (x < -32) x += 64; else if (x >= 32) x -= 64;
                        if (y < -32) y += 64; else if (y >= 32) y -= 64;
                        motion_vectors[k] = Mv{(int16_t)x, (int16_t)y};
                    } else {
                        motion_vectors[k] = Mv{halfpel_decode(running_options, hdr, hdr.width, pred.x, mvd[k].x, true),
                                               halfpel_decode(running_options, hdr, hdr.height, pred.y, mvd[k].y, false)};
                    }
                }
                if (!four) motion_vectors[1] = motion_vectors[2] = motion_vectors[3] = motion_vectors[0];
            }
            rec.mb_type = (uint8_t)mb_type;
            rec.quant = (uint8_t)in_force_quantizer;
            rec.coeff_index = (uint32_t)n_blocks;
            // room for the events of this macroblock (6 blocks x 64 at most)
            if (!wext && out.events.size() < n_events + 6 * 64) out.events.resize((n_events + 6 * 64) * 2);
            uint32_t *const events_now = wext ? out.events_ext : out.events.data();
            const uint32_t coded6 = ((uint32_t)luma << 2) | ((uint32_t)cb << 1) | (uint32_t)cr;   // bit 5 - b: block b
            // state.rs:287-381: the six blocks in order; an inter block without TCOEFs has no bits at all (block.rs:
            // 684-687), so an inter macroblock only visits its coded blocks (no branch per absent block)
            uint32_t todo = intra ? 0x3fu : coded6;
            // coded6 (bit 5 - b = block b, the order of the bitstream) -> the record's cbp (bit b)
            static const uint8_t kCbpOfCoded6[64] = {
#define C6(v) (uint8_t)((((v) >> 5) & 1) | ((((v) >> 4) & 1) << 1) | ((((v) >> 3) & 1) << 2) | ((((v) >> 2) & 1) << 3) | ((((v) >> 1) & 1) << 4) | (((v) & 1) << 5))
#define C6x8(v) C6(v), C6(v + 1), C6(v + 2), C6(v + 3), C6(v + 4), C6(v + 5), C6(v + 6), C6(v + 7)
                C6x8(0), C6x8(8), C6x8(16), C6x8(24), C6x8(32), C6x8(40), C6x8(48), C6x8(56)
#undef C6x8
#undef C6
            };
            if (!intra && coded6 && fast_blocks) {
                // every coded block of an inter macroblock in one loop (inter_macroblock_events)
                const uint32_t n_coded = (uint32_t)__builtin_popcount(coded6);
                const InterRun ir = sorenson_v1
                    ? inter_macroblock_events<true>(r.data(), r.position(), end64, n_coded, events_now + n_events,
                                                    first_event + n_blocks, event_base + (uint32_t)n_events, ht.tcoef13[sorenson_v1 ? 1 : 0])
                    : inter_macroblock_events<false>(r.data(), r.position(), end64, n_coded, events_now + n_events,
                                                     first_event + n_blocks, event_base + (uint32_t)n_events, ht.tcoef13[sorenson_v1 ? 1 : 0]);
                if (ir.rc != H263MI_OK) return finish(ir.rc);
                if (ir.done) {
                    r.rollback(ir.pos);
                    rec.cbp = kCbpOfCoded6[coded6];
                    n_events += ir.n_ev;
                    n_blocks += n_coded;
                    todo = 0;
                }
            }
            if (intra && fast_blocks) {
                // the six blocks of an intra macroblock as one list of items (intra_macroblock_events)
                uint8_t dcs[8];
                const InterRun ir = sorenson_v1
                    ? intra_macroblock_events<true>(r.data(), r.position(), end64, coded6, events_now + n_events,
                                                    first_event + n_blocks, event_base + (uint32_t)n_events, ht.tcoef13[sorenson_v1 ? 1 : 0], dcs)
                    : intra_macroblock_events<false>(r.data(), r.position(), end64, coded6, events_now + n_events,
                                                     first_event + n_blocks, event_base + (uint32_t)n_events, ht.tcoef13[sorenson_v1 ? 1 : 0], dcs);
                if (ir.rc != H263MI_OK) return finish(ir.rc);
                if (ir.done) {
                    r.rollback(ir.pos);
                    rec.cbp = kCbpOfCoded6[coded6];
                    for (int b = 0; b < 6; b++) rec.intradc[b] = dcs[b];
                    n_events += ir.n_ev;
                    n_blocks += (uint32_t)__builtin_popcount(coded6);
                    todo = 0;
                }
            }
            while (todo) {
                const int b = __builtin_clz(todo) - 26;                        // bit 5 - b, highest first
                todo &= ~(0x20u >> b);
                const bool coded = (coded6 >> (5 - b)) & 1u;
                // run-length expansion + de-zigzag of inverse_rle (rle.rs:117-136) as the events arrive;
                // dequantisation is left to the GPU.  A run that walks past zigzag 63 voids the block (rle.rs:125-127).
                int16_t *dense = nullptr;
                if (coded && want_dense) {
                    const size_t base = out.coeffs.size();
                    out.coeffs.resize(base + 64, 0);
                    dense = out.coeffs.data() + base;
                }
                uint32_t *const ev = events_now + n_events;            // a block places 64 events at most
                size_t zz = intra ? 1 : 0, n_ev = 0;
                bool overrun = false;
                uint8_t dc = 0;
                bool dc_taken = false, tcoef_left = coded;
                if (fast_blocks) {
                    // INTRADC and the events out of 64-bit windows while whole windows lie behind the cursor (block_events_fast)
                    size_t bpos = r.position();
                    if (intra && bpos <= end64) {
                        const uint32_t code = (uint32_t)(r.window_at(bpos) >> 56);
                        if (code == 0 || code == 128) return finish(H263MI_ERR_INVALID_INTRA_DC);      // IntraDc::from_u8, types.rs:930-936
                        dc = (uint8_t)code;
                        dc_taken = true;
                        bpos += 8;
                    }
                    if (tcoef_left && (dc_taken || !intra) && bpos <= end64) {
                        const BlockRun br = sorenson_v1 ? block_events_fast<true>(r.data(), bpos, end64, (uint32_t)zz, ev, t_tcoef)
                                                        : block_events_fast<false>(r.data(), bpos, end64, (uint32_t)zz, ev, t_tcoef);
                        if (br.rc != H263MI_OK) return finish(br.rc);
                        bpos = br.pos; zz = br.zz; n_ev = br.n_ev; overrun = br.overrun; tcoef_left = br.more;
                    }
                    r.rollback(bpos);
                }
                if ((intra && !dc_taken) || tcoef_left)
                rc = decode_block_to(r, sorenson, hdr.version, intra && !dc_taken, tcoef_left, dc, [&](bool, int run, int level) {
                    if (overrun) return;
                    zz += (size_t)run;
                    if (zz >= 64) { overrun = true; return; }
                    const uint32_t pos = kZigzagRaster[zz++];
                    if (dense) dense[pos] = (int16_t)level;
                    ev[n_ev++] = ((uint32_t)(uint16_t)(int16_t)level << 16) | pos;
                }, out.field_by_field);
                if (rc != H263MI_OK) return finish(rc);   // `?` in the reference: a block error fails the whole decode
                if (intra) rec.intradc[b] = dc;
                if (!coded) continue;
                rec.cbp |= (uint8_t)(1u << b);
                if (overrun) rec.kill |= (uint8_t)(1u << b);
                n_events += n_ev;
                first_event[++n_blocks] = event_base + (uint32_t)n_events;
            }
        }
        if (n_mbs >= total) {
            // more macroblocks than the picture holds: the reference indexes its level arrays out of bounds
            // here (a panic); reported as an invalid bitstream instead
            return finish(H263MI_ERR_INVALID_BITSTREAM);
        }
        for (int k = 0; k < 4; k++) {
            rec.mv[k][0] = motion_vectors[k].x;
            rec.mv[k][1] = motion_vectors[k].y;
            pv_at(mb_line, mb_col)[k] = motion_vectors[k];
        }
        any_inter = any_inter || mb_type_is_inter(rec.mb_type);
        if (!sparse_rec) {
            store_record(recs + n_mbs, rec);
        } else if (!uncoded) {
            // (a macroblock that is not coded has no record: that is what the absence of one says)
            uint32_t &gw = group_index[mb_line * groups_per_line + (mb_col >> 3)];
            if (!(gw & 0xffu)) gw = (uint32_t)n_rec << 8;
            gw |= 1u << (mb_col & 7);
            store_record(recs + n_rec, rec);
            n_rec++;
        }
        n_mbs++;
        if (++mb_col == mb_per_line) { mb_col = 0; mb_line++; }
    }
    out.bits_consumed = r.position();
    return finish(H263MI_OK);
}

}  // namespace bits
}  // namespace h263mi
