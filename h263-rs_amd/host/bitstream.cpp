// bitstream.cpp -- see bitstream.hpp.  Plain C++17 host code (no HIP): it is linked into libh263mi.so
// and, separately, into the CPU-only parser test library (tests/parser).
#include "bitstream.hpp"

#include <cstring>

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
    // gather up to 40 bits around the cursor into a 64-bit window
    uint64_t window = 0;
    const size_t byte = pos_ >> 3, nbytes = nbits_ >> 3;
    for (size_t k = 0; k < 5; k++) window = (window << 8) | (byte + k < nbytes ? p_[byte + k] : 0);
    const uint32_t shift = 40 - (uint32_t)(pos_ & 7) - n;
    return n ? (uint32_t)((window >> shift) & ((n == 32) ? 0xffffffffull : ((1ull << n) - 1))) : 0u;
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

// ---------------------------------------------------------------------------------------------------
// VLC tables
// ---------------------------------------------------------------------------------------------------
VlcTable::VlcTable(const VlcCode *codes, size_t n) : max_len_(0)
{
    for (size_t i = 0; i < n; i++) {
        int l = (int)strlen(codes[i].bits);
        if (l > max_len_) max_len_ = l;
    }
    lut_.assign((size_t)1 << max_len_, Slot{0, 0, 0, 0, 0});
    // For every max_len_-bit pattern: the length at which a bit-by-bit walk of the code tree stops --
    // either on a code word, or on the shortest prefix that no code word starts with (the tree's
    // "invalid" leaves).
    for (uint32_t pat = 0; pat < lut_.size(); pat++) {
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
                lut_[pat] = Slot{(uint8_t)l, 1, codes[which].v0, codes[which].v1, codes[which].v2};
                break;
            }
            if (!extendable) {
                lut_[pat] = Slot{(uint8_t)l, 0, 0, 0, 0};
                break;
            }
        }
    }
}

int VlcTable::decode(BitReader &r, VlcHit &hit) const
{
    const Slot &s = lut_[r.peek_padded((uint32_t)max_len_)];
    if (s.len > r.remaining()) {
        // the data ends inside the code word: the reference reads bit by bit and fails on the missing bit
        (void)r.skip_bits((uint32_t)r.remaining());
        return kEof;
    }
    (void)r.skip_bits(s.len);
    hit.valid = s.valid != 0;
    hit.v0 = s.v0; hit.v1 = s.v1; hit.v2 = s.v2;
    return H263MI_OK;
}

#include "vlc_tables.inc"

#define H263MI_TABLE(fn, arr)                                                   \
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

// ---------------------------------------------------------------------------------------------------
// block layer: parser/block.rs:670-755
// ---------------------------------------------------------------------------------------------------
int decode_block(BitReader &r, bool sorenson, int version, bool intra, bool tcoef_present, ParsedBlock &out)
{
    const size_t checkpoint = r.position();          // with_transaction (block.rs:682)
    out = ParsedBlock();
    int rc = H263MI_OK;
    do {
        if (intra) {
            uint32_t code;
            if ((rc = r.read_u8(code)) != H263MI_OK) break;
            if (code == 0 || code == 128) { rc = H263MI_ERR_INVALID_INTRA_DC; break; }   // IntraDc::from_u8, types.rs:930-936
            out.has_intradc = true;
            out.intradc = (uint8_t)code;
        }
        while (tcoef_present) {
            VlcHit h;
            if ((rc = tcoef_table().decode(r, h)) != H263MI_OK) break;
            if (!h.valid) { rc = H263MI_ERR_INVALID_SHORT_COEFFICIENT; break; }
            bool last;
            int run, level;
            bool is_short;
            if (h.v0 < 0) {                              // ESCAPE (block.rs:689-724)
                uint32_t width = 8, v;
                if (sorenson && version == 1) {          // Sorenson v1: 1 bit selects an 11- or 7-bit LEVEL
                    if ((rc = r.read_bits(1, v)) != H263MI_OK) break;
                    width = v ? 11 : 7;
                }
                if ((rc = r.read_bits(1, v)) != H263MI_OK) break;
                last = v == 1;
                if ((rc = r.read_bits(6, v)) != H263MI_OK) break;
                run = (int)v;
                int32_t lv;
                if ((rc = r.read_signed_bits(width, lv)) != H263MI_OK) break;
                if (lv == 0) { rc = H263MI_ERR_INVALID_LONG_COEFFICIENT; break; }
                // (the reference's second check, `level == i16::MAX << level_width`, can never hold for a
                // sign-extended LEVEL of that width: block.rs:708-715)
                level = lv;
                is_short = false;
            } else {
                uint32_t sign;
                if ((rc = r.read_bits(1, sign)) != H263MI_OK) break;
                last = h.v0 != 0;
                run = h.v1;
                level = sign ? -(int)h.v2 : (int)h.v2;
                is_short = true;
            }
            if (out.n_tcoef >= (int)(sizeof(out.tcoef) / sizeof(out.tcoef[0]))) {
                // more events than a block can place: every further one lands beyond zigzag 63 anyway;
                // keep parsing (the bitstream position matters) but stop storing
            } else {
                out.tcoef[out.n_tcoef].is_short = is_short;
                out.tcoef[out.n_tcoef].run = (uint8_t)run;
                out.tcoef[out.n_tcoef].level = (int16_t)level;
                out.n_tcoef++;
            }
            tcoef_present = !last;
        }
    } while (0);
    if (rc != H263MI_OK) r.rollback(checkpoint);
    return rc;
}

// ---------------------------------------------------------------------------------------------------
// picture layer, Sorenson branch: parser/picture.rs:611-661, 271-327, 577-596
// ---------------------------------------------------------------------------------------------------
int decode_picture_header(BitReader &r, uint32_t decoder_options, PictureHeader &out, bool &is_picture)
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
        uint32_t gob_id, v;
        if ((rc = r.read_bits(5, gob_id)) != H263MI_OK) break;
        if (!(decoder_options & H263MI_SORENSON_SPARK_BITSTREAM)) {
            // standard H.263 PTYPE / PLUSPTYPE paths: SURVEY section 8 row f-4, not built yet
            rc = H263MI_ERR_UNIMPLEMENTED_DECODING;
            break;
        }
        out.version = (int)gob_id;                   // "Sorenson abuses the GOB ID as a version field"
        if ((rc = r.read_u8(v)) != H263MI_OK) break;
        out.temporal_reference = (uint16_t)v;
        // decode_sorenson_ptype (picture.rs:271-327)
        uint32_t fmt;
        if ((rc = r.read_bits(3, fmt)) != H263MI_OK) break;
        out.format_valid = true;
        switch (fmt) {
        case 0:
        case 1: {
            const uint32_t n = fmt == 0 ? 8 : 16;
            uint32_t w, h;
            if ((rc = r.read_bits(n, w)) != H263MI_OK) break;
            if ((rc = r.read_bits(n, h)) != H263MI_OK) break;
            out.width = (uint16_t)w; out.height = (uint16_t)h;
            break;
        }
        case 2: out.width = 352; out.height = 288; break;     // FullCif     (types.rs:168-180)
        case 3: out.width = 176; out.height = 144; break;     // QuarterCif
        case 4: out.width = 128; out.height = 96; break;      // SubQcif
        case 5: out.width = 320; out.height = 240; break;
        case 6: out.width = 160; out.height = 120; break;
        default: out.format_valid = false; break;              // SourceFormat::Reserved
        }
        if (rc != H263MI_OK) break;
        if ((rc = r.read_bits(2, v)) != H263MI_OK) break;
        out.picture_type = (uint8_t)v;
        if ((rc = r.read_bits(1, v)) != H263MI_OK) break;
        out.use_deblocker = v == 1;
        if ((rc = r.read_bits(5, v)) != H263MI_OK) break;
        out.quantizer = (uint8_t)v;
        for (;;) {                                   // decode_pei (picture.rs:577-596)
            if ((rc = r.read_bits(1, v)) != H263MI_OK) break;
            if (!v) break;
            if ((rc = r.read_u8(v)) != H263MI_OK) break;
            out.extra.push_back((uint8_t)v);
        }
        if (rc != H263MI_OK) break;
        is_picture = true;
    } while (0);
    if (rc != H263MI_OK) r.rollback(checkpoint);
    return rc;
}

// ---------------------------------------------------------------------------------------------------
// motion vector prediction: decoder/cpu/mvd_pred.rs
// ---------------------------------------------------------------------------------------------------
namespace {
struct Mv { int16_t x, y; };

// HalfPel::median_of (types.rs:772-800): the reference's comparison chain, which is the median
int16_t median3(int16_t self, int16_t mhs, int16_t rhs)
{
    if (self > mhs) {
        if (rhs > mhs) return rhs > self ? self : rhs;
        return mhs;
    }
    if (mhs > rhs) return rhs > self ? rhs : self;
    return mhs;
}

// predict_candidate (mvd_pred.rs:27-67); pv = vectors of the macroblocks decoded so far
Mv predict_candidate(const std::vector<Mv> &pv /* 4 per MB */, const Mv cur[4], size_t mb_per_line, int index)
{
    const size_t current_mb = pv.size() / 4, col = current_mb % mb_per_line;
    const Mv zero{0, 0};
    Mv mv1;
    if (index == 0 || index == 2) mv1 = col == 0 ? zero : pv[(current_mb - 1) * 4 + (size_t)index + 1];
    else mv1 = cur[index - 1];

    const size_t line = current_mb / mb_per_line;
    const size_t last_line_mb = (line ? line - 1 : 0) * mb_per_line + col;
    Mv mv2;
    if (index <= 1) {
        if (line == 0) mv2 = mv1;
        else mv2 = last_line_mb < current_mb ? pv[last_line_mb * 4 + (size_t)index + 2] : mv1;
    } else {
        mv2 = cur[0];
    }
    const bool end_of_line = col == (mb_per_line ? mb_per_line - 1 : 0);
    Mv mv3;
    if (index <= 1) {
        if (end_of_line) mv3 = zero;
        else if (line == 0) mv3 = mv1;
        else mv3 = last_line_mb + 1 < current_mb ? pv[(last_line_mb + 1) * 4 + 2] : mv1;
    } else {
        mv3 = cur[1];
    }
    return Mv{median3(mv1.x, mv2.x, mv3.x), median3(mv1.y, mv2.y, mv3.y)};
}

// halfpel_decode (mvd_pred.rs:70-117) for the cases a Sorenson stream can reach: no
// UNRESTRICTED_MOTION_VECTORS option is ever set there, so the range is the standard [-32, 32) half-pels
int16_t halfpel_decode(int16_t predictor, int16_t mvd)
{
    int out = mvd + predictor;
    if (!(-32 <= out && out < 32)) {
        const int inv = mvd > 0 ? mvd - 64 : (mvd < 0 ? mvd + 64 : mvd);     // HalfPel::invert (types.rs:736-742)
        out = inv + predictor;
    }
    return (int16_t)out;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------
// whole picture: state.rs:138-427 up to the cut line
// ---------------------------------------------------------------------------------------------------
int parse_picture(const uint8_t *data, size_t len, uint32_t decoder_options, ParsedPicture &out)
{
    out = ParsedPicture();
    BitReader r(data, len);
    PictureHeader hdr;
    bool is_picture = false;
    int rc = decode_picture_header(r, decoder_options, hdr, is_picture);
    if (rc != H263MI_OK) return rc;
    if (!is_picture) return H263MI_ERR_MIDDLE_OF_BITSTREAM;                    // state.rs:143-145
    if (!hdr.format_valid || !hdr.width || !hdr.height) return H263MI_ERR_PICTURE_FORMAT_INVALID;   // state.rs:169-171
    const bool sorenson = (decoder_options & H263MI_SORENSON_SPARK_BITSTREAM) != 0;

    out.desc.width = hdr.width;
    out.desc.height = hdr.height;
    out.desc.picture_type = hdr.picture_type;
    out.desc.pquant = hdr.quantizer;
    out.desc.use_deblocker = hdr.use_deblocker ? 1 : 0;
    out.desc.temporal_reference = hdr.temporal_reference;

    const size_t mb_per_line = (hdr.width + 15u) / 16u, mb_height = (hdr.height + 15u) / 16u;   // state.rs:173-174
    const size_t total = mb_per_line * mb_height;
    int in_force_quantizer = hdr.quantizer;
    std::vector<Mv> predictor_vectors;             // 4 per decoded macroblock
    predictor_vectors.reserve(total * 4);

    for (;;) {                                       // state.rs:193-417
        const size_t mb_checkpoint = r.position();   // decode_macroblock runs in a transaction (macroblock.rs:454)
        int mrc = H263MI_OK;
        bool stuffing = false, uncoded = false;
        int mb_type = 0, cb = 0, cr = 0, luma = 0, dquant = 0;
        bool has_dquant = false;
        Mv mvd[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
        do {                                         // decode_macroblock (macroblock.rs:445-549)
            uint32_t v;
            uint32_t cod = 0;
            if (hdr.picture_type != H263MI_PICTURE_I && (mrc = r.read_bits(1, cod)) != H263MI_OK) break;
            if (cod) { uncoded = true; break; }
            VlcHit h;
            if (hdr.picture_type == H263MI_PICTURE_I) mrc = mcbpc_i_table().decode(r, h);
            else if (hdr.picture_type == H263MI_PICTURE_P) mrc = mcbpc_p_table().decode(r, h);
            else mrc = H263MI_ERR_UNIMPLEMENTED_DECODING;            // macroblock.rs:461-465
            if (mrc != H263MI_OK) break;
            if (!h.valid) { mrc = H263MI_ERR_INVALID_MACROBLOCK_HEADER; break; }
            if (h.v0 < 0) { stuffing = true; break; }
            mb_type = h.v0; cb = h.v1; cr = h.v2;
            const bool intra = mb_type == H263MI_MB_INTRA || mb_type == H263MI_MB_INTRA_Q;
            if ((mrc = cbpy_table().decode(r, h)) != H263MI_OK) break;
            if (!h.valid) { mrc = H263MI_ERR_INVALID_MACROBLOCK_CODED_BITS; break; }
            luma = intra ? h.v0 : (~h.v0 & 0xf);                     // macroblock.rs:479-489
            if (mb_type == H263MI_MB_INTER_Q || mb_type == H263MI_MB_INTRA_Q || mb_type == H263MI_MB_INTER4V_Q) {
                if ((mrc = r.read_bits(2, v)) != H263MI_OK) break;   // decode_dquant (macroblock.rs:257-271)
                static const int kDquant[4] = {-1, -2, 1, 2};
                dquant = kDquant[v];
                has_dquant = true;
            }
            if (!intra) {
                const int n_mv = (mb_type == H263MI_MB_INTER4V || mb_type == H263MI_MB_INTER4V_Q) ? 4 : 1;
                for (int k = 0; k < n_mv && mrc == H263MI_OK; k++) {
                    VlcHit hx, hy;                                   // decode_motion_vector (macroblock.rs:414-438)
                    if ((mrc = mvd_table().decode(r, hx)) != H263MI_OK) break;
                    if (!hx.valid) { mrc = H263MI_ERR_INVALID_MVD; break; }
                    if ((mrc = mvd_table().decode(r, hy)) != H263MI_OK) break;
                    if (!hy.valid) { mrc = H263MI_ERR_INVALID_MVD; break; }
                    mvd[k] = Mv{hx.v0, hy.v0};
                }
            }
        } while (0);
        if (mrc != H263MI_OK) {
            r.rollback(mb_checkpoint);
            // state.rs:387-412: macroblock errors would resynchronise to the next GOB in standard H.263 (not in
            // Sorenson mode); EOF ends the picture; anything else fails the decode
            if (mrc == kEof) break;
            return mrc;
        }
        if (stuffing) continue;                      // Macroblock::Stuffing (state.rs:206)

        h263mi_mb_record rec;
        memset(&rec, 0, sizeof rec);
        Mv motion_vectors[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
        if (uncoded) {
            // Macroblock::Uncoded: an I picture has no COD bit, so this is always a P picture (state.rs:207-216)
            rec.mb_type = H263MI_MB_INTER;
            rec.quant = (uint8_t)in_force_quantizer;
        } else {
            const int q = in_force_quantizer + (has_dquant ? dquant : 0);      // state.rs:226-227
            in_force_quantizer = q < 1 ? 1 : (q > 31 ? 31 : q);
            const bool intra = mb_type == H263MI_MB_INTRA || mb_type == H263MI_MB_INTRA_Q;
            if (!intra) {                                                      // state.rs:229-285
                const bool four = mb_type == H263MI_MB_INTER4V || mb_type == H263MI_MB_INTER4V_Q;
                for (int k = 0; k < (four ? 4 : 1); k++) {
                    const Mv pred = predict_candidate(predictor_vectors, motion_vectors, mb_per_line, k);
                    motion_vectors[k] = Mv{halfpel_decode(pred.x, mvd[k].x), halfpel_decode(pred.y, mvd[k].y)};
                }
                if (!four) motion_vectors[1] = motion_vectors[2] = motion_vectors[3] = motion_vectors[0];
            }
            rec.mb_type = (uint8_t)mb_type;
            rec.quant = (uint8_t)in_force_quantizer;
            rec.coeff_index = (uint32_t)(out.coeffs.size() / 64);
            const int coded[6] = {(luma >> 3) & 1, (luma >> 2) & 1, (luma >> 1) & 1, luma & 1, cb, cr};
            for (int b = 0; b < 6; b++) {                                      // state.rs:287-381
                ParsedBlock blk;
                rc = decode_block(r, sorenson, hdr.version, intra, coded[b] != 0, blk);
                if (rc != H263MI_OK) return rc;      // `?` in the reference: a block error fails the whole decode
                if (intra) rec.intradc[b] = blk.intradc;
                if (!coded[b]) continue;
                rec.cbp |= (uint8_t)(1u << b);
                const size_t base = out.coeffs.size();
                out.coeffs.resize(base + 64, 0);
                // run-length expansion + de-zigzag of inverse_rle (rle.rs:117-136); dequantisation is left
                // to the GPU.  A run that walks past zigzag 63 voids the block (rle.rs:125-127).
                size_t zz = intra ? 1 : 0;
                for (int t = 0; t < blk.n_tcoef; t++) {
                    zz += blk.tcoef[t].run;
                    if (zz >= 64) { rec.kill |= (uint8_t)(1u << b); break; }
                    out.coeffs[base + kZigzagRaster[zz]] = blk.tcoef[t].level;
                    zz++;
                }
            }
        }
        if (predictor_vectors.size() / 4 >= total) {
            // more macroblocks than the picture holds: the reference indexes its level arrays out of bounds
            // here (a panic); reported as an invalid bitstream instead
            return H263MI_ERR_INVALID_BITSTREAM;
        }
        for (int k = 0; k < 4; k++) {
            rec.mv[k][0] = motion_vectors[k].x;
            rec.mv[k][1] = motion_vectors[k].y;
            predictor_vectors.push_back(motion_vectors[k]);
        }
        out.mbs.push_back(rec);
    }
    out.bits_consumed = r.position();
    return H263MI_OK;
}

}  // namespace bits
}  // namespace h263mi
