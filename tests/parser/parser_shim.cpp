// parser_shim.cpp -- C entry points over h263-rs_amd/host/bitstream.{hpp,cpp} for the CPU-only parser
// tests (tests/test_parser.py).  Test infrastructure: the product reaches the parser through
// h263mi_decode_next_picture only.
#include <cstring>

#include "../../h263-rs_amd/host/bitstream.hpp"

using namespace h263mi::bits;

extern "C" {

// table: 0 TCOEF, 1 MCBPC-I, 2 MCBPC-P, 3 CBPY, 4 MVD.  out[i] = {status, v0, v1, v2}; status 1 valid, 0 invalid
// code, negative = error code (EOF).  Returns the number of entries written.
int pt_read_vlc(int table, const uint8_t *data, size_t len, int n, int32_t *out)
{
    const VlcTable *t = table == 0 ? &tcoef_table() : table == 1 ? &mcbpc_i_table() : table == 2 ? &mcbpc_p_table()
                        : table == 3 ? &cbpy_table() : &mvd_table();
    BitReader r(data, len);
    int i = 0;
    for (; i < n; i++) {
        VlcHit h{};
        int rc = t->decode(r, h);
        out[4 * i + 0] = rc != H263MI_OK ? rc : (h.valid ? 1 : 0);
        out[4 * i + 1] = h.v0; out[4 * i + 2] = h.v1; out[4 * i + 3] = h.v2;
        if (rc != H263MI_OK) { i++; break; }
    }
    return i;
}

// decode_block; out: [rc, has_intradc, intradc_code, n_tcoef, then (is_short, run, level) triples]
int pt_decode_block(const uint8_t *data, size_t len, int sorenson, int version, int intra, int tcoef_present,
                    int32_t *out, size_t *bits_used)
{
    BitReader r(data, len);
    ParsedBlock b;
    int rc = decode_block(r, sorenson != 0, version, intra != 0, tcoef_present != 0, b);
    out[0] = rc; out[1] = b.has_intradc; out[2] = b.intradc; out[3] = b.n_tcoef;
    for (int i = 0; i < b.n_tcoef; i++) {
        out[4 + 3 * i] = b.tcoef[i].is_short; out[5 + 3 * i] = b.tcoef[i].run; out[6 + 3 * i] = b.tcoef[i].level;
    }
    *bits_used = r.position();
    return rc;
}

// a small script interpreter over BitReader for the reader.rs tests:
// ops: 0 read_bits(n) 1 peek_bits(n) 2 skip_bits(n) 3 read_signed_bits(n) 4 recognize_start_code(in_error = n)
// result[i] = {rc, value}
void pt_reader_script(const uint8_t *data, size_t len, int n_ops, const int32_t *ops, int64_t *result)
{
    BitReader r(data, len);
    for (int i = 0; i < n_ops; i++) {
        const int op = ops[2 * i], arg = ops[2 * i + 1];
        uint32_t u = 0; int32_t s = 0; int sk = 0; int rc = 0; int64_t val = 0;
        switch (op) {
        case 0: rc = r.read_bits((uint32_t)arg, u); val = u; break;
        case 1: rc = r.peek_bits((uint32_t)arg, u); val = u; break;
        case 2: rc = r.skip_bits((uint32_t)arg); break;
        case 3: rc = r.read_signed_bits((uint32_t)arg, s); val = s; break;
        default: rc = r.recognize_start_code(arg != 0, sk); val = sk; break;
        }
        result[2 * i] = rc; result[2 * i + 1] = val;
    }
}

// whole picture -> records.  Returns rc; *n_mbs / *n_blocks the counts (capacity checked).
// ctx (in/out, may be null): the parser's view of the last decoded picture, as 16 opaque bytes + validity --
// updated when the parse succeeds, like the state commit of decode_next_picture.
static ParserContext g_ctx;
void pt_context_reset() { g_ctx = ParserContext(); }

int pt_parse_picture(const uint8_t *data, size_t len, uint32_t options, h263mi_picture_desc *desc, h263mi_mb_record *mbs,
                     size_t cap_mbs, int16_t *coeffs, size_t cap_blocks, size_t *n_mbs, size_t *n_blocks, size_t *bits,
                     int use_context)
{
    ParsedPicture p;
    int rc = parse_picture(data, len, options, use_context ? &g_ctx : nullptr, p);
    *n_mbs = p.mbs.size(); *n_blocks = p.coeffs.size() / 64; *bits = p.bits_consumed;
    if (rc != H263MI_OK) return rc;
    *desc = p.desc;
    if (p.mbs.size() > cap_mbs || p.coeffs.size() / 64 > cap_blocks) return H263MI_ERR_INVALID_ARGUMENT;
    if (!p.mbs.empty()) memcpy(mbs, p.mbs.data(), p.mbs.size() * sizeof(h263mi_mb_record));
    if (!p.coeffs.empty()) memcpy(coeffs, p.coeffs.data(), p.coeffs.size() * sizeof(int16_t));
    if (use_context) g_ctx = p.next;
    return rc;
}

// the product's form of the same parse (want_dense = false): block offsets and events (LEVEL << 16 | raster position), for the
// hand-derived known answers (tests/golden/macroblock_content_known_answers.json)
int pt_parse_picture_events(const uint8_t *data, size_t len, uint32_t options, uint32_t *first_event, size_t cap_blocks, uint32_t *events,
                            size_t cap_events, size_t *n_blocks, size_t *n_events)
{
    ParsedPicture p;
    p.want_dense = false;
    const int rc = parse_picture(data, len, options, nullptr, p);
    if (rc != H263MI_OK) return rc;
    *n_blocks = p.n_coded_blocks;
    *n_events = p.events.size();
    if (p.n_coded_blocks + 1 > cap_blocks || p.events.size() > cap_events) return H263MI_ERR_INVALID_ARGUMENT;
    memcpy(first_event, p.block_first_event.data(), (p.n_coded_blocks + 1) * sizeof(uint32_t));
    if (!p.events.empty()) memcpy(events, p.events.data(), p.events.size() * sizeof(uint32_t));
    return rc;
}

// parse_picture with the caller's limit on the picture size (ParsedPicture::size_fits): the return code, and how many 32-bit
// words the parser's own arrays were sized to -- a picture beyond the limit must not have sized any of them for itself
static uint32_t g_max_w = 0, g_max_h = 0;
static bool pt_size_fits(uint32_t w, uint32_t h) { return w <= g_max_w && h <= g_max_h; }
int pt_parse_picture_limited(const uint8_t *data, size_t len, uint32_t options, uint32_t max_w, uint32_t max_h, size_t *words)
{
    ParsedPicture p;
    p.want_dense = false;
    g_max_w = max_w; g_max_h = max_h;
    p.size_fits = max_w ? &pt_size_fits : nullptr;
    const int rc = parse_picture(data, len, options, nullptr, p);
    *words = p.scratch.capacity() + p.block_first_event.capacity() + p.mbs.capacity() * (sizeof(h263mi_mb_record) / 4);
    return rc;
}

// Both forms of the parser on the same bytes: the windowed fast paths against the field-by-field transcription
// (ParsedPicture::field_by_field).  Returns 0 when every output agrees -- return code, bits consumed, records, dense
// coefficients, events, block index, the context for the next picture -- else a positive number naming the first
// difference.  *rc_out = the (field-by-field) return code.
int pt_compare_parser_paths(const uint8_t *data, size_t len, uint32_t options, int *rc_out)
{
    ParsedPicture a, b;
    b.field_by_field = true;
    const int ra = parse_picture(data, len, options, nullptr, a), rb = parse_picture(data, len, options, nullptr, b);
    *rc_out = rb;
    if (ra != rb) return 1;
    if (ra != H263MI_OK) {                               // after an error the outputs mean nothing -- but the code must agree
        ParsedPicture c;
        c.want_dense = false;
        return parse_picture(data, len, options, nullptr, c) == rb ? 0 : 11;
    }
    if (a.bits_consumed != b.bits_consumed) return 2;
    if (memcmp(&a.desc, &b.desc, sizeof a.desc)) return 3;
    if (a.mbs.size() != b.mbs.size() || (a.mbs.size() && memcmp(a.mbs.data(), b.mbs.data(), a.mbs.size() * sizeof(h263mi_mb_record)))) return 4;
    if (a.coeffs != b.coeffs) return 5;
    if (a.events != b.events) return 6;
    if (a.block_first_event != b.block_first_event || a.n_coded_blocks != b.n_coded_blocks) return 7;
    if (a.next.have_last != b.next.have_last || a.next.last_format != b.next.last_format ||
        a.next.last_header_options != b.next.last_header_options) return 8;
    // ... and the form the product runs: events only (want_dense = false), where the blocks go through the stand-alone event
    // loops (block_events_short / _fast, inter_macroblock_events)
    ParsedPicture c;
    c.want_dense = false;
    const int rcc = parse_picture(data, len, options, nullptr, c);
    if (rcc != rb) return 11;
    if (c.bits_consumed != b.bits_consumed) return 12;
    if (c.mbs.size() != b.mbs.size() || (c.mbs.size() && memcmp(c.mbs.data(), b.mbs.data(), c.mbs.size() * sizeof(h263mi_mb_record)))) return 14;
    if (c.events != b.events) return 16;
    if (c.block_first_event != b.block_first_event || c.n_coded_blocks != b.n_coded_blocks) return 17;
    return 0;
}

// The records written into a caller's array (ParsedPicture::mbs_ext, what h263mi_batch_decode_next_pictures does with
// its pinned staging) against the records in the vector: 0 when return code, counts, records and everything else
// agree.  cap: capacity of the caller's array in records; a picture that does not fit must fall back to the vector.
// *used_ext says which form the parser chose.
int pt_compare_record_destinations(const uint8_t *data, size_t len, uint32_t options, size_t cap, int *rc_out, int *used_ext)
{
    ParsedPicture a, b;
    std::vector<h263mi_mb_record> ext(cap + 2);
    // guard records behind the array: the parser must not write past cap
    memset(ext.data(), 0xA5, ext.size() * sizeof(h263mi_mb_record));
    b.mbs_ext = ext.data();
    b.mbs_ext_cap = cap;
    const int ra = parse_picture(data, len, options, nullptr, a), rb = parse_picture(data, len, options, nullptr, b);
    *rc_out = ra;
    *used_ext = b.mbs_ext_used ? 1 : 0;
    if (ra != rb) return 1;
    for (size_t k = cap; k < ext.size(); k++) {
        const uint8_t *g = reinterpret_cast<const uint8_t *>(&ext[k]);
        for (size_t j = 0; j < sizeof(h263mi_mb_record); j++)
            if (g[j] != 0xA5) return 9;
    }
    if (ra != H263MI_OK) return 0;
    if (a.bits_consumed != b.bits_consumed) return 2;
    if (memcmp(&a.desc, &b.desc, sizeof a.desc)) return 3;
    if (a.mbs.size() != b.n_records() || (a.mbs.size() && memcmp(a.mbs.data(), b.records(), a.mbs.size() * sizeof(h263mi_mb_record)))) return 4;
    if (b.mbs_ext_used && !b.mbs.empty()) return 5;
    if (a.events != b.events) return 6;
    if (a.block_first_event != b.block_first_event || a.n_coded_blocks != b.n_coded_blocks) return 7;
    return 0;
}

// picture header only.  out: [rc, is_picture, picture_type, width, height, format_kind, options, has_plusptype,
// has_opptype, mv_range, quantizer, temporal_reference, n_extra, bits_used]
int pt_parse_header(const uint8_t *data, size_t len, uint32_t options, int use_context, int32_t *out)
{
    BitReader r(data, len);
    PictureHeader h;
    bool is_picture = false;
    int rc = decode_picture_header(r, options, use_context ? &g_ctx : nullptr, h, is_picture);
    out[0] = rc; out[1] = is_picture; out[2] = h.picture_type; out[3] = h.width; out[4] = h.height; out[5] = h.format.kind;
    out[6] = (int32_t)h.options; out[7] = h.has_plusptype; out[8] = h.has_opptype; out[9] = h.mv_range; out[10] = h.quantizer;
    out[11] = h.temporal_reference; out[12] = (int32_t)h.extra.size(); out[13] = (int32_t)r.position();
    return rc;
}

// read_umv: out = [rc, value, bits_used]
void pt_read_umv(const uint8_t *data, size_t len, int32_t *out)
{
    BitReader r(data, len);
    int v = 0;
    out[0] = r.read_umv(v); out[1] = v; out[2] = (int32_t)r.position();
}
}
