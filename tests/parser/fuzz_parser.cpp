// fuzz_parser.cpp -- deterministic mutation fuzzer of the host bitstream parser (TEST INFRASTRUCTURE).
//
// h263-rs_amd/host/bitstream.cpp replaces the reference's parser (h263/src/parser/*.rs, safe Rust) with C++ that eats
// untrusted bytes and writes records in place into a caller's (pinned) array.  This program is its sanitizer harness: built
// with -fsanitize=address,undefined (tests/parser/Makefile: fuzz_parser_asan), it takes the seed corpus of
// tools/gen_fuzz_corpus.py (tests/golden/parser_fuzz_corpus.bin), derives inputs from it -- bit flips, byte smashes,
// truncations, insertions, deletions, splices of two seeds, sweeps over the picture header's fields, runs of 0x00 / 0xff, for
// Sorenson and ITU-T flavoured streams alike -- and runs every input through
//
//   * the windowed fast paths and the field-by-field transcription of the reference's parser: same return code, same bits
//     consumed, same records / events / block index / next context (differential);
//   * the in-place record writer (ParsedPicture::mbs_ext) with the SMALLEST legal slot (exactly the picture's macroblocks)
//     and with one record less (must fall back to the vector), guard records behind the slot;
//   * a ParsedPicture that is REUSED from input to input, the way the product's per-stream parse buffers are, against a
//     fresh one;
//   * the input in a heap block of exactly its size (the windows read ahead of the cursor: a read past the end is an ASan
//     report);
//   * the SPARSE record form (ParsedPicture::sparse_records: records for the coded macroblocks only + an index word per
//     group of 8), into the parser's own array and in place into an exactly sized external one, against the dense records.
//
// and asserts on every input that still parses (rc == H263MI_OK):
//   * structure: macroblock count <= the picture's, known types, quantisers 1..31, kill within cbp, coeff_index = running
//     count of coded blocks, block_first_event ascending from 0 to the event count, <= 64 events per block at distinct
//     positions, no zero LEVEL, LEVELs within the escape width of the flavour (parser/block.rs:694-715), no DC event in an
//     intra block, vectors inside the range the picture's options allow (mvd_pred.rs:70-117);
//   * the EOF policy of state.rs:387-412: whatever ended the picture, nothing behind the consumed position mattered -- the
//     input cut at that position parses to the same result;
//   * standard-mode inputs never return a macroblock-header error (they resynchronise, state.rs:387-408), Sorenson inputs
//     never resynchronise;
// and for the layer functions (decode_block, decode_picture_header, read_umv, recognize_start_code) on raw mutated bytes:
//   * rollback (reader.rs:376-441, with_transaction): after an error the cursor is where it was.
//
// usage: fuzz_parser_asan corpus.bin [--inputs N] [--seed S] [--threads T] [--quiet]
// exit code 0 = no finding; a finding prints the input as hex and exits 1 (sanitizer reports abort on their own).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../h263-rs_amd/host/bitstream.hpp"

using namespace h263mi::bits;

namespace {

struct Seed { uint32_t options; std::vector<uint8_t> data; };

struct Rng {                                   // xorshift64*: the same inputs on every machine
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) { if (!s) s = 1; next(); next(); }
    uint64_t next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 0x2545F4914F6CDD1Dull; }
    uint32_t below(uint32_t n) { return n ? (uint32_t)(next() >> 33) % n : 0; }
    bool chance(uint32_t percent) { return below(100) < percent; }
};

bool load_corpus(const char *path, std::vector<Seed> &out)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    uint32_t n = 0;
    if (fread(&n, 4, 1, f) != 1 || n > 100000) { fclose(f); return false; }
    for (uint32_t i = 0; i < n; i++) {
        uint32_t opt = 0, len = 0;
        if (fread(&opt, 4, 1, f) != 1 || fread(&len, 4, 1, f) != 1 || len > (64u << 20)) { fclose(f); return false; }
        Seed s;
        s.options = opt;
        s.data.resize(len);
        if (len && fread(s.data.data(), 1, len, f) != len) { fclose(f); return false; }
        out.push_back(std::move(s));
    }
    fclose(f);
    return !out.empty();
}

// ---- mutations ----------------------------------------------------------------------------------------------------
void set_bits(std::vector<uint8_t> &d, size_t bitpos, uint32_t n, uint32_t value)
{
    for (uint32_t k = 0; k < n; k++) {
        const size_t b = bitpos + k;
        if (b / 8 >= d.size()) return;
        const uint8_t mask = (uint8_t)(0x80u >> (b & 7));
        if ((value >> (n - 1 - k)) & 1u) d[b / 8] |= mask;
        else d[b / 8] &= (uint8_t)~mask;
    }
}

// the fields of the two header flavours (picture.rs:611-661 Sorenson; 5.1 of the Recommendation / picture.rs:21-269): bit
// position and width; a sweep sets one of them to a random or an extreme value
struct Field { uint32_t pos, width; };
const Field kSorensonFields[] = {{0, 17}, {17, 5}, {22, 8}, {30, 3}, {33, 8}, {41, 8}, {33, 16}, {49, 16}, {33, 2}, {35, 1}, {36, 5},
                                 {41, 1}, {49, 2}, {51, 1}, {52, 5}, {57, 1}, {65, 2}, {67, 1}, {68, 5}, {73, 1}};
const Field kStandardFields[] = {{0, 17}, {17, 5}, {22, 8}, {30, 2}, {32, 3}, {35, 3}, {38, 5}, {38, 3}, {41, 18}, {59, 9}, {43, 5},
                                 {48, 1}, {49, 1}, {68, 1}, {69, 23}, {92, 5}, {97, 1}};

void mutate(Rng &rng, const std::vector<Seed> &corpus, std::vector<uint8_t> &d, uint32_t &options)
{
    const Seed &seed = corpus[rng.below((uint32_t)corpus.size())];
    d = seed.data;
    options = seed.options;
    if (rng.chance(3)) options ^= 1u;                                 // the wrong flavour for the bytes
    const uint32_t n_ops = 1 + rng.below(4);
    for (uint32_t op = 0; op < n_ops && !d.empty(); op++) {
        const uint32_t kind = rng.below(100);
        const size_t len = d.size();
        if (kind < 30) {                                              // bit flips, biased towards the front half the time
            const uint32_t flips = 1 + rng.below(4);
            for (uint32_t k = 0; k < flips; k++) {
                const size_t at = rng.chance(40) ? rng.below((uint32_t)std::min<size_t>(len, 24)) : rng.below((uint32_t)len);
                d[at] ^= (uint8_t)(1u << rng.below(8));
            }
        } else if (kind < 42) {                                       // byte smash
            d[rng.below((uint32_t)len)] = (uint8_t)rng.below(256);
        } else if (kind < 54) {                                       // truncation
            d.resize(rng.chance(30) ? rng.below(16) : rng.below((uint32_t)len + 1));
        } else if (kind < 62) {                                       // header field sweep
            const bool sor = (options & 1u) != 0;
            const Field &f = sor ? kSorensonFields[rng.below(sizeof kSorensonFields / sizeof(Field))]
                                 : kStandardFields[rng.below(sizeof kStandardFields / sizeof(Field))];
            const uint32_t pick = rng.below(4);
            const uint32_t all = f.width >= 32 ? ~0u : (1u << f.width) - 1u;
            set_bits(d, f.pos, f.width, pick == 0 ? 0u : pick == 1 ? all : pick == 2 ? 1u : (uint32_t)rng.next() & all);
        } else if (kind < 70) {                                       // splice: the tail of another seed
            const Seed &o = corpus[rng.below((uint32_t)corpus.size())];
            if (!o.data.empty()) {
                const size_t cut = rng.below((uint32_t)len + 1), from = rng.below((uint32_t)o.data.size());
                d.resize(cut);
                d.insert(d.end(), o.data.begin() + (long)from, o.data.end());
            }
        } else if (kind < 78) {                                       // insertion of random / constant bytes
            const size_t at = rng.below((uint32_t)len + 1), n = 1 + rng.below(8);
            const uint32_t fill = rng.below(3);
            std::vector<uint8_t> ins(n);
            for (uint8_t &b : ins) b = fill == 0 ? 0x00 : fill == 1 ? 0xff : (uint8_t)rng.below(256);
            d.insert(d.begin() + (long)at, ins.begin(), ins.end());
        } else if (kind < 86) {                                       // deletion
            const size_t at = rng.below((uint32_t)len), n = std::min<size_t>(1 + rng.below(8), len - at);
            d.erase(d.begin() + (long)at, d.begin() + (long)(at + n));
        } else if (kind < 93) {                                       // a run of 0x00 / 0xff (start codes, COD = 1 runs)
            const size_t at = rng.below((uint32_t)len), n = std::min<size_t>(1 + rng.below(40), len - at);
            memset(d.data() + at, rng.chance(50) ? 0x00 : 0xff, n);
        } else if (kind < 97) {                                       // bit shift of the tail: everything behind moves by 1..7 bits
            const size_t at = rng.below((uint32_t)len);
            const uint32_t sh = 1 + rng.below(7);
            for (size_t k = at; k + 1 < d.size(); k++) d[k] = (uint8_t)((d[k] << sh) | (d[k + 1] >> (8 - sh)));
        } else {                                                      // oversized: 16-bit custom dimensions in a Sorenson header
            if ((options & 1u) && d.size() > 9) {
                set_bits(d, 30, 3, 1);
                set_bits(d, 33, 16, rng.chance(50) ? 0xffffu : (uint32_t)rng.below(65536));
                set_bits(d, 49, 16, rng.chance(50) ? 0xffffu : (uint32_t)rng.below(65536));
            }
        }
    }
    if (d.size() > (1u << 20)) d.resize(1u << 20);
}

// ---- checks -------------------------------------------------------------------------------------------------------
struct Finding { std::string what; };

[[noreturn]] void report(const char *what, const std::vector<uint8_t> &d, uint32_t options, uint64_t index)
{
    fprintf(stderr, "FINDING at input %llu (options %u, %zu bytes): %s\n", (unsigned long long)index, options, d.size(), what);
    for (size_t i = 0; i < d.size() && i < 4096; i++) fprintf(stderr, "%02x", d[i]);
    fprintf(stderr, "\n");
    fflush(stderr);
    _Exit(1);
}

bool fits_small(uint32_t w, uint32_t h) { return (uint64_t)w * h <= 1920ull * 1088ull; }   // (bounds the fuzzer's own memory)

bool same_pictures(const ParsedPicture &a, const ParsedPicture &b, int &why)
{
    why = 0;
    if (a.bits_consumed != b.bits_consumed) { why = 2; return false; }
    if (memcmp(&a.desc, &b.desc, sizeof a.desc)) { why = 3; return false; }
    if (a.n_records() != b.n_records() ||
        (a.n_records() && memcmp(a.records(), b.records(), a.n_records() * sizeof(h263mi_mb_record)))) { why = 4; return false; }
    if (a.any_inter != b.any_inter || a.n_macroblocks != b.n_macroblocks) { why = 5; return false; }
    if (a.events != b.events) { why = 6; return false; }
    if (a.block_first_event != b.block_first_event || a.n_coded_blocks != b.n_coded_blocks) { why = 7; return false; }
    if (a.next.have_last != b.next.have_last || a.next.last_format != b.next.last_format ||
        a.next.last_header_options != b.next.last_header_options || a.next.last_header_format != b.next.last_header_format) { why = 8; return false; }
    return true;
}

const char *check_structure(const ParsedPicture &p, uint32_t options)
{
    const size_t total = (size_t)((p.desc.width + 15) / 16) * ((p.desc.height + 15) / 16);
    if (!p.desc.width || !p.desc.height) return "a picture without a size parsed";
    if (p.n_records() > total) return "more macroblocks than the picture holds";
    const h263mi_mb_record *r = p.records();
    const bool sorenson = (options & 1u) != 0;
    if (p.block_first_event.size() != p.n_coded_blocks + 1 || p.block_first_event[0] != 0) return "block index does not start at 0";
    if (p.block_first_event[p.n_coded_blocks] != p.events.size()) return "block index does not end at the event count";
    size_t blocks = 0;
    for (size_t i = 0; i < p.n_records(); i++) {
        const h263mi_mb_record &m = r[i];
        if (m.mb_type > 5) return "unknown macroblock type";
        if (m.quant < 1 || m.quant > 31) return "quantiser outside 1..31";
        if (m.cbp & ~0x3f) return "cbp beyond six blocks";
        if (m.kill & ~m.cbp) return "kill outside cbp";
        const bool intra = m.mb_type == 3 || m.mb_type == 4;
        const int n_coded = __builtin_popcount(m.cbp);
        if (n_coded && m.coeff_index != blocks) return "coeff_index is not the running count of coded blocks";
        if (intra) {
            for (int b = 0; b < 6; b++)
                if (m.intradc[b] == 0 || m.intradc[b] == 128) return "illegal INTRADC code in a record";   // types.rs:930-936
            for (int k = 0; k < 4; k++)
                if (m.mv[k][0] || m.mv[k][1]) return "an intra macroblock with a vector";
        }
        for (int b = 0; b < n_coded; b++) {
            const uint32_t first = p.block_first_event[blocks + (size_t)b], next = p.block_first_event[blocks + (size_t)b + 1];
            if (first > next || next - first > 64) return "block index not ascending / more than 64 events";
            uint64_t seen = 0;
            for (uint32_t e = first; e < next; e++) {
                const uint32_t ev = p.events[e];
                if (ev & 0xffc0u) return "event with bits between position and LEVEL";
                const uint64_t bit = 1ull << (ev & 63u);
                if (seen & bit) return "two events at one position";
                seen |= bit;
                const int level = (int16_t)(ev >> 16);
                if (level == 0) return "zero LEVEL";
                if (sorenson ? (level < -1024 || level > 1023) : (level < -128 || level > 127)) return "LEVEL beyond the escape width";
                if (intra && (ev & 63u) == 0) return "a DC event in an intra block";
            }
        }
        blocks += (size_t)n_coded;
        if (!(p.desc.picture_type == H263MI_PICTURE_I) && !intra) {
            // Sorenson and plain H.263: [-32, 32) half-pels after the wrap (mvd_pred.rs:70-117); Annex D: up to +-(2 x 4096)
            const int lim = sorenson ? 32 : 8192 + 64;
            for (int k = 0; k < 4; k++)
                if (m.mv[k][0] < -lim || m.mv[k][0] >= lim || m.mv[k][1] < -lim || m.mv[k][1] >= lim) return "vector outside its range";
        }
        if (p.desc.picture_type == H263MI_PICTURE_I && !intra) return "an inter macroblock in the records of an I picture";
    }
    if (blocks != p.n_coded_blocks) return "coded blocks of the records != n_coded_blocks";
    return nullptr;
}

// ParsedPicture::sparse_records against the dense records of the same input: a macroblock with a record has that record; one
// without is "not coded" in the dense array (INTER, nothing coded, zero vectors: state.rs:207-216; its quantiser -- never
// used -- is the one field the sparse form does not carry); `first` of every group = the number of records in front of it
const char *check_sparse(const ParsedPicture &dense, const ParsedPicture &sp)
{
    const size_t mbw = (dense.desc.width + 15) / 16, mbh = (dense.desc.height + 15) / 16, gpl = (mbw + 7) / 8;
    if (sp.group_index.size() != gpl * mbh) return "group index of the wrong size";
    if (sp.n_macroblocks != dense.n_records()) return "sparse: macroblock count differs";
    if (sp.any_inter != dense.any_inter) return "sparse: any_inter differs";
    const h263mi_mb_record *d = dense.records();
    size_t k = 0;
    for (size_t line = 0; line < mbh; line++)
        for (size_t g = 0; g < gpl; g++) {
            const uint32_t gw = sp.group_index[line * gpl + g];
            if ((gw & 0xffu) && (gw >> 8) != k) return "sparse: `first` of a group is not the number of records in front of it";
            for (size_t b = 0; b < 8; b++) {
                const size_t col = g * 8 + b, mb = line * mbw + col;
                const bool has = (gw >> b) & 1u;
                if (col >= mbw || mb >= dense.n_records()) {
                    if (has) return "sparse: a record for a macroblock the bitstream does not hold";
                    continue;
                }
                if (has) {
                    if (k >= sp.n_records()) return "sparse: more mask bits than records";
                    if (memcmp(&sp.records()[k], &d[mb], sizeof(h263mi_mb_record))) return "sparse: a record differs from the dense one";
                    k++;
                } else {
                    h263mi_mb_record skip{};
                    skip.mb_type = H263MI_MB_INTER;
                    skip.quant = d[mb].quant;
                    if (memcmp(&skip, &d[mb], sizeof skip)) return "sparse: a macroblock without a record is coded in the dense form";
                }
            }
        }
    if (k != sp.n_records()) return "sparse: records nobody points at";
    if (sp.events != dense.events || sp.block_first_event != dense.block_first_event) return "sparse: events differ";
    return nullptr;
}

struct Stats {
    std::atomic<uint64_t> inputs{0}, parsed_ok{0}, layer_checks{0};
    std::atomic<uint64_t> rc_hist[32];
    Stats() { for (auto &h : rc_hist) h = 0; }
};

struct Worker {
    const std::vector<Seed> &corpus;
    Stats &st;
    ParsedPicture reused_fast, reused_field, reused_sparse;      // live across inputs, like the product's per-stream buffers
    Worker(const std::vector<Seed> &c, Stats &s) : corpus(c), st(s) {}

    void one_input(uint64_t index, uint64_t base_seed)
    {
        Rng rng(base_seed * 0x100000001B3ull + index);
        std::vector<uint8_t> d;
        uint32_t options = 1;
        mutate(rng, corpus, d, options);
        // exactly-sized heap block: reading one byte past the data is a heap-buffer-overflow for ASan
        uint8_t *heap = (uint8_t *)malloc(d.size() ? d.size() : 1);
        if (d.size()) memcpy(heap, d.data(), d.size());
        const uint8_t *data = heap;
        const size_t len = d.size();

        ParsedPicture fresh_fast, fresh_field;
        for (ParsedPicture *p : {&fresh_fast, &fresh_field, &reused_fast, &reused_field}) {
            p->want_dense = false;
            p->size_fits = &fits_small;
            p->mbs_ext = nullptr;
            p->mbs_ext_cap = 0;
        }
        fresh_field.field_by_field = true;
        reused_field.field_by_field = true;
        const int ra = parse_picture(data, len, options, nullptr, fresh_fast);
        const int rb = parse_picture(data, len, options, nullptr, fresh_field);
        st.rc_hist[ra == 0 ? 0 : std::min(31, -ra)]++;
        if (ra != rb) report("fast and field-by-field parsers return different codes", d, options, index);
        const int rc_reused_fast = parse_picture(data, len, options, nullptr, reused_fast);
        const int rc_reused_field = parse_picture(data, len, options, nullptr, reused_field);
        if (rc_reused_fast != ra || rc_reused_field != ra) report("a reused ParsedPicture returns another code than a fresh one", d, options, index);
        // reference policy (state.rs:387-408): macroblock-header errors resynchronise in standard mode and only there
        if (!(options & 1u) && (ra == H263MI_ERR_INVALID_MACROBLOCK_HEADER || ra == H263MI_ERR_INVALID_MACROBLOCK_CODED_BITS))
            report("a standard-mode parse returned a macroblock-header error instead of resynchronising", d, options, index);
        {
            // the caller's WORD arrays (ParsedPicture::events_ext: what the batch entry's staging slot is): exactly the bounds the
            // parser asks for, no slack -- ASan guards the ends -- for EVERY input, parsed or not (what is written before an
            // error is found must stay inside too); then one word less, which must send everything back into the vectors
            const size_t mbw = (fresh_fast.desc.width + 15) / 16, mbh = (fresh_fast.desc.height + 15) / 16, total = mbw * mbh;
            const size_t groups = ((mbw + 7) / 8) * mbh;
            for (int shrink = 0; shrink < 2 && total; shrink++) {
                const size_t cap_e = h263mi::bits::event_words_bound(len, total) - (size_t)shrink;
                const size_t cap_f = h263mi::bits::block_offset_words_bound(total), cap_g = groups;
                uint32_t *ev = (uint32_t *)malloc(cap_e * 4), *fe = (uint32_t *)malloc(cap_f * 4), *gi = (uint32_t *)malloc((cap_g ? cap_g : 1) * 4);
                h263mi_mb_record *slot = (h263mi_mb_record *)malloc(total * sizeof(h263mi_mb_record));
                ParsedPicture w;
                w.want_dense = false;
                w.size_fits = &fits_small;
                w.sparse_records = true;
                w.mbs_ext = slot; w.mbs_ext_cap = total;
                w.events_ext = ev; w.events_ext_cap = cap_e;
                w.first_event_ext = fe; w.first_event_ext_cap = cap_f;
                w.group_index_ext = gi; w.group_index_ext_cap = cap_g;
                w.event_base = 0x01000000u + (uint32_t)(index & 0xffff);
                if (parse_picture(data, len, options, nullptr, w) != ra) report("the caller's word arrays change the return code", d, options, index);
                if (w.words_ext_used != (shrink == 0)) report("the caller's word arrays: wrong destination chosen", d, options, index);
                if (ra == H263MI_OK) {
                    ParsedPicture sp;
                    sp.want_dense = false;
                    sp.size_fits = &fits_small;
                    sp.sparse_records = true;
                    (void)parse_picture(data, len, options, nullptr, sp);
                    const uint32_t base = shrink ? 0u : w.event_base;
                    bool same = w.n_coded_blocks == sp.n_coded_blocks && w.n_event_words() == sp.events.size() && w.n_records() == sp.n_records() &&
                                w.bits_consumed == sp.bits_consumed && w.any_inter == sp.any_inter && w.n_macroblocks == sp.n_macroblocks;
                    for (size_t k = 0; same && k <= sp.n_coded_blocks; k++) same = w.first_event_words()[k] == sp.block_first_event[k] + base;
                    same = same && (sp.events.empty() || !memcmp(w.event_words(), sp.events.data(), sp.events.size() * 4));
                    same = same && (!groups || !memcmp(w.group_index_words(), sp.group_index.data(), groups * 4));
                    same = same && (!sp.n_records() || !memcmp(w.records(), sp.records(), sp.n_records() * sizeof(h263mi_mb_record)));
                    if (!same) report("the caller's word arrays hold something else than the vectors", d, options, index);
                    if (!shrink && (!w.events.empty() || !w.block_first_event.empty() || !w.group_index.empty())) report("words in both destinations", d, options, index);
                }
                free(ev); free(fe); free(gi); free(slot);
            }
        }
        if (ra == H263MI_OK) {
            st.parsed_ok++;
            int why = 0;
            if (!same_pictures(fresh_fast, fresh_field, why)) report("fast and field-by-field parsers differ in their outputs", d, options, index);
            if (!same_pictures(fresh_fast, reused_fast, why) || !same_pictures(fresh_fast, reused_field, why))
                report("a reused ParsedPicture gives other outputs than a fresh one", d, options, index);
            if (const char *bad = check_structure(fresh_fast, options)) report(bad, d, options, index);
            {
                // the sparse record form (what the batch entry sends over the link) against the dense one
                ParsedPicture sp;
                sp.want_dense = false;
                sp.size_fits = &fits_small;
                sp.sparse_records = true;
                if (parse_picture(data, len, options, nullptr, sp) != ra) report("sparse records: another return code", d, options, index);
                if (const char *bad = check_sparse(fresh_fast, sp)) report(bad, d, options, index);
                // ... and with a reused object, as the product's per-stream buffers are
                // ... into the caller's array (the smallest legal one: the batch entry's staging slot), with a reused object
                const size_t total_mbs = (size_t)((fresh_fast.desc.width + 15) / 16) * ((fresh_fast.desc.height + 15) / 16);
                h263mi_mb_record *slot = (h263mi_mb_record *)malloc((total_mbs ? total_mbs : 1) * sizeof(h263mi_mb_record));
                reused_sparse.want_dense = false;
                reused_sparse.size_fits = &fits_small;
                reused_sparse.sparse_records = true;
                reused_sparse.mbs_ext = slot;
                reused_sparse.mbs_ext_cap = total_mbs;
                if (parse_picture(data, len, options, nullptr, reused_sparse) != ra) report("sparse records (reused): another return code", d, options, index);
                if (!reused_sparse.mbs_ext_used) report("sparse records: the caller's array was not used", d, options, index);
                if (const char *bad = check_sparse(fresh_fast, reused_sparse)) report(bad, d, options, index);
                reused_sparse.mbs_ext = nullptr;
                free(slot);
            }
            if (fresh_fast.bits_consumed > len * 8) report("more bits consumed than there are", d, options, index);
            // in-place record writer: the smallest legal slot, and one record less
            const size_t total = (size_t)((fresh_fast.desc.width + 15) / 16) * ((fresh_fast.desc.height + 15) / 16);
            for (int shrink = 0; shrink < 2; shrink++) {
                const size_t cap = total - (size_t)shrink;
                if (shrink && !total) continue;
                h263mi_mb_record *slot = (h263mi_mb_record *)malloc((cap ? cap : 1) * sizeof(h263mi_mb_record));   // no slack: ASan guards the end
                ParsedPicture ext;
                ext.want_dense = false;
                ext.size_fits = &fits_small;
                ext.mbs_ext = slot;
                ext.mbs_ext_cap = cap;
                const int re = parse_picture(data, len, options, nullptr, ext);
                if (re != ra) report("the in-place record writer changes the return code", d, options, index);
                if (ext.mbs_ext_used != (shrink == 0)) report("the in-place record writer chose the wrong destination", d, options, index);
                if (!same_pictures(fresh_fast, ext, why)) report("the in-place record writer changes the outputs", d, options, index);
                if (ext.mbs_ext_used && !ext.mbs.empty()) report("records in both destinations", d, options, index);
                free(slot);
            }
            // EOF policy: nothing behind the consumed position mattered
            const size_t cut = (fresh_fast.bits_consumed + 7) / 8;
            if (cut < len) {
                uint8_t *h2 = (uint8_t *)malloc(cut ? cut : 1);
                memcpy(h2, data, cut);
                ParsedPicture pre;
                pre.want_dense = false;
                pre.size_fits = &fits_small;
                const int rp = parse_picture(h2, cut, options, nullptr, pre);
                if (rp != H263MI_OK) report("the input cut at the consumed position no longer parses", d, options, index);
                // (the cut may end the picture a few stuffing / COD bits earlier or later inside the last byte: the records up
                // to the original's count must agree, and the cut parse can only have consumed up to its own end)
                if (pre.bits_consumed > cut * 8) report("cut input: more bits consumed than there are", d, options, index);
                const size_t n = std::min(pre.n_records(), fresh_fast.n_records());
                if (n && memcmp(pre.records(), fresh_fast.records(), n * sizeof(h263mi_mb_record)))
                    report("the input cut at the consumed position gives other records", d, options, index);
                if (pre.n_records() < fresh_fast.n_records()) report("the input cut at the consumed position lost macroblocks", d, options, index);
                free(h2);
            }
            // the next picture of the stream sees this one's context (standard headers take their format from it)
            ParsedPicture follow;
            follow.want_dense = false;
            follow.size_fits = &fits_small;
            ParsedPicture follow_field;
            follow_field.want_dense = false;
            follow_field.size_fits = &fits_small;
            follow_field.field_by_field = true;
            const Seed &nxt = corpus[rng.below((uint32_t)corpus.size())];
            const int f1 = parse_picture(nxt.data.data(), nxt.data.size(), nxt.options, &fresh_fast.next, follow);
            const int f2 = parse_picture(nxt.data.data(), nxt.data.size(), nxt.options, &fresh_fast.next, follow_field);
            if (f1 != f2 || (f1 == H263MI_OK && !same_pictures(follow, follow_field, why)))
                report("with a context: fast and field-by-field parsers differ", nxt.data, nxt.options, index);
        }
        // ---- the layer functions on the raw bytes: an error leaves the cursor where it was (with_transaction)
        if (len) {
            for (int k = 0; k < 4; k++) {
                BitReader r(data, len);
                const size_t start = rng.below((uint32_t)(len * 8));
                r.rollback(start);
                ParsedBlock blk;
                const bool sor = rng.chance(60), intra = rng.chance(50);
                const int rc = decode_block(r, sor, (int)rng.below(2), intra, rng.chance(80), blk);
                if (rc != H263MI_OK && r.position() != start) report("decode_block: the cursor moved on an error", d, options, index);
                if (rc == H263MI_OK && r.position() > len * 8) report("decode_block: cursor past the end", d, options, index);
                if (rc == H263MI_OK && (blk.n_tcoef < 0 || blk.n_tcoef > 72)) report("decode_block: event count out of range", d, options, index);
                st.layer_checks++;
            }
            {
                BitReader r(data, len);
                const size_t start = rng.chance(70) ? 0 : rng.below((uint32_t)(len * 8));
                r.rollback(start);
                PictureHeader h;
                bool is_picture = false;
                const int rc = decode_picture_header(r, options, rng.chance(50) ? &reused_fast.next : nullptr, h, is_picture);
                if ((rc != H263MI_OK || !is_picture) && r.position() != start) report("decode_picture_header: the cursor moved without a picture header", d, options, index);
                BitReader u(data, len);
                u.rollback(start);
                int v = 0;
                const int urc = u.read_umv(v);
                if (urc == H263MI_OK && (v <= -8192 || v >= 8192)) report("read_umv: value out of range", d, options, index);
                if (u.position() > len * 8) report("read_umv: cursor past the end", d, options, index);
                int skipped = 0;
                BitReader s2(data, len);
                s2.rollback(start);
                const int src = s2.recognize_start_code(rng.chance(50), skipped);
                if (s2.position() != start) report("recognize_start_code: the cursor moved", d, options, index);
                if (src == H263MI_OK && skipped >= 0 && start + (size_t)skipped + 17 > len * 8) report("recognize_start_code: start code beyond the data", d, options, index);
                st.layer_checks++;
            }
        }
        free(heap);
        st.inputs++;
    }
};

}  // namespace

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s corpus.bin [--inputs N] [--seed S] [--threads T] [--quiet]\n", argv[0]);
        return 2;
    }
    uint64_t inputs = 10000, seed = 1;
    unsigned threads = 1;
    bool quiet = false;
    for (int i = 2; i < argc; i++) {
        if (!strcmp(argv[i], "--inputs") && i + 1 < argc) inputs = strtoull(argv[++i], nullptr, 10);
        else if (!strcmp(argv[i], "--seed") && i + 1 < argc) seed = strtoull(argv[++i], nullptr, 10);
        else if (!strcmp(argv[i], "--threads") && i + 1 < argc) threads = (unsigned)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--quiet")) quiet = true;
    }
    std::vector<Seed> corpus;
    if (!load_corpus(argv[1], corpus)) { fprintf(stderr, "cannot read the corpus %s\n", argv[1]); return 2; }
    // the un-mutated seeds must parse: a corpus that does not is a broken test
    for (size_t i = 0; i < corpus.size(); i++) {
        ParsedPicture p;
        p.want_dense = false;
        const int rc = parse_picture(corpus[i].data.data(), corpus[i].data.size(), corpus[i].options, nullptr, p);
        if (rc != H263MI_OK) { fprintf(stderr, "seed %zu does not parse (rc %d)\n", i, rc); return 2; }
        if (const char *bad = check_structure(p, corpus[i].options)) { fprintf(stderr, "seed %zu: %s\n", i, bad); return 2; }
    }
    Stats st;
    std::atomic<uint64_t> next{0};
    if (threads < 1) threads = 1;
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; t++)
        pool.emplace_back([&]() {
            Worker w(corpus, st);
            for (;;) {
                const uint64_t i = next.fetch_add(1);
                if (i >= inputs) break;
                w.one_input(i, seed);
                if (!quiet && i && i % 1000000 == 0) { fprintf(stderr, "  %llu inputs\n", (unsigned long long)i); fflush(stderr); }
            }
        });
    for (auto &t : pool) t.join();
    printf("fuzz_parser: %llu inputs from %zu seeds (seed %llu): %llu still parsed, %llu layer checks, 0 findings\n",
           (unsigned long long)st.inputs.load(), corpus.size(), (unsigned long long)seed, (unsigned long long)st.parsed_ok.load(),
           (unsigned long long)st.layer_checks.load());
    printf("return codes:");
    for (int k = 0; k < 32; k++)
        if (st.rc_hist[k].load()) printf(" %d:%llu", -k, (unsigned long long)st.rc_hist[k].load());
    printf("\n");
    return 0;
}
