// state.cpp -- the H263State mirror (h263/src/decoder/state.rs:16-490): a batch of one stream fed with host records or
// coded pictures, its DecodedPicture accessors (picture.rs:61-142) and the rendering a consumer composes behind them.
#include "batch.h"

#include <algorithm>
#include <cstring>
#include <memory>
#include <new>

using namespace h263mi;

struct h263mi_state {
    uint32_t options = 0;
    h263mi_backend_cfg cfg{};
    h263mi_batch *b = nullptr;
    h263mi_picture_desc last_desc{};
    bool has_last = false;
    bits::ParserContext parser_ctx;   // header + format of the last picture decoded from a bitstream (state.rs:143-167)
    bits::ParsedPicture parsed;       // parse results of h263mi_decode_next_picture: kept, so that its buffers are reused
    // staging: two slots (pinned host + device) used alternately, so that filling slot i+1 on the host
    // overlaps the H2D copy and the kernel of slot i
    struct Staging {
        MbRecord *h_mbs = nullptr;  int16_t *h_coeffs = nullptr;     // pinned
        MbRecord *d_mbs = nullptr;  int16_t *d_coeffs = nullptr;
        uint32_t *h_events = nullptr, *d_events = nullptr;           // sparse transport: block offsets, then events
        size_t cap_mbs = 0, cap_blocks = 0, cap_events = 0;
        hipEvent_t done = nullptr;  // recorded after the kernel that reads the slot
    } stg[2];
    unsigned next_slot = 0;
    uint8_t *d_rgba = nullptr;  size_t cap_rgba = 0;

    void free_staging()
    {
        for (Staging &g : stg) {
            if (g.h_mbs) (void)hipHostFree(g.h_mbs);
            if (g.h_coeffs) (void)hipHostFree(g.h_coeffs);
            if (g.d_mbs) (void)hipFree(g.d_mbs);
            if (g.d_coeffs) (void)hipFree(g.d_coeffs);
            if (g.h_events) (void)hipHostFree(g.h_events);
            if (g.d_events) (void)hipFree(g.d_events);
            if (g.done) (void)hipEventDestroy(g.done);
            g = Staging();
        }
        if (d_rgba) (void)hipFree(d_rgba);
        d_rgba = nullptr;
        cap_rgba = 0;
    }
    ~h263mi_state()
    {
        DeviceGuard g(cfg.device_id);
        if (b) (void)hipStreamSynchronize(b->stream);
        free_staging();
        delete b;
    }
};

// n_event_words > 0: sparse transport -- no dense blocks anywhere, the reconstruction waves read the events
static int state_ensure_staging(h263mi_state::Staging &g, size_t n_mbs, size_t n_blocks, size_t n_event_words, const HostPlacement &where)
{
    PlacementScope near_device(where);           // pinned memory on the NUMA node of the state's GPU (worker_pool.h)
    if (n_mbs > g.cap_mbs) {
        if (g.h_mbs) (void)hipHostFree(g.h_mbs);
        if (g.d_mbs) (void)hipFree(g.d_mbs);
        g.h_mbs = nullptr; g.d_mbs = nullptr; g.cap_mbs = 0;
        HIP_TRY(hipHostMalloc((void **)&g.h_mbs, n_mbs * sizeof(MbRecord), hipHostMallocDefault));
        HIP_TRY(hipMalloc((void **)&g.d_mbs, n_mbs * sizeof(MbRecord)));
        g.cap_mbs = n_mbs;
    }
    if (!n_event_words && (n_blocks > g.cap_blocks || !g.h_coeffs)) {
        if (g.h_coeffs) (void)hipHostFree(g.h_coeffs);
        if (g.d_coeffs) (void)hipFree(g.d_coeffs);
        g.h_coeffs = nullptr; g.d_coeffs = nullptr; g.cap_blocks = 0;
        size_t cap = std::max(n_blocks, g.cap_blocks);
        cap = cap + cap / 2 + 64;
        if (!n_event_words) HIP_TRY(hipHostMalloc((void **)&g.h_coeffs, cap * 128, hipHostMallocDefault));
        HIP_TRY(hipMalloc((void **)&g.d_coeffs, cap * 128));
        g.cap_blocks = cap;
    }
    if (n_event_words > g.cap_events) {
        if (g.h_events) (void)hipHostFree(g.h_events);
        if (g.d_events) (void)hipFree(g.d_events);
        g.h_events = nullptr; g.d_events = nullptr; g.cap_events = 0;
        const size_t cap = n_event_words + n_event_words / 2 + 256;
        HIP_TRY(hipHostMalloc((void **)&g.h_events, cap * sizeof(uint32_t), hipHostMallocDefault));
        HIP_TRY(hipMalloc((void **)&g.d_events, cap * sizeof(uint32_t)));
        g.cap_events = cap;
    }
    if (!g.done) HIP_TRY(hipEventCreateWithFlags(&g.done, hipEventDisableTiming));
    return H263MI_OK;
}

// state.rs:421-483 from host records; the coefficients come either as dense blocks (`coeffs`) or as events
// (`first_event` + `events`, expanded on the device)
static int submit_records(h263mi_state *s, const h263mi_picture_desc *desc, const h263mi_mb_record *mbs, size_t n_mbs,
                          const int16_t *coeffs, size_t n_coeff_blocks, const uint32_t *first_event, const uint32_t *events,
                          size_t n_events, bool from_parser = false);

extern "C" {

int h263mi_state_new(uint32_t decoder_options, const h263mi_backend_cfg *cfg, h263mi_state **out)
{
    if (!out) return H263MI_ERR_INVALID_ARGUMENT;
    const int dev = cfg ? cfg->device_id : 0;
    RC_TRY(check_device(dev));
    h263mi_state *s = new (std::nothrow) h263mi_state();
    if (!s) return H263MI_ERR_OUT_OF_MEMORY;
    s->options = decoder_options;
    if (cfg) s->cfg = *cfg;
    s->cfg.device_id = dev;
    s->cfg.flags &= ~(H263MI_CFG_OVERLAP_POST | H263MI_CFG_PIPELINE_POST);   // batches only: h263mi_render_rgba copies back on the main stream
    *out = s;
    return H263MI_OK;
}

void h263mi_state_free(h263mi_state *s) { delete s; }

int h263mi_state_is_sorenson(const h263mi_state *s) { return s && (s->options & H263MI_SORENSON_SPARK_BITSTREAM) ? 1 : 0; }

int h263mi_state_reset(h263mi_state *s)
{
    if (!s) return H263MI_ERR_INVALID_ARGUMENT;
    s->has_last = false;
    s->parser_ctx = bits::ParserContext();
    if (s->b) {
        DeviceGuard g(s->b->device);
        return s->b->forget_pictures();
    }
    return H263MI_OK;
}

int h263mi_state_cleanup_buffers(h263mi_state *s)
{
    // The store never holds more than the last and the reference picture (two frame sets),
    // which is exactly what cleanup_buffers (state.rs:81-98) leaves behind.
    return s ? H263MI_OK : H263MI_ERR_INVALID_ARGUMENT;
}

}  // extern "C"

static int submit_records(h263mi_state *s, const h263mi_picture_desc *desc, const h263mi_mb_record *mbs, size_t n_mbs,
                          const int16_t *coeffs, size_t n_coeff_blocks, const uint32_t *first_event, const uint32_t *events,
                          size_t n_events, bool from_parser)
{
    // from_parser: the arrays are what bits::parse_picture just wrote -- valid by construction (quantisers, types, block
    // indices, one event per position), so the per-record and per-event checks a caller's arrays get are skipped
    const bool sparse = first_event != nullptr;
    if (!s || !desc || (!mbs && n_mbs) || (!sparse && !coeffs && n_coeff_blocks) || (sparse && !events && n_events))
        return H263MI_ERR_INVALID_ARGUMENT;
    if (sparse && n_coeff_blocks && !from_parser) {
        // offsets must be monotone and end at n_events: checked here, the kernel trusts them
        if (first_event[0] != 0 || first_event[n_coeff_blocks] != n_events) return H263MI_ERR_INVALID_ARGUMENT;
        // ... and a block's events name every position at most once (the device places them in no particular order)
        for (size_t i = 0; i < n_coeff_blocks; i++) {
            if (first_event[i] > first_event[i + 1] || first_event[i + 1] > n_events || first_event[i + 1] - first_event[i] > 64)
                return H263MI_ERR_INVALID_ARGUMENT;
            uint64_t seen = 0;
            for (uint32_t e = first_event[i]; e < first_event[i + 1]; e++) {
                const uint64_t bit = 1ull << (events[e] & 63u);
                if (seen & bit) return H263MI_ERR_INVALID_ARGUMENT;
                seen |= bit;
            }
        }
    }
    if (!desc->width || !desc->height || !layout_fits(desc->width, desc->height)) return H263MI_ERR_PICTURE_FORMAT_INVALID;
    if (desc->picture_type > H263MI_PICTURE_RESERVED) return H263MI_ERR_INVALID_ARGUMENT;
    const FrameLayout L = make_layout(desc->width, desc->height);
    const size_t total = (size_t)L.mbw * L.mbh;
    if (n_mbs > total) return H263MI_ERR_INVALID_ARGUMENT;

    // ---- everything that can fail is checked before any state changes (state.rs:142, 464-487)
    bool any_inter = n_mbs < total;   // missing macroblocks are padded as Inter (state.rs:421-427)
    for (size_t i = 0; i < n_mbs; i++) {
        const h263mi_mb_record &m = mbs[i];
        if (mb_is_inter(m.mb_type)) any_inter = true;
        if (from_parser) {
            if (any_inter) break;                // (nothing else to learn from a parser's records)
            continue;
        }
        if (m.mb_type > H263MI_MB_INTER4V_Q || m.quant < 1 || m.quant > 31 || (m.cbp & 0xC0) || (m.kill & 0xC0))
            return H263MI_ERR_INVALID_ARGUMENT;
        // (a record without coded blocks does not use its coeff_index)
        if (m.cbp && (size_t)m.coeff_index + (size_t)__builtin_popcount(m.cbp) > n_coeff_blocks) return H263MI_ERR_INVALID_ARGUMENT;
    }
    const bool same_size = s->b && s->b->L.width == L.width && s->b->L.height == L.height;
    const bool has_ref = s->b && s->b->ss[0].has_ref && s->b->ss[0].cur >= 0;
    if (any_inter && !has_ref) return H263MI_ERR_UNCODED_IFRAME_BLOCKS;              // gather.rs:149
    // A size change under inter prediction indexes the new planes with the reference's strides in
    // the reference (gather.rs:150,183: out-of-bounds panic or garbage); reported as an error here.
    if (any_inter && !same_size) return H263MI_ERR_PICTURE_FORMAT_INVALID;

    DeviceGuard g(s->cfg.device_id);
    if (!g.ok) return H263MI_ERR_NO_DEVICE;
    // A picture of another size gets a frame store of its own.  The old one -- the state's last picture -- is given up only
    // once the new picture's launch has been queued: everything from here to there can fail (allocations, copies, the
    // launch), and the reference mutates its state after the last fallible call only (state.rs:464-487).
    std::unique_ptr<h263mi_batch> fresh;
    if (!same_size) {
        h263mi_batch *nb = nullptr;
        RC_TRY(batch_create(1, L.width, L.height, &s->cfg, &nb));
        fresh.reset(nb);
    }
    h263mi_batch *b = same_size ? s->b : fresh.get();
    h263mi_state::Staging &g2 = s->stg[s->next_slot & 1];
    const size_t event_words = sparse ? n_coeff_blocks + 1 + n_events : 0;
    RC_TRY(state_ensure_staging(g2, total, n_coeff_blocks ? n_coeff_blocks : 1, event_words, b->placement));
    HIP_TRY(hipEventSynchronize(g2.done));       // the kernel that read this slot two pictures ago is done

    // (h263mi_decode_next_picture has its parser write the records straight into this slot)
    if (n_mbs && mbs != g2.h_mbs) memcpy(g2.h_mbs, mbs, n_mbs * sizeof(MbRecord));
    for (size_t i = n_mbs; i < total; i++) {     // state.rs:421-427: Inter, mv (0,0), nothing coded
        MbRecord pad;
        memset(&pad, 0, sizeof pad);
        pad.mb_type = H263MI_MB_INTER;
        pad.quant = 1;
        g2.h_mbs[i] = pad;
    }
    HIP_TRY(hipMemcpyAsync(g2.d_mbs, g2.h_mbs, total * sizeof(MbRecord), hipMemcpyHostToDevice, b->stream));
    h263mi_coeff_source src;
    if (sparse && n_coeff_blocks) {
        memcpy(g2.h_events, first_event, (n_coeff_blocks + 1) * sizeof(uint32_t));
        if (n_events) memcpy(g2.h_events + n_coeff_blocks + 1, events, n_events * sizeof(uint32_t));
        HIP_TRY(hipMemcpyAsync(g2.d_events, g2.h_events, event_words * sizeof(uint32_t), hipMemcpyHostToDevice, b->stream));
        src.first_event = g2.d_events;           // (read by the reconstruction waves themselves)
        src.events = g2.d_events + n_coeff_blocks + 1;
        src.n_events = (uint32_t)n_events;
    } else if (n_coeff_blocks) {
        memcpy(g2.h_coeffs, coeffs, n_coeff_blocks * 128);
        HIP_TRY(hipMemcpyAsync(g2.d_coeffs, g2.h_coeffs, n_coeff_blocks * 128, hipMemcpyHostToDevice, b->stream));
    }

    src.coeffs = g2.d_coeffs;
    src.pool_blocks = n_coeff_blocks;
    src.checked = true;
    RC_TRY(b->submit(desc->picture_type, g2.d_mbs, src));
    // ---- the launch is queued: from here on nothing fails any more, the state changes (state.rs:464-483)
    if (fresh) {
        delete s->b;
        s->b = fresh.release();
    }
    if (hipEventRecord(g2.done, b->stream) != hipSuccess) (void)hipStreamSynchronize(b->stream);   // (the slot is reused two pictures on)
    s->next_slot++;
    s->last_desc = *desc;
    s->has_last = true;
    return H263MI_OK;
}

extern "C" {

int h263mi_submit_picture(h263mi_state *s, const h263mi_picture_desc *desc, const h263mi_mb_record *mbs, size_t n_mbs,
                          const int16_t *coeffs, size_t n_coeff_blocks)
{
    return submit_records(s, desc, mbs, n_mbs, coeffs, n_coeff_blocks, nullptr, nullptr, 0);
}

int h263mi_submit_picture_events(h263mi_state *s, const h263mi_picture_desc *desc, const h263mi_mb_record *mbs,
                                 size_t n_mbs, const uint32_t *block_first_event, size_t n_coeff_blocks,
                                 const uint32_t *events, size_t n_events)
{
    if (!block_first_event) return H263MI_ERR_INVALID_ARGUMENT;
    if (n_coeff_blocks > 0xffffffffu / 8u) return H263MI_ERR_INVALID_ARGUMENT;
    return submit_records(s, desc, mbs, n_mbs, nullptr, n_coeff_blocks, block_first_event, events, n_events);
}

int h263mi_decode_next_picture(h263mi_state *s, const uint8_t *data, size_t len, size_t *consumed)
{
    if (!s || (!data && len)) return H263MI_ERR_INVALID_ARGUMENT;
    if (consumed) *consumed = 0;
    // serial half on the host (state.rs:143-427) ...
    bits::ParsedPicture &pic = s->parsed;                // (kept between calls: no allocation per picture)
    pic.want_dense = false;                              // the coefficients travel as events
    pic.size_fits = &picture_size_fits;
    pic.mbs_ext = nullptr;
    pic.mbs_ext_cap = 0;
    if (s->b) {
        // A stream rarely changes its size: the records are parsed straight into the pinned staging slot the next submit
        // copies from (sized for the last picture; a picture with more macroblocks falls back to the parser's own array).
        // The slot was last read by the copy of two pictures ago.
        DeviceGuard g(s->cfg.device_id);
        h263mi_state::Staging &g2 = s->stg[s->next_slot & 1];
        const size_t total = (size_t)s->b->L.mbw * s->b->L.mbh;
        if (g.ok && state_ensure_staging(g2, total, 1, 1, s->b->placement) == H263MI_OK && hipEventSynchronize(g2.done) == hipSuccess) {
            pic.mbs_ext = g2.h_mbs;
            pic.mbs_ext_cap = total;
        }
    }
    RC_TRY(bits::parse_picture(data, len, s->options, &s->parser_ctx, pic));
    // ... everything from the cut line on (state.rs:421-483) on the GPU.  Nothing has touched the state so
    // far, so every error above leaves it unchanged, like the reader transaction of state.rs:142.
    if (pic.n_coded_blocks > 0xffffffffu / 8u) return H263MI_ERR_INVALID_ARGUMENT;
    RC_TRY(submit_records(s, &pic.desc, pic.records(), pic.n_records(), nullptr, pic.n_coded_blocks, pic.block_first_event.data(),
                          pic.events.data(), pic.events.size(), /*from_parser=*/true));
    s->parser_ctx = pic.next;
    if (consumed) *consumed = pic.bits_consumed / 8;     // reader.commit() drains whole bytes (reader.rs:391-394)
    return H263MI_OK;
}

int h263mi_parse_picture_header(const h263mi_state *s, const uint8_t *data, size_t len, h263mi_picture_desc *out)
{
    if (!s || !out || (!data && len)) return H263MI_ERR_INVALID_ARGUMENT;
    bits::BitReader r(data, len);
    bits::PictureHeader h;
    bool is_picture = false;
    RC_TRY(bits::decode_picture_header(r, s->options, &s->parser_ctx, h, is_picture));
    if (!is_picture) return H263MI_ERR_MIDDLE_OF_BITSTREAM;
    memset(out, 0, sizeof *out);
    out->width = h.width;
    out->height = h.height;
    out->picture_type = h.picture_type;
    out->pquant = h.quantizer;
    out->use_deblocker = h.use_deblocker ? 1 : 0;
    out->temporal_reference = h.temporal_reference;
    return H263MI_OK;
}

static int fill_view(const h263mi_state *s, h263mi_frame_view *out)
{
    const h263mi_batch *b = s->b;
    const uint8_t *f = b->frames[b->ss[0].cur];
    memset(out, 0, sizeof *out);
    out->width = (uint16_t)b->L.width;
    out->height = (uint16_t)b->L.height;
    out->chroma_width = (uint16_t)b->L.cwidth;
    out->chroma_height = (uint16_t)b->L.cheight;
    out->temporal_reference = s->last_desc.temporal_reference;
    out->picture_type = s->last_desc.picture_type;
    out->pquant = s->last_desc.pquant;
    out->use_deblocker = s->last_desc.use_deblocker;
    out->dev_y = f;
    out->dev_cb = f + b->L.off_cb;
    out->dev_cr = f + b->L.off_cr;
    out->dev_pitch_y = b->L.pitch_y;
    out->dev_pitch_c = b->L.pitch_c;
    return H263MI_OK;
}

int h263mi_get_last_picture(const h263mi_state *s, h263mi_frame_view *out)
{
    if (!s || !out) return H263MI_ERR_INVALID_ARGUMENT;
    if (!s->has_last || !s->b || s->b->ss[0].cur < 0) return H263MI_ERR_NO_PICTURE;
    return fill_view(s, out);
}

int h263mi_get_reference_picture(const h263mi_state *s, h263mi_frame_view *out)
{
    if (!s || !out) return H263MI_ERR_INVALID_ARGUMENT;
    // state.rs:72-78: None without a reference, otherwise the entry of *last_picture*
    if (!s->has_last || !s->b || s->b->ss[0].cur < 0 || !s->b->ss[0].has_ref) return H263MI_ERR_NO_PICTURE;
    return fill_view(s, out);
}

int h263mi_copy_yuv(const h263mi_state *s, uint8_t *y, uint8_t *cb, uint8_t *cr)
{
    if (!s) return H263MI_ERR_INVALID_ARGUMENT;
    if (!s->has_last || !s->b) return H263MI_ERR_NO_PICTURE;
    DeviceGuard g(s->cfg.device_id);
    return s->b->copy_yuv(0, y, cb, cr);
}

// `strength` of the rendering calls: 0..12, or H263MI_STRENGTH_FROM_HEADER = what the last picture's own header asks for
// (QUANT_TO_STRENGTH[quantizer] when USE_DEBLOCKER is set: deblock.rs:5-8, picture.rs:61-64)
static int state_strength(const h263mi_state *s, uint8_t strength, h263mi_batch::Strengths &st)
{
    if (strength == H263MI_STRENGTH_FROM_HEADER) {
        st.uniform = strength_from_header(s->last_desc);
        return H263MI_OK;
    }
    if (strength > 12) return H263MI_ERR_INVALID_ARGUMENT;
    st.uniform = strength;
    return H263MI_OK;
}

int h263mi_render_rgba(const h263mi_state *cs, uint8_t strength, uint8_t *rgba)
{
    h263mi_state *s = const_cast<h263mi_state *>(cs);
    if (!s || !rgba) return H263MI_ERR_INVALID_ARGUMENT;
    if (!s->has_last || !s->b) return H263MI_ERR_NO_PICTURE;
    DeviceGuard g(s->cfg.device_id);
    h263mi_batch *b = s->b;
    const size_t bytes = (size_t)b->L.width * b->L.height * 4;
    if (bytes > s->cap_rgba) {
        if (s->d_rgba) (void)hipFree(s->d_rgba);
        s->d_rgba = nullptr; s->cap_rgba = 0;
        HIP_TRY(hipMalloc((void **)&s->d_rgba, bytes));
        s->cap_rgba = bytes;
    }
    h263mi_batch::Strengths st;
    RC_TRY(state_strength(s, strength, st));
    RC_TRY(b->render(st, s->d_rgba, nullptr));
    HIP_TRY(hipMemcpyAsync(rgba, s->d_rgba, bytes, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return H263MI_OK;
}

int h263mi_render_rgba_pinned(const h263mi_state *cs, uint8_t strength, uint8_t *rgba_pinned)
{
    h263mi_state *s = const_cast<h263mi_state *>(cs);
    if (!s || !rgba_pinned) return H263MI_ERR_INVALID_ARGUMENT;
    if (!s->has_last || !s->b) return H263MI_ERR_NO_PICTURE;
    DeviceGuard g(s->cfg.device_id);
    h263mi_batch *b = s->b;
    // the device's view of the caller's page-locked buffer: the kernel stores RGBA straight into it
    void *dev = nullptr;
    if (hipHostGetDevicePointer(&dev, rgba_pinned, 0) != hipSuccess || !dev) {
        (void)hipGetLastError();
        return H263MI_ERR_INVALID_ARGUMENT;          // not from h263mi_host_alloc / h263mi_host_register
    }
    h263mi_batch::Strengths st;
    RC_TRY(state_strength(s, strength, st));
    RC_TRY(b->render(st, static_cast<uint8_t *>(dev), nullptr));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return H263MI_OK;
}

}  // extern "C"
