// batch_staging.cpp -- host data -> pinned staging -> one launch: the batch entry points that take HOST records
// (h263mi_batch_submit_host[_events]) and coded pictures (h263mi_batch_decode_next_pictures[_ex|_ps]: N x
// H263State::decode_next_picture, state.rs:138-141, the serial parses on the batch's host threads).  SURVEY section 8 rows
// f-1 / f-2.
#include "batch.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>

using namespace h263mi;

// =========================================================================================
// the staging slots
// =========================================================================================
int h263mi_batch::ensure_record_staging(HostStaging &g2)
{
    const size_t total = (size_t)n * L.mbw * L.mbh;
    PlacementScope near_device(placement);       // pinned memory on the NUMA node of this batch's GPU (worker_pool.h)
    // each piece on its own, so that a failed allocation leaves nothing half-initialised for the next call
    if (!g2.h_mbs) HIP_TRY(hipHostMalloc((void **)&g2.h_mbs, total * sizeof(MbRecord), hipHostMallocDefault));
    if (!g2.d_mbs) HIP_TRY(hipMalloc((void **)&g2.d_mbs, total * sizeof(MbRecord)));
    if (!g2.done) HIP_TRY(hipEventCreateWithFlags(&g2.done, hipEventDisableTiming));
    return H263MI_OK;
}

int h263mi_batch::ensure_host_staging(HostStaging &g2, size_t n_blocks, size_t n_event_words)
{
    PlacementScope near_device(placement);
    const size_t head = head_words();
    if (head + n_event_words > g2.cap_words) {
        if (g2.h_words) (void)hipHostFree(g2.h_words);
        if (g2.d_words) (void)hipFree(g2.d_words);
        g2.h_words = nullptr; g2.d_words = nullptr; g2.cap_words = 0;
        const size_t cap = head + n_event_words + n_event_words / 2 + 256;
        HIP_TRY(hipHostMalloc((void **)&g2.h_words, cap * sizeof(uint32_t), hipHostMallocDefault));
        HIP_TRY(hipMalloc((void **)&g2.d_words, cap * sizeof(uint32_t)));
        g2.cap_words = cap;
    }
    g2.h_base = reinterpret_cast<uint64_t *>(g2.h_words);  g2.d_base = reinterpret_cast<uint64_t *>(g2.d_words);
    g2.h_index = g2.h_words + 4 * (size_t)n;               g2.d_index = g2.d_words + 4 * (size_t)n;
    g2.h_events = g2.h_words + head;                       g2.d_events = g2.d_words + head;
    RC_TRY(ensure_record_staging(g2));
    // with sparse transport there are no dense blocks anywhere: the reconstruction waves read the events
    if (!n_event_words && (n_blocks > g2.cap_blocks || !g2.h_coeffs)) {
        if (g2.h_coeffs) (void)hipHostFree(g2.h_coeffs);
        if (g2.d_coeffs) (void)hipFree(g2.d_coeffs);
        g2.h_coeffs = nullptr; g2.d_coeffs = nullptr;
        size_t cap = std::max(n_blocks, g2.cap_blocks);
        cap = cap + cap / 2 + 64;
        g2.cap_blocks = 0;
        HIP_TRY(hipHostMalloc((void **)&g2.h_coeffs, cap * 128, hipHostMallocDefault));
        if (hipMalloc((void **)&g2.d_coeffs, cap * 128) != hipSuccess) {
            if (g2.h_coeffs) (void)hipHostFree(g2.h_coeffs);
            g2.h_coeffs = nullptr;
            return H263MI_ERR_OUT_OF_MEMORY;
        }
        g2.cap_blocks = cap;
    }
    return H263MI_OK;
}

void h263mi_batch::release_staging()
{
    for (HostStaging &g2 : host_stg) {
        if (g2.h_mbs) (void)hipHostFree(g2.h_mbs);
        if (g2.d_mbs) (void)hipFree(g2.d_mbs);
        if (g2.h_coeffs) (void)hipHostFree(g2.h_coeffs);
        if (g2.d_coeffs) (void)hipFree(g2.d_coeffs);
        if (g2.h_words) (void)hipHostFree(g2.h_words);
        if (g2.d_words) (void)hipFree(g2.d_words);
        if (g2.done) (void)hipEventDestroy(g2.done);
        g2 = HostStaging();
    }
}

namespace h263mi {

int batch_submit_host(h263mi_batch *b, uint8_t picture_type, const h263mi_mb_record *const *mbs, const uint32_t *n_mbs,
                      const int16_t *const *coeffs, const uint32_t *n_coeff_blocks, const uint32_t *const *first_event,
                      const uint32_t *const *events, const uint32_t *n_events, bool from_parser, uint32_t pack_threads,
                      const uint8_t *types, bool deferred_post, const uint32_t *const *group_index, const DirectWords *direct)
{
    // group_index (from the parser only): SPARSE RECORDS -- mbs[i] holds the n_mbs[i] records of stream i's coded macroblocks
    // (written in place at the head of the stream's part of the staging slot) and group_index[i] one word per group of 8
    // macroblocks (bits::ParsedPicture::sparse_records, ReconArgs::mb_group_index): what crosses the link is the head of every
    // stream's part, as long as the longest of them -- one 2-D copy; a third of the bytes of the dense arrays on real content.
    // (Packing the streams' records one behind the other for a plain copy was measured too: the call as a whole 0.45 -> 0.48 ms
    // -- the packing pass and the parser without its non-temporal stores cost more than the plain copy saves; and so was letting
    // the waves read the records out of the pinned slot over the link, no copy at all: +-0.  In the steady state a call IS its
    // parse phase: 0.43-0.53 ms on 16 threads against 0.03 ms of packing and 0.01 ms of enqueueing
    // (profiles/r05_j_e2e_per_call_packed_records.txt, r05_m_*).  The packing went last: `direct`, below.)
    const bool sparse_rec = group_index != nullptr && from_parser;
    // from_parser: the arrays are what bits::parse_picture just wrote (h263mi_batch_decode_next_pictures) -- valid by
    // construction, so the per-record checks a caller's arrays get are skipped; pack_threads: the caller's thread budget
    const bool sparse = first_event != nullptr;
    if (!b || !mbs || !n_mbs || !n_coeff_blocks || (!sparse && !coeffs) || (sparse && (!events || !n_events)) ||
        picture_type > H263MI_PICTURE_RESERVED)
        return H263MI_ERR_INVALID_ARGUMENT;
    const size_t per = (size_t)b->L.mbw * b->L.mbh;
    size_t blocks = 0, n_ev = 0;
    for (uint32_t i = 0; i < b->n; i++) {
        if (n_mbs[i] > per || (n_mbs[i] && !mbs[i])) return H263MI_ERR_INVALID_ARGUMENT;
        if (n_coeff_blocks[i] && !direct) {
            if (!sparse && !coeffs[i]) return H263MI_ERR_INVALID_ARGUMENT;
            if (sparse && (!first_event[i] || first_event[i][0] != 0 || first_event[i][n_coeff_blocks[i]] != n_events[i] ||
                           (n_events[i] && !events[i])))
                return H263MI_ERR_INVALID_ARGUMENT;
        }
        // block offsets inside a stream's share of the pool are 32-bit byte offsets on the device (recon_block_limit)
        if (n_coeff_blocks[i] > (1u << 25)) return H263MI_ERR_INVALID_ARGUMENT;
        blocks += n_coeff_blocks[i];
        if (sparse) n_ev += n_events[i];
    }
    if (blocks > 0xffffffffu / 8u || n_ev > kMaxEventWords) return H263MI_ERR_INVALID_ARGUMENT;
    DeviceGuard g(b->device);
    if (!g.ok) return H263MI_ERR_NO_DEVICE;
    h263mi_batch::HostStaging &g2 = b->host_stg[b->host_slot & 1];
    if (direct && !(from_parser && sparse && group_index)) return H263MI_ERR_INVALID_ARGUMENT;
    // (direct: the slot was sized before the parser wrote into it, batch_decode_next_pictures)
    const size_t event_words = direct ? (size_t)b->n * (direct->pitch_blocks + direct->pitch_events) : sparse ? blocks + 1 + n_ev : 0;
    if (direct && (b->head_words() + event_words > g2.cap_words || event_words > kMaxEventWords)) return H263MI_ERR_INVALID_ARGUMENT;
    if (!direct) RC_TRY(b->ensure_host_staging(g2, blocks ? blocks : 1, event_words));
    const auto t_wait0 = std::chrono::steady_clock::now();
    HIP_TRY(hipEventSynchronize(g2.done));       // the kernel that read this slot two pictures ago is done
    const auto t_pack0 = std::chrono::steady_clock::now();

    MbRecord pad;                                // state.rs:421-427: Inter, mv (0,0), nothing coded
    memset(&pad, 0, sizeof pad);
    pad.mb_type = H263MI_MB_INTER;
    pad.quant = 1;
    std::vector<uint32_t> ev_base(b->n + 1, 0);
    size_t at = 0, rec_at = 0, most_blocks = 0, most_events = 0;
    const size_t groups_pp = (size_t)recon_tiles_x(b->L) * b->L.mbh;
    for (uint32_t i = 0; i < b->n; i++) {        // coeff_index of stream i counts from its own first block
        g2.h_base[i] = direct ? (uint64_t)i * direct->pitch_blocks : at;
        at += n_coeff_blocks[i];
        if (direct && b->ss[i].active) {
            most_blocks = std::max<size_t>(most_blocks, n_coeff_blocks[i]);
            most_events = std::max<size_t>(most_events, n_events[i]);
        }
        ev_base[i + 1] = ev_base[i] + (sparse ? n_events[i] : 0);
        g2.h_base[b->n + i] = (uint64_t)i * per; // (sparse records: stream i's first record)
        if (sparse_rec && b->ss[i].active && n_mbs[i] > rec_at) rec_at = n_mbs[i];
    }
    const size_t records_sent = sparse_rec ? rec_at : 0;         // records of the stream that has the most
    uint32_t *h_first = g2.h_events, *h_ev = sparse ? g2.h_events + blocks + 1 : nullptr;
    std::atomic<bool> offsets_ok{true}, records_ok{true};
    // packing is a host memcpy of every record byte: a few threads, or one core caps the rate below the PCIe link
    auto pack = [&](uint32_t first, uint32_t last, uint32_t step) {
        for (uint32_t i = first; i < last; i += step) {
            if (!b->ss[i].active) continue;      // sits the call out: its records are never read (STREAM_RECON_SKIP)
            MbRecord *dst = g2.h_mbs + (size_t)i * per;
            if (sparse_rec) {
                if (group_index[i]) memcpy(g2.h_index + (size_t)i * groups_pp, group_index[i], groups_pp * sizeof(uint32_t));
                else memset(g2.h_index + (size_t)i * groups_pp, 0, groups_pp * sizeof(uint32_t));      // (no record at all)
            }
            for (uint32_t k = 0; k < n_mbs[i] && !from_parser; k++) {   // the same checks as h263mi_submit_picture
                const MbRecord &m = mbs[i][k];
                // (a record without coded blocks does not use its coeff_index)
                if (m.mb_type > H263MI_MB_INTER4V_Q || m.quant < 1 || m.quant > 31 || (m.cbp & 0xC0) || (m.kill & 0xC0) ||
                    (m.cbp && (uint64_t)m.coeff_index + (uint64_t)__builtin_popcount(m.cbp) > n_coeff_blocks[i]))
                    records_ok.store(false, std::memory_order_relaxed);
            }
            // (h263mi_batch_decode_next_pictures has its parser write the records straight into this slot)
            if (n_mbs[i] && mbs[i] != dst) memcpy(dst, mbs[i], (size_t)n_mbs[i] * sizeof(MbRecord));
            for (size_t k = n_mbs[i]; k < per && !sparse_rec; k++) dst[k] = pad;      // (sparse records: no record = not coded)
            if (!n_coeff_blocks[i]) continue;
            if (!sparse) {
                memcpy(g2.h_coeffs + g2.h_base[i] * 64, coeffs[i], (size_t)n_coeff_blocks[i] * 128);
            } else {
                uint32_t *fo = h_first + g2.h_base[i];
                bool ascending = true;
                for (uint32_t k = 0; k < n_coeff_blocks[i]; k++) {
                    ascending = ascending && first_event[i][k] <= first_event[i][k + 1] && first_event[i][k + 1] <= n_events[i];
                    fo[k] = first_event[i][k] + ev_base[i];
                }
                // a caller's events: at most 64 per block, every position once (the device places them in no particular order)
                for (uint32_t k = 0; k < n_coeff_blocks[i] && ascending && !from_parser; k++) {
                    uint64_t seen = 0;
                    const uint32_t e0 = first_event[i][k], e1 = first_event[i][k + 1];
                    if (e1 - e0 > 64) ascending = false;
                    for (uint32_t e = e0; e < e1 && ascending; e++) {
                        const uint64_t bit = 1ull << (events[i][e] & 63u);
                        if (seen & bit) ascending = false;
                        seen |= bit;
                    }
                }
                if (!ascending) offsets_ok.store(false, std::memory_order_relaxed);
                if (n_events[i]) memcpy(h_ev + ev_base[i], events[i], (size_t)n_events[i] * sizeof(uint32_t));
            }
        }
    };
    const size_t bytes = (sparse_rec ? records_sent * b->n : (size_t)b->n * per) * sizeof(MbRecord) + (sparse ? event_words * 4 : blocks * 128);
    const uint32_t n_thr = bytes < (4u << 20) ? 1u
                         : std::min<uint32_t>({pack_threads ? pack_threads : 8u, b->n, std::max(1u, std::thread::hardware_concurrency())});
    if (direct) {
        // nothing to pack: records, index, block offsets and events are where the copies read them
    } else if (n_thr <= 1) {
        pack(0, b->n, 1);
    } else {
        // (thread t packs the streams t, t + T, ...: the ones it has just parsed, see StreamDeal)
        b->workers(n_thr).run(n_thr, [&](unsigned t) { pack(t, b->n, n_thr); }, b->pool_spin_us);
    }
    if (!offsets_ok.load() || !records_ok.load()) return H263MI_ERR_INVALID_ARGUMENT;      // nothing has been queued yet
    const auto t_enq0 = std::chrono::steady_clock::now();
    RC_TRY(b->time_close());                     // the copies below are not part of any kernel's time
    // (A stream of their own for these copies -- beside the kernel of the call before -- was measured in round 5 and dropped:
    // the wait for the staging slot went from 0.15 ms to 0.01 ms per call, and the call as a whole from 0.60 to 0.68 ms: the
    // copies are blit kernels, they then share the CUs with k_frame and the host's memory with the parser threads.
    // profiles/r05_g_e2e_per_call*.txt)
    hipStream_t cs = b->stream;
    const auto enqueue_copies = [&]() -> int {
        if (sparse_rec) {
            // sparse records: the head of every stream's part in one 2-D copy (the index words travel with the small things below)
            if (records_sent)
                HIP_TRY(hipMemcpy2DAsync(g2.d_mbs, per * sizeof(MbRecord), g2.h_mbs, per * sizeof(MbRecord), records_sent * sizeof(MbRecord),
                                         b->n, hipMemcpyHostToDevice, cs));
        }
        // the records of the streams that take part, one copy per run of neighbouring streams (all of them: one copy)
        for (uint32_t i = 0; i < b->n && !sparse_rec;) {
            if (!b->ss[i].active) { i++; continue; }
            uint32_t j = i + 1;
            while (j < b->n && b->ss[j].active) j++;
            HIP_TRY(hipMemcpyAsync(g2.d_mbs + (size_t)i * per, g2.h_mbs + (size_t)i * per, (size_t)(j - i) * per * sizeof(MbRecord),
                                   hipMemcpyHostToDevice, cs));
            i = j;
        }
        if (direct) {
            // the bases and the record index in one copy, the used heads of the streams' block offsets and events in a 2-D
            // copy each
            HIP_TRY(hipMemcpyAsync(g2.d_words, g2.h_words, b->head_words() * sizeof(uint32_t), hipMemcpyHostToDevice, cs));
            if (blocks) {
                uint32_t *const h_ev0 = g2.h_events + (size_t)b->n * direct->pitch_blocks;
                uint32_t *const d_ev0 = g2.d_events + (size_t)b->n * direct->pitch_blocks;
                HIP_TRY(hipMemcpy2DAsync(g2.d_events, direct->pitch_blocks * sizeof(uint32_t), g2.h_events, direct->pitch_blocks * sizeof(uint32_t),
                                         (most_blocks + 1) * sizeof(uint32_t), b->n, hipMemcpyHostToDevice, cs));
                if (most_events)
                    HIP_TRY(hipMemcpy2DAsync(d_ev0, direct->pitch_events * sizeof(uint32_t), h_ev0, direct->pitch_events * sizeof(uint32_t),
                                             most_events * sizeof(uint32_t), b->n, hipMemcpyHostToDevice, cs));
            }
            return H263MI_OK;
        }
        // the bases, the record index and the events: one copy (HostStaging::h_words)
        if (sparse && blocks) h_first[blocks] = (uint32_t)n_ev;
        HIP_TRY(hipMemcpyAsync(g2.d_words, g2.h_words, (b->head_words() + (sparse && blocks ? event_words : 0)) * sizeof(uint32_t),
                               hipMemcpyHostToDevice, cs));
        if (!sparse && blocks) {
            HIP_TRY(hipMemcpyAsync(g2.d_coeffs, g2.h_coeffs, blocks * 128, hipMemcpyHostToDevice, cs));
        }
        return H263MI_OK;
    };
    {
        const int crc = enqueue_copies();
        if (b->trace_host) b->host_ms[4] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enq0).count();
        if (crc != H263MI_OK) {
            // nothing is launched: no copy may still be reading this slot's host memory when a later call fills it again
            (void)hipStreamSynchronize(cs);
            return crc;
        }
    }
    // where the launch finds it all (the waves bounds-check what they read: the sizes are known here)
    h263mi_coeff_source src;
    src.coeffs = g2.d_coeffs;
    src.coeff_base = g2.d_base;
    if (sparse && blocks) {
        // the reconstruction waves read the events themselves (recon_kernel.inl: coeff_row_from_events); round 2 had a
        // kernel of its own (k_expand) rebuild dense blocks in HBM first
        src.first_event = g2.d_events;
        src.events = direct ? g2.d_events + (size_t)b->n * direct->pitch_blocks : g2.d_events + blocks + 1;
        src.n_events = direct ? (uint32_t)((size_t)b->n * direct->pitch_events) : (uint32_t)n_ev;
    }
    src.pool_blocks = direct ? (uint64_t)b->n * direct->pitch_blocks : blocks;
    src.checked = true;
    if (sparse_rec) {
        src.group_index = g2.d_index;
        src.mb_base = g2.d_base + b->n;
    }
    {
        const auto t_sub0 = std::chrono::steady_clock::now();
        const int src_rc = b->submit(picture_type, g2.d_mbs, src, /*with_post=*/deferred_post && b->pending.valid, types);
        if (b->trace_host) b->host_ms[5] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_sub0).count();
        if (src_rc != H263MI_OK) {
            (void)hipStreamSynchronize(cs);      // (no copy left behind that reads this slot)
            return src_rc;
        }
    }
    // ---- the launch is queued and the streams have advanced: nothing below may turn that into an error
    // (timing: the bracket ends right behind the launch -- closed at the next call it would hold the time the device
    // idles while the host parses the next pictures)
    (void)b->time_close();
    if (hipEventRecord(g2.done, b->stream) != hipSuccess) (void)hipStreamSynchronize(b->stream);   // (the slot is reused two calls on)
    b->host_slot++;
    if (b->trace_host) {
        const auto t_end = std::chrono::steady_clock::now();
        b->host_ms[1] += std::chrono::duration<double, std::milli>(t_pack0 - t_wait0).count();
        b->host_ms[2] += std::chrono::duration<double, std::milli>(t_enq0 - t_pack0).count();
        b->host_ms[3] += std::chrono::duration<double, std::milli>(t_end - t_enq0).count();
        b->host_calls++;
        if (b->trace_each)                       // H263MI_TRACE_E2E=2: one line per call
            fprintf(stderr, "h263mi call %u: wait %.3f pack %.3f enqueue %.3f ms (%zu blocks, %zu events)\n", b->host_calls,
                    std::chrono::duration<double, std::milli>(t_pack0 - t_wait0).count(),
                    std::chrono::duration<double, std::milli>(t_enq0 - t_pack0).count(),
                    std::chrono::duration<double, std::milli>(t_end - t_enq0).count(), blocks, n_ev);
    }
    return H263MI_OK;
}

// N x decode_next_picture.  stream_rc == nullptr: all or nothing (any stream's error fails the call, nothing changes).
// stream_rc != nullptr: every stream is its own H263State -- a stream that fails keeps its state (state.rs:142) and gets
// its error code, a stream without data (data[i] == nullptr) is left alone, the others advance.
// st: the post-filter strength of the pictures of this call; st.per_stream is filled in here when `from_header` is set:
// each picture with what its own header asks for (host_common.h: strength_from_header).
static int batch_decode_next_pictures(h263mi_batch *b, uint32_t decoder_options, const uint8_t *const *data, const size_t *len,
                                      size_t *consumed, uint32_t n_threads, int *stream_rc, h263mi_batch::Strengths st,
                                      bool from_header, uint8_t *d_rgba, uint8_t *d_deblocked)
{
    if (!b || !data || !len) return H263MI_ERR_INVALID_ARGUMENT;
    const uint32_t n = b->n;
    // data[i] == NULL: in the _ex form (stream_rc given) the stream has no picture in this call; in the plain form every
    // stream decodes, and NULL with length 0 is an empty reader (the parser answers with its end-of-stream error, as it does
    // for a non-NULL pointer with length 0 in either form)
    static const uint8_t kEmptyReader[1] = {0};
    std::vector<const uint8_t *> data_fixed;
    for (uint32_t i = 0; i < n; i++) {
        if (!data[i] && len[i]) return H263MI_ERR_INVALID_ARGUMENT;
        if (!data[i] && !stream_rc) {
            if (data_fixed.empty()) data_fixed.assign(data, data + n);
            data_fixed[i] = kEmptyReader;
        }
    }
    if (!data_fixed.empty()) data = data_fixed.data();
    if (b->parser_ctx.size() != n) b->parser_ctx.assign(n, bits::ParserContext());
    if (b->parsed.size() != n) b->parsed.resize(n);
    // The records are parsed straight into the pinned staging slot this call will copy from (stream i at i * mbs per
    // picture): no second pass over them.  The slot was last read by the copy of two calls ago.
    DeviceGuard g(b->device);
    if (!g.ok) return H263MI_ERR_NO_DEVICE;
    h263mi_batch::HostStaging &g2 = b->host_stg[b->host_slot & 1];
    RC_TRY(b->ensure_record_staging(g2));
    {
        // (this is where a call waits when the GPU stream -- copies + kernel of two calls ago -- is the slower side)
        const auto t_wait = std::chrono::steady_clock::now();
        HIP_TRY(hipEventSynchronize(g2.done));
        if (b->trace_host) b->host_ms[1] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_wait).count();
    }
    const size_t per = (size_t)b->L.mbw * b->L.mbh;
    // H263MI_SPARSE_RECORDS=0: dense record arrays over the link, as rounds 2-4 sent them (A/B switch)
    static const bool sparse_rec = !(getenv("H263MI_SPARSE_RECORDS") && getenv("H263MI_SPARSE_RECORDS")[0] == '0');
    // DIRECT WORDS (see DirectWords): every stream's events, block offsets and group index are parsed straight into the
    // staging slot, at pitches that hold the worst case of this call's pictures -- as long as that worst case is a sensible
    // amount of pinned memory (a 1080p key frame of 100 KB: 1 MB per stream; the 2.3 MB test key frames take the packed form).
    // H263MI_DIRECT_WORDS=0: the packed form always (A/B switch).
    static const bool direct_allowed = !(getenv("H263MI_DIRECT_WORDS") && getenv("H263MI_DIRECT_WORDS")[0] == '0');
    constexpr size_t kDirectEventBytesMax = (size_t)128 << 20;
    DirectWords dw{0, 0};
    bool direct = sparse_rec && direct_allowed;
    if (direct) {
        size_t longest = 0;
        for (uint32_t i = 0; i < n; i++)
            if (data[i] && b->ss[i].active) longest = std::max(longest, len[i]);
        dw.pitch_blocks = (bits::block_offset_words_bound(per) + 15) & ~(size_t)15;
        dw.pitch_events = (bits::event_words_bound(longest, per) + 15) & ~(size_t)15;
        direct = (size_t)n * dw.pitch_events * sizeof(uint32_t) <= kDirectEventBytesMax &&
                 (size_t)n * (dw.pitch_events + dw.pitch_blocks) <= kMaxEventWords;
        if (direct) RC_TRY(b->ensure_host_staging(g2, 1, (size_t)n * (dw.pitch_blocks + dw.pitch_events)));
    }
    const size_t groups_pp = (size_t)recon_tiles_x(b->L) * b->L.mbh;
    // ---- the serial half of decode_next_picture (state.rs:143-427), one stream per task, on n_threads host threads
    std::vector<int> rcs(n, H263MI_OK);
    const HostThreadPlan plan = host_thread_plan(n, n_threads);
    const uint32_t n_thr = plan.threads;
    b->pool_spin_us = plan.spin_us;
    StreamDeal deal(n);
    auto work = [&](unsigned t) {
        deal.run(t, n_thr, [&](uint32_t i) {
            if (!data[i] || !b->ss[i].active) return;            // no picture for this stream in this call
            bits::ParsedPicture &pic = b->parsed[i];
            pic.want_dense = false;                              // the coefficients travel as events
            pic.size_fits = &picture_size_fits;
            pic.sparse_records = sparse_rec;                     // records for the coded macroblocks only (round 5)
            pic.mbs_ext = g2.h_mbs + (size_t)i * per;
            pic.mbs_ext_cap = per;
            pic.events_ext = direct ? g2.h_events + (size_t)n * dw.pitch_blocks + (size_t)i * dw.pitch_events : nullptr;
            pic.events_ext_cap = direct ? dw.pitch_events : 0;
            pic.first_event_ext = direct ? g2.h_events + (size_t)i * dw.pitch_blocks : nullptr;
            pic.first_event_ext_cap = direct ? dw.pitch_blocks : 0;
            pic.group_index_ext = direct ? g2.h_index + (size_t)i * groups_pp : nullptr;
            pic.group_index_ext_cap = direct ? groups_pp : 0;
            pic.event_base = direct ? (uint32_t)((size_t)i * dw.pitch_events) : 0u;
            int rc = bits::parse_picture(data[i], len[i], decoder_options, &b->parser_ctx[i], pic);
            if (rc == H263MI_OK && (pic.desc.width != b->L.width || pic.desc.height != b->L.height)) rc = H263MI_ERR_PICTURE_FORMAT_INVALID;
            // (a picture of the batch's size fits the pitches by construction: anything else is a fault of this library)
            if (rc == H263MI_OK && direct && !pic.words_ext_used) rc = H263MI_ERR_INTERNAL_DECODER_ERROR;
            // gather.rs:149: an inter macroblock without a reference picture is Error::UncodedIFrameBlocks -- found here,
            // before anything is queued, so that the stream (parser state included) stays as it was (macroblocks the picture
            // does not code are padded as Inter, state.rs:421-427: the parser's any_inter covers them)
            if (rc == H263MI_OK && !(b->ss[i].has_ref && b->ss[i].cur >= 0) && pic.any_inter) rc = H263MI_ERR_UNCODED_IFRAME_BLOCKS;
            rcs[i] = rc;
        });
    };
    const auto t_parse0 = std::chrono::steady_clock::now();
    if (n_thr == 1) work(0);
    else b->workers(n_thr).run(n_thr, work, plan.spin_us);
    if (b->trace_host) {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_parse0).count();
        b->host_ms[0] += ms;
        if (b->trace_each) fprintf(stderr, "h263mi parse phase: %.3f ms on %u threads\n", ms, n_thr);
    }
    std::vector<uint8_t> takes_part(n), types(n, H263MI_PICTURE_P);
    int first_error = H263MI_OK;
    uint32_t n_ok = 0;
    for (uint32_t i = 0; i < n; i++) {
        takes_part[i] = data[i] && b->ss[i].active && rcs[i] == H263MI_OK;
        if (rcs[i] != H263MI_OK && first_error == H263MI_OK) first_error = rcs[i];
        if (takes_part[i]) {
            types[i] = b->parsed[i].desc.picture_type;
            n_ok++;
        }
        if (stream_rc) stream_rc[i] = rcs[i];
    }
    // all or nothing: the batch -- frames, reference bookkeeping and what it remembers of the picture headers -- is
    // unchanged (state.rs:142)
    if (!stream_rc && first_error != H263MI_OK) return first_error;
    if (consumed)
        for (uint32_t i = 0; i < n; i++) consumed[i] = 0;
    if (!n_ok) return first_error;
    std::vector<const h263mi_mb_record *> mbs(n);
    std::vector<const uint32_t *> first(n), events(n), gidx(n, nullptr);
    std::vector<uint32_t> n_mbs(n, 0), n_blocks(n, 0), n_events(n, 0);
    static const uint32_t kNoEvents[1] = {0};
    for (uint32_t i = 0; i < n; i++) {
        const bits::ParsedPicture &pic = b->parsed[i];
        mbs[i] = g2.h_mbs + (size_t)i * per;
        first[i] = kNoEvents;
        events[i] = nullptr;
        if (!takes_part[i]) continue;
        mbs[i] = pic.records();
        n_mbs[i] = (uint32_t)pic.n_records();
        gidx[i] = pic.group_index_words();
        first[i] = pic.first_event_words();
        events[i] = pic.event_words();
        n_blocks[i] = (uint32_t)pic.n_coded_blocks;
        n_events[i] = (uint32_t)pic.n_event_words();
    }
    // the streams that take part in THIS call (restored below: h263mi_batch_set_active is the caller's)
    std::vector<uint8_t> was_active(n);
    for (uint32_t i = 0; i < n; i++) {
        was_active[i] = b->ss[i].active;
        b->ss[i].active = takes_part[i] != 0;
    }
    if (from_header)
        for (uint32_t i = 0; i < n; i++) st.per_stream[i] = takes_part[i] ? strength_from_header(b->parsed[i].desc) : (uint8_t)0;
    const bool deferred = b->pipeline_post && (d_rgba || d_deblocked);
    int rc = batch_submit_host(b, H263MI_PICTURE_P, mbs.data(), n_mbs.data(), nullptr, n_blocks.data(), first.data(), events.data(),
                               n_events.data(), /*from_parser=*/true, n_thr, types.data(), deferred, sparse_rec ? gidx.data() : nullptr,
                               direct ? &dw : nullptr);
    int render_rc = H263MI_OK;
    if (rc == H263MI_OK) {
        // the pictures are decoded: what the streams remember of their headers moves on with them, whatever happens to the
        // rendering below (a failed rendering is reported, but it does not un-decode anything)
        for (uint32_t i = 0; i < n; i++) {
            if (!takes_part[i]) continue;
            b->parser_ctx[i] = b->parsed[i].next;
            if (consumed) consumed[i] = b->parsed[i].bits_consumed / 8;      // reader.commit() drains whole bytes
        }
        if (deferred) render_rc = b->note_pending(st, d_rgba, d_deblocked);
        else if (d_rgba || d_deblocked) render_rc = b->render(st, d_rgba, d_deblocked, /*only_active=*/true);
    }
    for (uint32_t i = 0; i < n; i++) b->ss[i].active = was_active[i] != 0;
    RC_TRY(rc);
    RC_TRY(render_rc);
    return stream_rc ? H263MI_OK : first_error;
}

}  // namespace h263mi

extern "C" {

int h263mi_batch_submit_host(h263mi_batch *b, uint8_t picture_type, const h263mi_mb_record *const *mbs,
                             const uint32_t *n_mbs, const int16_t *const *coeffs, const uint32_t *n_coeff_blocks)
{
    return batch_submit_host(b, picture_type, mbs, n_mbs, coeffs, n_coeff_blocks, nullptr, nullptr, nullptr);
}

int h263mi_batch_submit_host_events(h263mi_batch *b, uint8_t picture_type, const h263mi_mb_record *const *mbs,
                                    const uint32_t *n_mbs, const uint32_t *const *block_first_event,
                                    const uint32_t *n_coeff_blocks, const uint32_t *const *events, const uint32_t *n_events)
{
    if (!block_first_event) return H263MI_ERR_INVALID_ARGUMENT;
    return batch_submit_host(b, picture_type, mbs, n_mbs, nullptr, n_coeff_blocks, block_first_event, events, n_events);
}

uint32_t h263mi_default_parser_threads(uint32_t n_streams, uint32_t *cpu_quota)
{
    const HostThreadPlan p = host_thread_plan(n_streams, 0);
    if (cpu_quota) *cpu_quota = p.quota_cpus;
    return p.threads;
}

void h263mi_set_ranks_per_node(uint32_t ranks) { set_ranks_per_node(ranks); }

int h263mi_batch_decode_next_pictures(h263mi_batch *b, uint32_t decoder_options, const uint8_t *const *data,
                                      const size_t *len, size_t *consumed, uint32_t n_threads)
{
    return batch_decode_next_pictures(b, decoder_options, data, len, consumed, n_threads, nullptr, h263mi_batch::Strengths(), false,
                                      nullptr, nullptr);
}

int h263mi_batch_decode_next_pictures_ps(h263mi_batch *b, uint32_t decoder_options, const uint8_t *const *data,
                                         const size_t *len, size_t *consumed, uint32_t n_threads, int *stream_rc,
                                         uint8_t strength, const uint8_t *strengths, uint8_t *d_rgba, uint8_t *d_deblocked)
{
    if (!b) return H263MI_ERR_INVALID_ARGUMENT;
    h263mi_batch::Strengths st;
    RC_TRY(make_strengths(strength, strengths, b->n, /*from_header_allowed=*/true, st));
    return batch_decode_next_pictures(b, decoder_options, data, len, consumed, n_threads, stream_rc, st,
                                      !strengths && strength == H263MI_STRENGTH_FROM_HEADER, d_rgba, d_deblocked);
}

int h263mi_batch_decode_next_pictures_ex(h263mi_batch *b, uint32_t decoder_options, const uint8_t *const *data,
                                         const size_t *len, size_t *consumed, uint32_t n_threads, int *stream_rc,
                                         uint8_t strength, uint8_t *d_rgba, uint8_t *d_deblocked)
{
    return h263mi_batch_decode_next_pictures_ps(b, decoder_options, data, len, consumed, n_threads, stream_rc, strength, nullptr,
                                                d_rgba, d_deblocked);
}

// where the host side of the batch was placed (bench report, tests): the NUMA node of its device (-1 = unknown / off), the
// node its pinned staging memory really lies on (-1 = no staging yet / unknown) and how many CPUs its pool's threads are
// confined to (0 = left alone / no pool yet); cpus (may be NULL) receives up to cpus_cap of their numbers
int h263mi_batch_host_placement(const h263mi_batch *b, int *device_numa_node, int *staging_numa_node, uint32_t *n_pool_cpus,
                                uint16_t *cpus, uint32_t cpus_cap)
{
    if (!b) return H263MI_ERR_INVALID_ARGUMENT;
    if (device_numa_node) *device_numa_node = b->placement.node;
    if (staging_numa_node) {
        const h263mi_batch::HostStaging &g2 = b->host_stg[0];
        *staging_numa_node = g2.h_mbs ? numa_node_of_address(g2.h_mbs) : -1;
    }
    uint32_t k = 0;
    if (b->pool) {
        const cpu_set_t set = b->pool->confined_to();
        for (int c = 0; c < CPU_SETSIZE; c++) {
            if (!CPU_ISSET(c, &set)) continue;
            if (cpus && k < cpus_cap) cpus[k] = (uint16_t)c;
            k++;
        }
    }
    if (n_pool_cpus) *n_pool_cpus = k;
    return H263MI_OK;
}

// TEST HOOK (no HIP call, usable without a device): what a batch on device `device` of a host with the PCI functions pci_ids[0 ..
// n_devices) would get as its host placement when `ranks` processes share the node, on the sysfs tree under `sysfs_root`
// (NULL = H263MI_SYSFS_ROOT or /sys).  *node = the NUMA node (-1: none), cpus / *n_cpus as in h263mi_batch_host_placement.
// tests/test_numa_placement.py and the 8-rank stand-in of bench.py use it on a made-up two-socket topology.
int h263mi_debug_host_placement(const char *const *pci_ids, uint32_t n_devices, int device, uint32_t ranks, const char *sysfs_root,
                                int *node, uint16_t *cpus, uint32_t cpus_cap, uint32_t *n_cpus)
{
    if (!pci_ids || !node || !n_cpus) return H263MI_ERR_INVALID_ARGUMENT;
    std::vector<std::string> ids;
    for (uint32_t d = 0; d < n_devices; d++) ids.push_back(pci_ids[d] ? pci_ids[d] : "");
    const HostPlacement p = host_placement(ids, device, ranks ? ranks : 1, sysfs_root);
    *node = p.node;
    uint32_t k = 0;
    for (int c = 0; c < CPU_SETSIZE && p.have_cpus; c++) {
        if (!CPU_ISSET(c, &p.cpus)) continue;
        if (cpus && k < cpus_cap) cpus[k] = (uint16_t)c;
        k++;
    }
    *n_cpus = k;
    return H263MI_OK;
}

}  // extern "C"
