// mixed_set.cpp -- streams of DIFFERENT picture sizes behind one call (h263mi_mixed_*)
//
// The reference resolves the picture format per H263State and per picture (state.rs:157-176): a server holds QCIF, CIF
// and 1080p streams side by side, and a stream may change its size at an I picture.  A fixed-geometry batch (above)
// takes one size; h263mi_mixed keeps one such batch per size CLASS -- created when the first stream of that size
// shows up -- and a call decodes every class that has pictures in it with ONE launch (k_frame on pipelined classes), back
// to back on the same HIP stream.  A stream belongs to the class of its last picture; an I picture of another size moves
// it (its old slot is given up once the new picture's launch is queued), a picture of another size with inter
// macroblocks is that stream's H263MI_ERR_PICTURE_FORMAT_INVALID (the reference indexes the new planes with the old
// strides there, gather.rs:150,183).
//
// Round 5: SLOTS BY MEMBERSHIP.  A class owns as many slots as it has (had) members -- a power of two, doubled when a
// stream joins a full class, halved when three quarters of it stand empty -- and a map stream <-> slot; its frame store
// is 2 x slots frames and its launches cover its slots, not the streams of the whole set (round 4: every class held two
// frames for every stream of the set: 63 QCIF streams and one 1080p stream cost 64 x 2 x 3.1 MB for the one).  Growing
// and shrinking move the members' frames into a new batch (device-to-device copies behind a sync of the old one: rare).
// =========================================================================================
#include "batch.h"

#include <algorithm>
#include <cstring>
#include <new>

using namespace h263mi;

struct h263mi_mixed {
    uint32_t n = 0;
    h263mi_backend_cfg cfg{};
    struct SizeClass {
        uint32_t w = 0, h = 0;
        h263mi_batch *b = nullptr;              // b->n slots
        std::vector<int> stream_of_slot;        // -1 = free
        bool submitted = false;                 // took part in the current call
        uint32_t members() const
        {
            uint32_t k = 0;
            for (int s : stream_of_slot) k += s >= 0 ? 1u : 0u;
            return k;
        }
    };
    std::vector<SizeClass> classes;
    struct Want { uint32_t w, h; };             // the size a stream's picture of the current call has
    std::vector<int> cls;                       // per stream: index into `classes`, -1 = no picture yet
    std::vector<int> slot;                      // per stream: its slot in that class
    std::vector<int> late_rc;                   // per stream: a device error found while its class was rebuilt, reported by the next sync
    std::vector<bits::ParserContext> parser_ctx;
    std::vector<bits::ParsedPicture> parsed;
    HostPlacement placement;                    // the NUMA node of the set's device (worker_pool.h)
    std::unique_ptr<WorkerPool> pool;
    // What the frame stores of all classes together may take (0 = no limit).  The sizes come out of untrusted bitstreams:
    // without a limit one hostile key frame of 16 384 x 16 384 asks for 800 MB per stream that sends one.
    uint64_t limit_bytes = 0;
    ~h263mi_mixed()
    {
        pool.reset();                           // the host threads first
        for (SizeClass &c : classes) delete c.b;
    }
    static uint64_t slots_bytes(uint32_t w, uint32_t h, uint32_t slots) { return 2ull * slots * make_layout(w, h).frame_bytes; }
    uint64_t store_bytes() const
    {
        uint64_t sum = 0;
        for (const SizeClass &c : classes)
            if (c.b) sum += slots_bytes(c.w, c.h, c.b->n);
        return sum;
    }
    WorkerPool &workers(unsigned want)
    {
        if (!pool || pool->size() < want) pool.reset(new WorkerPool(want - 1, &placement));
        return *pool;
    }
    static uint32_t pow2_at_least(uint32_t v)
    {
        uint32_t p = 1;
        while (p < v) p <<= 1;
        return p;
    }
    int find_class(uint32_t w, uint32_t h) const
    {
        for (size_t k = 0; k < classes.size(); k++)
            if (classes[k].b && classes[k].w == w && classes[k].h == h) return (int)k;
        return -1;
    }
    // classes nobody belongs to and nobody is about to join give up their batch: the dimensions come out of untrusted
    // bitstreams, and a stream that changes its size with every key frame must not make the set grow
    // (target[i] == -2: stream i joins a class of size want[i] that has not been resolved yet -- an existing class of that
    // size is wanted too: dropping it now would free its frame store and staging only to make them again a moment later)
    void drop_empty_classes(const std::vector<int> &target, const std::vector<Want> &want)
    {
        for (size_t k = 0; k < classes.size(); k++) {
            if (!classes[k].b || classes[k].members()) continue;
            bool wanted = false;
            for (uint32_t i = 0; i < n && !wanted; i++)
                wanted = target[i] == (int)k || (target[i] == -2 && want[i].w == classes[k].w && want[i].h == classes[k].h);
            if (wanted) continue;
            // (a rendering of a picture a departed stream left behind is delivered first; the destructor waits for it)
            if (classes[k].b->pending.valid) (void)classes[k].b->flush_pending();
            delete classes[k].b;
            classes[k] = SizeClass();
        }
    }
    // a batch of `slots` slots for class k in the place of the one it has (or of none): the members move over, slot by slot
    // from 0 on.  The old batch is synced first (a device error found there is kept in late_rc for h263mi_mixed_sync).
    int rebuild_class(size_t k, uint32_t slots)
    {
        SizeClass &c = classes[k];
        h263mi_batch *ob = c.b;
        const uint64_t others = store_bytes() - (ob ? slots_bytes(c.w, c.h, ob->n) : 0);
        // (old and new exist side by side for the length of the copies)
        if (limit_bytes && others + slots_bytes(c.w, c.h, slots) + (ob ? slots_bytes(c.w, c.h, ob->n) : 0) > limit_bytes)
            return H263MI_ERR_OUT_OF_MEMORY;
        h263mi_batch *nb = nullptr;
        RC_TRY(batch_create(slots, c.w, c.h, &cfg, &nb));
        std::vector<int> moved(slots, -1);
        if (ob) {
            std::vector<int> rcs(ob->n, H263MI_OK);
            const int src = ob->sync(rcs.data());           // (delivers a deferred rendering too)
            if (src != H263MI_OK && src != H263MI_ERR_UNCODED_IFRAME_BLOCKS && src != H263MI_ERR_INVALID_ARGUMENT) {
                delete nb;
                return src;
            }
            uint32_t at = 0;
            for (uint32_t s = 0; s < ob->n; s++) {
                const int i = c.stream_of_slot[s];
                if (i < 0) continue;
                if (rcs[s] != H263MI_OK && late_rc[i] == H263MI_OK) late_rc[i] = rcs[s];
                for (int set = 0; set < 2; set++) {
                    const hipError_t e = fault_now() ? hipErrorOutOfMemory
                        : hipMemcpyAsync(nb->frames[set] + (size_t)at * nb->L.frame_bytes, ob->frames[set] + (size_t)s * ob->L.frame_bytes,
                                         ob->L.frame_bytes, hipMemcpyDeviceToDevice, ob->stream);
                    if (e != hipSuccess) {
                        (void)hipStreamSynchronize(ob->stream);
                        delete nb;
                        return map_hip_error(e);
                    }
                }
                nb->ss[at] = ob->ss[s];
                moved[at] = i;
                at++;
            }
            // the copies read the old frame store: it may go when they are done
            if (hipStreamSynchronize(ob->stream) != hipSuccess) {
                delete nb;
                return H263MI_ERR_HIP;
            }
            for (uint32_t s = 0; s < slots; s++) {
                const int i = moved[s];
                if (i < 0) continue;
                if (cls[i] == (int)k) slot[i] = (int)s;
            }
            delete ob;
        }
        c.b = nb;
        c.stream_of_slot = moved;
        return H263MI_OK;
    }
    uint32_t live_classes() const
    {
        uint32_t k = 0;
        for (const SizeClass &c : classes) k += c.b ? 1u : 0u;
        return k;
    }
};

extern "C" {

int h263mi_mixed_create(uint32_t n_streams, const h263mi_backend_cfg *cfg, h263mi_mixed **out)
{
    if (!out || !n_streams) return H263MI_ERR_INVALID_ARGUMENT;
    const int dev = cfg ? cfg->device_id : 0;
    RC_TRY(check_device(dev));
    h263mi_mixed *m = new (std::nothrow) h263mi_mixed();
    if (!m) return H263MI_ERR_OUT_OF_MEMORY;
    m->n = n_streams;
    if (cfg) m->cfg = *cfg;
    m->cfg.device_id = dev;
    m->cfg.flags &= ~H263MI_CFG_OVERLAP_POST;     // (one HIP stream carries the classes' launches back to back)
    {
        // default limit: half of the device's memory
        DeviceGuard g(dev);
        size_t free_b = 0, total_b = 0;
        if (g.ok && hipMemGetInfo(&free_b, &total_b) == hipSuccess) m->limit_bytes = total_b / 2;
        if (g.ok) m->placement = placement_of_device(dev);
    }
    m->cls.assign(n_streams, -1);
    m->slot.assign(n_streams, -1);
    m->late_rc.assign(n_streams, H263MI_OK);
    m->parser_ctx.assign(n_streams, bits::ParserContext());
    m->parsed.resize(n_streams);
    *out = m;
    return H263MI_OK;
}

void h263mi_mixed_destroy(h263mi_mixed *m) { delete m; }

int h263mi_mixed_stream_size(const h263mi_mixed *m, uint32_t stream, uint16_t *width, uint16_t *height)
{
    if (!m || stream >= m->n) return H263MI_ERR_INVALID_ARGUMENT;
    const int c = m->cls[stream];
    const bool has = c >= 0 && m->classes[c].b->ss[m->slot[stream]].cur >= 0;
    if (width) *width = has ? (uint16_t)m->classes[c].w : 0;
    if (height) *height = has ? (uint16_t)m->classes[c].h : 0;
    return has ? H263MI_OK : H263MI_ERR_NO_PICTURE;
}

uint32_t h263mi_mixed_size_classes(const h263mi_mixed *m) { return m ? m->live_classes() : 0; }

int h263mi_mixed_set_memory_limit(h263mi_mixed *m, uint64_t bytes)
{
    if (!m) return H263MI_ERR_INVALID_ARGUMENT;
    m->limit_bytes = bytes;
    return H263MI_OK;
}

uint64_t h263mi_mixed_frame_store_bytes(const h263mi_mixed *m) { return m ? m->store_bytes() : 0; }

int h263mi_mixed_decode_next_pictures(h263mi_mixed *m, uint32_t decoder_options, const uint8_t *const *data, const size_t *len,
                                      size_t *consumed, uint32_t n_threads, int *stream_rc, uint8_t strength,
                                      uint8_t *const *d_rgba, const size_t *rgba_capacity, h263mi_picture_desc *descs)
{
    return h263mi_mixed_decode_next_pictures_ps(m, decoder_options, data, len, consumed, n_threads, stream_rc, strength, nullptr, d_rgba,
                                                rgba_capacity, descs);
}

int h263mi_mixed_decode_next_pictures_ps(h263mi_mixed *m, uint32_t decoder_options, const uint8_t *const *data, const size_t *len,
                                         size_t *consumed, uint32_t n_threads, int *stream_rc, uint8_t strength,
                                         const uint8_t *strengths, uint8_t *const *d_rgba, const size_t *rgba_capacity,
                                         h263mi_picture_desc *descs)
{
    if (!m || !data || !len || !stream_rc || (d_rgba && !rgba_capacity)) return H263MI_ERR_INVALID_ARGUMENT;
    const uint32_t n = m->n;
    // the strength of every stream's picture: the caller's (one for all, or one per stream of the set), or what the picture's
    // own header asks for (filled in per class below, once the headers have been parsed)
    h263mi_batch::Strengths set_strength;
    RC_TRY(make_strengths(strength, strengths, n, /*from_header_allowed=*/true, set_strength));
    const bool from_header = !strengths && strength == H263MI_STRENGTH_FROM_HEADER;
    for (uint32_t i = 0; i < n; i++)
        if (!data[i] && len[i]) return H263MI_ERR_INVALID_ARGUMENT;
    DeviceGuard g(m->cfg.device_id);
    if (!g.ok) return H263MI_ERR_NO_DEVICE;
    // ---- the serial half of decode_next_picture (state.rs:143-427) per stream, on the host threads
    static const bool mixed_sparse = !(getenv("H263MI_SPARSE_RECORDS") && getenv("H263MI_SPARSE_RECORDS")[0] == '0');
    std::vector<int> rcs(n, H263MI_OK);
    const HostThreadPlan plan = host_thread_plan(n, n_threads);
    const uint32_t n_thr = plan.threads;
    StreamDeal deal(n);
    auto work = [&](unsigned t) {
        deal.run(t, n_thr, [&](uint32_t i) {
            if (!data[i]) return;
            bits::ParsedPicture &pic = m->parsed[i];
            pic.want_dense = false;
            pic.size_fits = &picture_size_fits;
            pic.sparse_records = mixed_sparse;   // records for the coded macroblocks only (see batch_submit_host)
            pic.mbs_ext = nullptr;               // (the class -- and with it the staging slot -- is known after the header)
            pic.mbs_ext_cap = 0;
            rcs[i] = bits::parse_picture(data[i], len[i], decoder_options, &m->parser_ctx[i], pic);
        });
    };
    if (n_thr == 1) work(0);
    else m->workers(n_thr).run(n_thr, work, plan.spin_us);

    // ---- which size each picture has; what must be refused before anything is queued
    std::vector<int> target(n, -1);
    typedef h263mi_mixed::Want Want;
    std::vector<Want> want(n, Want{0, 0});
    for (uint32_t i = 0; i < n; i++) {
        if (consumed) consumed[i] = 0;
        stream_rc[i] = rcs[i];
        if (!data[i] || rcs[i] != H263MI_OK) continue;
        const bits::ParsedPicture &pic = m->parsed[i];
        const uint32_t w = pic.desc.width, h = pic.desc.height;
        int rc = H263MI_OK;
        if (!w || !h || !layout_fits(w, h)) rc = H263MI_ERR_PICTURE_FORMAT_INVALID;
        const bool any_inter = pic.any_inter;       // (the parser's: inter types, macroblocks not coded, macroblocks not reached)
        const int c_old = m->cls[i];
        const bool same = c_old >= 0 && m->classes[c_old].w == w && m->classes[c_old].h == h;
        const h263mi_batch::StreamState *st_old = c_old >= 0 ? &m->classes[c_old].b->ss[m->slot[i]] : nullptr;
        const bool has_ref = st_old && st_old->has_ref && st_old->cur >= 0;
        if (rc == H263MI_OK && any_inter && !has_ref) rc = H263MI_ERR_UNCODED_IFRAME_BLOCKS;          // gather.rs:149
        if (rc == H263MI_OK && any_inter && !same) rc = H263MI_ERR_PICTURE_FORMAT_INVALID;           // (see the head of this section)
        if (rc == H263MI_OK && d_rgba && d_rgba[i] && rgba_capacity[i] < (size_t)w * h * 4) rc = H263MI_ERR_INVALID_ARGUMENT;
        if (rc == H263MI_OK) {
            want[i] = Want{w, h};
            target[i] = same ? c_old : -2;       // -2: joins a class of that size (existing or new), resolved below
        }
        stream_rc[i] = rc;
    }
    // ---- the joiners of every size: a class with room for them (made, grown, or -- three quarters empty -- shrunk)
    std::vector<int> new_slot(n, -1);            // the slot a joiner is about to take in its target class
    for (uint32_t i = 0; i < n; i++) {
        if (target[i] != -2) continue;
        const uint32_t w = want[i].w, h = want[i].h;
        std::vector<uint32_t> joiners;
        for (uint32_t j = i; j < n; j++)
            if (target[j] == -2 && want[j].w == w && want[j].h == h) joiners.push_back(j);
        int k = m->find_class(w, h);
        // (the picture of a stream that moves away lives in its old slot until the new one's launch is queued: stayers,
        // leavers and joiners all need a slot during this call)
        const uint32_t have = k >= 0 ? m->classes[k].members() : 0u;
        const uint32_t need = have + (uint32_t)joiners.size();
        int rc = H263MI_OK;
        if (k < 0) {
            m->drop_empty_classes(target, want);
            int place = -1;
            for (size_t q = 0; q < m->classes.size(); q++)
                if (!m->classes[q].b) { place = (int)q; break; }
            if (place < 0) { m->classes.push_back(h263mi_mixed::SizeClass()); place = (int)m->classes.size() - 1; }
            m->classes[place].w = w;
            m->classes[place].h = h;
            rc = m->rebuild_class((size_t)place, h263mi_mixed::pow2_at_least(need));
            if (rc != H263MI_OK) m->classes[place] = h263mi_mixed::SizeClass();
            k = place;
        } else if (need > m->classes[k].b->n) {
            rc = m->rebuild_class((size_t)k, h263mi_mixed::pow2_at_least(need));
        }
        for (uint32_t j : joiners) {
            if (rc != H263MI_OK) { target[j] = -1; stream_rc[j] = rc; continue; }
            target[j] = k;
        }
        if (rc != H263MI_OK) continue;
        h263mi_mixed::SizeClass &c = m->classes[k];
        uint32_t s = 0;
        for (uint32_t j : joiners) {
            while (s < c.b->n && c.stream_of_slot[s] >= 0) s++;
            new_slot[j] = (int)s;
            c.stream_of_slot[s] = (int)j;        // (tentative: given back below when the class's launch does not happen)
            s++;
        }
    }
    // (a class that stands three quarters empty gives the room back in h263mi_mixed_sync -- where everything has been waited
    // for anyway -- not here: sizes come out of untrusted bitstreams, and streams that alternate between two sizes could
    // otherwise force a sync, device-to-device copies and an allocation out of every decode call)
    for (h263mi_mixed::SizeClass &c : m->classes) c.submitted = false;

    // ---- one launch per class that has pictures
    // (a class whose RENDERING fails does not stop the others: every class that has pictures is decoded, and the first
    // rendering error is what the call returns at the end -- no stream is left with H263MI_OK and no decoded picture)
    int call_rc = H263MI_OK;
    static const uint32_t kNoEvents[1] = {0};
    for (size_t k = 0; k < m->classes.size(); k++) {
        h263mi_mixed::SizeClass &c = m->classes[k];
        h263mi_batch *b = c.b;
        if (!b) continue;                        // (a class that was given up)
        const uint32_t slots = b->n;
        std::vector<const h263mi_mb_record *> mbs(slots, nullptr);
        std::vector<const uint32_t *> first(slots, kNoEvents), events(slots, nullptr), gidx(slots, nullptr);
        std::vector<uint32_t> n_mbs(slots, 0), n_blocks(slots, 0), n_events(slots, 0);
        std::vector<uint8_t> types(slots, H263MI_PICTURE_P), was_active(slots, 0);
        std::vector<uint8_t *> out_ptrs(slots, nullptr);
        h263mi_batch::Strengths cst;             // ... of this class's slots
        if (from_header || strengths) cst.per_stream.assign(slots, 0);
        else cst.uniform = set_strength.uniform;
        uint32_t members = 0;
        bool any_out = false;
        for (uint32_t s = 0; s < slots; s++) {
            const int i = c.stream_of_slot[s];
            was_active[s] = b->ss[s].active;
            b->ss[s].active = false;
            if (i < 0 || target[i] != (int)k) continue;
            const bool joins = m->cls[i] != (int)k;
            if (joins ? new_slot[i] != (int)s : m->slot[i] != (int)s) continue;
            const bits::ParsedPicture &pic = m->parsed[i];
            members++;
            mbs[s] = pic.records();
            n_mbs[s] = (uint32_t)pic.n_records();
            first[s] = pic.block_first_event.data();
            events[s] = pic.events.data();
            gidx[s] = pic.group_index.data();
            n_blocks[s] = (uint32_t)pic.n_coded_blocks;
            n_events[s] = (uint32_t)pic.events.size();
            types[s] = pic.desc.picture_type;
            if (d_rgba && d_rgba[i]) { out_ptrs[s] = d_rgba[i]; any_out = true; }
            if (!cst.per_stream.empty()) cst.per_stream[s] = from_header ? strength_from_header(pic.desc) : set_strength.of((uint32_t)i);
            // a stream that arrives from another class starts afresh here (it brings an I picture)
            if (joins) b->ss[s] = h263mi_batch::StreamState();
            b->ss[s].active = true;
        }
        int rc = H263MI_OK;
        if (members) {
            const bool deferred = b->pipeline_post && any_out;
            rc = batch_submit_host(b, H263MI_PICTURE_P, mbs.data(), n_mbs.data(), nullptr, n_blocks.data(), first.data(), events.data(),
                                   n_events.data(), /*from_parser=*/true, n_thr, types.data(), deferred, mixed_sparse ? gidx.data() : nullptr);
            if (rc == H263MI_OK) {
                c.submitted = true;
                // the pictures are decoded: the streams move to this class, their parser state moves on (state.rs:464-483)
                for (uint32_t i = 0; i < n; i++) {
                    if (target[i] != (int)k) continue;
                    const int c_old = m->cls[i];
                    if (c_old != (int)k) {
                        if (c_old >= 0) {
                            // the slot the stream leaves: given up now, not earlier (its last picture lived there)
                            h263mi_mixed::SizeClass &oc = m->classes[c_old];
                            const int os = m->slot[i];
                            const bool a = oc.b->ss[os].active;
                            oc.b->ss[os] = h263mi_batch::StreamState();
                            oc.b->ss[os].active = a;
                            oc.stream_of_slot[os] = -1;
                            // (a rendering of the old picture that is still pending there is delivered all the same: it names
                            // the frame set, and the frames themselves are not touched)
                        }
                        m->cls[i] = (int)k;
                        m->slot[i] = new_slot[i];
                    }
                    m->parser_ctx[i] = m->parsed[i].next;
                    if (consumed) consumed[i] = m->parsed[i].bits_consumed / 8;
                    if (descs) descs[i] = m->parsed[i].desc;
                }
                int render_rc = H263MI_OK;
                if (deferred) render_rc = b->note_pending(cst, nullptr, nullptr, out_ptrs.data());
                else if (any_out) render_rc = b->render(cst, nullptr, nullptr, /*only_active=*/true, out_ptrs.data());
                if (render_rc != H263MI_OK && call_rc == H263MI_OK) call_rc = render_rc;
            } else {
                // the class's launch did not happen: its members keep their state and get the error, the joiners their old place
                for (uint32_t i = 0; i < n; i++) {
                    if (target[i] != (int)k) continue;
                    stream_rc[i] = rc;
                    if (m->cls[i] != (int)k && new_slot[i] >= 0) {
                        b->ss[new_slot[i]] = h263mi_batch::StreamState();
                        c.stream_of_slot[new_slot[i]] = -1;
                    }
                }
            }
        }
        for (uint32_t s = 0; s < slots; s++) b->ss[s].active = was_active[s] != 0;
    }
    // a class that is waiting to render its previous pictures and had nothing to decode in this call renders them now
    // (on a pipelined class the rendering rides in the NEXT launch of that class: without one it would wait for the sync)
    for (h263mi_mixed::SizeClass &c : m->classes)
        if (c.b && !c.submitted && c.b->pending.valid) {
            const int rc = c.b->flush_pending();
            if (rc != H263MI_OK && call_rc == H263MI_OK) call_rc = rc;
        }
    return call_rc;
}

int h263mi_mixed_sync(h263mi_mixed *m, int *stream_rc)
{
    if (!m) return H263MI_ERR_INVALID_ARGUMENT;
    DeviceGuard g(m->cfg.device_id);
    if (!g.ok) return H263MI_ERR_NO_DEVICE;
    int first_error = H263MI_OK;
    for (uint32_t i = 0; i < m->n; i++) {
        // (device errors found while a class was rebuilt: delivered here, once)
        if (stream_rc) stream_rc[i] = m->late_rc[i];
        if (m->late_rc[i] != H263MI_OK && first_error == H263MI_OK) first_error = m->late_rc[i];
        m->late_rc[i] = H263MI_OK;
    }
    for (size_t k = 0; k < m->classes.size(); k++) {
        h263mi_mixed::SizeClass &c = m->classes[k];
        if (!c.b) continue;
        std::vector<int> rcs(c.b->n, H263MI_OK);
        const int rc = c.b->sync(rcs.data());
        if (rc != H263MI_OK && first_error == H263MI_OK) first_error = rc;
        for (uint32_t s = 0; s < c.b->n; s++) {
            const int i = c.stream_of_slot[s];
            if (stream_rc && i >= 0 && rcs[s] != H263MI_OK && stream_rc[i] == H263MI_OK) stream_rc[i] = rcs[s];
        }
    }
    // a class that stands three quarters empty gives the room back: everything it had queued has just been waited for, so the
    // rebuild's own sync is free and its device-to-device copies have the device to themselves.  (A failure leaves the class
    // as it was -- roomy, but whole.)
    for (size_t k = 0; k < m->classes.size(); k++) {
        h263mi_mixed::SizeClass &c = m->classes[k];
        if (!c.b || c.b->n < 8) continue;
        const uint32_t mem = c.members();
        if (mem && mem * 4 <= c.b->n) (void)m->rebuild_class(k, h263mi_mixed::pow2_at_least(mem));
    }
    return first_error;
}

int h263mi_mixed_copy_yuv(h263mi_mixed *m, uint32_t stream, uint8_t *y, uint8_t *cb, uint8_t *cr)
{
    if (!m || stream >= m->n) return H263MI_ERR_INVALID_ARGUMENT;
    if (m->cls[stream] < 0) return H263MI_ERR_NO_PICTURE;
    DeviceGuard g(m->cfg.device_id);
    return m->classes[m->cls[stream]].b->copy_yuv((uint32_t)m->slot[stream], y, cb, cr);
}

int h263mi_mixed_reset_stream(h263mi_mixed *m, uint32_t stream)
{
    if (!m || stream >= m->n) return H263MI_ERR_INVALID_ARGUMENT;
    DeviceGuard g(m->cfg.device_id);
    if (!g.ok) return H263MI_ERR_NO_DEVICE;
    int rc = H263MI_OK;
    if (m->cls[stream] >= 0) {
        h263mi_mixed::SizeClass &c = m->classes[m->cls[stream]];
        rc = c.b->forget_stream((uint32_t)m->slot[stream]);
        c.stream_of_slot[m->slot[stream]] = -1;          // the stream leaves its class: the slot is free
    }
    m->cls[stream] = -1;
    m->slot[stream] = -1;
    m->late_rc[stream] = H263MI_OK;
    m->parser_ctx[stream] = bits::ParserContext();
    return rc;
}

}  // extern "C"
