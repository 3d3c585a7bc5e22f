// batch.cpp -- h263mi_batch: the device-resident frame store of N streams, the reference bookkeeping of
// state.rs:464-483 per stream, submit / render / sync, launch timing, and the entry points that take DEVICE records
// (h263mi_batch_submit / _decode / _decode_events / _render_rgba / _sync ...).  Compiled with hipcc; every compute path
// launches the gfx950 kernels of kernels.hip -- there is no CPU fallback.
#include "batch.h"

#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>

using namespace h263mi;

// =========================================================================================
// frame store and per-stream state
// =========================================================================================
int h263mi_batch::alloc(uint32_t n_streams, uint32_t w, uint32_t h)
{
    n = n_streams;
    L = make_layout(w, h);
    // both frame sets in one allocation
    {
        const size_t set_bytes = (size_t)n * L.frame_bytes;
        // H263MI_EXP_FRAME_SKEW (experiment, a multiple of 16): the whole frame store starts that many bytes past a
        // 64-byte line, so that no row of any plane is line-aligned (profiles/README.md r03_zz: aligned RGBA runs
        // are 10 % slower than runs that start 16 bytes into a line -- the same for the planes?)
        const char *skew_env = getenv("H263MI_EXP_FRAME_SKEW");
        frame_skew = skew_env ? ((size_t)atoi(skew_env) & 0xff0u) : 0;
        HIP_TRY(hipMalloc((void **)&frames[0], 2 * set_bytes + 4096));
        frames[0] += frame_skew;
        frames[1] = frames[0] + set_bytes;
        if (getenv("H263MI_TRACE_ALLOC"))
            fprintf(stderr, "h263mi frame store: %p .. +%zu\n", (void *)frames[0], 2 * set_bytes);
        HIP_TRY(hipMemsetAsync(frames[0], 0, 2 * set_bytes, stream));
    }
    if (!d_status) {
        HIP_TRY(hipMalloc((void **)&d_status, (size_t)n * sizeof(uint32_t)));
        HIP_TRY(hipHostMalloc((void **)&h_status, (size_t)n * sizeof(uint32_t), hipHostMallocDefault));
        HIP_TRY(hipMalloc((void **)&d_state, (size_t)n * kStateSlots * sizeof(uint32_t)));
        HIP_TRY(hipHostMalloc((void **)&h_state, (size_t)n * kStateSlots * sizeof(uint32_t), hipHostMallocDefault));
        for (hipEvent_t &e : state_copied) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    HIP_TRY(hipMemsetAsync(d_status, 0, (size_t)n * sizeof(uint32_t), stream));
    ss.assign(n, StreamState());
    pending.set.assign(n, -1);
    return H263MI_OK;
}

bool h263mi_batch::any_picture() const
{
    for (const StreamState &t : ss)
        if (t.cur >= 0) return true;
    return false;
}

bool h263mi_batch::uniform() const
{
    for (const StreamState &t : ss)
        if (!t.active || t.cur != ss[0].cur || t.has_ref != ss[0].has_ref) return false;
    return true;
}

bool h263mi_batch::pending_uniform() const
{
    for (int8_t v : pending.set)
        if (v != pending.set[0]) return false;
    return true;
}

int h263mi_batch::push_stream_words(const std::vector<uint32_t> &words, const uint32_t **d_out, hipStream_t on)
{
    const unsigned slot = state_slot++ % kStateSlots;
    HIP_TRY(hipEventSynchronize(state_copied[slot]));         // (its previous copy has left the host buffer)
    uint32_t *h = h_state + (size_t)slot * n, *d = d_state + (size_t)slot * n;
    memcpy(h, words.data(), (size_t)n * sizeof(uint32_t));
    RC_TRY(time_close());                                      // a copy is not part of any kernel's time
    HIP_TRY(hipMemcpyAsync(d, h, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, on));
    HIP_TRY(hipEventRecord(state_copied[slot], on));
    *d_out = d;
    return H263MI_OK;
}

int h263mi_batch::make_ptr_ring()
{
    HIP_TRY(hipHostMalloc((void **)&h_ptrs, (size_t)n * kPtrSlots * sizeof(uint8_t *), hipHostMallocDefault));
    HIP_TRY(hipMalloc((void **)&d_ptrs, (size_t)n * kPtrSlots * sizeof(uint8_t *)));
    for (hipEvent_t &e : ptrs_copied) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return H263MI_OK;
}

void h263mi_batch::release_ptr_ring()
{
    if (d_ptrs) (void)hipFree(d_ptrs);
    if (h_ptrs) (void)hipHostFree(h_ptrs);
    d_ptrs = nullptr;
    h_ptrs = nullptr;
    for (hipEvent_t &e : ptrs_copied) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    ptrs_ready = false;
}

int h263mi_batch::push_rgba_ptrs(uint8_t *const *host_ptrs, uint8_t *const **d_out, hipStream_t on)
{
    if (!ptrs_ready) {
        // all or nothing: a failure half-way (the host block is there, the device block or an event is not) frees what
        // was made, so that the next call starts over instead of synchronising on an event that does not exist
        const int rc = make_ptr_ring();
        if (rc != H263MI_OK) {
            release_ptr_ring();
            return rc;
        }
        ptrs_ready = true;
    }
    const unsigned slot = ptrs_slot++ % kPtrSlots;
    HIP_TRY(hipEventSynchronize(ptrs_copied[slot]));
    uint8_t **h = h_ptrs + (size_t)slot * n, **d = d_ptrs + (size_t)slot * n;
    memcpy(h, host_ptrs, (size_t)n * sizeof(uint8_t *));
    RC_TRY(time_close());
    HIP_TRY(hipMemcpyAsync(d, h, (size_t)n * sizeof(uint8_t *), hipMemcpyHostToDevice, on));
    HIP_TRY(hipEventRecord(ptrs_copied[slot], on));
    *d_out = d;
    return H263MI_OK;
}

int h263mi_batch::forget_pictures()
{
    const int rc = flush_pending();        // what was asked to be rendered still is
    for (StreamState &t : ss) {
        const bool active = t.active;
        t = StreamState();
        t.active = active;
    }
    parser_ctx.clear();
    return rc;
}

int h263mi_batch::forget_stream(uint32_t i)
{
    RC_TRY(flush_pending());
    const bool active = ss[i].active;
    ss[i] = StreamState();
    ss[i].active = active;
    if (i < parser_ctx.size()) parser_ctx[i] = bits::ParserContext();
    return H263MI_OK;
}

void h263mi_batch::release_frames()
{
    if (frames[0]) (void)hipFree(frames[0] - frame_skew);         // (one allocation holds both sets)
    frames[0] = frames[1] = nullptr;
}

h263mi_batch::~h263mi_batch()
{
    DeviceGuard g(device);
    (void)hipStreamSynchronize(stream);
    if (trace_host && host_calls)
        fprintf(stderr, "h263mi batch (%u streams): %u host submits; ms per call: parse %.3f, wait for slot %.3f, pack %.3f, "
                        "enqueue %.3f (copies %.3f, launch %.3f)\n", n, host_calls, host_ms[0] / host_calls, host_ms[1] / host_calls,
                host_ms[2] / host_calls, host_ms[3] / host_calls, host_ms[4] / host_calls, host_ms[5] / host_calls);
    pool.reset();                               // the host threads first: nothing of theirs may outlive the staging memory
    release_frames();
    if (post_stream) {
        (void)hipStreamSynchronize(post_stream);
        (void)hipStreamDestroy(post_stream);
    }
    if (ev_recon_done) (void)hipEventDestroy(ev_recon_done);
    for (hipEvent_t e : ev_post_done)
        if (e) (void)hipEventDestroy(e);
    if (d_status) (void)hipFree(d_status);
    if (h_status) (void)hipHostFree(h_status);
    if (d_state) (void)hipFree(d_state);
    if (h_state) (void)hipHostFree(h_state);
    release_ptr_ring();
    for (hipEvent_t e : state_copied)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
    release_staging();
}

// =========================================================================================
// launch timing (h263mi_batch_timing_begin / _end).  Consecutive launches of the same kernel form a CHAIN that is
// bracketed by ONE pair of events -- begin in front of the first launch, end behind the last -- and the chain's time
// is shared out over its launches: an event pair around every single launch put a 6 us bubble between two launches
// (2 % of a frame index of the 64-stream bench; tools/probes/timing_overhead.py).  A chain ends where the kernel
// changes and in front of anything else that is queued on the stream (copies, the status read of sync), so only
// launches -- and the gaps between back-to-back launches -- are inside.
// =========================================================================================
int h263mi_batch::time_close()
{
    if (chain_kernel < 0) return H263MI_OK;
    const int k = chain_kernel;
    chain_kernel = -1;
    HIP_TRY(hipEventRecord(ev_pool[ev_used + 1], stream_of(k)));
    ev_ranges.push_back(TimedChain{ev_used, k, chain_launches});
    ev_used += 2;
    return H263MI_OK;
}

int h263mi_batch::time_begin(int kernel_id)
{
    if (!timing) return H263MI_OK;
    if (chain_kernel == kernel_id) {
        chain_launches++;
        return H263MI_OK;
    }
    RC_TRY(time_close());
    if (ev_used + 2 > ev_pool.size()) {
        for (int i = 0; i < 2; i++) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            ev_pool.push_back(e);
        }
    }
    HIP_TRY(hipEventRecord(ev_pool[ev_used], stream_of(kernel_id)));
    chain_kernel = kernel_id;
    chain_launches = 1;
    return H263MI_OK;
}

// =========================================================================================
// the work
// =========================================================================================
int h263mi_batch::submit(uint8_t picture_type, const MbRecord *d_mbs, const h263mi_coeff_source &src, bool with_post, const uint8_t *types)
{
    if (!with_post) RC_TRY(flush_pending());
    ReconArgs a{};
    a.L = L;
    a.mbs = d_mbs;
    a.coeffs = src.coeffs;
    a.block_first_event = src.first_event;
    a.events = src.events;
    a.n_events = src.n_events ? src.n_events : 0xffffffffu;
    a.mb_group_index = src.group_index;
    a.mb_base = src.mb_base;
    a.groups_per_picture = recon_tiles_x(L) * L.mbh;
    a.coeff_base = src.coeff_base;
    a.status = d_status;
    a.coeff_pool_blocks = src.pool_blocks;
    a.coeff_checked = src.checked ? 1u : 0u;
    a.n_pictures = n;
    a.mbs_per_picture = L.mbw * L.mbh;
    a.tiles_x = recon_tiles_x(L);
    a.tiles_y = recon_tiles_y(L);
    a.frame_set[0] = frames[0];
    a.frame_set[1] = frames[1];
    PostArgs pa{};
    if (with_post) pa = post_args(0, pending.strength.of(0), pending.rgba, pending.planes);
    if (with_post) pa.rgba_ptrs = pending.rgba_ptrs;       // (read in the per-stream branch of the kernel only)
    // one strength for every picture of the launch, or one per stream (then it travels in the streams' words)
    const bool all_same = uniform() && (!with_post || (pending_uniform() && pending.set[0] >= 0 && !pending.rgba_ptrs &&
                                                       pending.strength.same_for_all()));
    int out0 = 0;
    std::vector<uint32_t> words;                 // one STREAM_* word per stream when they differ (empty: they do not)
    const bool words_inline = n <= STREAM_WORDS_INLINE;       // ... which then travel in the launch's kernel arguments
    if (all_same) {
        const int cur = ss[0].cur;
        out0 = cur < 0 ? 0 : (cur ^ 1);
        // get_reference_picture() hands out the LAST picture whenever a reference exists (state.rs:72-78)
        a.ref = frames[cur < 0 ? 1 : cur];
        a.cur = frames[out0];
        a.has_ref = (ss[0].has_ref && cur >= 0) ? 1u : 0u;
        if (with_post) pa.frames = frames[pending.set[0]];
    } else {
        words.resize(n);
        for (uint32_t i = 0; i < n; i++) {
            const StreamState &t = ss[i];
            uint32_t w = (t.cur != 0 ? STREAM_REF_SET1 : 0u) | ((t.has_ref && t.cur >= 0) ? STREAM_HAS_REF : 0u) |
                         (t.active ? 0u : STREAM_RECON_SKIP);
            if (!with_post || pending.set[i] < 0) w |= STREAM_POST_SKIP;
            else w |= (pending.set[i] == 1 ? STREAM_POST_SET1 : 0u) | ((uint32_t)pending.strength.of(i) << STREAM_STRENGTH_SHIFT);
            words[i] = w;
        }
        // (a batch of more than STREAM_WORDS_INLINE streams: the words go to device memory, a small copy in front of the launch)
        const uint32_t *d_words = nullptr;
        if (!words_inline) RC_TRY(push_stream_words(words, &d_words, stream));
        a.stream_state = d_words;
        a.ref = frames[0];                   // (never used with per-stream words; never null)
        a.cur = frames[1];
        if (with_post) {
            pa.stream_state = d_words;
            pa.frame_set[0] = frames[0];
            pa.frame_set[1] = frames[1];
            pa.frames = frames[0];
        }
    }
    // the set being overwritten was last read by the post-processing of the picture before the last one
    if (overlap_post) {
        HIP_TRY(hipStreamWaitEvent(stream, ev_post_done[out0], 0));
        if (!all_same) HIP_TRY(hipStreamWaitEvent(stream, ev_post_done[out0 ^ 1], 0));     // (streams write either set)
    }
    if (with_post) {
        RC_TRY(time_begin(2));
        const hipError_t e = launch_frame(a, pa, stream, (frame_launches++ & 1u) != 0, !words.empty() && words_inline ? words.data() : nullptr);
        if (e != hipSuccess) {               // the deferred post-processing must not get lost with the failed launch
            (void)flush_pending();
            return map_hip_error(e);
        }
        pending.valid = false;
    } else {
        RC_TRY(time_begin(0));
        HIP_TRY(launch_recon(a, stream, !words.empty() && words_inline ? words.data() : nullptr));
    }
    if (overlap_post) HIP_TRY(hipEventRecord(ev_recon_done, stream));
    // reference bookkeeping, state.rs:464-483, per stream
    for (uint32_t i = 0; i < n; i++) {
        StreamState &t = ss[i];
        if (!t.active) continue;
        const uint8_t type = types ? types[i] : picture_type;
        t.unsynced++;
        if (type == H263MI_PICTURE_I) t.has_ref = false;
        t.cur = (int8_t)(t.cur < 0 ? 0 : (t.cur ^ 1));
        if (type != H263MI_PICTURE_DISPOSABLE_P) t.has_ref = true;
    }
    return H263MI_OK;
}

PostArgs h263mi_batch::post_args(int set, uint8_t strength, uint8_t *d_rgba, uint8_t *d_planes) const
{
    PostArgs a{};
    a.L = L;
    a.frames = frames[set];
    a.rgba = d_rgba;
    a.planes_out = d_planes;
    a.n_pictures = n;
    a.strength = strength;
    set_post_tiles(a);
    a.luma_only = 0;
    return a;
}

int h263mi_batch::launch_post_sets(const std::vector<int8_t> &sets, const Strengths &strength, uint8_t *d_rgba, uint8_t *d_planes,
                                   hipStream_t on, uint8_t *const *rgba_ptrs)
{
    bool same = rgba_ptrs == nullptr && strength.same_for_all(), any = false;
    for (int8_t v : sets) {
        same = same && v == sets[0];
        any = any || v >= 0;
    }
    if (!any) return H263MI_OK;
    PostArgs a = post_args(sets[0] >= 0 ? sets[0] : 0, strength.of(0), d_rgba, d_planes);
    a.rgba_ptrs = rgba_ptrs;
    std::vector<uint32_t> words;
    const bool words_inline = n <= STREAM_WORDS_INLINE;
    if (!same) {
        words.resize(n);
        for (uint32_t i = 0; i < n; i++)
            words[i] = STREAM_RECON_SKIP | (sets[i] < 0 ? STREAM_POST_SKIP : (sets[i] == 1 ? STREAM_POST_SET1 : 0u)) |
                       ((uint32_t)strength.of(i) << STREAM_STRENGTH_SHIFT);
        const uint32_t *d_words = nullptr;
        if (!words_inline) RC_TRY(push_stream_words(words, &d_words, on));
        a.stream_state = d_words;
        a.frame_set[0] = frames[0];
        a.frame_set[1] = frames[1];
    }
    RC_TRY(time_begin(1));
    HIP_TRY(launch_post(a, on, !words.empty() && words_inline ? words.data() : nullptr));
    return H263MI_OK;
}

int h263mi_batch::note_pending(const Strengths &strength, uint8_t *d_rgba, uint8_t *d_planes, uint8_t *const *host_ptrs)
{
    pending.valid = false;
    pending.rgba_ptrs = nullptr;
    if (host_ptrs) RC_TRY(push_rgba_ptrs(host_ptrs, &pending.rgba_ptrs, stream));
    pending.valid = d_rgba || d_planes || host_ptrs;
    pending.strength = strength;
    pending.rgba = d_rgba;
    pending.planes = d_planes;
    for (uint32_t i = 0; i < n; i++)
        pending.set[i] = (ss[i].active && (!host_ptrs || host_ptrs[i])) ? ss[i].cur : (int8_t)-1;
    return H263MI_OK;
}

int h263mi_batch::flush_pending()
{
    if (!pending.valid) return H263MI_OK;
    pending.valid = false;
    return launch_post_sets(pending.set, pending.strength, pending.rgba, pending.planes, stream, pending.rgba_ptrs);
}

int h263mi_batch::render(const Strengths &strength, uint8_t *d_rgba, uint8_t *d_planes, bool only_active, uint8_t *const *host_ptrs)
{
    if (!any_picture()) return H263MI_ERR_NO_PICTURE;
    RC_TRY(flush_pending());
    std::vector<int8_t> sets(n);
    bool reads[2] = {false, false};
    for (uint32_t i = 0; i < n; i++) {
        sets[i] = ((only_active && !ss[i].active) || (host_ptrs && !host_ptrs[i])) ? (int8_t)-1 : ss[i].cur;
        if (sets[i] >= 0) reads[sets[i]] = true;
    }
    if (overlap_post) HIP_TRY(hipStreamWaitEvent(post_stream, ev_recon_done, 0));
    uint8_t *const *d_out_ptrs = nullptr;
    if (host_ptrs) RC_TRY(push_rgba_ptrs(host_ptrs, &d_out_ptrs, stream_of(1)));
    RC_TRY(launch_post_sets(sets, strength, d_rgba, d_planes, stream_of(1), d_out_ptrs));
    // a later reconstruction may overwrite a frame set only when every post-processing that reads it is done: streams
    // that have drifted apart read both sets
    if (overlap_post)
        for (int k = 0; k < 2; k++)
            if (reads[k]) HIP_TRY(hipEventRecord(ev_post_done[k], post_stream));
    return H263MI_OK;
}

int h263mi_batch::sync(int *stream_rc)
{
    RC_TRY(flush_pending());
    RC_TRY(time_close());
    if (overlap_post) HIP_TRY(hipStreamSynchronize(post_stream));
    HIP_TRY(hipMemcpyAsync(h_status, d_status, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    int first_error = H263MI_OK;
    for (uint32_t i = 0; i < n; i++) {
        StreamState &t = ss[i];
        const uint32_t st = h_status[i];
        int rc = H263MI_OK;
        if (st) {
            // A picture the device rejected must not become the stream's last / reference picture.  One picture since
            // the last good sync: the previous frame set is intact, go back to it.  More than one: the set it lived
            // in has been overwritten by the ping-pong, so no picture survives (like a reset of the stream).
            if (t.unsynced <= 1) {
                t.cur = t.good_cur;
                t.has_ref = t.good_has_ref;
            } else {
                t.cur = -1;
                t.has_ref = false;
            }
            rc = (st & STATUS_INTER_WITHOUT_REFERENCE) ? H263MI_ERR_UNCODED_IFRAME_BLOCKS : H263MI_ERR_INVALID_ARGUMENT;
            if (first_error == H263MI_OK) first_error = rc;
        }
        t.good_cur = t.cur;
        t.good_has_ref = t.has_ref;
        t.unsynced = 0;
        if (stream_rc) stream_rc[i] = rc;
    }
    if (first_error != H263MI_OK) HIP_TRY(hipMemsetAsync(d_status, 0, (size_t)n * sizeof(uint32_t), stream));
    return first_error;
}

int h263mi_batch::copy_yuv(uint32_t s, uint8_t *y, uint8_t *cb, uint8_t *cr)
{
    if (s >= n) return H263MI_ERR_INVALID_ARGUMENT;
    if (ss[s].cur < 0) return H263MI_ERR_NO_PICTURE;
    RC_TRY(time_close());
    const uint8_t *f = frames[ss[s].cur] + (size_t)s * L.frame_bytes;
    // DecodedPicture planes are exact-size and tightly packed (picture.rs:39-58)
    if (y) HIP_TRY(hipMemcpy2DAsync(y, L.width, f, L.pitch_y, L.width, L.height, hipMemcpyDeviceToHost, stream));
    if (cb) HIP_TRY(hipMemcpy2DAsync(cb, L.cwidth, f + L.off_cb, L.pitch_c, L.cwidth, L.cheight, hipMemcpyDeviceToHost, stream));
    if (cr) HIP_TRY(hipMemcpy2DAsync(cr, L.cwidth, f + L.off_cr, L.pitch_c, L.cwidth, L.cheight, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return H263MI_OK;
}

namespace h263mi {

int make_strengths(uint8_t strength, const uint8_t *strengths, uint32_t n, bool from_header_allowed, h263mi_batch::Strengths &out)
{
    out = h263mi_batch::Strengths();
    if (strengths) {
        out.per_stream.assign(strengths, strengths + n);
        for (uint8_t v : out.per_stream)
            if (v > 12) return H263MI_ERR_INVALID_ARGUMENT;
        return H263MI_OK;
    }
    if (strength == H263MI_STRENGTH_FROM_HEADER) {
        if (!from_header_allowed) return H263MI_ERR_INVALID_ARGUMENT;
        out.per_stream.assign(n, 0);             // (filled in per picture by the entry that has parsed the headers)
        return H263MI_OK;
    }
    if (strength > 12) return H263MI_ERR_INVALID_ARGUMENT;
    out.uniform = strength;
    return H263MI_OK;
}

// Where the host side of device `dev` belongs: the PCI addresses of the visible devices -> worker_pool.cpp (sysfs)
HostPlacement placement_of_device(int dev)
{
    // looked up once per (device, ranks per node, switches): a mixed-size set makes a batch whenever a class is made or
    // rebuilt -- out of untrusted bitstreams -- and the look-up reads a few hundred sysfs files
    static std::mutex cache_mutex;
    static std::map<std::string, HostPlacement> cache;
    const uint32_t ranks = host_thread_plan(1, 0).ranks;
    std::string key = std::to_string(dev) + "|" + std::to_string(ranks);
    for (const char *name : {"H263MI_NUMA", "H263MI_NUMA_NODE", "H263MI_SYSFS_ROOT"}) {
        const char *v = getenv(name);
        key += std::string("|") + (v ? v : "");
    }
    {
        std::lock_guard<std::mutex> l(cache_mutex);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return HostPlacement();
    std::vector<std::string> ids((size_t)count);
    for (int d = 0; d < count; d++) {
        char id[64] = {0};
        if (hipDeviceGetPCIBusId(id, (int)sizeof id, d) == hipSuccess) ids[(size_t)d] = id;
        else (void)hipGetLastError();
    }
    const HostPlacement p = host_placement(ids, dev, ranks);
    std::lock_guard<std::mutex> l(cache_mutex);
    cache[key] = p;
    return p;
}

int batch_create(uint32_t n_streams, uint32_t w, uint32_t h, const h263mi_backend_cfg *cfg, h263mi_batch **out)
{
    if (!out || !n_streams || !w || !h) return H263MI_ERR_INVALID_ARGUMENT;
    if (!layout_fits(w, h)) return H263MI_ERR_PICTURE_FORMAT_INVALID;        // before anything is allocated
    const int dev = cfg ? cfg->device_id : 0;
    RC_TRY(check_device(dev));
    DeviceGuard g(dev);
    if (!g.ok) return H263MI_ERR_NO_DEVICE;
    h263mi_batch *b = new (std::nothrow) h263mi_batch();
    if (!b) return H263MI_ERR_OUT_OF_MEMORY;
    b->device = dev;
    b->stream = cfg ? (hipStream_t)cfg->stream : nullptr;
    int rc = H263MI_OK;
    b->pipeline_post = cfg && (cfg->flags & H263MI_CFG_PIPELINE_POST);
    b->trusted_arrays = cfg && (cfg->flags & H263MI_CFG_TRUSTED_ARRAYS);
    b->placement = placement_of_device(dev);
    if (cfg && (cfg->flags & H263MI_CFG_OVERLAP_POST) && !b->pipeline_post) {
        b->overlap_post = true;
        if (fault_now() || hipStreamCreateWithFlags(&b->post_stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&b->ev_recon_done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&b->ev_post_done[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&b->ev_post_done[1], hipEventDisableTiming) != hipSuccess)
            rc = H263MI_ERR_HIP;
    }
    if (rc == H263MI_OK) rc = b->alloc(n_streams, w, h);
    if (rc != H263MI_OK) {
        delete b;
        return rc;
    }
    *out = b;
    return H263MI_OK;
}

// The allocation a device pointer of the caller lies in bounds what may be read through it: `bytes` receives what is left
// of it from `p` on.  H263MI_ERR_INVALID_ARGUMENT when the runtime does not know the pointer (then the caller must say how
// large its arrays are, or vouch for them: H263MI_CFG_TRUSTED_ARRAYS).
static int bytes_behind(const void *p, size_t *bytes)
{
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (fault_now()) return H263MI_ERR_OUT_OF_MEMORY;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)const_cast<void *>(p)) != hipSuccess || !base) {
        (void)hipGetLastError();
        return H263MI_ERR_INVALID_ARGUMENT;
    }
    const size_t off = (size_t)((const uint8_t *)p - (const uint8_t *)base);
    if (off > size) return H263MI_ERR_INVALID_ARGUMENT;
    *bytes = size - off;
    return H263MI_OK;
}

// The caller's device arrays of h263mi_batch_submit / _decode / _decode_events -> what the waves may read (ABI 7: checked
// unless the batch was made with H263MI_CFG_TRUSTED_ARRAYS).  Counts the caller gave are taken as they are; what it did not
// say is bounded by the allocation the pointer lies in, so that no record, offset or event of its arrays -- whatever they
// hold -- can take a wave outside memory the caller owns.  The fixed-size arrays (records, bases) must fit theirs.
static int bound_device_arrays(const h263mi_batch *b, const h263mi_mb_record *d_mbs, uint64_t coeff_pool_blocks, uint64_t n_events,
                               h263mi_coeff_source &src)
{
    src.pool_blocks = coeff_pool_blocks;
    src.n_events = (uint32_t)n_events;
    if (b->trusted_arrays) {
        src.checked = coeff_pool_blocks != 0;
        return H263MI_OK;
    }
    size_t left = 0;
    RC_TRY(bytes_behind(d_mbs, &left));
    if (left < (size_t)b->n * b->L.mbw * b->L.mbh * sizeof(MbRecord)) return H263MI_ERR_INVALID_ARGUMENT;
    if (src.coeff_base) {
        RC_TRY(bytes_behind(src.coeff_base, &left));
        if (left < (size_t)b->n * sizeof(uint64_t)) return H263MI_ERR_INVALID_ARGUMENT;
    }
    if (src.events) {
        // coded block k of the pool reads first_event[k] and [k + 1]: the offsets array bounds the pool
        RC_TRY(bytes_behind(src.first_event, &left));
        const uint64_t blocks_max = left / sizeof(uint32_t) ? left / sizeof(uint32_t) - 1 : 0;
        if (!coeff_pool_blocks) src.pool_blocks = blocks_max;
        else if (coeff_pool_blocks > blocks_max) return H263MI_ERR_INVALID_ARGUMENT;
        RC_TRY(bytes_behind(src.events, &left));
        const uint64_t words_max = std::min<uint64_t>(left / sizeof(uint32_t), kMaxEventWords);
        if (!n_events) src.n_events = (uint32_t)words_max;
        else if (n_events > words_max) return H263MI_ERR_INVALID_ARGUMENT;
        if (!src.n_events || !src.pool_blocks) return H263MI_ERR_INVALID_ARGUMENT;
    } else if (src.coeffs) {
        RC_TRY(bytes_behind(src.coeffs, &left));
        const uint64_t blocks_max = left / 128;
        if (!coeff_pool_blocks) src.pool_blocks = blocks_max;
        else if (coeff_pool_blocks > blocks_max) return H263MI_ERR_INVALID_ARGUMENT;
        if (!src.pool_blocks) return H263MI_ERR_INVALID_ARGUMENT;
    } else {
        src.pool_blocks = 0;                     // no pool at all: every coded block is outside it
    }
    src.checked = true;
    return H263MI_OK;
}

// The caller's OUTPUT buffers of the same entry points: where the runtime knows the allocation a pointer lies in, it must hold
// what the launch will write (n pictures of RGBA / of filtered planes); a pointer it does not know (the device view of
// registered host memory, another runtime's memory) is taken as it is.  Checked batches only.
static int bound_output_buffers(const h263mi_batch *b, const uint8_t *d_rgba, const uint8_t *d_deblocked)
{
    if (b->trusted_arrays) return H263MI_OK;
    const size_t rgba_bytes = (size_t)b->n * b->L.width * b->L.height * 4;
    const size_t plane_bytes = (size_t)b->n * ((size_t)b->L.width * b->L.height + 2 * (size_t)b->L.cwidth * b->L.cheight);
    size_t left = 0;
    if (d_rgba) {
        const int rc = bytes_behind(d_rgba, &left);
        if (rc == H263MI_ERR_OUT_OF_MEMORY || (rc == H263MI_OK && left < rgba_bytes)) return rc == H263MI_OK ? H263MI_ERR_INVALID_ARGUMENT : rc;
    }
    if (d_deblocked) {
        const int rc = bytes_behind(d_deblocked, &left);
        if (rc == H263MI_ERR_OUT_OF_MEMORY || (rc == H263MI_OK && left < plane_bytes)) return rc == H263MI_OK ? H263MI_ERR_INVALID_ARGUMENT : rc;
    }
    return H263MI_OK;
}

}  // namespace h263mi

// =========================================================================================
// C ABI: batches over device records
// =========================================================================================
extern "C" {

int h263mi_batch_create(uint32_t n_streams, uint16_t width, uint16_t height, const h263mi_backend_cfg *cfg,
                        h263mi_batch **out)
{
    return batch_create(n_streams, width, height, cfg, out);
}

void h263mi_batch_destroy(h263mi_batch *b) { delete b; }

uint32_t h263mi_batch_mbs_per_picture(const h263mi_batch *b) { return b ? b->L.mbw * b->L.mbh : 0; }

int h263mi_batch_submit(h263mi_batch *b, uint8_t picture_type, const h263mi_mb_record *d_mbs, const int16_t *d_coeffs,
                        const uint64_t *d_coeff_base)
{
    if (!b || !d_mbs || picture_type > H263MI_PICTURE_RESERVED) return H263MI_ERR_INVALID_ARGUMENT;
    DeviceGuard g(b->device);
    h263mi_coeff_source src;
    src.coeffs = d_coeffs;
    src.coeff_base = d_coeff_base;
    RC_TRY(bound_device_arrays(b, d_mbs, 0, 0, src));       // (no size argument here: the pool is bounded by its allocation)
    return b->submit(picture_type, d_mbs, src);
}

// decode + post-process in one call: the common tail of h263mi_batch_decode[_events][_ps]
static int batch_decode_device(h263mi_batch *b, uint8_t picture_type, const h263mi_mb_record *d_mbs, const h263mi_coeff_source &src,
                               const h263mi_batch::Strengths &st, uint8_t *d_rgba, uint8_t *d_deblocked)
{
    if (b->pipeline_post) {
        // this picture's reconstruction and the previous picture's post-processing in one launch; this picture's
        // post-processing waits for the next call (or the next sync)
        RC_TRY(b->submit(picture_type, d_mbs, src, /*with_post=*/b->pending.valid));
        return b->note_pending(st, d_rgba, d_deblocked);
    }
    RC_TRY(b->submit(picture_type, d_mbs, src));
    if (!d_rgba && !d_deblocked) return H263MI_OK;
    return b->render(st, d_rgba, d_deblocked, /*only_active=*/true);
}

int h263mi_batch_decode_ps(h263mi_batch *b, uint8_t picture_type, const h263mi_mb_record *d_mbs, const int16_t *d_coeffs,
                           const uint64_t *d_coeff_base, uint64_t coeff_pool_blocks, uint8_t strength, const uint8_t *strengths,
                           uint8_t *d_rgba, uint8_t *d_deblocked)
{
    if (!b || !d_mbs || picture_type > H263MI_PICTURE_RESERVED) return H263MI_ERR_INVALID_ARGUMENT;
    h263mi_batch::Strengths st;
    RC_TRY(make_strengths(strength, strengths, b->n, /*from_header_allowed=*/false, st));
    DeviceGuard g(b->device);
    h263mi_coeff_source src;
    src.coeffs = d_coeffs;
    src.coeff_base = d_coeff_base;
    RC_TRY(bound_device_arrays(b, d_mbs, coeff_pool_blocks, 0, src));
    RC_TRY(bound_output_buffers(b, d_rgba, d_deblocked));
    return batch_decode_device(b, picture_type, d_mbs, src, st, d_rgba, d_deblocked);
}

int h263mi_batch_decode(h263mi_batch *b, uint8_t picture_type, const h263mi_mb_record *d_mbs, const int16_t *d_coeffs,
                        const uint64_t *d_coeff_base, uint64_t coeff_pool_blocks, uint8_t strength, uint8_t *d_rgba,
                        uint8_t *d_deblocked)
{
    return h263mi_batch_decode_ps(b, picture_type, d_mbs, d_coeffs, d_coeff_base, coeff_pool_blocks, strength, nullptr, d_rgba, d_deblocked);
}

/* the same with the coefficients as sparse events already in device memory (what the host entry points copy there) */
int h263mi_batch_decode_events_ps(h263mi_batch *b, uint8_t picture_type, const h263mi_mb_record *d_mbs,
                                  const uint32_t *d_block_first_event, const uint32_t *d_events, const uint64_t *d_coeff_base,
                                  uint64_t coeff_pool_blocks, uint64_t n_events, uint8_t strength, const uint8_t *strengths,
                                  uint8_t *d_rgba, uint8_t *d_deblocked)
{
    if (!b || !d_mbs || !d_block_first_event || !d_events || picture_type > H263MI_PICTURE_RESERVED || n_events > kMaxEventWords)
        return H263MI_ERR_INVALID_ARGUMENT;
    h263mi_batch::Strengths st;
    RC_TRY(make_strengths(strength, strengths, b->n, /*from_header_allowed=*/false, st));
    DeviceGuard g(b->device);
    h263mi_coeff_source src;
    src.first_event = d_block_first_event;
    src.events = d_events;
    src.coeff_base = d_coeff_base;
    RC_TRY(bound_device_arrays(b, d_mbs, coeff_pool_blocks, n_events, src));
    RC_TRY(bound_output_buffers(b, d_rgba, d_deblocked));
    return batch_decode_device(b, picture_type, d_mbs, src, st, d_rgba, d_deblocked);
}

int h263mi_batch_decode_events(h263mi_batch *b, uint8_t picture_type, const h263mi_mb_record *d_mbs,
                               const uint32_t *d_block_first_event, const uint32_t *d_events, const uint64_t *d_coeff_base,
                               uint64_t coeff_pool_blocks, uint64_t n_events, uint8_t strength, uint8_t *d_rgba,
                               uint8_t *d_deblocked)
{
    return h263mi_batch_decode_events_ps(b, picture_type, d_mbs, d_block_first_event, d_events, d_coeff_base, coeff_pool_blocks, n_events,
                                         strength, nullptr, d_rgba, d_deblocked);
}

int h263mi_batch_render_rgba_ps(h263mi_batch *b, uint8_t strength, const uint8_t *strengths, uint8_t *d_rgba, uint8_t *d_deblocked)
{
    if (!b || (!d_rgba && !d_deblocked)) return H263MI_ERR_INVALID_ARGUMENT;
    h263mi_batch::Strengths st;
    RC_TRY(make_strengths(strength, strengths, b->n, /*from_header_allowed=*/false, st));
    DeviceGuard g(b->device);
    RC_TRY(bound_output_buffers(b, d_rgba, d_deblocked));
    return b->render(st, d_rgba, d_deblocked);
}

int h263mi_batch_render_rgba(h263mi_batch *b, uint8_t strength, uint8_t *d_rgba, uint8_t *d_deblocked)
{
    return h263mi_batch_render_rgba_ps(b, strength, nullptr, d_rgba, d_deblocked);
}

int h263mi_batch_sync(h263mi_batch *b)
{
    if (!b) return H263MI_ERR_INVALID_ARGUMENT;
    DeviceGuard g(b->device);
    return b->sync();
}

int h263mi_batch_reset(h263mi_batch *b)
{
    if (!b) return H263MI_ERR_INVALID_ARGUMENT;
    DeviceGuard g(b->device);                    // (a deferred post-processing may be launched: on the batch's device)
    if (!g.ok) return H263MI_ERR_NO_DEVICE;
    return b->forget_pictures();
}

int h263mi_batch_reset_stream(h263mi_batch *b, uint32_t stream)
{
    if (!b || stream >= b->n) return H263MI_ERR_INVALID_ARGUMENT;
    DeviceGuard g(b->device);
    if (!g.ok) return H263MI_ERR_NO_DEVICE;
    return b->forget_stream(stream);
}

int h263mi_batch_set_active(h263mi_batch *b, const uint8_t *active)
{
    if (!b) return H263MI_ERR_INVALID_ARGUMENT;
    for (uint32_t i = 0; i < b->n; i++) b->ss[i].active = active ? active[i] != 0 : true;
    return H263MI_OK;
}

int h263mi_batch_sync_streams(h263mi_batch *b, int *stream_rc)
{
    if (!b) return H263MI_ERR_INVALID_ARGUMENT;
    DeviceGuard g(b->device);
    return b->sync(stream_rc);
}

int h263mi_batch_stream_has_picture(const h263mi_batch *b, uint32_t stream)
{
    return b && stream < b->n && b->ss[stream].cur >= 0 ? 1 : 0;
}

int h263mi_batch_copy_yuv(h263mi_batch *b, uint32_t stream, uint8_t *y, uint8_t *cb, uint8_t *cr)
{
    if (!b) return H263MI_ERR_INVALID_ARGUMENT;
    DeviceGuard g(b->device);
    return b->copy_yuv(stream, y, cb, cr);
}

int h263mi_batch_timing_begin(h263mi_batch *b)
{
    if (!b) return H263MI_ERR_INVALID_ARGUMENT;
    b->timing = true;
    b->ev_used = 0;
    b->ev_ranges.clear();
    b->chain_kernel = -1;
    return H263MI_OK;
}

int h263mi_batch_timing_reserve(h263mi_batch *b, uint32_t n_launches)
{
    if (!b) return H263MI_ERR_INVALID_ARGUMENT;
    DeviceGuard g(b->device);
    while (b->ev_pool.size() < 2 * (size_t)n_launches) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        b->ev_pool.push_back(e);
    }
    b->ev_ranges.reserve(n_launches);
    return H263MI_OK;
}

int h263mi_batch_timing_end(h263mi_batch *b, h263mi_kernel_times *out)
{
    if (!b || !out) return H263MI_ERR_INVALID_ARGUMENT;
    DeviceGuard g(b->device);
    RC_TRY(b->time_close());
    b->timing = false;
    if (b->overlap_post) HIP_TRY(hipStreamSynchronize(b->post_stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    *out = h263mi_kernel_times{};
    for (auto &r : b->ev_ranges) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, b->ev_pool[r.first], b->ev_pool[r.first + 1]));
        if (r.kernel == 0) {
            out->recon_ms += ms;
            out->recon_launches += r.launches;
        } else if (r.kernel == 1) {
            out->post_ms += ms;
            out->post_launches += r.launches;
        } else {
            out->frame_ms += ms;
            out->frame_launches += r.launches;
        }
    }
    b->ev_ranges.clear();
    b->ev_used = 0;
    return H263MI_OK;
}

}  // extern "C"
