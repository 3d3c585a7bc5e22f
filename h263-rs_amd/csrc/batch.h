// batch.h -- h263mi_batch: N independent streams (N x H263State, state.rs:16-50) on one GPU: the device-resident frame
// store, the per-stream reference bookkeeping of state.rs:464-483, the deferred post-processing of pipeline mode, launch
// timing, and the pinned staging slots of the entry points that take host data.
//   batch.cpp          frame store, submit / render / sync, timing, the entry points over DEVICE records
//   batch_staging.cpp  host records and bitstreams -> pinned staging -> one launch (h263mi_batch_submit_host*,
//                      h263mi_batch_decode_next_pictures*)
//   mixed_set.cpp      streams of different picture sizes: one batch per size class
//   state.cpp          the H263State mirror: a batch of one stream fed with host records
#pragma once

#include <memory>
#include <vector>

#include "../host/bitstream.hpp"
#include "host_common.h"
#include "worker_pool.h"

// Where the coefficients of one submit live, how far they may be read, and -- sparse records -- where the records are.
// Passed to submit() by value for the one launch it is for (rounds 2-5 parked these in "next submit" members of the batch,
// which an early return could leave behind for the submit after).
struct h263mi_coeff_source {
    const int16_t *coeffs = nullptr;             // dense transport: the pool
    const uint32_t *first_event = nullptr;       // sparse transport: block offsets ...
    const uint32_t *events = nullptr;            // ... and events (level << 16 | x + 8 * y)
    uint32_t n_events = 0;                       // words in `events`; 0 = unknown (trusted arrays only)
    const uint64_t *coeff_base = nullptr;        // per picture base (blocks), or nullptr: all 0
    uint64_t pool_blocks = 0;                    // blocks in the pool (dense) / entries of first_event minus one (events)
    bool checked = false;                        // the waves refuse to read beyond pool_blocks / n_events
    const uint32_t *group_index = nullptr;       // sparse RECORDS (ReconArgs::mb_group_index), with ...
    const uint64_t *mb_base = nullptr;           // ... each picture's first record
};

struct h263mi_batch {
    int device = 0;
    hipStream_t stream = nullptr;
    static constexpr unsigned kPtrSlots = 4;
    // H263MI_CFG_OVERLAP_POST: k_post runs on a second stream so that the post-processing of picture i overlaps
    // the reconstruction of picture i+1 (k_recon is VALU-heavy, k_post store-heavy).  Legal with two frame sets:
    // post(i) reads set i; recon(i+1) reads set i and overwrites the set of picture i-1, which post(i-1) must have
    // finished reading -- both dependencies are HIP events.
    hipStream_t post_stream = nullptr;
    hipEvent_t ev_recon_done = nullptr, ev_post_done[2] = {nullptr, nullptr};   // post events per frame set
    bool overlap_post = false;
    // H263MI_CFG_PIPELINE_POST: h263mi_batch_decode defers the post-processing of a picture to the launch that
    // reconstructs the NEXT one (k_frame: both read the same frame set, see kernels.hip); `pending` is that deferred
    // half.  Flushed (as a plain k_post launch) by sync, render, submit, reset.
    bool pipeline_post = false;
    // H263MI_CFG_TRUSTED_ARRAYS: the caller vouches for the device arrays it hands to h263mi_batch_submit / _decode /
    // _decode_events; without it (the default, ABI 7) every such array is bounded -- by the counts the caller gives, else by the
    // allocation the pointer lies in -- and the waves read nothing beyond
    bool trusted_arrays = false;
    // The post-filter strength is a property of the PICTURE (its quantiser and its USE_DEBLOCKER flag: deblock.rs:5-8,
    // picture.rs:61-64, types.rs:94-96,216), so every stream of a call may have its own (ABI 7).
    struct Strengths {
        uint8_t uniform = 0;                   // every stream, unless ...
        std::vector<uint8_t> per_stream;       // ... this holds one value per stream (empty = uniform)
        uint8_t of(uint32_t i) const { return per_stream.empty() ? uniform : per_stream[i]; }
        bool same_for_all() const
        {
            for (uint8_t v : per_stream)
                if (v != per_stream[0]) return false;
            return true;
        }
    };
    struct PendingPost {
        bool valid = false;
        Strengths strength;
        uint8_t *rgba = nullptr, *planes = nullptr;
        uint8_t *const *rgba_ptrs = nullptr;   // DEVICE array of per-stream output pointers (a batch inside a mixed-size set)
        std::vector<int8_t> set;               // per stream: frame set it reads, -1 = nothing to post-process
    } pending;
    // per-stream output pointers for the kernels: ring of pinned host slots + device arrays, like the state words
    uint8_t **h_ptrs = nullptr, **d_ptrs = nullptr;
    hipEvent_t ptrs_copied[kPtrSlots] = {nullptr, nullptr, nullptr, nullptr};
    unsigned ptrs_slot = 0;
    bool ptrs_ready = false;            // the ring above exists completely (push_rgba_ptrs makes it on first use)
    uint32_t n = 0;
    h263mi::FrameLayout L{};
    uint8_t *frames[2] = {nullptr, nullptr};   // ping-pong frame sets, n * frame_bytes each
    // Every stream of the batch is its own H263State (state.rs:16-50): its own last picture, its own reference flag,
    // its own errors.  As long as all streams agree (the common case: they advance in lock step and nothing fails) the
    // kernels get one set of pointers; once they differ, a word per stream (dev_common.h: STREAM_*).
    struct StreamState {
        int8_t cur = -1;                       // frame set holding the stream's last picture, -1 = none
        bool has_ref = false;                  // state.rs:29-31 reference_picture.is_some()
        int8_t good_cur = -1;                  // ... as of the last successful sync (what an error falls back to)
        bool good_has_ref = false;
        uint32_t unsynced = 0;                 // pictures submitted since then
        bool active = true;                    // takes part in the next submit (h263mi_batch_set_active)
    };
    std::vector<StreamState> ss;
    uint32_t *d_status = nullptr;              // one word per stream
    uint32_t *h_status = nullptr;              // pinned
    // per-stream words for the kernels: a small ring of pinned host slots + one device array per slot
    static constexpr unsigned kStateSlots = 4;
    uint32_t *h_state = nullptr, *d_state = nullptr;
    hipEvent_t state_copied[kStateSlots] = {nullptr, nullptr, nullptr, nullptr};
    unsigned state_slot = 0;
    // (what sync() falls back to when the device reports an error -- state.rs:142, 464-487: an error leaves the state
    // unchanged -- is each stream's good_cur / good_has_ref, valid as long as at most one picture was submitted for the
    // stream since: the frame set it names is the one the ping-pong has not overwritten yet)
    unsigned frame_launches = 0;               // k_frame launches so far: odd ones walk the pictures backwards
    // host-record staging for h263mi_batch_submit_host: two slots (pinned host + device) used alternately, so
    // that packing picture i+1 overlaps the copy and the kernel of picture i
    struct HostStaging {
        h263mi::MbRecord *h_mbs = nullptr, *d_mbs = nullptr;
        int16_t *h_coeffs = nullptr, *d_coeffs = nullptr;
        // ONE buffer (pinned host + device) for everything small that goes with a call, so that it crosses the link in one copy:
        // [base: 2n x u64 -- [0, n) coefficient base per stream, [n, 2n) record base (sparse records)]
        // [index: n x groups per picture x u32 -- sparse records, one word per group of 8 macroblocks]
        // [events: rebased block offsets, then the events -- sparse coefficient transport]
        uint32_t *h_words = nullptr, *d_words = nullptr;
        size_t cap_words = 0;
        uint64_t *h_base = nullptr, *d_base = nullptr;       // (into h_words / d_words)
        uint32_t *h_index = nullptr, *d_index = nullptr;
        uint32_t *h_events = nullptr, *d_events = nullptr;
        size_t cap_blocks = 0;
        hipEvent_t done = nullptr;             // recorded after the kernel that reads the slot
    } host_stg[2];
    unsigned host_slot = 0;
    // h263mi_batch_decode_next_pictures: what each stream remembers of its last picture header (state.rs:143-167)
    // and the parse results of the current call (kept between calls so that their buffers are reused)
    std::vector<h263mi::bits::ParserContext> parser_ctx;
    std::vector<h263mi::bits::ParsedPicture> parsed;
    // Where the host side of this batch runs: the NUMA node of its device (worker_pool.h).  Looked up when the batch is made.
    h263mi::HostPlacement placement;
    std::unique_ptr<h263mi::WorkerPool> pool;  // host threads of the entry points that take host data
    long pool_spin_us = h263mi::WorkerPool::kSpinUsDefault;    // (HostThreadPlan::spin_us of the call that is being packed)
    h263mi::WorkerPool &workers(unsigned want)
    {
        if (!pool || pool->size() < want) pool.reset(new h263mi::WorkerPool(want - 1, &placement));
        return *pool;
    }
    // H263MI_TRACE_E2E=1: where the host time of h263mi_batch_decode_next_pictures goes (printed when the batch is
    // destroyed): [0] parser threads, [1] waiting for the staging slot, [2] packing into pinned staging, [3] enqueueing
    // copies and launches
    double host_ms[6] = {0, 0, 0, 0, 0, 0};     // ... [4] of [3]: the copies, [5] of [3]: submit (state words, launch)
    size_t frame_skew = 0;
    unsigned host_calls = 0;
    bool trace_host = getenv("H263MI_TRACE_E2E") != nullptr;
    bool trace_each = trace_host && getenv("H263MI_TRACE_E2E")[0] == '2';
    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    struct TimedChain { size_t first; int kernel; uint32_t launches; };   // (index of the begin event, kernel id, launches)
    std::vector<TimedChain> ev_ranges;
    int chain_kernel = -1;
    uint32_t chain_launches = 0;

    ~h263mi_batch();
    int alloc(uint32_t n_streams, uint32_t w, uint32_t h);

    // ---- views of the per-stream state
    bool any_picture() const;
    // every stream takes part and all agree on (cur, has_ref): the kernels need no per-stream words
    bool uniform() const;
    bool pending_uniform() const;
    // hand the kernels one word per stream: fills the next slot of the ring and queues its copy
    // `on`: the stream whose kernel reads the words (the copy is ordered in front of that kernel by being on its stream)
    int push_stream_words(const std::vector<uint32_t> &words, const uint32_t **d_out, hipStream_t on);
    int make_ptr_ring();
    void release_ptr_ring();
    // hand the post-processing one output pointer per stream (host array of n DEVICE pointers): next ring slot + its copy
    int push_rgba_ptrs(uint8_t *const *host_ptrs, uint8_t *const **d_out, hipStream_t on);
    int forget_pictures();
    // one stream forgets its pictures (the seeking rule of state.rs:134-137 for a single H263State of the batch)
    int forget_stream(uint32_t i);
    void release_frames();

    // ---- staging (batch_staging.cpp)
    // the fixed-size part of a staging slot: the records of every stream, the event that says when the slot may be written again
    int ensure_record_staging(HostStaging &g2);
    // words in front of the events in HostStaging::h_words: the two base arrays and the sparse-record index
    size_t head_words() const { return 4 * (size_t)n + (size_t)n * h263mi::recon_tiles_x(L) * L.mbh; }
    int ensure_host_staging(HostStaging &g2, size_t n_blocks, size_t n_event_words = 0);
    void release_staging();

    // ---- launch timing (h263mi_batch_timing_begin / _end); kernel ids: 0 k_recon, 1 k_post, 2 k_frame
    hipStream_t stream_of(int kernel_id) const { return (kernel_id == 1 && overlap_post) ? post_stream : stream; }
    int time_close();
    int time_begin(int kernel_id);

    // ---- the work
    // state.rs:432-483 for every stream of the batch that takes part.  types: one picture type per stream, or nullptr:
    // `picture_type` for all.  with_post: run the deferred post-processing (pending) in the same launch (pipeline mode).
    int submit(uint8_t picture_type, const h263mi::MbRecord *d_mbs, const h263mi_coeff_source &src, bool with_post = false,
               const uint8_t *types = nullptr);
    h263mi::PostArgs post_args(int set, uint8_t strength, uint8_t *d_rgba, uint8_t *d_planes) const;
    // k_post over `sets` (per stream: the frame set to read, -1 = skip the stream); rgba_ptrs: DEVICE array of per-stream
    // output pointers instead of d_rgba (or nullptr)
    int launch_post_sets(const std::vector<int8_t> &sets, const Strengths &strength, uint8_t *d_rgba, uint8_t *d_planes, hipStream_t on,
                         uint8_t *const *rgba_ptrs = nullptr);
    // pipeline mode: the post-processing of the pictures just submitted is deferred to the next launch.
    // host_ptrs (or nullptr): n DEVICE pointers, the RGBA buffer of each stream (nullptr = none for it) instead of d_rgba.
    int note_pending(const Strengths &strength, uint8_t *d_rgba, uint8_t *d_planes, uint8_t *const *host_ptrs = nullptr);
    // the deferred post-processing of pipeline mode, as a launch of its own
    int flush_pending();
    // only_active: the rendering half of a decode call -- streams that sat the call out (h263mi_batch_set_active, no data,
    // a picture that failed to parse) keep their part of the output buffers untouched, as the pipelined form (note_pending)
    // does; h263mi_batch_render_rgba renders every stream's last picture.
    int render(const Strengths &strength, uint8_t *d_rgba, uint8_t *d_planes, bool only_active = false, uint8_t *const *host_ptrs = nullptr);
    // stream_rc (may be null): per stream 0, H263MI_ERR_UNCODED_IFRAME_BLOCKS or H263MI_ERR_INVALID_ARGUMENT
    int sync(int *stream_rc = nullptr);
    int copy_yuv(uint32_t s, uint8_t *y, uint8_t *cb, uint8_t *cr);
};

namespace h263mi {

int batch_create(uint32_t n_streams, uint32_t w, uint32_t h, const h263mi_backend_cfg *cfg, h263mi_batch **out);
// where the host side of device `dev`'s work belongs (worker_pool.h): the PCI addresses of the visible devices -> sysfs
HostPlacement placement_of_device(int dev);

// The `strength` / `strengths` pair of the ABI 7 entry points -> Strengths.  strengths != nullptr: one value per stream
// (0..12 each); else strength: 0..12 for every stream, or -- where the entry has parsed the headers (from_header_allowed) --
// H263MI_STRENGTH_FROM_HEADER: per_stream is sized and the entry fills it in per picture.  H263MI_ERR_INVALID_ARGUMENT for
// anything else.
int make_strengths(uint8_t strength, const uint8_t *strengths, uint32_t n, bool from_header_allowed, h263mi_batch::Strengths &out);

// DIRECT WORDS (round 5, h263mi_batch_decode_next_pictures only): the parser has written every stream's block offsets, events
// and group index straight into the staging slot -- stream i's block offsets at h_events + i * pitch_blocks, its events at
// h_events + n * pitch_blocks + i * pitch_events, the offsets counting from i * pitch_events -- so nothing is packed: the
// used head of every stream's part crosses the link in one 2-D copy per array.  The pitches are the worst case of the
// call's pictures (bits::event_words_bound), known from their lengths before a bit is parsed.
struct DirectWords {
    size_t pitch_blocks, pitch_events;
};

// one picture per stream from per-stream host arrays; coefficients dense (`coeffs`) or as events (`first_event`,
// `events`, `n_events`) -- batch_staging.cpp
int batch_submit_host(h263mi_batch *b, uint8_t picture_type, const h263mi_mb_record *const *mbs, const uint32_t *n_mbs,
                      const int16_t *const *coeffs, const uint32_t *n_coeff_blocks, const uint32_t *const *first_event,
                      const uint32_t *const *events, const uint32_t *n_events, bool from_parser = false, uint32_t pack_threads = 0,
                      const uint8_t *types = nullptr, bool deferred_post = false, const uint32_t *const *group_index = nullptr,
                      const DirectWords *direct = nullptr);

}  // namespace h263mi
