// backend.cpp -- host side of the C ABI (include/h263mi.h): device-resident frame store,
// batch of streams, the H263State mirror, and the plain-function deblock / bt601 entry
// points.  Compiled with hipcc; every compute path launches the gfx950 kernels of
// kernels.hip -- there is no CPU fallback.
#include <hip/hip_runtime.h>

#include <pthread.h>
#include <sched.h>

#include <new>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../host/bitstream.hpp"
#include "kernels.h"
#include "post_kernel.inl"   // tile constants only
#include "recon_kernel.inl"  // tile constants only
#include "synth.inl"

using namespace h263mi;

namespace {

int map_hip_error(hipError_t e)
{
    switch (e) {
    case hipSuccess: return H263MI_OK;
    case hipErrorNoDevice:
    case hipErrorInvalidDevice:
    case hipErrorInsufficientDriver:
    case hipErrorNotInitialized: return H263MI_ERR_NO_DEVICE;
    case hipErrorOutOfMemory: return H263MI_ERR_OUT_OF_MEMORY;
    default: return H263MI_ERR_HIP;
    }
}

// Fault injection for tests (h263mi_debug_fail_nth_hip_call): the n-th HIP call made through HIP_TRY from now on is not
// executed and reports hipErrorOutOfMemory instead.  Every allocation, copy, event operation and launch of the host entry
// points goes through HIP_TRY, so sweeping n over a call proves "on error the state is unchanged" (state.rs:142, 464-487)
// at every point at which the call can fail.  -1 = off (the product never sets it).
std::atomic<int> g_fail_countdown{-1};
bool fault_now()
{
    int v = g_fail_countdown.load(std::memory_order_relaxed);
    if (v < 0) return false;
    v = g_fail_countdown.fetch_sub(1, std::memory_order_relaxed);
    return v == 1;                                 // the countdown went 1 -> 0 with this call
}

#define HIP_TRY(expr)                                                      \
    do {                                                                   \
        hipError_t _e = fault_now() ? hipErrorOutOfMemory : (expr);        \
        if (_e != hipSuccess) return map_hip_error(_e);                    \
    } while (0)

#define RC_TRY(expr)                  \
    do {                              \
        int _rc = (expr);             \
        if (_rc != H263MI_OK) return _rc; \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; }
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// Host worker threads of a batch (parser tasks, packing into pinned staging): created once and parked between calls --
// h263mi_batch_decode_next_pictures used to start and join two sets of threads per frame index.  run(k, fn) executes
// fn(0) .. fn(k - 1), fn(0) on the calling thread, and returns when all are done; calls do not nest or overlap (a batch
// is driven from one thread at a time).
// A server calls the batch entries back to back, a millisecond apart: waking 15 parked threads through a condition
// variable cost 50-100 us of every call (twice: parser tasks, then packing).  A worker therefore SPINS on the generation
// counter for a short while after it has finished a task (spin_us) and only then parks; the caller spins likewise while
// it waits for the last worker.  An idle batch costs nothing: everybody is parked.
// Under a CPU-TIME quota (a container's cpu.max) spinning is paid for like parsing: see HostThreadPlan, which then runs more
// threads than the quota has CPUs and has them park at once.
class WorkerPool {
public:
    explicit WorkerPool(unsigned workers)
    {
        for (unsigned t = 0; t < workers; t++) threads_.emplace_back([this, t] { loop(t + 1); });
        keep_off_sibling_hyperthreads();
    }
    ~WorkerPool()
    {
        {
            std::lock_guard<std::mutex> l(m_);
            stop_.store(true, std::memory_order_release);
            generation_.store(((generation_.load(std::memory_order_relaxed) >> 32) + 1) << 32, std::memory_order_release);
        }
        wake_.notify_all();
        for (std::thread &t : threads_) t.join();
    }
    unsigned size() const { return (unsigned)threads_.size() + 1; }       // the caller counts
    // spin_us: how long a worker that has finished spins for the next task before it parks (see loop())
    void run(unsigned k, const std::function<void(unsigned)> &fn, long spin_us = kSpinUsDefault)
    {
        if (k > size()) k = size();
        if (k <= 1) { fn(0); return; }
        spin_us_.store(spin_us, std::memory_order_relaxed);
        fn_ = &fn;
        pending_.store(k - 1, std::memory_order_relaxed);
        {
            // generation and the number of threads it is for travel in ONE word: a worker that is late for a generation it
            // has no part in must not pair that generation with the next one's thread count
            std::lock_guard<std::mutex> l(m_);                            // (orders the bump against a worker about to park)
            const uint64_t gen = (generation_.load(std::memory_order_relaxed) >> 32) + 1;
            generation_.store((gen << 32) | k, std::memory_order_release);
        }
        if (parked_.load(std::memory_order_acquire)) wake_.notify_all();
        fn(0);
        for (unsigned spins = 0; pending_.load(std::memory_order_acquire) != 0; spins++) {
            if (spins < 20000) cpu_relax();
            else std::this_thread::yield();
        }
        fn_ = nullptr;
    }

private:
    // The parser is a chain of dependent table look-ups: two of its threads on the two hyperthreads of one core run at
    // 60-70 % each.  On a host with many more cores than worker threads (the GPU boxes: 16 CPUs of quota on 128 cores / 256
    // hyperthreads) the workers are therefore confined to ONE hyperthread per physical core -- the lowest-numbered of each
    // sibling set, within the affinity mask the process already has -- and the scheduler spreads them over distinct cores.
    // Every process makes the same choice, so eight ranks with 16 workers each still find 128 distinct cores.  Only the
    // pool's own threads are touched (never the caller's); H263MI_PIN_THREADS=0 leaves them alone.
    void keep_off_sibling_hyperthreads()
    {
        const char *env = getenv("H263MI_PIN_THREADS");
        if (env && env[0] == '0') return;
        cpu_set_t have, want;
        if (sched_getaffinity(0, sizeof have, &have) != 0) return;
        CPU_ZERO(&want);
        for (int cpu = 0; cpu < CPU_SETSIZE; cpu++) {
            if (!CPU_ISSET(cpu, &have)) continue;
            char path[128];
            snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", cpu);
            FILE *f = fopen(path, "r");
            int first = cpu;
            if (f) {
                if (fscanf(f, "%d", &first) != 1) first = cpu;      // "3,131" or "3-4": the list starts with its lowest member
                fclose(f);
            }
            if (first == cpu || !CPU_ISSET(first, &have)) CPU_SET(cpu, &want);
        }
        // only when that still leaves room to spread: at least twice as many cores as workers
        if ((unsigned)CPU_COUNT(&want) < 2 * (unsigned)threads_.size() + 2 || CPU_COUNT(&want) == CPU_COUNT(&have)) return;
        for (std::thread &t : threads_) (void)pthread_setaffinity_np(t.native_handle(), sizeof want, &want);
    }
public:
    static constexpr long kSpinUsDefault = 300;
private:
    std::atomic<long> spin_us_{kSpinUsDefault};
    static void cpu_relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    void loop(unsigned id)
    {
        uint64_t seen = 0;                               // generation << 32 | threads of that generation
        for (;;) {
            // spin for the next task, then park
            const auto t0 = std::chrono::steady_clock::now();
            unsigned polls = 0;
            uint64_t now;
            while ((now = generation_.load(std::memory_order_acquire)) == seen) {
                cpu_relax();
                if ((++polls & 255u) == 0 &&
                    std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >=
                        spin_us_.load(std::memory_order_relaxed)) {
                    std::unique_lock<std::mutex> l(m_);
                    parked_.fetch_add(1, std::memory_order_release);
                    wake_.wait(l, [&] { return generation_.load(std::memory_order_acquire) != seen; });
                    parked_.fetch_sub(1, std::memory_order_release);
                }
            }
            if (stop_.load(std::memory_order_acquire)) return;
            seen = now;
            if (id >= (unsigned)(now & 0xffffffffu)) continue;           // no part in this generation
            (*fn_)(id);                                  // (fn_ cannot change before this thread has reported back)
            pending_.fetch_sub(1, std::memory_order_release);
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable wake_;
    const std::function<void(unsigned)> *fn_ = nullptr;
    std::atomic<unsigned> pending_{0}, parked_{0};
    std::atomic<uint64_t> generation_{0};
    std::atomic<bool> stop_{false};
};

// Streams dealt to host threads with AFFINITY: thread t of T first takes the streams t, t + T, t + 2T, ... -- the same ones
// in every call, so that a stream's parse buffers and its slot of the staging memory (261 KB of records per 1080p picture)
// stay in that core's caches instead of migrating between cores from call to call -- and then helps out with whatever
// the other threads have not started yet (a stream is claimed with one atomic exchange).
struct StreamDeal {
    std::unique_ptr<std::atomic<uint8_t>[]> taken;
    uint32_t n = 0;
    explicit StreamDeal(uint32_t n_streams) : taken(new std::atomic<uint8_t>[n_streams]), n(n_streams)
    {
        for (uint32_t i = 0; i < n; i++) taken[i].store(0, std::memory_order_relaxed);
    }
    template <class F> void run(unsigned t, unsigned n_threads, F &&task)
    {
        for (uint32_t i = t; i < n; i += n_threads)
            if (!taken[i].exchange(1, std::memory_order_relaxed)) task(i);
        for (uint32_t k = 0; k < n; k++) {               // leftovers, starting behind the own ones
            const uint32_t i = (k + t) % n;
            if (!taken[i].load(std::memory_order_relaxed) && !taken[i].exchange(1, std::memory_order_relaxed)) task(i);
        }
    }
};

uint32_t recon_tiles_x(const FrameLayout &L) { return (L.mbw + TILE_MBX - 1) / TILE_MBX; }
uint32_t recon_tiles_y(const FrameLayout &L) { return (L.mbh + TILE_MBY - 1) / TILE_MBY; }
uint32_t post_tiles_y(const FrameLayout &L) { return (post_strips_y(L.height) + POST_STRIPS - 1) / POST_STRIPS; }
// Event indices are 32-bit on the device, 0xffffffff stands for "the caller did not say how many" (ReconArgs::n_events), and a
// lane looks up to 64 words past its first event before it compares with the block's end: the count stays clear of the top.
constexpr uint64_t kMaxEventWords = 0xffffff00ull;
// what the parser asks right behind a picture header (bits::ParsedPicture::size_fits): can the frame store hold such a picture?
bool picture_size_fits(uint32_t w, uint32_t h) { return layout_fits(w, h); }
// tile geometry of k_post for the layout in a.L (post_kernel.inl: post_tile_columns)
void set_post_tiles(PostArgs &a)
{
    a.tiles_x = post_tile_columns(a.L.width, &a.wrap);
    a.tiles_y = post_tiles_y(a.L);
}

}  // namespace

// =========================================================================================
// batch of streams
// =========================================================================================
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
    struct PendingPost {
        bool valid = false;
        uint8_t strength = 0;
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
    FrameLayout L{};
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
    const uint32_t *cur_first_event = nullptr, *cur_events = nullptr;   // sparse transport of the next submit (then cleared)
    const uint32_t *cur_group_index = nullptr;   // sparse RECORDS of the next submit (ReconArgs::mb_group_index; then cleared)
    const uint64_t *cur_mb_base = nullptr;
    uint32_t cur_n_events = 0;                 // ... and how many event words there are (0 = the caller did not say)
    uint64_t coeff_pool_blocks = 0;            // size of the pool the next submit reads ...
    bool coeff_checked = false;                // ... when the caller told us (host entry points do; device pointers do not)
    // (what sync() falls back to when the device reports an error -- state.rs:142, 464-487: an error leaves the state
    // unchanged -- is each stream's good_cur / good_has_ref, valid as long as at most one picture was submitted for the
    // stream since: the frame set it names is the one the ping-pong has not overwritten yet)
    unsigned frame_launches = 0;               // k_frame launches so far: odd ones walk the pictures backwards
    // host-record staging for h263mi_batch_submit_host: two slots (pinned host + device) used alternately, so
    // that packing picture i+1 overlaps the copy and the kernel of picture i
    struct HostStaging {
        MbRecord *h_mbs = nullptr, *d_mbs = nullptr;
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
    std::vector<bits::ParserContext> parser_ctx;
    std::vector<bits::ParsedPicture> parsed;
    std::unique_ptr<WorkerPool> pool;          // host threads of the entry points that take host data
    long pool_spin_us = WorkerPool::kSpinUsDefault;    // (HostThreadPlan::spin_us of the call that is being packed)
    WorkerPool &workers(unsigned want)
    {
        if (!pool || pool->size() < want) pool.reset(new WorkerPool(want - 1));
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

    int alloc(uint32_t n_streams, uint32_t w, uint32_t h)
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

    // ---- views of the per-stream state -----------------------------------------------------------------------
    bool any_picture() const
    {
        for (const StreamState &t : ss)
            if (t.cur >= 0) return true;
        return false;
    }
    // every stream takes part and all agree on (cur, has_ref): the kernels need no per-stream words
    bool uniform() const
    {
        for (const StreamState &t : ss)
            if (!t.active || t.cur != ss[0].cur || t.has_ref != ss[0].has_ref) return false;
        return true;
    }
    bool pending_uniform() const
    {
        for (int8_t v : pending.set)
            if (v != pending.set[0]) return false;
        return true;
    }
    // hand the kernels one word per stream: fills the next slot of the ring and queues its copy
    // `on`: the stream whose kernel reads the words (the copy is ordered in front of that kernel by being on its stream)
    int push_stream_words(const std::vector<uint32_t> &words, const uint32_t **d_out, hipStream_t on)
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

    int make_ptr_ring()
    {
        HIP_TRY(hipHostMalloc((void **)&h_ptrs, (size_t)n * kPtrSlots * sizeof(uint8_t *), hipHostMallocDefault));
        HIP_TRY(hipMalloc((void **)&d_ptrs, (size_t)n * kPtrSlots * sizeof(uint8_t *)));
        for (hipEvent_t &e : ptrs_copied) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        return H263MI_OK;
    }
    void release_ptr_ring()
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
    // hand the post-processing one output pointer per stream (host array of n DEVICE pointers): next ring slot + its copy
    int push_rgba_ptrs(uint8_t *const *host_ptrs, uint8_t *const **d_out, hipStream_t on)
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

    int forget_pictures()
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
    // one stream forgets its pictures (the seeking rule of state.rs:134-137 for a single H263State of the batch)
    int forget_stream(uint32_t i)
    {
        RC_TRY(flush_pending());
        const bool active = ss[i].active;
        ss[i] = StreamState();
        ss[i].active = active;
        if (i < parser_ctx.size()) parser_ctx[i] = bits::ParserContext();
        return H263MI_OK;
    }

    void release_frames()
    {
        if (frames[0]) (void)hipFree(frames[0] - frame_skew);         // (one allocation holds both sets)
        frames[0] = frames[1] = nullptr;
    }

    ~h263mi_batch()
    {
        DeviceGuard g(device);
        (void)hipStreamSynchronize(stream);
        if (trace_host && host_calls)
            fprintf(stderr, "h263mi batch (%u streams): %u host submits; ms per call: parse %.3f, wait for slot %.3f, pack %.3f, "
                            "enqueue %.3f (copies %.3f, launch %.3f)\n", n, host_calls, host_ms[0] / host_calls, host_ms[1] / host_calls,
                    host_ms[2] / host_calls, host_ms[3] / host_calls, host_ms[4] / host_calls, host_ms[5] / host_calls);
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
        for (HostStaging &g2 : host_stg) {
            if (g2.h_mbs) (void)hipHostFree(g2.h_mbs);
            if (g2.d_mbs) (void)hipFree(g2.d_mbs);
            if (g2.h_coeffs) (void)hipHostFree(g2.h_coeffs);
            if (g2.d_coeffs) (void)hipFree(g2.d_coeffs);
            if (g2.h_words) (void)hipHostFree(g2.h_words);
            if (g2.d_words) (void)hipFree(g2.d_words);
            if (g2.done) (void)hipEventDestroy(g2.done);
        }
    }

    // the fixed-size part of a staging slot: the records of every stream, the per-stream coefficient bases, the event
    // that says when the slot may be written again
    int ensure_record_staging(HostStaging &g2)
    {
        const size_t total = (size_t)n * L.mbw * L.mbh;
        // each piece on its own, so that a failed allocation leaves nothing half-initialised for the next call
        if (!g2.h_mbs) HIP_TRY(hipHostMalloc((void **)&g2.h_mbs, total * sizeof(MbRecord), hipHostMallocDefault));
        if (!g2.d_mbs) HIP_TRY(hipMalloc((void **)&g2.d_mbs, total * sizeof(MbRecord)));
        if (!g2.done) HIP_TRY(hipEventCreateWithFlags(&g2.done, hipEventDisableTiming));
        return H263MI_OK;
    }

    // words in front of the events in HostStaging::h_words: the two base arrays and the sparse-record index
    size_t head_words() const { return 4 * (size_t)n + (size_t)n * recon_tiles_x(L) * L.mbh; }
    int ensure_host_staging(HostStaging &g2, size_t n_blocks, size_t n_event_words = 0)
    {
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

    // kernel ids of the timing: 0 k_recon, 1 k_post, 2 k_frame
    hipStream_t stream_of(int kernel_id) const { return (kernel_id == 1 && overlap_post) ? post_stream : stream; }

    // Launch timing (h263mi_batch_timing_begin / _end).  Consecutive launches of the same kernel form a CHAIN that is
    // bracketed by ONE pair of events -- begin in front of the first launch, end behind the last -- and the chain's time
    // is shared out over its launches: an event pair around every single launch put a 6 us bubble between two launches
    // (2 % of a frame index of the 64-stream bench; tools/probes/timing_overhead.py).  A chain ends where the kernel
    // changes and in front of anything else that is queued on the stream (copies, the status read of sync), so only
    // launches -- and the gaps between back-to-back launches -- are inside.
    int chain_kernel = -1;
    uint32_t chain_launches = 0;
    int time_close()
    {
        if (chain_kernel < 0) return H263MI_OK;
        const int k = chain_kernel;
        chain_kernel = -1;
        HIP_TRY(hipEventRecord(ev_pool[ev_used + 1], stream_of(k)));
        ev_ranges.push_back(TimedChain{ev_used, k, chain_launches});
        ev_used += 2;
        return H263MI_OK;
    }
    int time_begin(int kernel_id)
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
    int time_end(int) { return H263MI_OK; }      // (the end of a chain is recorded when it is closed)

    // state.rs:432-483 for every stream of the batch that takes part.  types: one picture type per stream, or nullptr:
    // `picture_type` for all.  with_post: run the deferred post-processing (pending) in the same launch (pipeline mode).
    int submit(uint8_t picture_type, const MbRecord *d_mbs, const int16_t *d_coeffs, const uint64_t *d_coeff_base,
               bool with_post = false, const uint8_t *types = nullptr)
    {
        if (!with_post) RC_TRY(flush_pending());
        ReconArgs a{};
        a.L = L;
        a.mbs = d_mbs;
        a.coeffs = d_coeffs;
        a.block_first_event = cur_first_event;
        a.events = cur_events;
        a.n_events = cur_n_events ? cur_n_events : 0xffffffffu;
        cur_first_event = cur_events = nullptr;
        cur_n_events = 0;
        a.mb_group_index = cur_group_index;
        a.mb_base = cur_mb_base;
        a.groups_per_picture = recon_tiles_x(L) * L.mbh;
        cur_group_index = nullptr;
        cur_mb_base = nullptr;
        a.coeff_base = d_coeff_base;
        a.status = d_status;
        a.coeff_pool_blocks = coeff_pool_blocks;
        a.coeff_checked = coeff_checked ? 1u : 0u;
        a.n_pictures = n;
        a.mbs_per_picture = L.mbw * L.mbh;
        a.tiles_x = recon_tiles_x(L);
        a.tiles_y = recon_tiles_y(L);
        a.frame_set[0] = frames[0];
        a.frame_set[1] = frames[1];
        PostArgs pa{};
        if (with_post) pa = post_args(0, pending.strength, pending.rgba, pending.planes);
        if (with_post) pa.rgba_ptrs = pending.rgba_ptrs;       // (read in the per-stream branch of the kernel only)
        const bool all_same = uniform() && (!with_post || (pending_uniform() && pending.set[0] >= 0 && !pending.rgba_ptrs));
        int out0 = 0;
        if (all_same) {
            const int cur = ss[0].cur;
            out0 = cur < 0 ? 0 : (cur ^ 1);
            // get_reference_picture() hands out the LAST picture whenever a reference exists (state.rs:72-78)
            a.ref = frames[cur < 0 ? 1 : cur];
            a.cur = frames[out0];
            a.has_ref = (ss[0].has_ref && cur >= 0) ? 1u : 0u;
            if (with_post) pa.frames = frames[pending.set[0]];
        } else {
            std::vector<uint32_t> words(n);
            for (uint32_t i = 0; i < n; i++) {
                const StreamState &t = ss[i];
                uint32_t w = (t.cur != 0 ? STREAM_REF_SET1 : 0u) | ((t.has_ref && t.cur >= 0) ? STREAM_HAS_REF : 0u) |
                             (t.active ? 0u : STREAM_RECON_SKIP);
                if (!with_post || pending.set[i] < 0) w |= STREAM_POST_SKIP;
                else if (pending.set[i] == 1) w |= STREAM_POST_SET1;
                words[i] = w;
            }
            const uint32_t *d_words = nullptr;
            RC_TRY(push_stream_words(words, &d_words, stream));
            a.stream_state = d_words;
            a.ref = frames[0];                   // (never used with stream_state; never null)
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
            const hipError_t e = launch_frame(a, pa, stream, (frame_launches++ & 1u) != 0);
            if (e != hipSuccess) {               // the deferred post-processing must not get lost with the failed launch
                (void)flush_pending();
                return map_hip_error(e);
            }
            pending.valid = false;
            RC_TRY(time_end(2));
        } else {
            RC_TRY(time_begin(0));
            HIP_TRY(launch_recon(a, stream));
            RC_TRY(time_end(0));
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

    PostArgs post_args(int set, uint8_t strength, uint8_t *d_rgba, uint8_t *d_planes) const
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

    // k_post over `sets` (per stream: the frame set to read, -1 = skip the stream); rgba_ptrs: DEVICE array of per-stream
    // output pointers instead of d_rgba (or nullptr)
    int launch_post_sets(const std::vector<int8_t> &sets, uint8_t strength, uint8_t *d_rgba, uint8_t *d_planes, hipStream_t on,
                         uint8_t *const *rgba_ptrs = nullptr)
    {
        bool same = rgba_ptrs == nullptr, any = false;
        for (int8_t v : sets) {
            same = same && v == sets[0];
            any = any || v >= 0;
        }
        if (!any) return H263MI_OK;
        PostArgs a = post_args(sets[0] >= 0 ? sets[0] : 0, strength, d_rgba, d_planes);
        a.rgba_ptrs = rgba_ptrs;
        if (!same) {
            std::vector<uint32_t> words(n);
            for (uint32_t i = 0; i < n; i++)
                words[i] = STREAM_RECON_SKIP | (sets[i] < 0 ? STREAM_POST_SKIP : (sets[i] == 1 ? STREAM_POST_SET1 : 0u));
            const uint32_t *d_words = nullptr;
            RC_TRY(push_stream_words(words, &d_words, on));
            a.stream_state = d_words;
            a.frame_set[0] = frames[0];
            a.frame_set[1] = frames[1];
        }
        RC_TRY(time_begin(1));
        HIP_TRY(launch_post(a, on));
        RC_TRY(time_end(1));
        return H263MI_OK;
    }

    // pipeline mode: the post-processing of the pictures just submitted is deferred to the next launch.
    // host_ptrs (or nullptr): n DEVICE pointers, the RGBA buffer of each stream (nullptr = none for it) instead of d_rgba.
    int note_pending(uint8_t strength, uint8_t *d_rgba, uint8_t *d_planes, uint8_t *const *host_ptrs = nullptr)
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

    // the deferred post-processing of pipeline mode, as a launch of its own
    int flush_pending()
    {
        if (!pending.valid) return H263MI_OK;
        pending.valid = false;
        return launch_post_sets(pending.set, pending.strength, pending.rgba, pending.planes, stream, pending.rgba_ptrs);
    }

    // only_active: the rendering half of a decode call -- streams that sat the call out (h263mi_batch_set_active, no data,
    // a picture that failed to parse) keep their part of the output buffers untouched, as the pipelined form (note_pending)
    // does; h263mi_batch_render_rgba renders every stream's last picture.
    int render(uint8_t strength, uint8_t *d_rgba, uint8_t *d_planes, bool only_active = false, uint8_t *const *host_ptrs = nullptr)
    {
        if (!any_picture()) return H263MI_ERR_NO_PICTURE;
        if (strength > 12) return H263MI_ERR_INVALID_ARGUMENT;
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

    // stream_rc (may be null): per stream 0, H263MI_ERR_UNCODED_IFRAME_BLOCKS or H263MI_ERR_INVALID_ARGUMENT
    int sync(int *stream_rc = nullptr)
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

    int copy_yuv(uint32_t s, uint8_t *y, uint8_t *cb, uint8_t *cr)
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
};

static int check_device(int device_id)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return H263MI_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= count) return H263MI_ERR_NO_DEVICE;
    return H263MI_OK;
}

static int batch_create(uint32_t n_streams, uint32_t w, uint32_t h, const h263mi_backend_cfg *cfg, h263mi_batch **out)
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

// =========================================================================================
// H263State mirror: a batch of one stream fed with host records
// =========================================================================================
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
static int state_ensure_staging(h263mi_state::Staging &g, size_t n_mbs, size_t n_blocks, size_t n_event_words)
{
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

const uint8_t h263mi_quant_to_strength[32] = {0, 1, 1, 2, 2, 3, 3, 4, 4, 4,  5,  5,  6,  6,  7,  7,
                                              7, 8, 8, 8, 9, 9, 9, 10, 10, 10, 11, 11, 11, 12, 12, 12};

int h263mi_abi_version(void) { return H263MI_ABI_VERSION; }

const char *h263mi_strerror(int code)
{
    switch (code) {
    case H263MI_OK: return "ok";
    case H263MI_ERR_INTERNAL_DECODER_ERROR: return "the H.263 decoder failed internally, this is a bug";
    case H263MI_ERR_MIDDLE_OF_BITSTREAM: return "the H.263 bitstream doesn't start with a picture";
    case H263MI_ERR_INVALID_MACROBLOCK_HEADER: return "the H.263 bitstream contains an invalid macroblock header";
    case H263MI_ERR_INVALID_MACROBLOCK_CODED_BITS: return "the H.263 bitstream contains invalid macroblock coded bits";
    case H263MI_ERR_INVALID_INTRA_DC: return "the H.263 bitstream contains an invalid intra-dc coefficient";
    case H263MI_ERR_INVALID_SHORT_COEFFICIENT: return "the H.263 bitstream contains an invalid short ac coefficient";
    case H263MI_ERR_INVALID_LONG_COEFFICIENT: return "the H.263 bitstream contains an invalid long ac coefficient";
    case H263MI_ERR_INVALID_MVD: return "the H.263 bitstream contains an invalid motion vector";
    case H263MI_ERR_INVALID_PTYPE: return "the H.263 bitstream has an invalid picture type";
    case H263MI_ERR_INVALID_PLUS_PTYPE: return "the H.263 bitstream has an invalid extension picture type";
    case H263MI_ERR_INVALID_GOB_HEADER: return "the H.263 bitstream has an invalid group-of-blocks header";
    case H263MI_ERR_INVALID_BITSTREAM: return "the H.263 bitstream could not be decoded";
    case H263MI_ERR_PICTURE_FORMAT_MISSING: return "the decoded H.263 bitstream is missing it's picture format";
    case H263MI_ERR_PICTURE_FORMAT_INVALID: return "the decoded H.263 bitstream has an invalid picture format";
    case H263MI_ERR_UNCODED_IFRAME_BLOCKS: return "the decoded H.263 bitstream has uncoded iframe blocks";
    case H263MI_ERR_UNHANDLED_IO_ERROR: return "an I/O error occured";
    case H263MI_ERR_UNIMPLEMENTED_DECODING: return "a feature in the H.263 bitstream being decoded is not yet supported";
    case H263MI_ERR_INVALID_ARGUMENT: return "invalid argument";
    case H263MI_ERR_NO_DEVICE: return "no usable HIP device (the MI355X back-end has no CPU fallback)";
    case H263MI_ERR_HIP: return "HIP runtime error";
    case H263MI_ERR_OUT_OF_MEMORY: return "out of memory";
    case H263MI_ERR_NO_PICTURE: return "no picture has been decoded yet";
    default: return "unknown error";
    }
}

// ---------------------------------------------------------------------------------------
// batch API
// ---------------------------------------------------------------------------------------
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
    b->coeff_checked = false;                    // the size of a caller-owned device pool is not known here
    b->coeff_pool_blocks = 0;
    return b->submit(picture_type, d_mbs, d_coeffs, d_coeff_base);
}

int h263mi_batch_decode(h263mi_batch *b, uint8_t picture_type, const h263mi_mb_record *d_mbs, const int16_t *d_coeffs,
                        const uint64_t *d_coeff_base, uint64_t coeff_pool_blocks, uint8_t strength, uint8_t *d_rgba,
                        uint8_t *d_deblocked)
{
    if (!b || !d_mbs || picture_type > H263MI_PICTURE_RESERVED || strength > 12) return H263MI_ERR_INVALID_ARGUMENT;
    DeviceGuard g(b->device);
    b->coeff_checked = coeff_pool_blocks != 0;
    b->coeff_pool_blocks = coeff_pool_blocks;
    if (b->pipeline_post) {
        // this picture's reconstruction and the previous picture's post-processing in one launch; this picture's
        // post-processing waits for the next call (or the next sync)
        RC_TRY(b->submit(picture_type, d_mbs, d_coeffs, d_coeff_base, /*with_post=*/b->pending.valid));
        return b->note_pending(strength, d_rgba, d_deblocked);
    }
    RC_TRY(b->submit(picture_type, d_mbs, d_coeffs, d_coeff_base));
    if (!d_rgba && !d_deblocked) return H263MI_OK;
    return b->render(strength, d_rgba, d_deblocked, /*only_active=*/true);
}

/* the same with the coefficients as sparse events already in device memory (what the host entry points copy there) */
int h263mi_batch_decode_events(h263mi_batch *b, uint8_t picture_type, const h263mi_mb_record *d_mbs,
                               const uint32_t *d_block_first_event, const uint32_t *d_events, const uint64_t *d_coeff_base,
                               uint64_t coeff_pool_blocks, uint64_t n_events, uint8_t strength, uint8_t *d_rgba,
                               uint8_t *d_deblocked)
{
    if (!b || !d_mbs || !d_block_first_event || !d_events || picture_type > H263MI_PICTURE_RESERVED || strength > 12 ||
        n_events > kMaxEventWords)
        return H263MI_ERR_INVALID_ARGUMENT;
    DeviceGuard g(b->device);
    b->coeff_checked = coeff_pool_blocks != 0;
    b->coeff_pool_blocks = coeff_pool_blocks;
    b->cur_first_event = d_block_first_event;
    b->cur_events = d_events;
    b->cur_n_events = (uint32_t)n_events;
    if (b->pipeline_post) {
        RC_TRY(b->submit(picture_type, d_mbs, nullptr, d_coeff_base, /*with_post=*/b->pending.valid));
        return b->note_pending(strength, d_rgba, d_deblocked);
    }
    RC_TRY(b->submit(picture_type, d_mbs, nullptr, d_coeff_base));
    if (!d_rgba && !d_deblocked) return H263MI_OK;
    return b->render(strength, d_rgba, d_deblocked, /*only_active=*/true);
}

}  // extern "C"

// DIRECT WORDS (round 5, h263mi_batch_decode_next_pictures only): the parser has written every stream's block offsets, events
// and group index straight into the staging slot -- stream i's block offsets at h_events + i * pitch_blocks, its events at
// h_events + n * pitch_blocks + i * pitch_events, the offsets counting from i * pitch_events -- so nothing is packed: the
// used head of every stream's part crosses the link in one 2-D copy per array.  The pitches are the worst case of the
// call's pictures (bits::event_words_bound), known from their lengths before a bit is parsed.
struct DirectWords {
    size_t pitch_blocks, pitch_events;
};

// one picture per stream from per-stream host arrays; coefficients dense (`coeffs`) or as events (`first_event`,
// `events`, `n_events`)
static int batch_submit_host(h263mi_batch *b, uint8_t picture_type, const h263mi_mb_record *const *mbs,
                             const uint32_t *n_mbs, const int16_t *const *coeffs, const uint32_t *n_coeff_blocks,
                             const uint32_t *const *first_event, const uint32_t *const *events, const uint32_t *n_events,
                             bool from_parser = false, uint32_t pack_threads = 0, const uint8_t *types = nullptr,
                             bool deferred_post = false, const uint32_t *const *group_index = nullptr,
                             const DirectWords *direct = nullptr)
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
    if (sparse && blocks) {
        // the reconstruction waves read the events themselves (recon_kernel.inl: coeff_row_from_events); round 2 had a
        // kernel of its own (k_expand) rebuild dense blocks in HBM first
        b->cur_first_event = g2.d_events;
        b->cur_events = direct ? g2.d_events + (size_t)b->n * direct->pitch_blocks : g2.d_events + blocks + 1;
        b->cur_n_events = direct ? (uint32_t)((size_t)b->n * direct->pitch_events) : (uint32_t)n_ev;
    }
    b->coeff_pool_blocks = direct ? (uint64_t)b->n * direct->pitch_blocks : blocks;
    b->coeff_checked = true;
    if (sparse_rec) {
        b->cur_group_index = g2.d_index;
        b->cur_mb_base = g2.d_base + b->n;
    }
    {
        const auto t_sub0 = std::chrono::steady_clock::now();
        const int src = b->submit(picture_type, g2.d_mbs, g2.d_coeffs, g2.d_base, /*with_post=*/deferred_post && b->pending.valid, types);
        if (b->trace_host) b->host_ms[5] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_sub0).count();
        if (src != H263MI_OK) {
            (void)hipStreamSynchronize(cs);      // (no copy left behind that reads this slot)
            return src;
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

}  // extern "C"

// What the parser tasks of a call may use.  Two different limits:
//   * CPUs: the hardware threads and the affinity mask -- more runnable threads than that only take turns;
//   * CPU TIME: a container's quota (cgroup cpu.max: so many CPU-seconds per second, on a host that may have many more CPUs).
//     A quota does not limit how many threads run at once, it limits what they use together -- and a worker that spins for
//     its next task uses its CPU like one that parses.  A call is parse phase + a serial rest (packing, queueing, the caller),
//     so `quota` spinning threads hold the whole quota while a fifth of it does nothing; more than `quota` spinning threads
//     overdraw it and the kernel freezes the process for the rest of the scheduler period (32 spinning threads on a 16-CPU
//     quota: 72 k pictures/s end to end instead of 106 k).  Threads that PARK the moment they run out of work use what they
//     parse with: then half as many threads again as the quota has CPUs shorten the parse phase (64 streams: 3 pictures per
//     thread instead of 4) inside the same CPU time: 106 k -> 115-118 k pictures/s on the GPU boxes (16-CPU quota on a
//     256-thread host; profiles/r05_x_host_thread_plan.txt).  Beyond that the wake-ups cost more than the shorter phase gives.
// The launcher's LOCAL_WORLD_SIZE (torch.distributed.run, mpirun wrappers) divides both limits: the ranks of one job that
// share a node share its CPUs.
struct HostThreadPlan {
    uint32_t threads;      // parser threads of a call
    long spin_us;          // how long an idle worker spins before it parks
    uint32_t cpus;         // CPUs the process may run on at once (hardware, affinity; per rank)
    uint32_t quota_cpus;   // CPU-time quota in CPUs (per rank), 0 = none
};
static void host_cpu_limits(uint32_t &cpus, uint32_t &quota_cpus)
{
    cpus = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = std::min<uint32_t>(cpus, (uint32_t)std::max(1, CPU_COUNT(&set)));
    quota_cpus = 0;
    const char *cpu_max = getenv("H263MI_CGROUP_CPU_MAX");       // (tests: a file in the format of cgroup v2's cpu.max)
    if (FILE *f = fopen(cpu_max ? cpu_max : "/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32];
        long period = 0;
        if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0)
            quota_cpus = (uint32_t)std::max(1L, atol(quota) / period);
        fclose(f);
    }
    if (const char *lw = getenv("LOCAL_WORLD_SIZE")) {
        const long ranks = atol(lw);
        if (ranks > 1) {
            cpus = std::max<uint32_t>(1u, cpus / (uint32_t)ranks);
            if (quota_cpus) quota_cpus = std::max<uint32_t>(1u, quota_cpus / (uint32_t)ranks);
        }
    }
}
// n_tasks: the streams of the call; requested: the caller's n_threads (0 = choose)
static HostThreadPlan host_thread_plan(uint32_t n_tasks, uint32_t requested)
{
    HostThreadPlan p{};
    host_cpu_limits(p.cpus, p.quota_cpus);
    const bool quota_binds = p.quota_cpus && p.quota_cpus < p.cpus;
    static const bool oversubscribe = !(getenv("H263MI_QUOTA_OVERSUBSCRIBE") && getenv("H263MI_QUOTA_OVERSUBSCRIBE")[0] == '0');
    if (requested) {
        p.threads = requested;
    } else if (quota_binds && oversubscribe) {
        // the fewest threads that give the rounds of (quota + quota / 2) threads: 64 streams on a 16-CPU quota -> 3 rounds -> 22
        const uint32_t cap = std::min(p.cpus, p.quota_cpus + p.quota_cpus / 2);
        const uint32_t rounds = (n_tasks + cap - 1) / std::max(1u, cap);
        p.threads = rounds ? (n_tasks + rounds - 1) / rounds : 1;
    } else {
        p.threads = quota_binds ? p.quota_cpus : p.cpus;
    }
    p.threads = std::max(1u, std::min({p.threads, n_tasks ? n_tasks : 1u, 256u}));
    // more threads than the quota pays for: they must not spin
    p.spin_us = (quota_binds && p.threads > p.quota_cpus) ? 0 : WorkerPool::kSpinUsDefault;
    if (const char *e = getenv("H263MI_SPIN_US")) p.spin_us = atol(e);            // (probes)
    return p;
}

// N x decode_next_picture.  stream_rc == nullptr: all or nothing (any stream's error fails the call, nothing changes).
// stream_rc != nullptr: every stream is its own H263State -- a stream that fails keeps its state (state.rs:142) and gets
// its error code, a stream without data (data[i] == nullptr) is left alone, the others advance.
static int batch_decode_next_pictures(h263mi_batch *b, uint32_t decoder_options, const uint8_t *const *data, const size_t *len,
                                      size_t *consumed, uint32_t n_threads, int *stream_rc, uint8_t strength, uint8_t *d_rgba,
                                      uint8_t *d_deblocked)
{
    if (!b || !data || !len || strength > 12) return H263MI_ERR_INVALID_ARGUMENT;
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
        if (deferred) render_rc = b->note_pending(strength, d_rgba, d_deblocked);
        else if (d_rgba || d_deblocked) render_rc = b->render(strength, d_rgba, d_deblocked, /*only_active=*/true);
    }
    for (uint32_t i = 0; i < n; i++) b->ss[i].active = was_active[i] != 0;
    RC_TRY(rc);
    RC_TRY(render_rc);
    return stream_rc ? H263MI_OK : first_error;
}

extern "C" {

uint32_t h263mi_default_parser_threads(uint32_t n_streams, uint32_t *cpu_quota)
{
    const HostThreadPlan p = host_thread_plan(n_streams, 0);
    if (cpu_quota) *cpu_quota = p.quota_cpus;
    return p.threads;
}

int h263mi_batch_decode_next_pictures(h263mi_batch *b, uint32_t decoder_options, const uint8_t *const *data,
                                      const size_t *len, size_t *consumed, uint32_t n_threads)
{
    return batch_decode_next_pictures(b, decoder_options, data, len, consumed, n_threads, nullptr, 0, nullptr, nullptr);
}

int h263mi_batch_decode_next_pictures_ex(h263mi_batch *b, uint32_t decoder_options, const uint8_t *const *data,
                                         const size_t *len, size_t *consumed, uint32_t n_threads, int *stream_rc,
                                         uint8_t strength, uint8_t *d_rgba, uint8_t *d_deblocked)
{
    return batch_decode_next_pictures(b, decoder_options, data, len, consumed, n_threads, stream_rc, strength, d_rgba, d_deblocked);
}

int h263mi_batch_render_rgba(h263mi_batch *b, uint8_t strength, uint8_t *d_rgba, uint8_t *d_deblocked)
{
    if (!b || (!d_rgba && !d_deblocked)) return H263MI_ERR_INVALID_ARGUMENT;
    DeviceGuard g(b->device);
    return b->render(strength, d_rgba, d_deblocked);
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

// =========================================================================================
// streams of DIFFERENT picture sizes behind one call (h263mi_mixed_*)
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
    std::vector<int> cls;                       // per stream: index into `classes`, -1 = no picture yet
    std::vector<int> slot;                      // per stream: its slot in that class
    std::vector<int> late_rc;                   // per stream: a device error found while its class was rebuilt, reported by the next sync
    std::vector<bits::ParserContext> parser_ctx;
    std::vector<bits::ParsedPicture> parsed;
    std::unique_ptr<WorkerPool> pool;
    // What the frame stores of all classes together may take (0 = no limit).  The sizes come out of untrusted bitstreams:
    // without a limit one hostile key frame of 16 384 x 16 384 asks for 800 MB per stream that sends one.
    uint64_t limit_bytes = 0;
    ~h263mi_mixed()
    {
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
        if (!pool || pool->size() < want) pool.reset(new WorkerPool(want - 1));
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
    void drop_empty_classes(const std::vector<int> &target)
    {
        for (size_t k = 0; k < classes.size(); k++) {
            if (!classes[k].b || classes[k].members()) continue;
            bool wanted = false;
            for (uint32_t i = 0; i < n && !wanted; i++) wanted = target[i] == (int)k;
            if (wanted) continue;
            // (a rendering of a picture a departed stream left behind is delivered first; the destructor waits for it)
            if (classes[k].b->pending.valid) (void)classes[k].b->flush_pending();
            delete classes[k].b;
            classes[k] = SizeClass();
        }
    }
    // a batch of `slots` slots for class k in the place of the one it has (or of none): the members move over, slot by slot
    // from 0 on.  The old batch is synced first (a device error found there is kept in late_rc for h263mi_mixed_sync).
    // `tentative` (may be null): per stream, the slot a stream that is about to JOIN the class has been promised (such a
    // stream sits in stream_of_slot already but still lives in its old class: its `slot` entry is not ours to touch)
    int rebuild_class(size_t k, uint32_t slots, std::vector<int> *tentative = nullptr)
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
                else if (tentative) (*tentative)[i] = (int)s;
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
    if (!m || !data || !len || !stream_rc || strength > 12 || (d_rgba && !rgba_capacity)) return H263MI_ERR_INVALID_ARGUMENT;
    const uint32_t n = m->n;
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
    struct Want { uint32_t w, h; };
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
            // marks for drop_empty_classes: the joiners' targets are not set yet, every empty class is dropped
            m->drop_empty_classes(target);
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
    // a class that stands three quarters empty gives the room back (before its launch of this call, when it has one coming)
    for (size_t k = 0; k < m->classes.size(); k++) {
        h263mi_mixed::SizeClass &c = m->classes[k];
        if (!c.b || c.b->n < 8) continue;
        const uint32_t mem = c.members();
        // (the joiners of this call are in stream_of_slot already: they move along, their promised slots are updated; a
        // failure leaves the class as it was -- roomy, but whole)
        if (mem && mem * 4 <= c.b->n) (void)m->rebuild_class(k, h263mi_mixed::pow2_at_least(mem), &new_slot);
    }
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
                if (deferred) render_rc = b->note_pending(strength, nullptr, nullptr, out_ptrs.data());
                else if (any_out) render_rc = b->render(strength, nullptr, nullptr, /*only_active=*/true, out_ptrs.data());
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

// ---------------------------------------------------------------------------------------
// H263State
// ---------------------------------------------------------------------------------------
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
    RC_TRY(state_ensure_staging(g2, total, n_coeff_blocks ? n_coeff_blocks : 1, event_words));
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
    if (sparse && n_coeff_blocks) {
        memcpy(g2.h_events, first_event, (n_coeff_blocks + 1) * sizeof(uint32_t));
        if (n_events) memcpy(g2.h_events + n_coeff_blocks + 1, events, n_events * sizeof(uint32_t));
        HIP_TRY(hipMemcpyAsync(g2.d_events, g2.h_events, event_words * sizeof(uint32_t), hipMemcpyHostToDevice, b->stream));
        b->cur_first_event = g2.d_events;        // (read by the reconstruction waves themselves)
        b->cur_events = g2.d_events + n_coeff_blocks + 1;
        b->cur_n_events = (uint32_t)n_events;
    } else if (n_coeff_blocks) {
        memcpy(g2.h_coeffs, coeffs, n_coeff_blocks * 128);
        HIP_TRY(hipMemcpyAsync(g2.d_coeffs, g2.h_coeffs, n_coeff_blocks * 128, hipMemcpyHostToDevice, b->stream));
    }

    b->coeff_pool_blocks = n_coeff_blocks;
    b->coeff_checked = true;
    RC_TRY(b->submit(desc->picture_type, g2.d_mbs, g2.d_coeffs, nullptr));
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
        if (g.ok && state_ensure_staging(g2, total, 1, 1) == H263MI_OK && hipEventSynchronize(g2.done) == hipSuccess) {
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
    RC_TRY(b->render(strength, s->d_rgba, nullptr));
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
    RC_TRY(b->render(strength, static_cast<uint8_t *>(dev), nullptr));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return H263MI_OK;
}

// ---------------------------------------------------------------------------------------
// deblock::deblock and yuv::bt601::yuv420_to_rgba as plain functions over host buffers
// ---------------------------------------------------------------------------------------
struct TempBuf {
    void *p = nullptr;
    ~TempBuf() { if (p) (void)hipFree(p); }
};

// Device scratch of the plain-function entry points (deblock, yuv420_to_rgba): a caller in the style of Ruffle invokes
// them once per frame, so the frame and the output buffer on the device are kept per host thread and per device and only
// grow (round 2 paid two hipMalloc, a hipMemset and two hipFree per call).  What lies in the padding of the cached frame
// is never used by the kernel (bytes outside the picture are loaded from clamped addresses and dropped).
struct PlainScratch {
    int device = -1;
    uint8_t *frame = nullptr, *out = nullptr;
    size_t frame_cap = 0, out_cap = 0;
    void release()
    {
        if (frame) (void)hipFree(frame);
        if (out) (void)hipFree(out);
        frame = out = nullptr;
        frame_cap = out_cap = 0;
    }
    ~PlainScratch() { release(); }
    int reserve(int dev, size_t frame_bytes, size_t out_bytes)
    {
        if (dev != device) {
            release();
            device = dev;
        }
        if (frame_bytes > frame_cap) {
            if (frame) (void)hipFree(frame);
            frame = nullptr;
            frame_cap = 0;
            const size_t cap = frame_bytes + frame_bytes / 4;
            HIP_TRY(hipMalloc((void **)&frame, cap));
            HIP_TRY(hipMemset(frame, 0, cap));
            frame_cap = cap;
        }
        if (out_bytes > out_cap) {
            if (out) (void)hipFree(out);
            out = nullptr;
            out_cap = 0;
            const size_t cap = out_bytes + out_bytes / 4;
            HIP_TRY(hipMalloc((void **)&out, cap));
            out_cap = cap;
        }
        return H263MI_OK;
    }
};
static thread_local PlainScratch tls_plain;

int h263mi_deblock_on(const h263mi_backend_cfg *cfg, const uint8_t *data, size_t len, size_t width, uint8_t strength, uint8_t *out)
{
    // preconditions of deblock.rs:30,306 (debug_asserts in the reference)
    if (!data || !out || !width || len % width != 0 || strength < 1 || strength > 12) return H263MI_ERR_INVALID_ARGUMENT;
    const size_t height = len / width;
    if (!height || width > 65535 || height > 65535) return H263MI_ERR_INVALID_ARGUMENT;
    if (!layout_fits(width, height)) return H263MI_ERR_OUT_OF_MEMORY;          // frame offsets are 32-bit on the device
    const int dev = cfg ? cfg->device_id : 0;
    hipStream_t stream = cfg ? (hipStream_t)cfg->stream : nullptr;
    RC_TRY(check_device(dev));
    DeviceGuard g(dev);
    const FrameLayout L = make_layout((uint32_t)width, (uint32_t)height);
    RC_TRY(tls_plain.reserve(dev, L.frame_bytes, len));
    HIP_TRY(hipMemcpy2DAsync(tls_plain.frame, L.pitch_y, data, width, width, height, hipMemcpyHostToDevice, stream));
    PostArgs a{};
    a.L = L;
    a.L.cwidth = a.L.cheight = 0;
    a.frames = tls_plain.frame;
    a.rgba = nullptr;
    a.planes_out = tls_plain.out;
    a.n_pictures = 1;
    a.strength = strength;
    set_post_tiles(a);
    a.luma_only = 1;
    HIP_TRY(launch_post(a, stream));
    HIP_TRY(hipMemcpyAsync(out, tls_plain.out, len, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return H263MI_OK;
}

int h263mi_deblock(const uint8_t *data, size_t len, size_t width, uint8_t strength, uint8_t *out)
{
    return h263mi_deblock_on(nullptr, data, len, width, strength, out);
}

int h263mi_bt601_yuv420_to_rgba_on(const h263mi_backend_cfg *cfg, const uint8_t *y, size_t y_len, const uint8_t *chroma_b,
                                   const uint8_t *chroma_r, size_t c_len, size_t y_width, uint8_t *rgba_out)
{
    if (y_len == 0) return H263MI_OK;                       // bt601.rs:107-112: empty in, empty out
    if (!y || !chroma_b || !chroma_r || !rgba_out || !y_width || y_len % y_width != 0) return H263MI_ERR_INVALID_ARGUMENT;
    const size_t h = y_len / y_width, cw = (y_width + 1) / 2, ch = (h + 1) / 2;   // bt601.rs:115-126
    if (c_len != cw * ch || y_width > 65535 || h > 65535) return H263MI_ERR_INVALID_ARGUMENT;
    if (!layout_fits(y_width, h)) return H263MI_ERR_OUT_OF_MEMORY;
    const int dev = cfg ? cfg->device_id : 0;
    hipStream_t stream = cfg ? (hipStream_t)cfg->stream : nullptr;
    RC_TRY(check_device(dev));
    DeviceGuard g(dev);
    const FrameLayout L = make_layout((uint32_t)y_width, (uint32_t)h);
    RC_TRY(tls_plain.reserve(dev, L.frame_bytes, y_len * 4));
    uint8_t *f = tls_plain.frame;
    HIP_TRY(hipMemcpy2DAsync(f, L.pitch_y, y, y_width, y_width, h, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpy2DAsync(f + L.off_cb, L.pitch_c, chroma_b, cw, cw, ch, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpy2DAsync(f + L.off_cr, L.pitch_c, chroma_r, cw, cw, ch, hipMemcpyHostToDevice, stream));
    PostArgs a{};
    a.L = L;
    a.frames = f;
    a.rgba = tls_plain.out;
    a.planes_out = nullptr;
    a.n_pictures = 1;
    a.strength = 0;
    set_post_tiles(a);
    a.luma_only = 0;
    HIP_TRY(launch_post(a, stream));
    HIP_TRY(hipMemcpyAsync(rgba_out, tls_plain.out, y_len * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return H263MI_OK;
}

int h263mi_bt601_yuv420_to_rgba(const uint8_t *y, size_t y_len, const uint8_t *chroma_b, const uint8_t *chroma_r,
                                size_t c_len, size_t y_width, uint8_t *rgba_out)
{
    return h263mi_bt601_yuv420_to_rgba_on(nullptr, y, y_len, chroma_b, chroma_r, c_len, y_width, rgba_out);
}

// ---------------------------------------------------------------------------------------
// device memory helpers + synthetic records
// ---------------------------------------------------------------------------------------
int h263mi_device_count(int *count)
{
    if (!count) return H263MI_ERR_INVALID_ARGUMENT;
    *count = 0;
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) {
        *count = 0;
        return H263MI_ERR_NO_DEVICE;
    }
    return H263MI_OK;
}

int h263mi_device_malloc(int device_id, size_t bytes, void **out)
{
    if (!out) return H263MI_ERR_INVALID_ARGUMENT;
    RC_TRY(check_device(device_id));
    DeviceGuard g(device_id);
    HIP_TRY(hipMalloc(out, bytes ? bytes : 1));
    return H263MI_OK;
}

int h263mi_device_free(int device_id, void *p)
{
    RC_TRY(check_device(device_id));
    DeviceGuard g(device_id);
    HIP_TRY(hipFree(p));
    return H263MI_OK;
}

int h263mi_device_memcpy_h2d(int device_id, void *dst, const void *src, size_t bytes)
{
    RC_TRY(check_device(device_id));
    DeviceGuard g(device_id);
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return H263MI_OK;
}

int h263mi_device_memcpy_d2h(int device_id, void *dst, const void *src, size_t bytes)
{
    RC_TRY(check_device(device_id));
    DeviceGuard g(device_id);
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return H263MI_OK;
}

int h263mi_debug_fail_nth_hip_call(int n)
{
    if (n > 0) {
        g_fail_countdown.store(n, std::memory_order_relaxed);
        return n;
    }
    const int left = g_fail_countdown.exchange(-1, std::memory_order_relaxed);
    return left < 0 ? 0 : left;
}

int h263mi_host_alloc(size_t bytes, void **out)
{
    if (!out) return H263MI_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return H263MI_ERR_NO_DEVICE;
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocPortable | hipHostMallocMapped));
    return H263MI_OK;
}

int h263mi_host_free(void *p)
{
    if (!p) return H263MI_OK;
    HIP_TRY(hipHostFree(p));
    return H263MI_OK;
}

int h263mi_host_register(void *p, size_t bytes)
{
    if (!p || !bytes) return H263MI_ERR_INVALID_ARGUMENT;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return H263MI_ERR_NO_DEVICE;
    HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped));
    return H263MI_OK;
}

int h263mi_host_unregister(void *p)
{
    if (!p) return H263MI_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipHostUnregister(p));
    return H263MI_OK;
}

int h263mi_device_synchronize(int device_id)
{
    RC_TRY(check_device(device_id));
    DeviceGuard g(device_id);
    HIP_TRY(hipDeviceSynchronize());
    return H263MI_OK;
}

static int probe_bandwidth(const h263mi_backend_cfg *cfg, int mode, size_t bytes, int reps, double *gb_per_s, int *best_shape)
{
    if (!gb_per_s || mode < 0 || mode > 2 || bytes < (1u << 20) || reps < 1) return H263MI_ERR_INVALID_ARGUMENT;
    const int dev = cfg ? cfg->device_id : 0;
    RC_TRY(check_device(dev));
    DeviceGuard g(dev);
    hipStream_t stream = cfg ? (hipStream_t)cfg->stream : nullptr;
    bytes &= ~(size_t)15;
    TempBuf in, out;
    if (mode != 2) {
        HIP_TRY(hipMalloc(&in.p, bytes));
        HIP_TRY(hipMemsetAsync(in.p, 1, bytes, stream));
    }
    HIP_TRY(hipMalloc(&out.p, mode == 1 ? 16 : bytes));
    HIP_TRY(hipMemsetAsync(out.p, 0, mode == 1 ? 16 : bytes, stream));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) {
        (void)hipEventDestroy(e0);
        return H263MI_ERR_HIP;
    }
    // every launch shape of the mode (kernels.hip: probe_shapes), the fastest one is the box's ceiling
    hipError_t e = hipSuccess;
    float best_ms = 0.f;
    for (int shape = 0; shape < probe_shapes(mode) && e == hipSuccess; shape++) {
        e = launch_probe(mode, shape, in.p, out.p, bytes, stream);          // warm-up
        if (e == hipSuccess) e = hipEventRecord(e0, stream);
        for (int i = 0; i < reps && e == hipSuccess; i++) e = launch_probe(mode, shape, in.p, out.p, bytes, stream);
        if (e == hipSuccess) e = hipEventRecord(e1, stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float t = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&t, e0, e1);
        if (e == hipSuccess && t > 0.f && (best_ms == 0.f || t < best_ms)) {
            best_ms = t;
            if (best_shape) *best_shape = shape;
        }
    }
    const float ms = best_ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIP_TRY(e);
    const double moved = (mode == 0 ? 2.0 : 1.0) * (double)bytes * reps;
    *gb_per_s = ms > 0.f ? moved / (ms * 1e-3) / 1e9 : 0.0;
    return H263MI_OK;
}

int h263mi_probe_bandwidth(const h263mi_backend_cfg *cfg, int mode, size_t bytes, int reps, double *gb_per_s)
{
    return probe_bandwidth(cfg, mode, bytes, reps, gb_per_s, nullptr);
}

int h263mi_probe_bandwidth_shape(const h263mi_backend_cfg *cfg, int mode, size_t bytes, int reps, double *gb_per_s,
                                 const char **shape_name)
{
    int shape = 0;
    const int rc = probe_bandwidth(cfg, mode, bytes, reps, gb_per_s, &shape);
    if (shape_name) *shape_name = rc == H263MI_OK ? probe_shape_name(mode, shape) : "";
    return rc;
}

int h263mi_synth_picture_host(int kind, uint16_t width, uint16_t height, uint32_t stream_id, uint32_t frame_idx,
                              h263mi_mb_record *mbs, int16_t *coeffs, size_t coeff_capacity_blocks,
                              size_t *n_coeff_blocks)
{
    if (kind < 0 || kind > H263MI_SYNTH_P || !width || !height || !mbs) return H263MI_ERR_INVALID_ARGUMENT;
    const FrameLayout L = make_layout(width, height);
    const uint32_t n = L.mbw * L.mbh;
    size_t used = 0;
    for (uint32_t i = 0; i < n; i++) {
        MbRecord r = synth_mb_header(kind, stream_id, frame_idx, i);
        r.coeff_index = (uint32_t)used;
        for (int blk = 0; blk < 6; blk++) {
            if (!((r.cbp >> blk) & 1)) continue;
            if (coeffs) {
                if (used >= coeff_capacity_blocks) return H263MI_ERR_INVALID_ARGUMENT;
                synth_block_coeffs(kind, stream_id, frame_idx, i, blk, coeffs + used * 64);
            }
            used++;
        }
        mbs[i] = r;
    }
    if (n_coeff_blocks) *n_coeff_blocks = used;
    return H263MI_OK;
}

int h263mi_synth_batch_device(const h263mi_backend_cfg *cfg, int kind, uint16_t width, uint16_t height,
                              uint32_t n_streams, uint32_t first_stream_id, uint32_t frame_idx, h263mi_mb_record *d_mbs,
                              int16_t *d_coeffs, size_t coeff_capacity_blocks, uint64_t *d_coeff_base,
                              size_t *total_blocks)
{
    return h263mi_synth_batch_device_strided(cfg, kind, width, height, n_streams, first_stream_id, 1, frame_idx, d_mbs, d_coeffs,
                                             coeff_capacity_blocks, d_coeff_base, total_blocks);
}

int h263mi_synth_batch_device_strided(const h263mi_backend_cfg *cfg, int kind, uint16_t width, uint16_t height,
                                      uint32_t n_streams, uint32_t first_stream_id, uint32_t stream_stride, uint32_t frame_idx,
                                      h263mi_mb_record *d_mbs, int16_t *d_coeffs, size_t coeff_capacity_blocks,
                                      uint64_t *d_coeff_base, size_t *total_blocks)
{
    if (kind < 0 || kind > H263MI_SYNTH_P || !width || !height || !n_streams || !d_mbs || !d_coeffs || !d_coeff_base)
        return H263MI_ERR_INVALID_ARGUMENT;
    const int dev = cfg ? cfg->device_id : 0;
    RC_TRY(check_device(dev));
    DeviceGuard g(dev);
    hipStream_t stream = cfg ? (hipStream_t)cfg->stream : nullptr;
    const FrameLayout L = make_layout(width, height);
    SynthArgs a{};
    a.kind = kind;
    a.n_streams = n_streams;
    a.first_stream_id = first_stream_id;
    a.stream_stride = stream_stride;
    a.frame_idx = frame_idx;
    a.mbs_per_picture = L.mbw * L.mbh;
    a.mbs = d_mbs;
    a.coeffs = d_coeffs;
    a.coeff_base = d_coeff_base;
    TempBuf counts, totals;
    HIP_TRY(hipMalloc(&counts.p, (size_t)n_streams * a.mbs_per_picture * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&totals.p, (size_t)n_streams * sizeof(uint32_t)));
    a.counts = (uint32_t *)counts.p;
    a.totals = (uint32_t *)totals.p;
    HIP_TRY(launch_synth_headers(a, stream));
    std::vector<uint32_t> h_totals(n_streams);
    HIP_TRY(hipMemcpyAsync(h_totals.data(), totals.p, n_streams * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    std::vector<uint64_t> bases(n_streams);
    uint64_t run = 0;
    for (uint32_t p = 0; p < n_streams; p++) {
        bases[p] = run;
        run += h_totals[p];
    }
    if (total_blocks) *total_blocks = (size_t)run;
    if (run > coeff_capacity_blocks) return H263MI_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipMemcpyAsync(d_coeff_base, bases.data(), n_streams * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    HIP_TRY(launch_synth_coeffs(a, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return H263MI_OK;
}

}  // extern "C"
