// worker_pool.h -- the host threads of a batch (parser tasks, packing into pinned staging), how many of them a call uses
// (HostThreadPlan) and where they run (HostPlacement: the NUMA node of the batch's GPU).  No HIP in here: tests/tsan builds
// this file and its users with g++ -fsanitize=thread.
//
// The reference is single-threaded safe Rust (`&mut self`, state.rs:138-141): one H263State, one thread.  A batch is N such
// states advancing together; their serial parses are independent and run side by side on these threads.
#pragma once

#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace h263mi {

// Where the host side of one GPU's work belongs (round 6).  On a two-socket node every GPU hangs off one socket: parser
// threads on the other socket write the pinned staging memory across the inter-socket link and the DMA engine reads it back
// across it.  A batch therefore looks up the NUMA node of its device (sysfs: numa_node of the PCI function), confines its pool's
// threads to that node's CPUs (within the affinity mask the process already has) and has its staging memory placed there
// (PlacementScope, batch_staging.cpp).  H263MI_NUMA=0 switches all of it off; H263MI_NUMA_NODE=k forces node k (A/B runs:
// the far socket on purpose).  Unknown topology (no sysfs entry, node -1, a single node) = no placement, as before.
struct HostPlacement {
    int node = -1;                 // NUMA node of the device, -1 = unknown / switched off
    bool have_cpus = false;        // `cpus` holds that node's CPUs within the process's affinity mask (and is not empty)
    cpu_set_t cpus;
    HostPlacement() { CPU_ZERO(&cpus); }
};
// pci_ids[d]: "0000:c1:00.0" as hipDeviceGetPCIBusId prints it for device d ("" = unknown); device: the batch's; ranks: the
// processes that share the node (HostThreadPlan::ranks) -- with more than one, the devices 0 .. ranks - 1 that hang off the same
// node take disjoint slices of its cores in device order.  sysfs_root: nullptr = H263MI_SYSFS_ROOT or "/sys" (tests bring a
// tree of their own).
HostPlacement host_placement(const std::vector<std::string> &pci_ids, int device, uint32_t ranks, const char *sysfs_root = nullptr);
HostPlacement host_placement_for_device(const char *pci_bus_id, const char *sysfs_root = nullptr);
// "0-15,128-143" -> set (the format of /sys/devices/system/node/nodeK/cpulist); false when nothing could be parsed
bool parse_cpu_list(const char *text, cpu_set_t *out);
// While one of these lives, memory the calling thread allocates (and first touches) is placed on the placement's node when
// there is room (set_mempolicy MPOL_PREFERRED; nothing happens for node -1).  Around the hipHostMalloc of a staging slot.
class PlacementScope {
public:
    explicit PlacementScope(const HostPlacement &p);
    ~PlacementScope();
    PlacementScope(const PlacementScope &) = delete;
    PlacementScope &operator=(const PlacementScope &) = delete;
private:
    bool active_ = false;
};
// the NUMA node the page at `p` lives on (get_mempolicy MPOL_F_NODE | MPOL_F_ADDR), -1 = unknown; the page is touched first
int numa_node_of_address(const void *p);

// Host worker threads of a batch: created once and parked between calls -- h263mi_batch_decode_next_pictures used to start and
// join two sets of threads per frame index.  run(k, fn) executes fn(0) .. fn(k - 1), fn(0) on the calling thread, and returns
// when all are done; calls do not nest or overlap (a batch is driven from one thread at a time).
// A server calls the batch entries back to back, a millisecond apart: waking 15 parked threads through a condition
// variable cost 50-100 us of every call (twice: parser tasks, then packing).  A worker therefore SPINS on the generation
// counter for a short while after it has finished a task (spin_us) and only then parks; the caller spins likewise while
// it waits for the last worker.  An idle batch costs nothing: everybody is parked.
// Under a CPU-TIME quota (a container's cpu.max) spinning is paid for like parsing: see HostThreadPlan, which then runs more
// threads than the quota has CPUs and has them park at once.
//
// Ordering (checked under ThreadSanitizer, tests/tsan): the task (fn_, spin_us_, pending_) is published by the release store
// of generation_ and read behind its acquire load; a worker's results are published by its release decrement of pending_ and
// read behind the caller's acquire load of 0.  The generation is bumped under the mutex so that a worker which has found the
// old value and is about to wait cannot miss the notify (lost wake-up).
class WorkerPool {
public:
    static constexpr long kSpinUsDefault = 300;
    // placement (may be null): confine the pool's threads to these CPUs (the device's NUMA node)
    explicit WorkerPool(unsigned workers, const HostPlacement *placement = nullptr);
    ~WorkerPool();
    unsigned size() const { return (unsigned)threads_.size() + 1; }       // the caller counts
    // spin_us: how long a worker that has finished spins for the next task before it parks (see loop())
    void run(unsigned k, const std::function<void(unsigned)> &fn, long spin_us = kSpinUsDefault);
    // the CPUs the pool's threads were confined to (tests, bench report): empty set = left alone
    cpu_set_t confined_to() const { return confined_; }

private:
    void place_threads(const HostPlacement *placement);
    void loop(unsigned id);
    static void cpu_relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable wake_;
    const std::function<void(unsigned)> *fn_ = nullptr;
    std::atomic<long> spin_us_{kSpinUsDefault};
    std::atomic<unsigned> pending_{0}, parked_{0};
    std::atomic<uint64_t> generation_{0};
    std::atomic<bool> stop_{false};
    cpu_set_t confined_;
};

// Streams dealt to host threads with AFFINITY: thread t of T first takes the streams t, t + T, t + 2T, ... -- the same ones
// in every call, so that a stream's parse buffers and its slot of the staging memory (261 KB of records per 1080p picture)
// stay in that core's caches instead of migrating between cores from call to call -- and then helps out with whatever
// the other threads have not started yet (a stream is claimed with one atomic exchange; what the owner of a stream wrote
// is published to the caller by WorkerPool::run's pending_ protocol, not by this flag).
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
// Both limits are divided by the number of processes that share the node: what h263mi_set_ranks_per_node said, else
// H263MI_RANKS_PER_NODE, else the launcher's LOCAL_WORLD_SIZE (torch.distributed.run, mpirun wrappers).
struct HostThreadPlan {
    uint32_t threads;      // parser threads of a call
    long spin_us;          // how long an idle worker spins before it parks
    uint32_t cpus;         // CPUs the process may run on at once (hardware, affinity; per rank)
    uint32_t quota_cpus;   // CPU-time quota in CPUs (per rank), 0 = none
    uint32_t ranks;        // processes the node is shared with (>= 1)
};
// n_tasks: the streams of the call; requested: the caller's n_threads (0 = choose).  The limits of the host (affinity mask,
// cgroup quota, environment switches) are read ONCE per process; a call only derives its thread count from them.
HostThreadPlan host_thread_plan(uint32_t n_tasks, uint32_t requested);
// 0 = back to the environment (H263MI_RANKS_PER_NODE, LOCAL_WORLD_SIZE)
void set_ranks_per_node(uint32_t ranks);

}  // namespace h263mi
