// worker_pool.cpp -- see worker_pool.h.  No HIP in here.
#include "worker_pool.h"

#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace h263mi {

// ---------------------------------------------------------------------------------------------------------------------
// placement
// ---------------------------------------------------------------------------------------------------------------------
bool parse_cpu_list(const char *text, cpu_set_t *out)
{
    CPU_ZERO(out);
    bool any = false;
    const char *p = text;
    while (p && *p) {
        while (*p == ' ' || *p == ',' || *p == '\n' || *p == '\t') p++;
        if (!*p) break;
        char *end = nullptr;
        const long lo = strtol(p, &end, 10);
        if (end == p || lo < 0) return false;
        long hi = lo;
        p = end;
        if (*p == '-') {
            hi = strtol(p + 1, &end, 10);
            if (end == p + 1 || hi < lo) return false;
            p = end;
        }
        for (long c = lo; c <= hi && c < CPU_SETSIZE; c++) {
            CPU_SET((int)c, out);
            any = true;
        }
    }
    return any;
}

static bool read_small_file(const std::string &path, char *buf, size_t cap)
{
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return false;
    const size_t n = fread(buf, 1, cap - 1, f);
    fclose(f);
    buf[n] = 0;
    return n > 0;
}

static std::string sysfs_root_or_default(const char *sysfs_root)
{
    if (sysfs_root && sysfs_root[0]) return sysfs_root;
    if (const char *e = getenv("H263MI_SYSFS_ROOT")) return e;          // (tests: a topology of their own)
    return "/sys";
}

static int numa_node_of_pci(const std::string &root, const std::string &pci_bus_id)
{
    if (pci_bus_id.empty()) return -1;
    std::string id = pci_bus_id;
    for (char &c : id) c = (char)tolower((unsigned char)c);             // sysfs spells the address in lower case
    char buf[64];
    if (!read_small_file(root + "/bus/pci/devices/" + id + "/numa_node", buf, sizeof buf)) return -1;
    return atoi(buf);
}

HostPlacement host_placement(const std::vector<std::string> &pci_ids, int device, uint32_t ranks, const char *sysfs_root)
{
    HostPlacement p;
    if (const char *e = getenv("H263MI_NUMA"))
        if (e[0] == '0') return p;
    const std::string root = sysfs_root_or_default(sysfs_root);
    int node = -1;
    bool forced = false;
    if (const char *e = getenv("H263MI_NUMA_NODE")) {                   // A/B runs: the far socket on purpose
        node = atoi(e);
        forced = true;
    } else if (device >= 0 && (size_t)device < pci_ids.size()) {
        node = numa_node_of_pci(root, pci_ids[(size_t)device]);
    }
    if (node < 0) return p;
    char buf[4096];
    cpu_set_t node_cpus, have;
    if (!read_small_file(root + "/devices/system/node/node" + std::to_string(node) + "/cpulist", buf, sizeof buf) ||
        !parse_cpu_list(buf, &node_cpus))
        return p;
    if (sched_getaffinity(0, sizeof have, &have) != 0) return p;
    CPU_AND(&node_cpus, &node_cpus, &have);
    if (CPU_COUNT(&node_cpus) == 0) return p;                           // (the process may not run there: leave it alone)
    p.node = node;
    p.cpus = node_cpus;
    p.have_cpus = true;
    // Ranks that share the node share its CPUs: the ranks of a job use the devices 0 .. ranks - 1 (LOCAL_RANK = device), so
    // the devices among those that hang off THIS node take disjoint, equal slices of its CPUs in device order -- no rank
    // parses on another's cores, with no communication between them.  (Hyperthread siblings are numbered far apart -- k and
    // k + cores -- so a slice is cut out of the list of CORES, each with all its siblings.)
    if (!forced && ranks > 1 && device >= 0 && (uint32_t)device < ranks) {
        uint32_t on_node = 0, my_pos = 0;
        for (uint32_t d = 0; d < ranks && d < pci_ids.size(); d++) {
            if (numa_node_of_pci(root, pci_ids[d]) != node) continue;
            if ((int)d < device) my_pos++;
            on_node++;
        }
        if (on_node > 1) {
            // cores of the node = sets of siblings, identified by their lowest CPU
            std::vector<int> core_first;
            std::vector<cpu_set_t> core_set;
            for (int cpu = 0; cpu < CPU_SETSIZE; cpu++) {
                if (!CPU_ISSET(cpu, &node_cpus)) continue;
                cpu_set_t sib;
                CPU_ZERO(&sib);
                if (!read_small_file(root + "/devices/system/cpu/cpu" + std::to_string(cpu) + "/topology/thread_siblings_list", buf, sizeof buf) ||
                    !parse_cpu_list(buf, &sib))
                    CPU_SET(cpu, &sib);
                int first = cpu;
                for (int c = 0; c < cpu; c++)
                    if (CPU_ISSET(c, &sib) && CPU_ISSET(c, &node_cpus)) { first = c; break; }
                size_t k = 0;
                for (; k < core_first.size(); k++)
                    if (core_first[k] == first) break;
                if (k == core_first.size()) {
                    core_first.push_back(first);
                    cpu_set_t empty;
                    CPU_ZERO(&empty);
                    core_set.push_back(empty);
                }
                CPU_SET(cpu, &core_set[k]);
            }
            const size_t cores = core_first.size();
            if (cores >= on_node) {
                const size_t lo = cores * my_pos / on_node, hi = cores * (my_pos + 1) / on_node;
                cpu_set_t slice;
                CPU_ZERO(&slice);
                for (size_t k = lo; k < hi; k++) CPU_OR(&slice, &slice, &core_set[k]);
                if (CPU_COUNT(&slice) > 0) p.cpus = slice;
            }
        }
    }
    return p;
}

HostPlacement host_placement_for_device(const char *pci_bus_id, const char *sysfs_root)
{
    std::vector<std::string> ids;
    ids.push_back(pci_bus_id ? pci_bus_id : "");
    return host_placement(ids, 0, 1, sysfs_root);
}

// set_mempolicy(2) without libnuma
static long mempolicy(int mode, const unsigned long *mask, unsigned long maxnode)
{
#if defined(SYS_set_mempolicy)
    return syscall(SYS_set_mempolicy, mode, mask, maxnode);
#else
    (void)mode; (void)mask; (void)maxnode;
    return -1;
#endif
}

PlacementScope::PlacementScope(const HostPlacement &p)
{
    if (p.node < 0 || p.node >= 1024) return;
    unsigned long mask[1024 / (8 * sizeof(unsigned long))] = {0};
    mask[(size_t)p.node / (8 * sizeof(unsigned long))] |= 1ul << ((size_t)p.node % (8 * sizeof(unsigned long)));
    constexpr int kMpolPreferred = 1;
    active_ = mempolicy(kMpolPreferred, mask, 1024) == 0;               // (preferred, not bind: a full node falls back)
}

PlacementScope::~PlacementScope()
{
    constexpr int kMpolDefault = 0;
    if (active_) (void)mempolicy(kMpolDefault, nullptr, 0);
}

int numa_node_of_address(const void *p)
{
#if defined(SYS_get_mempolicy)
    int node = -1;
    constexpr unsigned long kMpolFNode = 1, kMpolFAddr = 2;
    if (syscall(SYS_get_mempolicy, &node, nullptr, 0ul, p, kMpolFNode | kMpolFAddr) != 0) return -1;
    return node;
#else
    (void)p;
    return -1;
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// the pool
// ---------------------------------------------------------------------------------------------------------------------
WorkerPool::WorkerPool(unsigned workers, const HostPlacement *placement)
{
    CPU_ZERO(&confined_);
    for (unsigned t = 0; t < workers; t++) threads_.emplace_back([this, t] { loop(t + 1); });
    place_threads(placement);
}

WorkerPool::~WorkerPool()
{
    {
        std::lock_guard<std::mutex> l(m_);
        stop_.store(true, std::memory_order_release);
        generation_.store(((generation_.load(std::memory_order_relaxed) >> 32) + 1) << 32, std::memory_order_release);
    }
    wake_.notify_all();
    for (std::thread &t : threads_) t.join();
}

void WorkerPool::run(unsigned k, const std::function<void(unsigned)> &fn, long spin_us)
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
#if defined(H263MI_TSAN_BREAK_GENERATION_ORDER)
        // DELIBERATELY BROKEN (tests/tsan only): the task is published with a relaxed store -- the workers read fn_ / pending_
        // without a happens-before edge to the writes above.  ThreadSanitizer must report it (test_tsan.py), which proves that
        // the clean run of the real ordering means something.
        const uint64_t gen = (generation_.load(std::memory_order_relaxed) >> 32) + 1;
        generation_.store((gen << 32) | k, std::memory_order_relaxed);
#else
        const uint64_t gen = (generation_.load(std::memory_order_relaxed) >> 32) + 1;
        generation_.store((gen << 32) | k, std::memory_order_release);
#endif
    }
    if (parked_.load(std::memory_order_acquire)) wake_.notify_all();
    fn(0);
    for (unsigned spins = 0; pending_.load(std::memory_order_acquire) != 0; spins++) {
        if (spins < 20000) cpu_relax();
        else std::this_thread::yield();
    }
    fn_ = nullptr;
}

// The parser is a chain of dependent table look-ups: two of its threads on the two hyperthreads of one core run at
// 60-70 % each.  On a host with many more cores than worker threads (the GPU boxes: 16 CPUs of quota on 128 cores / 256
// hyperthreads) the workers are therefore confined to ONE hyperthread per physical core -- the lowest-numbered of each
// sibling set, within the CPUs they may use -- and the scheduler spreads them over distinct cores.  "The CPUs they may use":
// the affinity mask the process already has, cut down to the NUMA node (or the rank's slice of it) of the batch's device when
// that is known (HostPlacement).  Only the pool's own threads are touched (never the caller's); H263MI_PIN_THREADS=0 leaves
// them alone altogether.
void WorkerPool::place_threads(const HostPlacement *placement)
{
    const char *env = getenv("H263MI_PIN_THREADS");
    if (env && env[0] == '0') return;
    cpu_set_t have, want;
    if (sched_getaffinity(0, sizeof have, &have) != 0) return;
    bool narrowed = false;
    if (placement && placement->have_cpus) {
        cpu_set_t both;
        CPU_AND(&both, &have, &placement->cpus);
        if (CPU_COUNT(&both) > 0 && !CPU_EQUAL(&both, &have)) {
            have = both;
            narrowed = true;
        }
    }
    CPU_ZERO(&want);
    const std::string root = sysfs_root_or_default(nullptr);
    for (int cpu = 0; cpu < CPU_SETSIZE; cpu++) {
        if (!CPU_ISSET(cpu, &have)) continue;
        char buf[256];
        int first = cpu;
        if (read_small_file(root + "/devices/system/cpu/cpu" + std::to_string(cpu) + "/topology/thread_siblings_list", buf, sizeof buf))
            first = atoi(buf);                                   // "3,131" or "3-4": the list starts with its lowest member
        if (first == cpu || !CPU_ISSET(first, &have)) CPU_SET(cpu, &want);
    }
    // one hyperthread per core only when that still leaves room to spread: at least twice as many cores as workers
    const bool one_per_core = (unsigned)CPU_COUNT(&want) >= 2 * (unsigned)threads_.size() + 2 && !CPU_EQUAL(&want, &have);
    if (!one_per_core && !narrowed) return;
    const cpu_set_t &use = one_per_core ? want : have;
    for (std::thread &t : threads_) (void)pthread_setaffinity_np(t.native_handle(), sizeof use, &use);
    confined_ = use;
}

void WorkerPool::loop(unsigned id)
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

// ---------------------------------------------------------------------------------------------------------------------
// how many threads
// ---------------------------------------------------------------------------------------------------------------------
namespace {

struct HostLimits {
    uint32_t cpus = 1, quota_cpus = 0, env_ranks = 1;
    bool oversubscribe = true;
    long spin_us_override = -1;
};

// read once per process: the affinity mask, the cgroup's cpu.max and the environment do not change under a running decoder
// (and getenv on every decode call raced with a caller's setenv)
const HostLimits &host_limits()
{
    static const HostLimits limits = [] {
        HostLimits l;
        l.cpus = std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) l.cpus = std::min<uint32_t>(l.cpus, (uint32_t)std::max(1, CPU_COUNT(&set)));
        const char *cpu_max = getenv("H263MI_CGROUP_CPU_MAX");       // (tests: a file in the format of cgroup v2's cpu.max)
        if (FILE *f = fopen(cpu_max ? cpu_max : "/sys/fs/cgroup/cpu.max", "r")) {
            char quota[32];
            long period = 0;
            if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0)
                l.quota_cpus = (uint32_t)std::max(1L, atol(quota) / period);
            fclose(f);
        }
        // processes that share the node: said outright, or -- the fallback -- what the launcher exports
        const char *r = getenv("H263MI_RANKS_PER_NODE");
        if (!r) r = getenv("LOCAL_WORLD_SIZE");
        if (r && atol(r) > 1) l.env_ranks = (uint32_t)atol(r);
        const char *o = getenv("H263MI_QUOTA_OVERSUBSCRIBE");
        l.oversubscribe = !(o && o[0] == '0');
        if (const char *e = getenv("H263MI_SPIN_US")) l.spin_us_override = atol(e);        // (probes)
        return l;
    }();
    return limits;
}

std::atomic<uint32_t> g_ranks_per_node{0};           // h263mi_set_ranks_per_node; 0 = the environment

}  // namespace

void set_ranks_per_node(uint32_t ranks) { g_ranks_per_node.store(ranks, std::memory_order_relaxed); }

HostThreadPlan host_thread_plan(uint32_t n_tasks, uint32_t requested)
{
    const HostLimits &l = host_limits();
    HostThreadPlan p{};
    const uint32_t said = g_ranks_per_node.load(std::memory_order_relaxed);
    p.ranks = std::max(1u, said ? said : l.env_ranks);
    p.cpus = std::max(1u, l.cpus / p.ranks);
    p.quota_cpus = l.quota_cpus ? std::max(1u, l.quota_cpus / p.ranks) : 0u;
    const bool quota_binds = p.quota_cpus && p.quota_cpus < p.cpus;
    if (requested) {
        p.threads = requested;
    } else if (quota_binds && l.oversubscribe) {
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
    if (l.spin_us_override >= 0) p.spin_us = l.spin_us_override;
    return p;
}

}  // namespace h263mi
