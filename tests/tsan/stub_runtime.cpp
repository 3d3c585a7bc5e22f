// tests/tsan: the stub HIP runtime (hip_stub/hip/hip_runtime.h) and stub kernel launchers.  Test infrastructure only.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "../../h263-rs_amd/csrc/kernels.h"

namespace {
std::mutex g_m;
std::map<uintptr_t, size_t> g_allocs;            // "device" allocations: base -> size (hipMemGetAddressRange)
thread_local int tl_device = 0;
std::atomic<uint64_t> g_sink{0};                 // what the stub kernels "compute": keeps their reads alive

hipError_t alloc(void **p, size_t bytes)
{
    void *m = malloc(bytes ? bytes : 1);
    if (!m) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> l(g_m);
    g_allocs[(uintptr_t)m] = bytes ? bytes : 1;
    *p = m;
    return hipSuccess;
}
hipError_t release(void *p)
{
    if (!p) return hipSuccess;
    {
        std::lock_guard<std::mutex> l(g_m);
        g_allocs.erase((uintptr_t)p);
    }
    free(p);
    return hipSuccess;
}
}  // namespace

hipError_t hipGetDeviceCount(int *count) { *count = 2; return hipSuccess; }
hipError_t hipGetDevice(int *dev) { *dev = tl_device; return hipSuccess; }
hipError_t hipSetDevice(int dev) { tl_device = dev; return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipDeviceGetPCIBusId(char *id, int len, int dev) { snprintf(id, (size_t)len, "0000:%02x:00.0", 0xc1 + dev); return hipSuccess; }
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) { *free_b = *total_b = (size_t)1 << 34; return hipSuccess; }
hipError_t hipMalloc(void **p, size_t bytes) { return alloc(p, bytes); }
hipError_t hipFree(void *p) { return release(p); }
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { return alloc(p, bytes); }
hipError_t hipHostFree(void *p) { return release(p); }
hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void *) { return hipSuccess; }
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned) { *dev = host; return hipSuccess; }
hipError_t hipMemGetAddressRange(hipDeviceptr_t *base, size_t *size, hipDeviceptr_t p)
{
    std::lock_guard<std::mutex> l(g_m);
    auto it = g_allocs.upper_bound((uintptr_t)p);
    if (it == g_allocs.begin()) return hipErrorInvalidValue;
    --it;
    if ((uintptr_t)p >= it->first + it->second) return hipErrorInvalidValue;
    *base = (void *)it->first;
    *size = it->second;
    return hipSuccess;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind) { memcpy(dst, src, bytes); return hipSuccess; }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind, hipStream_t) { memcpy(dst, src, bytes); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t)
{
    for (size_t r = 0; r < height; r++) memcpy((char *)dst + r * dpitch, (const char *)src + r * spitch, width);
    return hipSuccess;
}
hipError_t hipMemset(void *p, int v, size_t bytes) { memset(p, v, bytes); return hipSuccess; }
hipError_t hipMemsetAsync(void *p, int v, size_t bytes, hipStream_t) { memset(p, v, bytes); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)malloc(1); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.01f; return hipSuccess; }

// ---- the "kernels": they read what a launch would read first -- the per-stream words, bases and the sparse-record index the
// host threads have just written and the caller has just copied -- so that ThreadSanitizer sees those reads
namespace h263mi {

static void touch_recon(const ReconArgs &a)
{
    uint64_t s = 0;
    if (a.coeff_base)
        for (uint32_t i = 0; i < a.n_pictures; i++) s += a.coeff_base[i];
    if (a.stream_state)
        for (uint32_t i = 0; i < a.n_pictures; i++) s += a.stream_state[i];
    if (a.mb_group_index)
        for (size_t i = 0; i < (size_t)a.n_pictures * a.groups_per_picture; i++) s += a.mb_group_index[i];
    g_sink.fetch_add(s, std::memory_order_relaxed);
}
static void touch_words(const uint32_t *words, uint32_t n)
{
    uint64_t s = 0;
    for (uint32_t i = 0; words && i < n; i++) s += words[i];
    g_sink.fetch_add(s, std::memory_order_relaxed);
}
hipError_t launch_recon(const ReconArgs &a, hipStream_t, const uint32_t *words) { touch_recon(a); touch_words(words, a.n_pictures); return hipSuccess; }
hipError_t launch_frame(const ReconArgs &a, const PostArgs &p, hipStream_t, bool, const uint32_t *words)
{
    touch_recon(a);
    touch_words(words, a.n_pictures);
    if (p.stream_state) g_sink.fetch_add(p.stream_state[0], std::memory_order_relaxed);
    return hipSuccess;
}
hipError_t launch_post(const PostArgs &p, hipStream_t, const uint32_t *words)
{
    touch_words(words, p.n_pictures);
    if (p.stream_state) g_sink.fetch_add(p.stream_state[0], std::memory_order_relaxed);
    return hipSuccess;
}
hipError_t launch_synth_headers(const SynthArgs &, hipStream_t) { return hipSuccess; }
hipError_t launch_synth_coeffs(const SynthArgs &, hipStream_t) { return hipSuccess; }
int probe_shapes(int) { return 1; }
const char *probe_shape_name(int, int) { return "stub"; }
hipError_t launch_probe(int, int, const void *, void *, size_t, hipStream_t) { return hipSuccess; }

}  // namespace h263mi
