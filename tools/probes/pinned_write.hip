// pinned_write.hip -- how fast do CPU threads WRITE page-locked host memory (hipHostMalloc, the staging the parser threads
// fill) compared with ordinary memory?  T threads store 32-byte records, 261 KB each per "picture", into a 16.7 MB buffer.
//   hipcc -O2 -pthread tools/probes/pinned_write.hip -o /tmp/pinned_write && /tmp/pinned_write
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
struct Rec { uint32_t w[8]; };
static double run(Rec *buf, int T, int iters)
{
    std::atomic<bool> go{false};
    std::atomic<int> ready{0};
    std::vector<std::thread> th;
    const size_t per = 8160;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            ready++;
            while (!go.load()) std::this_thread::yield();
            for (int i = 0; i < iters; i++) {
                Rec *dst = buf + (size_t)((t * 4 + i) % 64) * per;
                for (size_t k = 0; k < per; k++) { Rec r = {{(uint32_t)k, 1, 2, 3, 4, 5, 6, (uint32_t)i}}; dst[k] = r; }
            }
        });
    while (ready.load() < T) std::this_thread::yield();
    const auto t0 = std::chrono::steady_clock::now();
    go = true;
    for (auto &x : th) x.join();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
int main()
{
    const size_t bytes = 64 * 8160 * sizeof(Rec);
    Rec *pinned = nullptr, *pinned_nc = nullptr, *plain = (Rec *)aligned_alloc(4096, bytes);
    if (hipHostMalloc((void **)&pinned, bytes, hipHostMallocDefault) != hipSuccess) return 1;
    if (hipHostMalloc((void **)&pinned_nc, bytes, hipHostMallocNonCoherent) != hipSuccess) pinned_nc = nullptr;
    memset(plain, 0, bytes); memset(pinned, 0, bytes); if (pinned_nc) memset(pinned_nc, 0, bytes);
    for (int T : {1, 4, 16}) {
        const int iters = 400;
        const double a = run(plain, T, iters), b = run(pinned, T, iters), c = pinned_nc ? run(pinned_nc, T, iters) : 0;
        printf("%2d threads: 261 KB of records written in %.1f us (malloc), %.1f us (hipHostMalloc default), %.1f us (non-coherent)\n", T,
               a / iters * 1e6, b / iters * 1e6, c / iters * 1e6);
    }
    return 0;
}
