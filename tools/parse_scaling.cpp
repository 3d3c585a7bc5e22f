// parse_scaling.cpp -- does the host parser scale with threads on this machine?  T threads parse the same coded picture
// into buffers of their own, `iters` times each, no GPU anywhere: pictures/s in total and per thread, T = 1, 2, 4, 8, 12,
// 16, 20, 24, 32.  `ext` = the records go to a caller's array of 64 x the picture (as the batch entry's pinned staging:
// a different 261 KB each time) instead of the parser's own (cache-resident) array.
//   g++ -O3 -std=c++17 -pthread -Iinclude -o /tmp/parse_scaling tools/parse_scaling.cpp h263-rs_amd/host/bitstream.cpp
//   /tmp/parse_scaling picture.bin [iters]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include "../h263-rs_amd/host/bitstream.hpp"
using namespace h263mi::bits;
int main(int argc, char **argv)
{
    if (argc < 2) return 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    std::vector<uint8_t> d;
    uint8_t buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
    fclose(f);
    const int iters = argc > 2 ? atoi(argv[2]) : 400;
    for (int ext = 0; ext < 2; ext++)
        for (int T : {1, 2, 4, 8, 12, 16, 20, 24, 32}) {
            std::atomic<int> ready{0};
            std::atomic<bool> go{false};
            std::vector<std::thread> th;
            std::vector<double> secs(T);
            for (int t = 0; t < T; t++)
                th.emplace_back([&, t] {
                    ParsedPicture p;
                    p.want_dense = false;
                    std::vector<h263mi_mb_record> big;
                    parse_picture(d.data(), d.size(), 1, nullptr, p);
                    const size_t per = p.n_records();
                    if (ext) big.resize(per * 64);
                    ready++;
                    while (!go.load()) std::this_thread::yield();
                    const auto t0 = std::chrono::steady_clock::now();
                    for (int i = 0; i < iters; i++) {
                        if (ext) { p.mbs_ext = big.data() + (size_t)(i % 64) * per; p.mbs_ext_cap = per; }
                        parse_picture(d.data(), d.size(), 1, nullptr, p);
                    }
                    secs[t] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                });
            while (ready.load() < T) std::this_thread::yield();
            const auto t0 = std::chrono::steady_clock::now();
            go = true;
            for (auto &x : th) x.join();
            const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            double slowest = 0;
            for (double s : secs) slowest = s > slowest ? s : slowest;
            printf("%s %2d threads: %8.0f pictures/s in total, %6.0f per thread (slowest thread %.1f us per picture)\n",
                   ext ? "records -> 16.7 MB ring " : "records -> own array    ", T, T * iters / wall, iters / wall, slowest / iters * 1e6);
        }
    return 0;
}
