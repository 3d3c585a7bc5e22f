// parse_rate.cpp -- throughput of the host bitstream parser (h263-rs_amd/host/bitstream.cpp) on coded pictures
// read from files.  Build and run (CPU only):
//   g++ -O3 -std=c++17 -Iinclude -o /tmp/parse_rate tools/parse_rate.cpp h263-rs_amd/host/bitstream.cpp
//   /tmp/parse_rate picture1.bin [picture2.bin ...]        (Sorenson Spark pictures, e.g. from tests/sorenson_enc.py)
// Measured here (8 cores container, one thread): a 1080p I picture of 39 209 coded blocks (1.8 MB) parses in 24 ms,
// a 1080p P picture with 25 % coded blocks (160 KB) in 2.6 ms: 60-75 MB/s of bitstream per core.
#include <chrono>
#include <cstdio>
#include <vector>
#include "../h263-rs_amd/host/bitstream.hpp"
using namespace h263mi::bits;
int main(int argc, char **argv)
{
    for (int a = 1; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb");
        std::vector<uint8_t> d;
        uint8_t buf[65536]; size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
        fclose(f);
        ParsedPicture p;
        parse_picture(d.data(), d.size(), 1, nullptr, p);
        const int iters = 50;
        auto t0 = std::chrono::steady_clock::now();
        size_t sink = 0;
        for (int i = 0; i < iters; i++) { parse_picture(d.data(), d.size(), 1, nullptr, p); sink += p.mbs.size(); }
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / iters;
        printf("%s: %zu bytes, %zu MBs, %zu blocks: %.3f ms per parse = %.0f pictures/s per core, %.1f MB/s\n", argv[a], d.size(), p.mbs.size(), p.coeffs.size() / 64, dt * 1e3, 1 / dt, d.size() / 1e6 / dt);
        if (!sink) return 1;
    }
}
