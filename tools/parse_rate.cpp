// parse_rate.cpp -- throughput of the host bitstream parser (h263-rs_amd/host/bitstream.cpp) on coded pictures
// read from files.  Build and run (CPU only):
//   g++ -O3 -std=c++17 -Iinclude -o /tmp/parse_rate tools/parse_rate.cpp h263-rs_amd/host/bitstream.cpp
//   /tmp/parse_rate picture1.bin [picture2.bin ...]        (Sorenson Spark pictures, e.g. from tests/sorenson_enc.py)
// Measured here (8-core container, 2.1 GHz Xeon, one thread, best of 60; the machine is shared, repeat the run and keep
// the minimum): a 1080p I picture of 39 231 coded blocks / 788 k events (2.3 MB) parses in 10.7 ms (13.6 ns per event),
// a 1080p P picture with 25 % coded blocks (150 KB, 45 k events) in 1.07 ms.
// Round 1: 39.3 ms and 3.45 ms.  Round 2, first pass (14.1 / 1.77 ms): one 32-bit window per TCOEF event, a two-level
// VLC table, no division in the vector prediction, parse buffers that keep their capacity from picture to picture.
// Second pass (10.7 / 1.07 ms): short code or ESCAPE selected by mask arithmetic instead of a branch (the largest
// single step: on these streams the kind of an event is random, every other one cost a misprediction), the macroblock
// header out of one 64-bit window, branch-free median, only the coded blocks of an inter macroblock visited, outputs
// written in place.  tests/test_parser_paths.py holds the fast paths against the field-by-field form.
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
        p.want_dense = false;                      // as the library does: the coefficients travel as events
        parse_picture(d.data(), d.size(), 1, nullptr, p);
        const int iters = 60;
        size_t sink = 0;
        double dt = 1e9, sum = 0;                    // best and mean of the individually timed parses
        for (int i = 0; i < iters; i++) {
            auto t0 = std::chrono::steady_clock::now();
            parse_picture(d.data(), d.size(), 1, nullptr, p);
            const double one = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            sink += p.mbs.size();
            sum += one;
            if (one < dt) dt = one;
        }
        (void)sum;
        printf("%s: %zu bytes, %zu MBs, %zu coded blocks, %zu events: %.3f ms per parse (best of 60) = %.0f pictures/s per core, %.1f MB/s, %.1f ns per event\n", argv[a], d.size(), p.mbs.size(), p.n_coded_blocks, p.events.size(), dt * 1e3, 1 / dt, d.size() / 1e6 / dt, dt * 1e9 / (double)(p.events.size() ? p.events.size() : 1));
        if (!sink) return 1;
    }
}
