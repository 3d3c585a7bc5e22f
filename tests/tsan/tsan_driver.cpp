// tests/tsan/tsan_driver.cpp -- the host pipeline of libh263mi under ThreadSanitizer.
//
// Built with g++ -fsanitize=thread from the PRODUCT's host sources (worker_pool.cpp, batch.cpp, batch_staging.cpp,
// mixed_set.cpp, state.cpp, device_util.cpp, host/bitstream.cpp) against a stub of the HIP runtime (hip_stub/, plain malloc
// for device and pinned memory, stub kernel launchers): what runs here is the real WorkerPool (spin-then-park, generation
// word), the real stream-affinity / stealing deal, the real parser writing records, events, block offsets and the group index
// of N streams straight into one pinned staging slot at per-stream pitches, and the real enqueue path that reads the slot.
// The reference needs none of this: it is single-threaded safe Rust (`&mut self`, state.rs:138-141).
//
// Scenarios (all through the C ABI):
//   * one batch, calls back to back with 1 .. 40 parser threads, random NULL streams, random pauses between the calls so that
//     the workers are met spinning, about to park and parked; under a fake 2-CPU quota (H263MI_CGROUP_CPU_MAX) so that calls
//     with more threads than the quota take the parking plan (spin 0) and the others the spinning plan;
//   * pools torn down in the middle of an idle spin and right after a call;
//   * two batches, a mixed-size set and one H263State driven from four threads at the same time (distinct objects may be);
//   * the packed (H263MI_DIRECT_WORDS=0) transport, and workers that never park (H263MI_SPIN_US = 50 ms, no quota, at most 6
//     threads: every hand-over goes through the generation word alone, never through the mutex): further runs of the test.
// Exit code 0 and no ThreadSanitizer report = pass.  The same driver built with -DH263MI_TSAN_BREAK_GENERATION_ORDER (the task is
// published with a relaxed store: worker_pool.cpp) MUST make ThreadSanitizer report a race: tests/test_tsan.py checks both.
// usage: tsan_driver <corpus.bin> [rounds [max threads]]      corpus: u32 streams, u32 frames, then per (stream, frame): u32 length, bytes
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <memory>
#include <random>
#include <thread>
#include <vector>

#include "../../include/h263mi.h"

static std::vector<std::vector<std::vector<uint8_t>>> g_corpus;      // [stream][frame]
static uint16_t g_w, g_h;
static std::atomic<int> g_failures{0};
static uint32_t g_max_threads = 40;                                  // (third argument: runs whose workers SPIN keep within the CPUs)

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) {                                                               \
            fprintf(stderr, "tsan_driver: %s:%d: %s failed\n", __FILE__, __LINE__, #cond); \
            g_failures.fetch_add(1);                                                 \
        }                                                                            \
    } while (0)

static bool load_corpus(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    uint32_t hdr[4];
    if (fread(hdr, 4, 4, f) != 4) return false;
    g_w = (uint16_t)hdr[2];
    g_h = (uint16_t)hdr[3];
    g_corpus.assign(hdr[0], std::vector<std::vector<uint8_t>>(hdr[1]));
    for (auto &s : g_corpus)
        for (auto &p : s) {
            uint32_t n = 0;
            if (fread(&n, 4, 1, f) != 1) return false;
            p.resize(n);
            if (n && fread(p.data(), 1, n, f) != n) return false;
        }
    fclose(f);
    return true;
}

// one batch of `n` streams: every call decodes the next picture of a random subset of them
static void drive_batch(unsigned seed, uint32_t n, int calls, bool pipeline, bool destroy_mid_spin)
{
    std::mt19937 rng(seed);
    h263mi_backend_cfg cfg{0, pipeline ? H263MI_CFG_PIPELINE_POST : 0u, nullptr};
    h263mi_batch *b = nullptr;
    CHECK(h263mi_batch_create(n, g_w, g_h, &cfg, &b) == H263MI_OK);
    if (!b) return;
    void *d_rgba = nullptr;
    CHECK(h263mi_device_malloc(0, (size_t)n * g_w * g_h * 4, &d_rgba) == H263MI_OK);
    std::vector<uint32_t> next(n, 0);
    const uint32_t frames = (uint32_t)g_corpus[0].size();
    static const uint32_t kThreads[] = {1, 2, 3, 5, 8, 13, 16, 24, 40};
    for (int c = 0; c < calls; c++) {
        std::vector<const uint8_t *> data(n, nullptr);
        std::vector<size_t> len(n, 0), used(n, 0);
        std::vector<int> rcs(n, 0);
        uint32_t with_data = 0;
        for (uint32_t s = 0; s < n; s++) {
            const bool key = next[s] == 0;
            if (!key && rng() % 4 == 0) continue;                          // no picture for this stream in this call
            if (next[s] >= frames) next[s] = 0;                            // (the corpus starts every stream with a key frame)
            const std::vector<uint8_t> &p = g_corpus[s % g_corpus.size()][next[s]];
            data[s] = p.data();
            len[s] = p.size();
            with_data++;
        }
        uint32_t threads = kThreads[rng() % (sizeof kThreads / sizeof kThreads[0])];
        if (threads > g_max_threads) threads = 1 + threads % g_max_threads;
        const int rc = h263mi_batch_decode_next_pictures_ex(b, H263MI_SORENSON_SPARK_BITSTREAM, data.data(), len.data(), used.data(),
                                                            threads, rcs.data(), H263MI_STRENGTH_FROM_HEADER, (uint8_t *)d_rgba, nullptr);
        CHECK(rc == H263MI_OK);
        for (uint32_t s = 0; s < n; s++) {
            if (!data[s]) continue;
            CHECK(rcs[s] == H263MI_OK);
            CHECK(used[s] == len[s] || used[s] + 1 == len[s]);      // (whole bytes: the last one may be half padding)
            if (rcs[s] == H263MI_OK) next[s]++;
        }
        if (c % 7 == 6) CHECK(h263mi_batch_sync(b) == H263MI_OK);
        // pauses of 0 .. 700 us: shorter, about as long and longer than the workers' 300 us spin
        const unsigned pause = rng() % 8;
        if (pause) std::this_thread::sleep_for(std::chrono::microseconds(100 * pause));
    }
    if (!destroy_mid_spin) std::this_thread::sleep_for(std::chrono::milliseconds(2));     // everybody parked
    CHECK(h263mi_batch_sync(b) == H263MI_OK);
    h263mi_batch_destroy(b);                                               // (pool teardown: mid-spin or parked)
    CHECK(h263mi_device_free(0, d_rgba) == H263MI_OK);
}

static void drive_mixed(unsigned seed, uint32_t n, int calls)
{
    std::mt19937 rng(seed);
    h263mi_backend_cfg cfg{0, H263MI_CFG_PIPELINE_POST, nullptr};
    h263mi_mixed *m = nullptr;
    CHECK(h263mi_mixed_create(n, &cfg, &m) == H263MI_OK);
    if (!m) return;
    std::vector<void *> bufs(n, nullptr);
    std::vector<size_t> caps(n, (size_t)g_w * g_h * 4);
    for (uint32_t s = 0; s < n; s++) CHECK(h263mi_device_malloc(0, caps[s], &bufs[s]) == H263MI_OK);
    std::vector<uint32_t> next(n, 0);
    const uint32_t frames = (uint32_t)g_corpus[0].size();
    for (int c = 0; c < calls; c++) {
        std::vector<const uint8_t *> data(n, nullptr);
        std::vector<size_t> len(n, 0), used(n, 0);
        std::vector<int> rcs(n, 0);
        for (uint32_t s = 0; s < n; s++) {
            if (next[s] && rng() % 3 == 0) continue;
            if (next[s] >= frames) next[s] = 0;
            const std::vector<uint8_t> &p = g_corpus[(s + 3) % g_corpus.size()][next[s]];
            data[s] = p.data();
            len[s] = p.size();
        }
        const int rc = h263mi_mixed_decode_next_pictures(m, H263MI_SORENSON_SPARK_BITSTREAM, data.data(), len.data(), used.data(),
                                                         1 + rng() % (g_max_threads < 12 ? g_max_threads : 12), rcs.data(), H263MI_STRENGTH_FROM_HEADER, (uint8_t *const *)bufs.data(),
                                                         caps.data(), nullptr);
        CHECK(rc == H263MI_OK);
        for (uint32_t s = 0; s < n; s++)
            if (data[s]) {
                CHECK(rcs[s] == H263MI_OK);
                if (rcs[s] == H263MI_OK) next[s]++;
            }
        if (c % 5 == 4) CHECK(h263mi_mixed_sync(m, nullptr) == H263MI_OK);
    }
    CHECK(h263mi_mixed_sync(m, nullptr) == H263MI_OK);
    h263mi_mixed_destroy(m);
    for (void *p : bufs) CHECK(h263mi_device_free(0, p) == H263MI_OK);
}

// one H263State fed coded pictures (h263mi_decode_next_picture: parse into the state's own staging slot), with the accessors
// and the rendering a consumer calls behind it
static void drive_state(unsigned seed, int pictures)
{
    std::mt19937 rng(seed);
    h263mi_state *st = nullptr;
    CHECK(h263mi_state_new(H263MI_SORENSON_SPARK_BITSTREAM, nullptr, &st) == H263MI_OK);
    if (!st) return;
    std::vector<uint8_t> rgba((size_t)g_w * g_h * 4), y((size_t)g_w * g_h), cb((size_t)g_w * g_h / 4 + 64), cr(cb.size());
    const std::vector<std::vector<uint8_t>> &stream = g_corpus[seed % g_corpus.size()];
    for (int k = 0; k < pictures; k++) {
        const std::vector<uint8_t> &p = stream[(size_t)k % stream.size()];
        size_t used = 0;
        CHECK(h263mi_decode_next_picture(st, p.data(), p.size(), &used) == H263MI_OK);
        h263mi_frame_view v;
        CHECK(h263mi_get_last_picture(st, &v) == H263MI_OK && v.width == g_w && v.height == g_h);
        if (rng() % 2) CHECK(h263mi_render_rgba(st, H263MI_STRENGTH_FROM_HEADER, rgba.data()) == H263MI_OK);
        if (rng() % 4 == 0) CHECK(h263mi_copy_yuv(st, y.data(), cb.data(), cr.data()) == H263MI_OK);
        if (rng() % 16 == 0) CHECK(h263mi_state_reset(st) == H263MI_OK && (k = (k / (int)stream.size() + 1) * (int)stream.size() - 1, true));
    }
    // a truncated picture is that call's error and leaves the state as it was
    const std::vector<uint8_t> &p0 = stream[0];
    size_t used = 0;
    CHECK(h263mi_decode_next_picture(st, p0.data(), 3, &used) != H263MI_OK);
    h263mi_state_free(st);
}

// the entries over DEVICE records (h263mi_batch_decode / _decode_events): the host side of "checked by default" -- sizes not
// given are taken from the allocations (here: the stub's registry of malloc'd blocks), arrays that cannot hold the batch are
// refused before anything is queued, H263MI_STRENGTH_FROM_HEADER has no meaning without a header
static void drive_device_arrays()
{
    const uint32_t n = 3;
    h263mi_backend_cfg cfg{0, H263MI_CFG_PIPELINE_POST, nullptr};
    h263mi_batch *b = nullptr;
    CHECK(h263mi_batch_create(n, g_w, g_h, &cfg, &b) == H263MI_OK);
    if (!b) return;
    const uint32_t per = h263mi_batch_mbs_per_picture(b);
    std::vector<h263mi_mb_record> mbs((size_t)n * per);
    std::vector<int16_t> co((size_t)n * per * 6 * 64);
    std::vector<uint64_t> base(n);
    size_t at = 0;
    for (uint32_t s = 0; s < n; s++) {
        size_t nb = 0;
        base[s] = at;
        CHECK(h263mi_synth_picture_host(H263MI_SYNTH_I_MIXED, g_w, g_h, s, 0, mbs.data() + (size_t)s * per, co.data() + at * 64,
                                        (size_t)per * 6, &nb) == H263MI_OK);
        at += nb;
    }
    void *d_m = nullptr, *d_c = nullptr, *d_b = nullptr, *d_rgba = nullptr, *d_short = nullptr;
    CHECK(h263mi_device_malloc(0, mbs.size() * sizeof mbs[0], &d_m) == H263MI_OK);
    CHECK(h263mi_device_malloc(0, at * 128, &d_c) == H263MI_OK);
    CHECK(h263mi_device_malloc(0, n * 8, &d_b) == H263MI_OK);
    CHECK(h263mi_device_malloc(0, (size_t)n * g_w * g_h * 4, &d_rgba) == H263MI_OK);
    CHECK(h263mi_device_malloc(0, mbs.size() * sizeof mbs[0] - 32, &d_short) == H263MI_OK);
    CHECK(h263mi_device_memcpy_h2d(0, d_m, mbs.data(), mbs.size() * sizeof mbs[0]) == H263MI_OK);
    CHECK(h263mi_device_memcpy_h2d(0, d_c, co.data(), at * 128) == H263MI_OK);
    CHECK(h263mi_device_memcpy_h2d(0, d_b, base.data(), n * 8) == H263MI_OK);
    const uint8_t strengths[3] = {0, 7, 12};
    // no size given: the pool ends where its allocation ends
    CHECK(h263mi_batch_decode_ps(b, H263MI_PICTURE_I, (const h263mi_mb_record *)d_m, (const int16_t *)d_c, (const uint64_t *)d_b, 0, 0,
                                 strengths, (uint8_t *)d_rgba, nullptr) == H263MI_OK);
    CHECK(h263mi_batch_sync(b) == H263MI_OK);
    // a size beyond the allocation, records one short, an output buffer too small, a strength out of range, no header to take it from
    CHECK(h263mi_batch_decode(b, H263MI_PICTURE_I, (const h263mi_mb_record *)d_m, (const int16_t *)d_c, (const uint64_t *)d_b, at + 1, 3,
                              (uint8_t *)d_rgba, nullptr) == H263MI_ERR_INVALID_ARGUMENT);
    CHECK(h263mi_batch_decode(b, H263MI_PICTURE_I, (const h263mi_mb_record *)d_short, (const int16_t *)d_c, (const uint64_t *)d_b, at, 3,
                              (uint8_t *)d_rgba, nullptr) == H263MI_ERR_INVALID_ARGUMENT);
    CHECK(h263mi_batch_decode(b, H263MI_PICTURE_I, (const h263mi_mb_record *)d_m, (const int16_t *)d_c, (const uint64_t *)d_b, at, 3,
                              (uint8_t *)d_b, nullptr) == H263MI_ERR_INVALID_ARGUMENT);
    CHECK(h263mi_batch_decode(b, H263MI_PICTURE_I, (const h263mi_mb_record *)d_m, (const int16_t *)d_c, (const uint64_t *)d_b, at, 13,
                              (uint8_t *)d_rgba, nullptr) == H263MI_ERR_INVALID_ARGUMENT);
    CHECK(h263mi_batch_decode(b, H263MI_PICTURE_I, (const h263mi_mb_record *)d_m, (const int16_t *)d_c, (const uint64_t *)d_b, at,
                              H263MI_STRENGTH_FROM_HEADER, (uint8_t *)d_rgba, nullptr) == H263MI_ERR_INVALID_ARGUMENT);
    // a pointer the runtime does not know (host memory) with no size: refused; on a trusted batch: the caller vouches
    CHECK(h263mi_batch_submit(b, H263MI_PICTURE_I, mbs.data(), co.data(), nullptr) == H263MI_ERR_INVALID_ARGUMENT);
    h263mi_batch_destroy(b);
    cfg.flags |= H263MI_CFG_TRUSTED_ARRAYS;
    CHECK(h263mi_batch_create(n, g_w, g_h, &cfg, &b) == H263MI_OK);
    if (b) {
        CHECK(h263mi_batch_decode(b, H263MI_PICTURE_I, (const h263mi_mb_record *)d_short, (const int16_t *)d_c, (const uint64_t *)d_b, 0, 3,
                                  (uint8_t *)d_rgba, nullptr) == H263MI_OK);       // (stub kernels read nothing of it)
        CHECK(h263mi_batch_sync(b) == H263MI_OK);
        h263mi_batch_destroy(b);
    }
    for (void *p : {d_m, d_c, d_b, d_rgba, d_short}) CHECK(h263mi_device_free(0, p) == H263MI_OK);
}

// the entries over HOST arrays of a caller (h263mi_submit_picture[_events], h263mi_batch_submit_host[_events]) fed GARBAGE: every
// record field, block index, block offset and event word random, in exact-size heap arrays (AddressSanitizer sees one element
// beyond them).  The host validates these before anything is staged: the answer is a refusal or -- garbage that happens to be a
// valid picture -- a decode; never a read outside the arrays, never a changed state behind a refusal.
static void drive_host_garbage(unsigned seed, int tries)
{
    std::mt19937 rng(seed);
    h263mi_state *st = nullptr;
    CHECK(h263mi_state_new(H263MI_SORENSON_SPARK_BITSTREAM, nullptr, &st) == H263MI_OK);
    h263mi_backend_cfg cfg{0, 0, nullptr};
    h263mi_batch *b = nullptr;
    CHECK(h263mi_batch_create(2, g_w, g_h, &cfg, &b) == H263MI_OK);
    if (!st || !b) return;
    const uint32_t per = h263mi_batch_mbs_per_picture(b);
    // a valid key picture first: garbage P pictures then have a reference to be refused against, or to predict from
    std::vector<h263mi_mb_record> key(per);
    std::vector<int16_t> key_co((size_t)per * 6 * 64);
    size_t key_blocks = 0;
    CHECK(h263mi_synth_picture_host(H263MI_SYNTH_I_MIXED, g_w, g_h, 0, 0, key.data(), key_co.data(), (size_t)per * 6, &key_blocks) == H263MI_OK);
    h263mi_picture_desc desc{};
    desc.width = g_w;
    desc.height = g_h;
    desc.picture_type = H263MI_PICTURE_I;
    desc.pquant = 8;
    CHECK(h263mi_submit_picture(st, &desc, key.data(), per, key_co.data(), key_blocks) == H263MI_OK);
    auto acceptable = [](int rc) {
        return rc == H263MI_OK || rc == H263MI_ERR_INVALID_ARGUMENT || rc == H263MI_ERR_UNCODED_IFRAME_BLOCKS ||
               rc == H263MI_ERR_PICTURE_FORMAT_INVALID;
    };
    int refused = 0, taken = 0;
    for (int t = 0; t < tries; t++) {
        const int flavour = (int)(rng() % 4);      // 0: all random; 1: plausible fields, random structure; 2, 3: a valid picture
                                                   // with (2) or without (3) a few wild fields
        const size_t n_mbs = rng() % 3 == 0 ? per : rng() % (per + 1);
        const size_t n_blocks = rng() % 40;
        // exact-size arrays (new[]: a redzone right behind the last element)
        std::unique_ptr<uint32_t[]> first(new uint32_t[n_blocks + 1]);
        uint32_t at = 0;
        for (size_t k = 0; k <= n_blocks; k++) {
            first[k] = flavour ? at : rng();
            at += (uint32_t)(rng() % 9);
        }
        const size_t n_events = flavour >= 2 ? first[n_blocks] : rng() % 200;
        if (flavour == 1 && rng() % 2) first[n_blocks] = (uint32_t)n_events;
        std::unique_ptr<h263mi_mb_record[]> mbs(new h263mi_mb_record[n_mbs ? n_mbs : 1]);
        std::unique_ptr<int16_t[]> co(new int16_t[(n_blocks ? n_blocks : 1) * 64]);
        std::unique_ptr<uint32_t[]> ev(new uint32_t[n_events ? n_events : 1]);
        const uint32_t wild = flavour == 3 ? 0u : (uint32_t)(flavour == 2 ? 2 * per : 50);       // (a wild value in one field of `wild`)
        for (size_t i = 0; i < n_mbs; i++) {
            h263mi_mb_record &m = mbs[i];
            uint32_t *w = reinterpret_cast<uint32_t *>(&m);
            for (size_t k = 0; k < sizeof m / 4; k++) w[k] = rng();
            if (flavour) {
                m.mb_type = (uint8_t)(rng() % 6);
                m.quant = (uint8_t)(1 + rng() % 31);
                m.cbp &= 0x3f;
                m.kill &= 0x3f;
                m.coeff_index = n_blocks ? (uint32_t)(rng() % n_blocks) : 0;
                if (flavour >= 2) {
                    if (n_blocks < 6) m.cbp = 0;
                    else m.coeff_index = (uint32_t)(rng() % (n_blocks - 5));
                }
                if (wild && rng() % wild == 0) m.coeff_index = rng() % 2 ? 0xffffffffu : (uint32_t)n_blocks;
                if (wild && rng() % wild == 0) m.quant = (uint8_t)(rng() % 2 ? 0 : 32);
                if (wild && rng() % wild == 0) m.mb_type = (uint8_t)(6 + rng() % 250);
                if (wild && rng() % wild == 0) m.cbp |= 0x40;
            }
        }
        for (size_t k = 0; k < (n_blocks ? n_blocks : 1) * 64; k++) co[k] = (int16_t)rng();
        for (size_t k = 0; k < n_events; k++) ev[k] = rng();
        if (flavour >= 2)                                  // a block's events name distinct positions
            for (size_t k = 0; k < n_blocks; k++)
                for (uint32_t e = first[k]; e < first[k + 1]; e++) ev[e] = (ev[e] & 0xffff0000u) | ((first[k] + 7u * (e - first[k])) & 63u);
        if (flavour == 2 && n_blocks && rng() % 3 == 0) first[rng() % (n_blocks + 1)] = rng() % 2 ? 0xffffffffu : (uint32_t)n_events + 1;
        if (flavour == 2 && n_events && rng() % 3 == 0) ev[rng() % n_events] = ev[0];
        desc.picture_type = (uint8_t)(flavour >= 2 ? rng() % 2 : rng() % 12);
        if (flavour < 3 && rng() % 20 == 0) desc.width = (uint16_t)rng();
        else desc.width = g_w;
        h263mi_frame_view before{}, after{};
        const int had = h263mi_get_last_picture(st, &before);
        int rc = rng() % 2 ? h263mi_submit_picture(st, &desc, mbs.get(), n_mbs, co.get(), n_blocks)
                           : h263mi_submit_picture_events(st, &desc, mbs.get(), n_mbs, first.get(), n_blocks, ev.get(), n_events);
        CHECK(acceptable(rc));
        if (rc != H263MI_OK) {
            refused++;
            // state.rs:142: on error the state is unchanged
            CHECK(h263mi_get_last_picture(st, &after) == had && (had != H263MI_OK || (after.width == before.width && after.dev_y == before.dev_y && after.temporal_reference == before.temporal_reference)));
        } else {
            taken++;
        }
        // the batch forms: stream 0 the garbage, stream 1 the key picture (or nothing)
        const h263mi_mb_record *bm[2] = {mbs.get(), key.data()};
        const uint32_t bn[2] = {(uint32_t)n_mbs, per};
        const int16_t *bc[2] = {co.get(), key_co.data()};
        const uint32_t bb[2] = {(uint32_t)n_blocks, (uint32_t)key_blocks};
        rc = h263mi_batch_submit_host(b, H263MI_PICTURE_I, bm, bn, bc, bb);
        CHECK(acceptable(rc));
        const uint32_t *bf[2] = {first.get(), nullptr};
        const uint32_t *be[2] = {ev.get(), nullptr};
        const uint32_t bn0[2] = {(uint32_t)n_mbs, 0}, bb0[2] = {(uint32_t)n_blocks, 0}, bne[2] = {(uint32_t)n_events, 0};
        const h263mi_mb_record *bm0[2] = {mbs.get(), nullptr};
        rc = h263mi_batch_submit_host_events(b, H263MI_PICTURE_I, bm0, bn0, bf, bb0, be, bne);
        CHECK(acceptable(rc));
        (void)h263mi_batch_sync(b);
    }
    CHECK(refused > 0);
    fprintf(stderr, "  host garbage: %d refused, %d decoded\n", refused, taken);
    h263mi_batch_destroy(b);
    h263mi_state_free(st);
}

int main(int argc, char **argv)
{
    if (argc < 2 || !load_corpus(argv[1])) {
        fprintf(stderr, "usage: tsan_driver <corpus.bin> [rounds]\n");
        return 2;
    }
    const int rounds = argc > 2 ? atoi(argv[2]) : 2;
    if (argc > 3) g_max_threads = (uint32_t)atoi(argv[3]);
    const uint32_t n = (uint32_t)g_corpus.size();
    drive_device_arrays();
    drive_host_garbage(900, 400);
    for (int r = 0; r < rounds; r++) {
        // one batch after the other: spinning and parking plans, teardown parked and mid-spin
        drive_batch(100 + r, n, 40, /*pipeline=*/true, /*destroy_mid_spin=*/false);
        drive_batch(200 + r, n, 25, /*pipeline=*/false, /*destroy_mid_spin=*/true);
        drive_batch(300 + r, 3, 30, /*pipeline=*/true, /*destroy_mid_spin=*/true);        // fewer streams than threads
        // distinct objects from distinct threads at the same time
        std::thread t1(drive_batch, 400 + r, n, 30, true, true);
        std::thread t2(drive_batch, 500 + r, n / 2 + 1, 30, false, false);
        std::thread t3(drive_mixed, 600 + r, n, 20);
        std::thread t4(drive_state, 700 + r, 40);
        t1.join();
        t2.join();
        t3.join();
        t4.join();
    }
    const int bad = g_failures.load();
    fprintf(stderr, "tsan_driver: %d rounds, %d check failures\n", rounds, bad);
    return bad ? 1 : 0;
}
