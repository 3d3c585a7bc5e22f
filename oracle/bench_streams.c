/*
 * bench_streams.c -- CPU BASELINE RUNNER (test / bench infrastructure, NOT product code).
 *
 * Times the C oracle (h263_oracle.c, the restatement of the reference CPU path) the way BASELINE.md section 3 and
 * SURVEY.md 8(d) specify the CPU baseline: native build (-O3 -march=native -ffp-contract=off), one independent
 * stream per thread like the reference's single-threaded-per-stream design (state.rs:16-38: one H263State per
 * stream), every thread with its own frame store and output buffers.  Per picture a thread runs what the consumer
 * of the reference runs: the tail of decode_next_picture (state.rs:421-458), deblock() on the three planes
 * (deblock.rs:305-315) and yuv420_to_rgba (bt601.rs:105-196).
 *
 * Only bench.py's cpu_baseline leg (through oracle/native_bench.py) and tests/ use this file.
 */
#define _POSIX_C_SOURCE 200809L
#include <malloc.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "h263_oracle.h"

typedef struct {
    const orc_mb_record *mbs;
    size_t n_mbs;
    const int16_t *coeffs;
    size_t n_blocks;
} orc_bench_picture;

typedef struct {
    int id, n_gops, n_frames;
    uint16_t w, h;
    uint8_t strength;
    int stages;                         /* ORC_STAGE_* bits: which stages run inside the clock */
    int simd;                           /* 1: deblock / BT.601 in their explicit 128-bit form (simd_stages.c) */
    const orc_bench_picture *pics;      /* n_frames pictures of this thread's stream */
    pthread_barrier_t *start, *stop;
    uint64_t checksum;
    int rc;
} worker_t;

enum { ORC_STAGE_RECON = 1, ORC_STAGE_DEBLOCK = 2, ORC_STAGE_RGBA = 4 };
int orc_deblock_simd(const uint8_t *data, size_t len, size_t width, uint8_t strength, uint8_t *out);
int orc_yuv420_to_rgba_simd(const uint8_t *y, size_t y_len, const uint8_t *cb, const uint8_t *cr, size_t c_len, size_t y_width,
                            uint8_t *rgba);

static uint64_t fold(uint64_t acc, const uint8_t *p, size_t n)
{
    /* a cheap digest of a few bytes: keeps the results alive without costing measurable time */
    for (size_t i = 0; i < n; i += 4099) acc = (acc ^ p[i]) * 0x100000001b3ull;
    return acc;
}

static void *worker(void *arg)
{
    worker_t *wk = (worker_t *)arg;
    const size_t w = wk->w, h = wk->h, cw = (w + 1) / 2, ch = (h + 1) / 2;
    const size_t ny = w * h, nc = cw * ch;
    uint8_t *buf[2][3], *filt[3], *rgba;
    int ok = 1;
    for (int s = 0; s < 2; s++) {
        buf[s][0] = (uint8_t *)malloc(ny);
        buf[s][1] = (uint8_t *)malloc(nc);
        buf[s][2] = (uint8_t *)malloc(nc);
        ok = ok && buf[s][0] && buf[s][1] && buf[s][2];
    }
    filt[0] = (uint8_t *)malloc(ny);
    filt[1] = (uint8_t *)malloc(nc);
    filt[2] = (uint8_t *)malloc(nc);
    rgba = (uint8_t *)malloc(ny * 4);
    ok = ok && filt[0] && filt[1] && filt[2] && rgba;
    if (ok) {                                             /* touch every page before the clock starts */
        for (int s = 0; s < 2; s++) { memset(buf[s][0], 0, ny); memset(buf[s][1], 0, nc); memset(buf[s][2], 0, nc); }
        memset(filt[0], 0, ny); memset(filt[1], 0, nc); memset(filt[2], 0, nc);
        memset(rgba, 0, ny * 4);
    }
    wk->rc = ok ? ORC_OK : ORC_ERR_INVALID_ARGUMENT;
    int (*const deblock)(const uint8_t *, size_t, size_t, uint8_t, uint8_t *) = wk->simd ? orc_deblock_simd : orc_deblock;
    int (*const to_rgba)(const uint8_t *, size_t, const uint8_t *, const uint8_t *, size_t, size_t, uint8_t *) =
        wk->simd ? orc_yuv420_to_rgba_simd : orc_yuv420_to_rgba;
    const int st = wk->stages;
    if (ok && !(st & ORC_STAGE_RECON)) {
        /* a stage timed on its own works on one fixed picture: the stream's first picture (and its second on top of it,
         * when there is one), decoded and -- for the conversion alone -- filtered before the clock starts */
        const orc_bench_picture *p = &wk->pics[0];
        int rc = orc_decode_picture(wk->w, wk->h, p->mbs, p->n_mbs, p->coeffs, p->n_blocks, NULL, NULL, NULL, buf[1][0], buf[1][1], buf[1][2]);
        if (rc == ORC_OK && wk->n_frames > 1) {
            p = &wk->pics[1];
            rc = orc_decode_picture(wk->w, wk->h, p->mbs, p->n_mbs, p->coeffs, p->n_blocks, buf[1][0], buf[1][1], buf[1][2],
                                    buf[0][0], buf[0][1], buf[0][2]);
        } else if (rc == ORC_OK) {
            for (int k = 0; k < 3; k++) memcpy(buf[0][k], buf[1][k], k ? nc : ny);
        }
        if (rc == ORC_OK) rc = orc_deblock(buf[0][0], ny, w, wk->strength, filt[0]);
        if (rc == ORC_OK) rc = orc_deblock(buf[0][1], nc, cw, wk->strength, filt[1]);
        if (rc == ORC_OK) rc = orc_deblock(buf[0][2], nc, cw, wk->strength, filt[2]);
        if (rc != ORC_OK) { wk->rc = rc; ok = 0; }
    }
    pthread_barrier_wait(wk->start);
    uint64_t acc = 0xcbf29ce484222325ull;
    for (int g = 0; ok && g < wk->n_gops; g++) {
        int cur = 0;
        for (int f = 0; f < wk->n_frames; f++) {
            const orc_bench_picture *p = &wk->pics[f];
            uint8_t **out = buf[cur], **ref = buf[cur ^ 1];
            const int has_ref = f > 0;                    /* frame 0 of a GOP is an I picture */
            int rc = ORC_OK;
            if (st & ORC_STAGE_RECON)
                rc = orc_decode_picture(wk->w, wk->h, p->mbs, p->n_mbs, p->coeffs, p->n_blocks,
                                        has_ref ? ref[0] : NULL, has_ref ? ref[1] : NULL, has_ref ? ref[2] : NULL,
                                        out[0], out[1], out[2]);
            else
                out = buf[0];                             /* the fixed picture */
            if (st & ORC_STAGE_DEBLOCK) {
                if (rc == ORC_OK) rc = deblock(out[0], ny, w, wk->strength, filt[0]);
                if (rc == ORC_OK) rc = deblock(out[1], nc, cw, wk->strength, filt[1]);
                if (rc == ORC_OK) rc = deblock(out[2], nc, cw, wk->strength, filt[2]);
            }
            if ((st & ORC_STAGE_RGBA) && rc == ORC_OK) rc = to_rgba(filt[0], ny, filt[1], filt[2], nc, w, rgba);
            if (rc != ORC_OK) { wk->rc = rc; ok = 0; break; }
            acc = fold(acc, out[0], ny);
            if (st & ORC_STAGE_DEBLOCK) acc = fold(acc, filt[0], ny);
            if (st & ORC_STAGE_RGBA) acc = fold(acc, rgba, ny * 4);
            if (st & ORC_STAGE_RECON) cur ^= 1;
        }
    }
    pthread_barrier_wait(wk->stop);
    wk->checksum = acc;
    for (int s = 0; s < 2; s++) for (int k = 0; k < 3; k++) free(buf[s][k]);
    for (int k = 0; k < 3; k++) free(filt[k]);
    free(rgba);
    return NULL;
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* n_threads streams decode n_gops x n_frames pictures each, concurrently; thread t takes the pictures of distinct
 * stream t % n_distinct (pics[(t % n_distinct) * n_frames + f], shared read-only).  Returns the wall-clock seconds
 * between the moment all threads are ready (buffers allocated and touched) and the moment the last one finishes,
 * or a negative error code.  checksums[t] (may be NULL) receives a digest of thread t's outputs. */
double orc_bench_stages(int n_threads, int n_gops, int n_frames, uint16_t w, uint16_t h, const orc_bench_picture *pics,
                        int n_distinct, uint8_t strength, int stages, int simd, uint64_t *checksums);
double orc_bench_streams(int n_threads, int n_gops, int n_frames, uint16_t w, uint16_t h,
                         const orc_bench_picture *pics, int n_distinct, uint8_t strength, uint64_t *checksums)
{
    return orc_bench_stages(n_threads, n_gops, n_frames, w, h, pics, n_distinct, strength,
                            ORC_STAGE_RECON | ORC_STAGE_DEBLOCK | ORC_STAGE_RGBA, 0, checksums);
}

/* The same with a choice of stages ("recon / deblock / yuv->rgba individually and end-to-end"):
 * `stages` = ORC_STAGE_* bits; a stage that runs without the reconstruction works on one fixed picture of the stream,
 * n_gops x n_frames times.  simd: deblock and BT.601 in the reference's explicit 128-bit shape (simd_stages.c) instead of
 * the oracle's one-lane-at-a-time restatement. */
double orc_bench_stages(int n_threads, int n_gops, int n_frames, uint16_t w, uint16_t h, const orc_bench_picture *pics,
                        int n_distinct, uint8_t strength, int stages, int simd, uint64_t *checksums)
{
    if (!(stages & 7) || (stages & ~7)) return -100.0;
    if (n_threads < 1 || n_gops < 1 || n_frames < 1 || n_distinct < 1 || !pics || !w || !h) return -100.0;
    /* Like the reference (state.rs:179-191), the oracle allocates its per-picture block arrays anew for every picture
     * (about 13 MB at 1080p).  With glibc's defaults each of those is an mmap + page-fault storm + munmap, and with one
     * stream per core the process-wide mmap lock, not the decoder, sets the pace (measured: 8 % parallel
     * efficiency on 128 cores).  Keep big blocks in the per-thread arenas instead, as a production build of the
     * reference would with a pooling allocator. */
    mallopt(M_MMAP_THRESHOLD, 32 << 20);      /* glibc's upper limit for this parameter */
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, 64 << 20);
    pthread_barrier_t start, stop;
    if (pthread_barrier_init(&start, NULL, (unsigned)n_threads + 1)) return -1.0;
    if (pthread_barrier_init(&stop, NULL, (unsigned)n_threads + 1)) return -1.0;
    worker_t *wk = (worker_t *)calloc((size_t)n_threads, sizeof *wk);
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof *th);
    if (!wk || !th) return -1.0;
    int started = 0;
    for (int t = 0; t < n_threads; t++) {
        wk[t].id = t; wk[t].n_gops = n_gops; wk[t].n_frames = n_frames;
        wk[t].w = w; wk[t].h = h; wk[t].strength = strength;
        wk[t].stages = stages; wk[t].simd = simd;
        wk[t].pics = pics + (size_t)(t % n_distinct) * (size_t)n_frames;
        wk[t].start = &start; wk[t].stop = &stop;
        if (pthread_create(&th[t], NULL, worker, &wk[t])) break;
        started++;
    }
    if (started != n_threads) return -2.0;                /* (threads already started stay blocked: fatal for the caller) */
    pthread_barrier_wait(&start);
    const double t0 = now_s();
    pthread_barrier_wait(&stop);
    const double t1 = now_s();
    double result = t1 - t0;
    for (int t = 0; t < n_threads; t++) {
        pthread_join(th[t], NULL);
        if (wk[t].rc != ORC_OK) result = (double)wk[t].rc;
        if (checksums) checksums[t] = wk[t].checksum;
    }
    pthread_barrier_destroy(&start);
    pthread_barrier_destroy(&stop);
    free(wk);
    free(th);
    return result;
}
