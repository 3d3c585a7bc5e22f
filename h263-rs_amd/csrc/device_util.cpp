// device_util.cpp -- the rest of the C ABI (include/h263mi.h): error strings, deblock::deblock and
// yuv::bt601::yuv420_to_rgba as plain functions over host buffers, device memory helpers, the fault-injection hook, the
// on-box bandwidth probes and the synthetic record generators of the bench.
#include <atomic>
#include <cstring>
#include <vector>

#include "host_common.h"
#include "synth.inl"

using namespace h263mi;

namespace h263mi {

// -1 = off (the product never sets it)
static std::atomic<int> g_fail_countdown{-1};
bool fault_now()
{
    int v = g_fail_countdown.load(std::memory_order_relaxed);
    if (v < 0) return false;
    v = g_fail_countdown.fetch_sub(1, std::memory_order_relaxed);
    return v == 1;                                 // the countdown went 1 -> 0 with this call
}

}  // namespace h263mi

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
