// host_common.h -- what every host file of the C ABI (include/h263mi.h) shares: HIP error mapping, the fault-injection
// hook behind HIP_TRY, the device guard, launch geometry.  Compiled with hipcc; tests/tsan compiles the same files with
// g++ -fsanitize=thread against a stub of the HIP runtime (no GPU work there, only the host threads' ordering is tested).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "kernels.h"
#include "post_kernel.inl"   // tile constants only
#include "recon_kernel.inl"  // tile constants only

namespace h263mi {

inline int map_hip_error(hipError_t e)
{
    switch (e) {
    case hipSuccess: return H263MI_OK;
    case hipErrorNoDevice:
    case hipErrorInvalidDevice:
    case hipErrorInsufficientDriver:
    case hipErrorNotInitialized: return H263MI_ERR_NO_DEVICE;
    case hipErrorOutOfMemory: return H263MI_ERR_OUT_OF_MEMORY;
    default: return H263MI_ERR_HIP;
    }
}

// Fault injection for tests (h263mi_debug_fail_nth_hip_call, device_util.cpp): the n-th HIP call made through HIP_TRY from
// now on is not executed and reports hipErrorOutOfMemory instead.  Every allocation, copy, event operation and launch of the
// host entry points goes through HIP_TRY, so sweeping n over a call proves "on error the state is unchanged" (state.rs:142,
// 464-487) at every point at which the call can fail.  Off unless a test switches it on.
bool fault_now();

#define HIP_TRY(expr)                                                          \
    do {                                                                       \
        hipError_t _e = ::h263mi::fault_now() ? hipErrorOutOfMemory : (expr);  \
        if (_e != hipSuccess) return ::h263mi::map_hip_error(_e);              \
    } while (0)

#define RC_TRY(expr)                  \
    do {                              \
        int _rc = (expr);             \
        if (_rc != H263MI_OK) return _rc; \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; }
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

inline int check_device(int device_id)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return H263MI_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= count) return H263MI_ERR_NO_DEVICE;
    return H263MI_OK;
}

inline uint32_t recon_tiles_x(const FrameLayout &L) { return (L.mbw + TILE_MBX - 1) / TILE_MBX; }
inline uint32_t recon_tiles_y(const FrameLayout &L) { return (L.mbh + TILE_MBY - 1) / TILE_MBY; }
inline uint32_t post_tiles_y(const FrameLayout &L) { return (post_strips_y(L.height) + POST_STRIPS - 1) / POST_STRIPS; }
// Event indices are 32-bit on the device, 0xffffffff stands for "the caller did not say how many" (ReconArgs::n_events), and a
// lane looks up to 64 words past its first event before it compares with the block's end: the count stays clear of the top.
constexpr uint64_t kMaxEventWords = 0xffffff00ull;
// what the parser asks right behind a picture header (bits::ParsedPicture::size_fits): can the frame store hold such a picture?
inline bool picture_size_fits(uint32_t w, uint32_t h) { return layout_fits(w, h); }
// tile geometry of k_post for the layout in a.L (post_kernel.inl: post_tile_columns)
inline void set_post_tiles(PostArgs &a)
{
    a.tiles_x = post_tile_columns(a.L.width, &a.wrap);
    a.tiles_y = post_tiles_y(a.L);
}

// the post-filter strength a picture asks for itself: what a consumer of the reference computes from the header fields
// DecodedPicture::as_header hands out (picture.rs:61-64) -- QUANT_TO_STRENGTH[quantizer] (deblock.rs:5-8; types.rs:94-96)
// when the header's USE_DEBLOCKER flag is set (types.rs:216, parser/picture.rs:322), no deblocking otherwise
inline uint8_t strength_from_header(const h263mi_picture_desc &d)
{
    return d.use_deblocker ? h263mi_quant_to_strength[d.pquant & 31u] : (uint8_t)0;
}

}  // namespace h263mi
