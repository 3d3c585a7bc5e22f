// rgba_store_shapes.hip -- how fast can the RGBA surfaces of 64 1080p pictures be WRITTEN, by store shape?
// k_frame's post waves write 531 MB of RGBA per launch as non-temporal 16-byte stores: one store instruction = two runs
// of 512 bytes (two picture rows, one tile wide) that start 16 bytes past a 64-byte line.  Timing builds say the stores
// cost 31 % of the launch (profiles/r03_d_timing_mc_rgba.txt) and that moving the runs onto whole lines makes the launch
// 3 % SLOWER (profiles/r03_o_ab_aligned_post_timing.txt).  This probe writes the same surfaces with nothing else going
// on, one wave per 128 x 32 tile in the XCD-dealt order of k_frame's work list, in several shapes.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/rgba_store_shapes.hip -o build/rgba_store_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int W = 1920, H = 1080, N = 64;
constexpr int TX = 15, TY = 34;                    // 128 x 32 tiles (the last row of tiles is cut at 1080)
constexpr size_t PIC = (size_t)W * H * 4;

// SHAPE 0: 2 rows x 512 B per store, runs start `shift` bytes past the tile's line-aligned origin (k_frame: 16)
// SHAPE 1: 4 rows x 256 B per store (16 lanes per row)
// SHAPE 2: 1 row x 1024 B per store: the wave takes a 256 x 16 tile instead (same pixels per wave)
// SHAPE 3: 2 rows x 512 B, rows r and r + 1 instead of r and r + 2
template <int SHAPE, bool NT>
__global__ __launch_bounds__(64) void k(uint8_t *rgba, int shift, int bands, uint32_t upp)
{
    const int lane = threadIdx.x;
    // the deal of k_frame: `bands` XCDs share a picture's tile list in contiguous chunks, 8 / bands pictures side by side
    const uint32_t xcd = blockIdx.x & 7, chunk = (upp + bands - 1) / bands;
    const uint32_t band = xcd & (bands - 1), side = xcd / bands, t = blockIdx.x >> 3, g = band * chunk + t;
    const uint32_t pic = blockIdx.y * (8 / bands) + side;
    if (t >= chunk || g >= upp || pic >= N) return;
    uint8_t *base = rgba + (size_t)pic * PIC;
    const u32x4 v = {lane + g, 2u, 3u, 0xff0000ffu};
    if (SHAPE == 2) {
        // 256 x 16 tiles: 8 columns x 68 rows of tiles; take them in the same list order
        const uint32_t tx = g % 8, ty = g / 8;
        if (tx * 256 >= (uint32_t)W) return;
        for (int r = 0; r < 16; r++) {
            const uint32_t y = ty * 16 + r, x = tx * 256 + lane * 4;
            if (y >= (uint32_t)H || x >= (uint32_t)W) continue;
            u32x4 *p = reinterpret_cast<u32x4 *>(base + ((size_t)y * W + x) * 4 + shift);
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
        return;
    }
    const uint32_t tx = g % TX, ty = g / TX;
    for (int it = 0; it < 16; it++) {
        uint32_t y, x;
        if (SHAPE == 0) { y = ty * 32 + (it >> 2) * 8 + ((it >> 1) & 1) * 4 + (it & 1) + (lane >> 5) * 2; x = tx * 128 + (lane & 31) * 4; }
        else if (SHAPE == 3) { y = ty * 32 + it * 2 + (lane >> 5); x = tx * 128 + (lane & 31) * 4; }
        else { y = ty * 32 + (it >> 1) * 4 + (lane >> 4); x = tx * 128 + (it & 1) * 64 + (lane & 15) * 4; }
        if (y >= (uint32_t)H) continue;
        u32x4 *p = reinterpret_cast<u32x4 *>(base + ((size_t)y * W + x) * 4 + shift);
        if (NT) __builtin_nontemporal_store(v, p); else *p = v;
    }
}

template <int SHAPE, bool NT>
static void run(const char *what, uint8_t *buf, int shift, int bands)
{
    const uint32_t upp = SHAPE == 2 ? 8 * 68 : TX * TY, chunk = (upp + bands - 1) / bands;
    const dim3 grid(chunk * 8, N / (8 / bands));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<SHAPE, NT>), grid, dim3(64), 0, 0, buf, shift, bands, upp);
    CK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0;
    for (int r = 0; r < 10; r++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<SHAPE, NT>), grid, dim3(64), 0, 0, buf, shift, bands, upp);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best; sum += ms;
    }
    const double bytes = (double)N * PIC;
    printf("%-64s %s bands %d: %.4f ms avg %.4f best = %7.1f GB/s\n", what, NT ? "nt   " : "plain", bands, sum / 10, best, bytes / (sum / 10) / 1e6);
}

int main()
{
    uint8_t *buf;
    CK(hipMalloc(&buf, (size_t)N * PIC + 4096));
    CK(hipMemset(buf, 0, (size_t)N * PIC + 4096));
    for (int bands : {4, 8}) {
        run<0, true>("2 rows (r, r+2) x 512 B, 16 B past a line  [k_frame]", buf, 16, bands);
        run<0, true>("2 rows (r, r+2) x 512 B, line aligned", buf, 0, bands);
        run<0, true>("2 rows (r, r+2) x 512 B, 32 B past a line", buf, 32, bands);
        run<3, true>("2 rows (r, r+1) x 512 B, 16 B past a line", buf, 16, bands);
        run<3, true>("2 rows (r, r+1) x 512 B, line aligned", buf, 0, bands);
        run<1, true>("4 rows x 256 B, 16 B past a line", buf, 16, bands);
        run<1, true>("4 rows x 256 B, line aligned", buf, 0, bands);
        run<2, true>("1 row x 1024 B (256 x 16 tiles), 16 B past a line", buf, 16, bands);
        run<2, true>("1 row x 1024 B (256 x 16 tiles), line aligned", buf, 0, bands);
        run<0, false>("2 rows (r, r+2) x 512 B, 16 B past a line", buf, 16, bands);
        run<0, false>("2 rows (r, r+2) x 512 B, line aligned", buf, 0, bands);
        run<2, false>("1 row x 1024 B (256 x 16 tiles), line aligned", buf, 0, bands);
    }
    return 0;
}
