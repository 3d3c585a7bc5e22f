// write_pattern.hip -- how fast can an MI355X write a 2-D image (RGBA frames, 7680-byte rows) as a
// function of the shape each wave writes?  k_post is bound by its 531 MB of RGBA stores per launch.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/write_pattern.hip -o write_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// image: n_pic pictures of rows x pitch bytes.  A wave writes a tile of tile_wb bytes x tile_rows rows
// (16 B per lane per store); tiles are numbered raster order inside a picture; workgroups of `wpw`
// waves take consecutive tiles.  xcd != 0: XCD-aware order (workgroup b handles chunk b%8).
__global__ void write_tiles(uint8_t *out, int n_pic, int rows, int pitch, int tile_wb, int tile_rows, int off,
                            int xcd_order, int persistent)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpw = blockDim.x >> 6;
    const int tx_n = pitch / tile_wb, ty_n = rows / tile_rows;
    const long tiles = (long)n_pic * tx_n * ty_n, wgs = (tiles + wpw - 1) / wpw;
    const long chunk = (wgs + 7) / 8;
    for (long b = blockIdx.x; b < (persistent ? wgs : (long)gridDim.x); b += gridDim.x) {
        long wg = xcd_order ? (b & 7) * chunk + (b >> 3) : b;
        if (wg >= wgs) continue;
        long t = wg * wpw + wave;
        if (t >= tiles) continue;
        int pic = t / (tx_n * ty_n), rem = t % (tx_n * ty_n), ty = rem / tx_n, tx = rem % tx_n;
        uint8_t *base = out + ((size_t)pic * rows + (size_t)ty * tile_rows) * pitch + (size_t)tx * tile_wb + off;
        const int lanes_per_row = tile_wb / 16;           // 32 for 512 B
        if (lanes_per_row <= 64) {
            const int rows_per_inst = 64 / lanes_per_row;
            for (int r = 0; r < tile_rows; r += rows_per_inst) {
                int rr = r + lane / lanes_per_row, c = (lane % lanes_per_row) * 16;
                *reinterpret_cast<uint4 *>(base + (size_t)rr * pitch + c) = make_uint4(lane, r, 2, 3);
            }
        } else {
            for (int r = 0; r < tile_rows; r++)
                for (int c = lane * 16; c < tile_wb; c += 1024)
                    *reinterpret_cast<uint4 *>(base + (size_t)r * pitch + c) = make_uint4(lane, r, 2, 3);
        }
    }
}

int main()
{
    const int n_pic = 64, rows = 1088, pitch = 7680;
    size_t bytes = (size_t)n_pic * rows * pitch + 4096;
    uint8_t *out;
    CK(hipMalloc(&out, bytes));
    CK(hipMemset(out, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Cfg { int tile_wb, tile_rows, wpw, off, xcd, persistent; const char *what; };
    const Cfg cfgs[] = {
        {512, 8, 4, 0, 1, 0, "wave 512Bx8, aligned, xcd order (k_post shape, aligned)"},
        {512, 8, 4, 16, 1, 0, "wave 512Bx8, +16B offset, xcd order (k_post today)"},
        {512, 8, 4, 0, 0, 0, "wave 512Bx8, aligned, round-robin order"},
        {512, 32, 4, 0, 1, 0, "wave 512Bx32 rows"},
        {1024, 8, 4, 0, 1, 0, "wave 1024Bx8"},
        {2048, 8, 4, 0, 1, 0, "wave 2048Bx8"},
        {7680, 1, 4, 0, 1, 0, "wave = 1 full row"},
        {7680, 8, 4, 0, 1, 0, "wave = 8 full rows"},
        {512, 8, 1, 0, 1, 0, "1 wave/WG 512Bx8"},
        {512, 8, 4, 0, 1, 1, "persistent 2048 WGs, 512Bx8"},
        {512, 8, 4, 0, 0, 1, "persistent 2048 WGs, 512Bx8, round-robin"},
        {2048, 8, 4, 0, 1, 1, "persistent 2048 WGs, 2048Bx8"},
        {7680, 8, 4, 0, 1, 1, "persistent 2048 WGs, 8 full rows"},
        {7680, 8, 4, 0, 0, 1, "persistent 2048 WGs, 8 full rows, round-robin"},
    };
    for (const Cfg &c : cfgs) {
        const int tx_n = pitch / c.tile_wb, ty_n = rows / c.tile_rows;
        const long tiles = (long)n_pic * tx_n * ty_n, wgs = (tiles + c.wpw - 1) / c.wpw;
        const long grid = c.persistent ? 2048 : ((wgs + 7) / 8) * 8;
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(write_tiles, dim3(grid), dim3(64 * c.wpw), 0, 0, out, n_pic, rows, pitch, c.tile_wb, c.tile_rows, c.off, c.xcd, c.persistent);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        double wrote = (double)tiles * c.tile_wb * c.tile_rows;
        printf("%-62s %7.3f ms  %7.1f GB/s\n", c.what, best, wrote / best / 1e6);
    }
    return 0;
}
