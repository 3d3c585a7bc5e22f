// fetch_calib.hip -- what does rocprofv3's FETCH_SIZE count, per access SHAPE, on gfx950?
// profiles/traffic_latest.json doubles FETCH_SIZE (the MI355X guide's calibration for wide 16-byte streaming reads,
// reproduced in profiles/r01_copy_bw.txt).  k_frame's reads are not that shape: 12-byte motion-compensation gathers (two
// neighbouring lanes per macroblock row, a vector per macroblock), 16-byte strip loads 4 bytes past a 16-byte boundary, 8-byte
// strip loads at 2-byte alignment, 16-byte record / coefficient loads.  Each kernel below reads a buffer far larger than
// the L2s and the infinity cache (default 2 GiB) in ONE of those shapes such that every 64-byte line of the buffer is
// touched exactly once -- so the bytes that must come from HBM are known: the buffer.  tools/fetch_calib.sh runs it under
// `rocprofv3 --pmc FETCH_SIZE` and stores counter-units per true byte for each shape.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/fetch_calib.hip -o /tmp/fetch_calib ; run: /tmp/fetch_calib [MiB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x4 __attribute__((aligned(4))) u32x4_a4;
typedef u32x3 __attribute__((aligned(4))) u32x3_a4;
typedef u32x2 __attribute__((aligned(2))) u32x2_a2;

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// (a) the calibrated shape: 16 bytes per lane, 1 KiB per wave-instruction, contiguous
__global__ __launch_bounds__(256) void k_calib_stream16(const uint8_t *buf, size_t bytes, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t o = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; o + 16 <= bytes; o += (size_t)gridDim.x * blockDim.x * 16) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(buf + o);
        acc ^= v.x ^ v.w;
    }
    if (acc == 0x12345u) sink[0] = acc;
}

// (b) k_recon's motion compensation: 12 bytes per lane at a dword-aligned address, two neighbouring lanes 8 bytes apart
// (bytes 0..19 of a 64-byte line, somewhere in it), the 32 lane pairs of an instruction on 32 different lines picked by
// a hash -- every line of the wave's region exactly once over the loop
__global__ __launch_bounds__(256) void k_calib_gather12(const uint8_t *buf, size_t bytes, uint32_t *sink)
{
    const size_t lines = bytes / 64, waves = (size_t)gridDim.x * blockDim.x / 64;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64, lane = threadIdx.x & 63;
    const size_t per_wave = lines / waves;                    // lines of this wave's region (a multiple of 32 by construction)
    const uint8_t *region = buf + wave * per_wave * 64;
    uint32_t acc = 0;
    for (size_t it = 0; it < per_wave / 32; it++) {
        // a permutation of the region's lines: line = (it + k * stride) mod per_wave with stride = per_wave / 32
        const size_t line = (it + (lane >> 1) * (per_wave / 32)) % per_wave;
        const uint32_t in_line = (hash((uint32_t)(line * 2654435761u)) % 11u) * 4u;      // 0..40: the pair's 20 bytes stay inside the line
        const u32x3 v = *reinterpret_cast<const u32x3_a4 *>(region + line * 64 + in_line + (lane & 1) * 8);
        acc ^= v.x ^ v.z;
    }
    if (acc == 0x12345u) sink[0] = acc;
}

// (c) k_post's strip fetch: 8 lanes x 16 bytes per row, starting 4 bytes past a 128-byte boundary (the last lane's load
// runs 4 bytes into the next 128 bytes), 8 rows per instruction; rows are `pitch` apart.  Covers a 2-D region once.
__global__ __launch_bounds__(256) void k_calib_strip16(const uint8_t *buf, size_t bytes, uint32_t *sink)
{
    const size_t pitch = 2048, rows = bytes / pitch;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64, lane = threadIdx.x & 63, waves = (size_t)gridDim.x * blockDim.x / 64;
    uint32_t acc = 0;
    // tiles of 128 bytes x 8 rows; tile t -> (row group t / 16, column t % 16); the 4-byte skew makes neighbouring tiles share a line
    for (size_t t = wave; t < (rows / 8) * 16; t += waves) {
        const size_t row = (t / 16) * 8 + (lane >> 3), col = (t % 16) * 128 + (lane & 7) * 16 + 4;
        if (col + 16 > pitch) continue;
        const u32x4 v = *reinterpret_cast<const u32x4_a4 *>(buf + row * pitch + col);
        acc ^= v.x ^ v.w;
    }
    if (acc == 0x12345u) sink[0] = acc;
}

// (d) k_post's chroma strip fetch: 8 lanes x 8 bytes per row at 2-byte alignment, 8 rows x 64 bytes per instruction
__global__ __launch_bounds__(256) void k_calib_strip8(const uint8_t *buf, size_t bytes, uint32_t *sink)
{
    const size_t pitch = 1024, rows = bytes / pitch;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64, lane = threadIdx.x & 63, waves = (size_t)gridDim.x * blockDim.x / 64;
    uint32_t acc = 0;
    for (size_t t = wave; t < (rows / 8) * 16; t += waves) {
        const size_t row = (t / 16) * 8 + (lane >> 3), col = (t % 16) * 64 + (lane & 7) * 8 + 2;
        if (col + 8 > pitch) continue;
        const u32x2 v = *reinterpret_cast<const u32x2_a2 *>(buf + row * pitch + col);
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x12345u) sink[0] = acc;
}

int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? (size_t)atoi(argv[1]) : 2048;
    // a size every shape divides evenly: 4096 workgroups x 4 waves x 32 lines x 64 bytes = 32 MiB granules
    const size_t bytes = (mib << 20) / (32u << 20) * (32u << 20);
    uint8_t *buf; uint32_t *sink;
    CK(hipMalloc(&buf, bytes + 4096)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, bytes + 4096));
    CK(hipDeviceSynchronize());
    const dim3 grid(4096), block(256);
    for (int rep = 0; rep < 3; rep++) {           // each launch reads the whole buffer once: nothing of it survives in the caches
        hipLaunchKernelGGL(k_calib_stream16, grid, block, 0, 0, buf, bytes, sink);
        hipLaunchKernelGGL(k_calib_gather12, grid, block, 0, 0, buf, bytes, sink);
        hipLaunchKernelGGL(k_calib_strip16, grid, block, 0, 0, buf, bytes, sink);
        hipLaunchKernelGGL(k_calib_strip8, grid, block, 0, 0, buf, bytes, sink);
    }
    CK(hipDeviceSynchronize());
    printf("bytes %zu\n", bytes);
    return 0;
}
