// gather_cost.hip -- what does a DIVERGENT vector load cost the CU's memory pipeline, as a function of how its 64 lanes
// fall onto cache lines?  k_recon's eight motion-compensation loads (12 bytes per lane, two neighbouring lanes per
// macroblock row, a vector per macroblock) are the largest item of k_frame's TA budget (tools/probes/vmem_rate.hip: 179
// cycles against 17 for the same load with one vector).  This probe separates the candidates: G consecutive lanes share
// one randomly placed row segment (G = 1, 2, 4, 8, 16), each lane loading W dwords (1, 2, 3, 4), from a working set that
// misses the L1 and hits the L2 -- as the reference rows do in k_frame.
// Build on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/probes/gather_cost.hip -o gpurun_out/gather_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int PITCH = 2048;            // bytes per row of a wave's private region
#ifndef REGION_KB
#define REGION_KB 8
#endif
constexpr int REGION = REGION_KB * 1024;   // bytes per wave.  8 KB: 16 waves per CU x 8 KB = 128 KB per CU (4 x the L1), 4 MB per XCD (its L2);
                                           // 256 KB: 1 GB in all -- every line comes from DRAM
constexpr int ITERS = 256;

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// G consecutive lanes form a group that reads one contiguous run of G * 4W bytes at a random (row, dword-aligned column)
template <int G, int W, bool STRADDLE>
__device__ __forceinline__ uint32_t gather_offset(uint32_t l, uint32_t q, uint32_t seed)
{
    const uint32_t grp = l / G, in = l % G;
    const uint32_t h = hash(seed * 977u + grp * 131u + q * 7919u);
    const uint32_t row = h % (uint32_t)(REGION / PITCH);
    // STRADDLE: any dword column (the run may cross a 64-byte line); else the run starts on a multiple of its own size
    const uint32_t run = G * W * 4u;
    uint32_t col = ((h >> 8) % ((PITCH - run - 64u) / 4u)) * 4u;
    if (!STRADDLE) col = col / 64u * 64u + (((h >> 20) % ((64u / run) ? (64u / run) : 1u)) * run) % 64u;
    return row * PITCH + col + in * W * 4u;
}

template <int G, int W, bool STRADDLE>
__global__ __launch_bounds__(256) void k(uint8_t *buf, uint32_t *sink)
{
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), l = threadIdx.x & 63;
    uint8_t *base = buf + (size_t)wave * REGION;
    uint32_t acc = 0;
    for (uint32_t kk = 0; kk < ITERS; kk += 8) {
        uint32_t off[8];
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) off[q] = gather_offset<G, W, STRADDLE>(l, kk + q, wave);
        u32x4 r4[8]; u32x3 r3[8]; u32x2 r2[8]; uint32_t r1[8];
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) {
            const uint32_t o = off[q];
            if (W == 4) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r4[q]) : "v"(o), "s"(base) : "memory");
            else if (W == 3) asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(r3[q]) : "v"(o), "s"(base) : "memory");
            else if (W == 2) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(r2[q]) : "v"(o), "s"(base) : "memory");
            else asm volatile("global_load_dword %0, %1, %2" : "=v"(r1[q]) : "v"(o), "s"(base) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) {
            if (W == 4) acc ^= r4[q].x; else if (W == 3) acc ^= r3[q].x; else if (W == 2) acc ^= r2[q].x; else acc ^= r1[q];
        }
    }
    if (acc == 0x12345u) sink[0] = acc;
}

template <int G, int W, bool STRADDLE>
static void run(uint8_t *buf, uint32_t *sink, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<G, W, STRADDLE>), dim3(blocks), dim3(256), 0, 0, buf, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k<G, W, STRADDLE>), dim3(blocks), dim3(256), 0, 0, buf, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= 3;
    const double instr_per_cu = (double)blocks * 4 * ITERS / 256.0;
    const double groups = 64.0 / G;
    printf("x%d  %2d lanes per run (%4d B runs, %s)  %4.0f runs/instr  %d waves/SIMD: %7.1f cycles per wave-instruction per CU (2.4 GHz) = %5.2f per run, useful %6.1f GB/s\n",
           W, G, G * W * 4, STRADDLE ? "any dword" : "in one line", groups, waves_per_simd, ms * 1e-3 * 2.4e9 / instr_per_cu,
           ms * 1e-3 * 2.4e9 / instr_per_cu / groups, (double)blocks * 4 * ITERS * 64.0 * W * 4 / ms / 1e6);
}

int main()
{
    const int max_waves = 256 * 4 * 4;
    uint8_t *buf; uint32_t *sink;
    CK(hipMalloc(&buf, (size_t)max_waves * REGION + 4096)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, (size_t)max_waves * REGION + 4096));
    for (int w : {1, 4}) {
        run<1, 1, false>(buf, sink, w); run<1, 2, false>(buf, sink, w); run<1, 3, true>(buf, sink, w); run<1, 4, false>(buf, sink, w);
        run<2, 1, false>(buf, sink, w); run<2, 2, false>(buf, sink, w); run<2, 3, true>(buf, sink, w); run<2, 4, false>(buf, sink, w); run<2, 4, true>(buf, sink, w);
        run<4, 1, false>(buf, sink, w); run<4, 2, false>(buf, sink, w); run<4, 2, true>(buf, sink, w); run<4, 4, false>(buf, sink, w);
        run<8, 1, false>(buf, sink, w); run<8, 2, false>(buf, sink, w); run<8, 2, true>(buf, sink, w);
        run<16, 1, false>(buf, sink, w); run<16, 1, true>(buf, sink, w); run<16, 4, true>(buf, sink, w);
        run<64, 4, false>(buf, sink, w);
    }
    return 0;
}
