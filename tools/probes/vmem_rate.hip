// vmem_rate.hip -- what one vector-memory instruction costs a CU's memory pipeline, by access shape.
// k_frame's waves spend half their life queueing at VMEM issue (tools/phase_profile.py: "issue loads" 28 %, "store" 16 %
// of a reconstruction wave's life) while HBM runs at 70 % of the copy rate and the vector ALUs at 69 %: the price
// list below says which of the kernel's access shapes are the expensive ones.
// Every wave re-reads / re-writes its own few KB (L1 / L2 resident), so the numbers are pipeline costs, not DRAM rates.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/vmem_rate.hip -o build/vmem_rate ; run: ./build/vmem_rate
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
typedef u32x2 __attribute__((aligned(4))) u32x2_a4;

constexpr int PITCH = 2048;            // bytes per row of a wave's private region
constexpr int REGION = 64 * 1024;      // bytes per wave
constexpr int ITERS = 512;

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// byte offset of lane l in iteration k for access shape `shape`
__device__ __forceinline__ uint32_t shape_offset(int shape, uint32_t l, uint32_t k, uint32_t wave_seed)
{
    const uint32_t rot = (k & 7u) * 128u;       // moves the access by whole lines so that nothing can be hoisted
    switch (shape) {
    case 0: return rot + l * 16u;                                        // x4, 16-byte aligned, 1 KiB contiguous
    case 1: return rot + 4u + l * 16u;                                   // x4, 4 bytes off
    case 2: return rot + 4u + (l & 7u) * 16u + (l >> 3) * PITCH;         // k_post luma fetch: 8 rows x 128 B, 4 bytes off
    case 3: return rot + 2u + (l & 7u) * 8u + (l >> 3) * PITCH;          // k_post chroma fetch: x2 at 2-byte alignment, 8 rows x 64 B
    case 4: return rot + (l & 15u) * 8u + (l >> 4) * PITCH;              // MC, one vector for the wave: x3, stride 8, 4 rows
    case 5: {                                                            // MC, a vector per macroblock (2 lanes), +-16 px / rows
        const uint32_t h = hash(wave_seed * 131u + (l >> 1) * 7u + (l >> 4) * 1009u + (k >> 3));
        const uint32_t dx = (h & 31u) & ~3u, dy = (h >> 8) & 31u;
        return rot + 64u + (l & 15u) * 8u + dx + ((l >> 4) * 5u + dy) * PITCH;
    }
    case 6: return rot + (l & 15u) * 8u + (l >> 4) * PITCH;              // recon store: x2, 4 rows x 128 B
    case 7: return rot + 16u + (l & 31u) * 16u + (l >> 5) * 2u * PITCH;  // RGBA store: x4, 2 runs of 512 B, 16 bytes past a line
    case 8: return rot + (l & 31u) * 16u + (l >> 5) * 2u * PITCH;        // the same, line aligned
    case 9: return rot + (l & 7u) * 16u + (l >> 3) * PITCH;              // x4 aligned, 8 rows x 128 B
    case 10: return rot + (l & 7u) * 8u + (l >> 3) * PITCH;              // x2 aligned, 8 rows x 64 B
    default: return rot + l * 4u;                                        // dword, 256 B contiguous
    }
}

// The accesses are inline assembly: the compiler can neither hoist nor merge them.  The 8 offsets a lane cycles through
// are computed before the loop; a wave keeps at most 8 loads in flight.
template <int SHAPE, int WIDTH, bool STORE, bool NT>
__global__ __launch_bounds__(256) void k(uint8_t *buf, uint32_t *sink)
{
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), l = threadIdx.x & 63;
    uint8_t *base = buf + (size_t)wave * REGION;     // wave-uniform: the accesses use scalar-base addressing like the kernels
    uint32_t off[8];
#pragma unroll
    for (uint32_t q = 0; q < 8; q++) off[q] = shape_offset(SHAPE, l, q, wave);
    u32x4 v4 = {l, 1, 2, 3};
    u32x2 v2 = {l, 1};
    uint32_t acc = 0;
    for (uint32_t kk = 0; kk < ITERS; kk += 8) {
        u32x4 r4[8]; u32x3 r3[8]; u32x2 r2[8]; uint32_t r1[8];
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) {
            const uint32_t o = off[q];
            if (STORE) {
                if (WIDTH == 4 && NT) asm volatile("global_store_dwordx4 %0, %1, %2 nt" :: "v"(o), "v"(v4), "s"(base) : "memory");
                else if (WIDTH == 4) asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(o), "v"(v4), "s"(base) : "memory");
                else asm volatile("global_store_dwordx2 %0, %1, %2" :: "v"(o), "v"(v2), "s"(base) : "memory");
            } else {
                if (WIDTH == 4) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r4[q]) : "v"(o), "s"(base) : "memory");
                else if (WIDTH == 3) asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(r3[q]) : "v"(o), "s"(base) : "memory");
                else if (WIDTH == 2) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(r2[q]) : "v"(o), "s"(base) : "memory");
                else asm volatile("global_load_dword %0, %1, %2" : "=v"(r1[q]) : "v"(o), "s"(base) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!STORE) {
#pragma unroll
            for (uint32_t q = 0; q < 8; q++) {
                if (WIDTH == 4) acc ^= r4[q].x; else if (WIDTH == 3) acc ^= r3[q].x; else if (WIDTH == 2) acc ^= r2[q].x; else acc ^= r1[q];
            }
        }
    }
    if (!STORE && acc == 0x12345u) sink[0] = acc;
}

template <int SHAPE, int WIDTH, bool STORE, bool NT>
static void run(const char *what, uint8_t *buf, uint32_t *sink, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd;        // 256-thread blocks = 4 waves: one per SIMD
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<SHAPE, WIDTH, STORE, NT>), dim3(blocks), dim3(256), 0, 0, buf, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k<SHAPE, WIDTH, STORE, NT>), dim3(blocks), dim3(256), 0, 0, buf, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= 3;
    const double instr_per_cu = (double)blocks * 4 * ITERS / 256.0;
    const double bytes = (double)blocks * 4 * ITERS * 64.0 * WIDTH * 4;
    printf("%-58s %d waves/SIMD: %7.1f cycles per wave-instruction per CU (2.4 GHz), %7.1f GB/s\n", what, waves_per_simd,
           ms * 1e-3 * 2.4e9 / instr_per_cu, bytes / ms / 1e6);
}

int main()
{
    const int max_waves = 256 * 4 * 4;
    uint8_t *buf; uint32_t *sink;
    CK(hipMalloc(&buf, (size_t)max_waves * REGION + 4096)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, (size_t)max_waves * REGION + 4096));
    for (int w : {1, 4}) {
        run<0, 4, false, false>("load x4, 16-B aligned, 1 KiB contiguous", buf, sink, w);
        run<1, 4, false, false>("load x4, 4 bytes off, 1 KiB contiguous", buf, sink, w);
        run<9, 4, false, false>("load x4 aligned, 8 rows x 128 B", buf, sink, w);
        run<2, 4, false, false>("load x4, 4 bytes off, 8 rows x 128 B (k_post luma)", buf, sink, w);
        run<10, 2, false, false>("load x2 aligned, 8 rows x 64 B", buf, sink, w);
        run<3, 2, false, false>("load x2, 2-B aligned, 8 rows x 64 B (k_post chroma)", buf, sink, w);
        run<4, 3, false, false>("load x3, stride 8, 4 rows (MC, one vector)", buf, sink, w);
        run<5, 3, false, false>("load x3, stride 8, vector per macroblock (MC)", buf, sink, w);
        run<11, 1, false, false>("load dword, 256 B contiguous", buf, sink, w);
        run<6, 2, true, false>("store x2, 4 rows x 128 B (recon planes)", buf, sink, w);
        run<0, 4, true, false>("store x4 aligned, 1 KiB contiguous", buf, sink, w);
        run<8, 4, true, false>("store x4, 2 runs of 512 B, line aligned", buf, sink, w);
        run<7, 4, true, false>("store x4, 2 runs of 512 B, 16 B past a line (RGBA)", buf, sink, w);
        run<7, 4, true, true>("store x4 nt, 2 runs of 512 B, 16 B past a line (RGBA, nt)", buf, sink, w);
        run<8, 4, true, true>("store x4 nt, 2 runs of 512 B, line aligned", buf, sink, w);
    }
    return 0;
}
