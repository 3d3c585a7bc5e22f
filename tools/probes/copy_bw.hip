// copy_bw.hip -- on-box HBM ceilings and FETCH_SIZE/WRITE_SIZE calibration for this repo's access
// shapes ("measure an on-box copy-kernel ceiling").
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/copy_bw.hip -o copy_bw ; run: ./copy_bw [MiB]
// Under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE each kernel moves a known byte count.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void copy16(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void copy4(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void read16(const uint4 *__restrict__ in, uint32_t *__restrict__ sink, size_t n)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = in[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void read4(const uint32_t *__restrict__ in, uint32_t *__restrict__ sink, size_t n)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= in[i];
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void write16(uint4 *__restrict__ out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ void write16_nt(uint4 *__restrict__ out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        __builtin_nontemporal_store((uint32_t)i, &out[i].x);
        __builtin_nontemporal_store(1u, &out[i].y);
        __builtin_nontemporal_store(2u, &out[i].z);
        __builtin_nontemporal_store(3u, &out[i].w);
    }
}
// one wave = 2 runs of 512 B at a 16-byte offset from the 128-byte lines (k_post's RGBA store shape)
__global__ void write16_off(uint4 *__restrict__ out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i + 1 < n; i += (size_t)gridDim.x * blockDim.x)
        out[i + 1] = make_uint4((uint32_t)i, 1, 2, 3);
}
// 1:2.67 read:write mix of k_post (read 1 byte of YUV planes per 2.67 bytes of RGBA written)
__global__ void expand(const uint32_t *__restrict__ in, uint4 *__restrict__ out, size_t n_out)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_out; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t v = in[i];     // 4 B in (luma) -> 16 B out; chroma adds 2 B more in the real kernel
        out[i] = make_uint4(v, v >> 8, v >> 16, v >> 24);
    }
}

template <class F> static float time_ms(F f, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char **argv)
{
    size_t mib = argc > 1 ? atoi(argv[1]) : 1024;
    size_t bytes = mib << 20;
    void *in, *out; uint32_t *sink;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    const int grid = 256 * 8, block = 256, reps = 10;
    float ms;
    ms = time_ms([&] { hipLaunchKernelGGL(copy16, dim3(grid), dim3(block), 0, 0, (const uint4 *)in, (uint4 *)out, bytes / 16); }, reps);
    printf("copy16  : %6zu MiB read + %6zu MiB written  %8.3f ms  %7.1f GB/s (read+write)\n", mib, mib, ms, 2.0 * bytes / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(copy4, dim3(grid), dim3(block), 0, 0, (const uint32_t *)in, (uint32_t *)out, bytes / 4); }, reps);
    printf("copy4   : %6zu MiB read + %6zu MiB written  %8.3f ms  %7.1f GB/s (read+write)\n", mib, mib, ms, 2.0 * bytes / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(read16, dim3(grid), dim3(block), 0, 0, (const uint4 *)in, sink, bytes / 16); }, reps);
    printf("read16  : %6zu MiB read                         %8.3f ms  %7.1f GB/s\n", mib, ms, 1.0 * bytes / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(read4, dim3(grid), dim3(block), 0, 0, (const uint32_t *)in, sink, bytes / 4); }, reps);
    printf("read4   : %6zu MiB read                         %8.3f ms  %7.1f GB/s\n", mib, ms, 1.0 * bytes / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(write16, dim3(grid), dim3(block), 0, 0, (uint4 *)out, bytes / 16); }, reps);
    printf("write16 : %6zu MiB written                      %8.3f ms  %7.1f GB/s\n", mib, ms, 1.0 * bytes / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(write16_nt, dim3(grid), dim3(block), 0, 0, (uint4 *)out, bytes / 16); }, reps);
    printf("write16n: %6zu MiB written (nontemporal)        %8.3f ms  %7.1f GB/s\n", mib, ms, 1.0 * bytes / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(write16_off, dim3(grid), dim3(block), 0, 0, (uint4 *)out, bytes / 16); }, reps);
    printf("write16o: %6zu MiB written (16 B off the lines) %8.3f ms  %7.1f GB/s\n", mib, ms, 1.0 * bytes / ms / 1e6);
    for (int g2 : {256, 512, 1024, 4096, 16384}) {
        ms = time_ms([&] { hipLaunchKernelGGL(write16, dim3(g2), dim3(block), 0, 0, (uint4 *)out, bytes / 16); }, reps);
        printf("write16 grid %5d: %8.3f ms  %7.1f GB/s\n", g2, ms, 1.0 * bytes / ms / 1e6);
    }
    ms = time_ms([&] { hipLaunchKernelGGL(expand, dim3(grid), dim3(block), 0, 0, (const uint32_t *)in, (uint4 *)out, bytes / 16); }, reps);
    printf("expand  : %6zu MiB read + %6zu MiB written  %8.3f ms  %7.1f GB/s (read+write)\n", mib / 4, mib, ms, 1.25 * bytes / ms / 1e6);
    return 0;
}
