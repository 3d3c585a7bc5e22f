// launch_rate.hip -- how long does MI355X take just to start and retire N short workgroups?
// (k_post r01: 35 840 workgroups of 256 threads per launch; is the wave launch rate a floor?)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int LDS> __global__ void k_empty(uint32_t *sink, int spin)
{
    __shared__ uint32_t s[LDS / 4 > 0 ? LDS / 4 : 1];
    uint32_t v = threadIdx.x;
    for (int i = 0; i < spin; i++) v = v * 1664525u + 1013904223u;
    if (LDS) s[threadIdx.x % (LDS / 4 > 0 ? LDS / 4 : 1)] = v;
    if (v == 0x12345678u) sink[0] = v + (LDS ? s[0] : 0);
}

template <class F> static float best_ms(F f)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (r && ms < best) best = ms;
    }
    return best;
}

int main()
{
    uint32_t *sink; CK(hipMalloc(&sink, 4));
    for (int spin : {0, 100, 400}) {
        for (int wgs : {2048, 35840, 143360}) {
            for (int threads : {64, 256}) {
                float ms0 = best_ms([&] { hipLaunchKernelGGL(k_empty<0>, dim3(wgs), dim3(threads), 0, 0, sink, spin); });
                float ms6 = best_ms([&] { hipLaunchKernelGGL(k_empty<6144>, dim3(wgs), dim3(threads), 0, 0, sink, spin); });
                printf("spin %4d  %6d WGs x %3d thr: no LDS %7.1f us (%5.2f waves/ns)   6 KB LDS %7.1f us\n", spin, wgs, threads,
                       ms0 * 1e3, wgs * (threads / 64) / (ms0 * 1e6), ms6 * 1e3);
            }
        }
    }
    return 0;
}
