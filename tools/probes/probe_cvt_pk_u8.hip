// probe_cvt_pk_u8.hip -- what exactly does v_cvt_pk_u8_f32 compute on gfx950?
// Candidate for the last step of an all-intra reconstruction wave: pixel = clamp((int)v, 0, 255) with v a float
// (idct.rs:189-194 with a zero prediction: `as i16` truncates toward zero, then the clamp) -- one instruction instead
// of v_cvt_i32_f32 + v_med3_i32, IF the conversion truncates toward zero and saturates at both ends.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/probe_cvt_pk_u8.hip -o build/probe_cvt_pk_u8
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k(const float *in, uint32_t *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r = 0xa5a5a5a5u;                      // the bytes that are not selected must survive
    const float v = in[i];
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(r) : "v"(v));      // into byte 1
    out[i] = r;
}

int main()
{
    std::vector<float> h;
    for (int k = -70000; k <= 70000; k++) h.push_back((float)k / 64.0f);
    const float special[] = {0.0f, -0.0f, 0.49999997f, 0.5f, 0.50000006f, 0.99999994f, 1.0f, 1.5f, 2.5f, 254.5f, 254.99998f, 255.0f,
                             255.5f, 256.0f, 1e9f, -1e9f, INFINITY, -INFINITY, NAN, -0.5f, -0.99999994f, -1.0f, 3.4e38f, 1e-40f};
    for (float s : special) h.push_back(s);
    const int n = (int)h.size();
    float *d_in; uint32_t *d_out;
    CK(hipMalloc(&d_in, n * 4)); CK(hipMalloc(&d_out, n * 4));
    CK(hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, d_in, d_out, n);
    std::vector<uint32_t> o(n);
    CK(hipMemcpy(o.data(), d_out, n * 4, hipMemcpyDeviceToHost));
    int bad_trunc = 0, bad_rne = 0, bad_keep = 0, shown = 0;
    for (int i = 0; i < n; i++) {
        const float v = h[i];
        const uint32_t got = (o[i] >> 8) & 0xffu;
        if ((o[i] & 0xffff00ffu) != 0xa5a500a5u) bad_keep++;
        if (v != v) { printf("NaN -> %u\n", got); continue; }
        const double t = v < 0 ? ceil((double)v) : floor((double)v);
        const uint32_t want_trunc = t < 0 ? 0u : (t > 255 ? 255u : (uint32_t)t);
        const double r = nearbyint((double)v);
        const uint32_t want_rne = r < 0 ? 0u : (r > 255 ? 255u : (uint32_t)r);
        if (got != want_trunc) { bad_trunc++; if (shown < 12) { printf("  %.9g -> %u (truncation would give %u, nearest-even %u)\n", v, got, want_trunc, want_rne); shown++; } }
        if (got != want_rne) bad_rne++;
    }
    printf("%d inputs: %d differ from clamp(trunc(v), 0, 255), %d differ from clamp(nearest-even(v), 0, 255), %d touched the other bytes\n",
           n, bad_trunc, bad_rne, bad_keep);
    return 0;
}
