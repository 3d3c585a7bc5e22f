// probe_ashr_pk_hi.hip -- does v_ashr_pk_u8_i32 with op_sel:[0,0,0,1] write bits [31:16] of its destination and keep
// bits [15:0]?  (The plain form writes [15:0] and keeps [31:16]: probe_ashr_pk.hip.)  If so, four saturated bytes
// R | G<<8 | B<<16 | A<<24 take two instructions and no byte permute.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/probe_ashr_pk_hi.hip -o probe_ashr_pk_hi
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void k(const int *in, unsigned *out, int n)
{
    int i = threadIdx.x;
    if (i >= n) return;
    const int a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
    unsigned r = 0xdeadbeefu;
    asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 16" : "+v"(r) : "v"(a), "v"(b));
    asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 16 op_sel:[0,0,0,1]" : "+v"(r) : "v"(c), "v"(d));
    out[2 * i] = r;
    unsigned q = 0xdeadbeefu;                      // the other order: high half first
    asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 16 op_sel:[0,0,0,1]" : "+v"(q) : "v"(c), "v"(d));
    asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 16" : "+v"(q) : "v"(a), "v"(b));
    out[2 * i + 1] = q;
}

static unsigned sat(int v) { v >>= 16; return v < 0 ? 0u : v > 255 ? 255u : (unsigned)v; }

int main()
{
    const int seeds[] = {-43541, -1, 0, 65535, 65536, 1 << 20, 255 << 16, 256 << 16, 0x7fffffff, (int)0x80000000, -70001, 12345678};
    const int n = sizeof(seeds) / sizeof(seeds[0]);
    int h[4 * 16];
    for (int i = 0; i < n; i++) { h[4 * i] = seeds[i]; h[4 * i + 1] = seeds[(i + 3) % n]; h[4 * i + 2] = seeds[(i + 5) % n]; h[4 * i + 3] = 255 << 16; }
    int *d_in; unsigned *d_o;
    hipMalloc(&d_in, sizeof h); hipMalloc(&d_o, 2 * n * 4);
    hipMemcpy(d_in, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, d_o, n);
    unsigned o[32];
    hipMemcpy(o, d_o, 2 * n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) {
        const unsigned want = sat(h[4 * i]) | sat(h[4 * i + 1]) << 8 | sat(h[4 * i + 2]) << 16 | sat(h[4 * i + 3]) << 24;
        printf("lo-then-hi=0x%08x hi-then-lo=0x%08x want=0x%08x %s\n", o[2 * i], o[2 * i + 1], want,
               (o[2 * i] == want && o[2 * i + 1] == want) ? "" : "<-- differs");
        bad += o[2 * i] != want || o[2 * i + 1] != want;
    }
    printf(bad ? "op_sel high-half write: NOT as hoped (%d rows differ)\n" : "op_sel high-half write: works (%d rows differ)\n", bad);
    return 0;
}
