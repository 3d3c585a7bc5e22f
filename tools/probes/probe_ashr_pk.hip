// probe_ashr_pk.hip -- what does gfx950's v_ashr_pk_u8_i32 return for negative inputs?
// hipcc (ROCm 7.2) selects it for clamp(x >> n, 0, 255) pairs; the BT.601 golden of the reference
// (yuv/src/bt601.rs:206-207, Y = 15 -> 0) failed on an MI355X with that lowering.
// RESULT (MI355X, ROCm 7.2, profiles/r01_probe_ashr_pk.txt): the instruction saturates correctly
// (negative -> 0, > 255 -> 255) and packs the two bytes into bits [15:0], but it leaves bits [31:16]
// of the destination register untouched, while hipcc's lowering of `clamp(a>>n,0,255) | clamp(b>>n,0,255)<<8`
// consumes the register as if those bits were zero.  When source and destination share a VGPR the
// stale upper half (a >> 16, unclamped) leaks into the result -- the "shift-then-clamp(C)" column.
// post_kernel.inl therefore never leaves the choice to the compiler: bt601_pack issues the instruction itself (inline
// asm), once per half of the destination (probe_ashr_pk_hi.hip).
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/probe_ashr_pk.hip -o probe_ashr_pk
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void k(const int *in, unsigned *out_asm, unsigned *out_c, int n)
{
    int i = threadIdx.x;
    if (i >= n) return;
    int a = in[i], b = in[i] + 70000;
    unsigned r;
    asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 16" : "=v"(r) : "v"(a), "v"(b));
    out_asm[i] = r;
    int ca = a >> 16, cb = b >> 16;
    ca = ca < 0 ? 0 : (ca > 255 ? 255 : ca);
    cb = cb < 0 ? 0 : (cb > 255 ? 255 : cb);
    out_c[i] = (unsigned)ca | ((unsigned)cb << 8);
}

int main()
{
    const int vals[] = {-43541, -1, 0, 65535, 65536, 1 << 20, 255 << 16, 256 << 16, 0x7fffffff, (int)0x80000000, -70001, -65536};
    const int n = sizeof(vals) / sizeof(vals[0]);
    int *d_in; unsigned *d_a, *d_c;
    hipMalloc(&d_in, sizeof vals); hipMalloc(&d_a, n * 4); hipMalloc(&d_c, n * 4);
    hipMemcpy(d_in, vals, sizeof vals, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, d_a, d_c, n);
    unsigned a[32], c[32];
    hipMemcpy(a, d_a, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c, d_c, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++)
        printf("in0=%11d in1=%11d  v_ashr_pk_u8_i32=0x%04x  shift-then-clamp(C)=0x%04x %s\n", vals[i], vals[i] + 70000,
               a[i], c[i], a[i] == c[i] ? "" : "<-- differs");
    return 0;
}
