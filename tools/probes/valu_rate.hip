// valu_rate.hip -- issue cost (cycles per wave-instruction per SIMD, all SIMDs busy, 8 waves/SIMD) of the
// VALU instructions the kernels of this repo lean on.  k_recon / k_post are instruction-bound, so the real
// price of each opcode on gfx950 decides which formulation is cheapest.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define ITERS 2048
#define BODY(ASM)                                                                                                   \
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    uint32_t b = seed + threadIdx.x, c = seed * 3 + 1;                                                             \
    for (int i = 0; i < ITERS; i++) {                                                                               \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                                        \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
    }                                                                                                               \
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345679u) sink[0] = a0;

#define K(NAME, ASM) __global__ void NAME(uint32_t *sink, uint32_t seed) { BODY(ASM) }

#define A_ADD(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define A_MAD24(n) "v_mad_i32_i24 %" #n ", %" #n ", %8, %9\n"
#define A_MUL24(n) "v_mul_i32_i24 %" #n ", %" #n ", %8\n"
#define A_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define A_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define A_MED3I(n) "v_med3_i32 %" #n ", %" #n ", %8, %9\n"
#define A_MED3F(n) "v_med3_f32 %" #n ", %" #n ", %8, %9\n"
#define A_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 8, 8\n"
#define A_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 8, %8\n"
#define A_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define A_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define A_MULF(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define A_ADDF(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define A_CVTFI(n) "v_cvt_f32_i32 %" #n ", %" #n "\n"
#define A_CVTIF(n) "v_cvt_i32_f32 %" #n ", %" #n "\n"
#define A_LERP(n) "v_lerp_u8 %" #n ", %" #n ", %8, %9\n"
#define A_ALIGNB(n) "v_alignbyte_b32 %" #n ", %" #n ", %8, 1\n"
#define A_PKADD16(n) "v_pk_add_i16 %" #n ", %" #n ", %8\n"
#define A_PKMAX16(n) "v_pk_max_i16 %" #n ", %" #n ", %8\n"
#define A_PKMUL16(n) "v_pk_mul_lo_u16 %" #n ", %" #n ", %8\n"
#define A_BFI(n) "v_bfi_b32 %" #n ", %8, %9, %" #n "\n"
#define A_SAD(n) "v_sad_u8 %" #n ", %" #n ", %8, %9\n"
#define A_ASHRPK(n) "v_ashr_pk_u8_i32 %" #n ", %" #n ", %8, 16\n"
#define A_DOT4(n) "v_dot4_i32_i8 %" #n ", %" #n ", %8, %9\n"
#define A_SATPK(n) "v_sat_pk_u8_i16 %" #n ", %" #n "\n"
#define A_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define A_OR(n) "v_or_b32 %" #n ", %" #n ", %8\n"
#define A_LSHL(n) "v_lshlrev_b32 %" #n ", 3, %" #n "\n"
#define A_LSHR(n) "v_lshrrev_b32 %" #n ", 3, %" #n "\n"
#define A_ASHR(n) "v_ashrrev_i32 %" #n ", 3, %" #n "\n"
#define A_MAXI(n) "v_max_i32 %" #n ", %" #n ", %8\n"
#define A_MINF(n) "v_min_f32 %" #n ", %" #n ", %8\n"
#define A_SUB(n) "v_sub_u32 %" #n ", %" #n ", %8\n"
#define A_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define A_MULU24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define A_MADU24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n"
#define A_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 2, %8\n"
#define A_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define A_CVTUB0(n) "v_cvt_f32_ubyte0 %" #n ", %" #n "\n"
#define A_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define A_FMAC(n) "v_fmac_f32 %" #n ", %8, %9\n"
#define A_CND64(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[20:21]\n"
#define A_CMP(n) "v_cmp_lt_i32 vcc, %" #n ", %8\n"
#define A_CMPCND(n) "v_cmp_lt_i32 vcc, %" #n ", %8\nv_cndmask_b32 %" #n ", %" #n ", %9, vcc\n"
#define A_ADDCO(n) "v_add_co_u32 %" #n ", vcc, %" #n ", %8\n"
#define A_PKADDF(n) "v_pk_add_u16 %" #n ", %" #n ", %8\n"
#define A_SDWA(n) "v_add_u32_sdwa %" #n ", %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define A_MAX3(n) "v_max3_i32 %" #n ", %" #n ", %8, %9\n"
#define A_XAD(n) "v_xad_u32 %" #n ", %" #n ", %8, %9\n"
#define A_MULHI(n) "v_mul_hi_u32 %" #n ", %" #n ", %8\n"
#define A_CVTPKU8(n) "v_cvt_pk_u8_f32 %" #n ", %" #n ", 1, %8\n"

K(k_add, A_ADD) K(k_mad24, A_MAD24) K(k_mul24, A_MUL24) K(k_mullo, A_MULLO) K(k_perm, A_PERM) K(k_med3i, A_MED3I)
K(k_med3f, A_MED3F) K(k_bfe, A_BFE) K(k_lshlor, A_LSHLOR) K(k_add3, A_ADD3) K(k_cndmask, A_CNDMASK) K(k_mulf, A_MULF)
K(k_addf, A_ADDF) K(k_cvtfi, A_CVTFI) K(k_cvtif, A_CVTIF) K(k_lerp, A_LERP) K(k_alignb, A_ALIGNB) K(k_pkadd16, A_PKADD16)
K(k_pkmax16, A_PKMAX16) K(k_pkmul16, A_PKMUL16) K(k_bfi, A_BFI) K(k_sad, A_SAD) K(k_ashrpk, A_ASHRPK) K(k_dot4, A_DOT4)
K(k_satpk, A_SATPK)
K(k_and, A_AND) K(k_or, A_OR) K(k_lshl, A_LSHL) K(k_lshr, A_LSHR) K(k_ashr, A_ASHR) K(k_maxi, A_MAXI) K(k_minf, A_MINF)
K(k_sub, A_SUB) K(k_mov, A_MOV) K(k_mulu24, A_MULU24) K(k_madu24, A_MADU24) K(k_lshladd, A_LSHLADD) K(k_andor, A_ANDOR)
K(k_cvtub0, A_CVTUB0) K(k_fma, A_FMA) K(k_fmac, A_FMAC) K(k_cnd64, A_CND64) K(k_cmp, A_CMP) K(k_cmpcnd, A_CMPCND)
K(k_addco, A_ADDCO) K(k_pkaddu16, A_PKADDF) K(k_sdwa, A_SDWA) K(k_max3, A_MAX3) K(k_xad, A_XAD) K(k_mulhi, A_MULHI)
K(k_cvtpku8, A_CVTPKU8)

// 64-bit register pair forms
__global__ void k_pkmulf(uint32_t *sink, uint32_t seed)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[8], b = {1.0000001f, 0.9999999f};
    for (int k = 0; k < 8; k++) a[k] = f2{(float)threadIdx.x + k, (float)seed};
    for (int i = 0; i < ITERS; i++)
        asm volatile("v_pk_mul_f32 %0, %0, %8\nv_pk_mul_f32 %1, %1, %8\nv_pk_mul_f32 %2, %2, %8\nv_pk_mul_f32 %3, %3, %8\n"
                     "v_pk_mul_f32 %4, %4, %8\nv_pk_mul_f32 %5, %5, %8\nv_pk_mul_f32 %6, %6, %8\nv_pk_mul_f32 %7, %7, %8\n"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));
    float s = 0; for (int k = 0; k < 8; k++) s += a[k].x + a[k].y;
    if (s == 1234.5f) sink[0] = 1;
}
__global__ void k_lshl64(uint32_t *sink, uint32_t seed)
{
    unsigned long long a[8];
    for (int k = 0; k < 8; k++) a[k] = ((unsigned long long)seed << 32) | (threadIdx.x + k);
    for (int i = 0; i < ITERS; i++)
        asm volatile("v_lshrrev_b64 %0, 1, %0\nv_lshrrev_b64 %1, 1, %1\nv_lshrrev_b64 %2, 1, %2\nv_lshrrev_b64 %3, 1, %3\n"
                     "v_lshrrev_b64 %4, 1, %4\nv_lshrrev_b64 %5, 1, %5\nv_lshrrev_b64 %6, 1, %6\nv_lshrrev_b64 %7, 1, %7\n"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
    unsigned long long s = 0; for (int k = 0; k < 8; k++) s ^= a[k];
    if (s == 0x1234567812345678ull) sink[0] = 1;
}
__global__ void k_salu(uint32_t *sink, uint32_t seed)
{
    uint32_t a = seed, b = seed + 1, c = seed + 2, d = seed + 3;
    for (int i = 0; i < ITERS; i++)
        asm volatile("s_add_u32 %0, %0, %1\ns_add_u32 %1, %1, %2\ns_add_u32 %2, %2, %3\ns_add_u32 %3, %3, %0\n"
                     "s_add_u32 %0, %0, %1\ns_add_u32 %1, %1, %2\ns_add_u32 %2, %2, %3\ns_add_u32 %3, %3, %0\n"
                     : "+s"(a), "+s"(b), "+s"(c), "+s"(d));
    if ((a ^ b ^ c ^ d) == 0x12345679u) sink[0] = a;
}

typedef void (*kern_t)(uint32_t *, uint32_t);
int main()
{
    uint32_t *sink; CK(hipMalloc(&sink, 4));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const double clk_ghz = prop.clockRate / 1e6;
    struct { const char *name; kern_t k; } ks[] = {
        {"v_add_u32", k_add}, {"v_mad_i32_i24", k_mad24}, {"v_mul_i32_i24", k_mul24}, {"v_mul_lo_u32", k_mullo},
        {"v_perm_b32", k_perm}, {"v_med3_i32", k_med3i}, {"v_med3_f32", k_med3f}, {"v_bfe_u32", k_bfe},
        {"v_lshl_or_b32", k_lshlor}, {"v_add3_u32", k_add3}, {"v_cndmask_b32", k_cndmask}, {"v_mul_f32", k_mulf},
        {"v_add_f32", k_addf}, {"v_cvt_f32_i32", k_cvtfi}, {"v_cvt_i32_f32", k_cvtif}, {"v_lerp_u8", k_lerp},
        {"v_alignbyte_b32", k_alignb}, {"v_pk_add_i16", k_pkadd16}, {"v_pk_max_i16", k_pkmax16},
        {"v_pk_mul_lo_u16", k_pkmul16}, {"v_bfi_b32", k_bfi}, {"v_sad_u8", k_sad}, {"v_ashr_pk_u8_i32", k_ashrpk},
        {"v_dot4_i32_i8", k_dot4}, {"v_sat_pk_u8_i16", k_satpk}, {"v_pk_mul_f32", k_pkmulf}, {"v_lshrrev_b64", k_lshl64},
        {"s_add_u32 (SALU)", k_salu},
        {"v_and_b32", k_and}, {"v_or_b32", k_or}, {"v_lshlrev_b32", k_lshl}, {"v_lshrrev_b32", k_lshr}, {"v_ashrrev_i32", k_ashr},
        {"v_max_i32", k_maxi}, {"v_min_f32", k_minf}, {"v_sub_u32", k_sub}, {"v_mov_b32", k_mov}, {"v_mul_u32_u24", k_mulu24},
        {"v_mad_u32_u24", k_madu24}, {"v_lshl_add_u32", k_lshladd}, {"v_and_or_b32", k_andor}, {"v_cvt_f32_ubyte0", k_cvtub0},
        {"v_fma_f32", k_fma}, {"v_fmac_f32", k_fmac}, {"v_cndmask_b32_e64 sgpr", k_cnd64}, {"v_cmp_lt_i32", k_cmp},
        {"v_cmp + v_cndmask (2 instr)", k_cmpcnd}, {"v_add_co_u32", k_addco}, {"v_pk_add_u16", k_pkaddu16},
        {"v_add_u32_sdwa", k_sdwa}, {"v_max3_i32", k_max3}, {"v_xad_u32", k_xad}, {"v_mul_hi_u32", k_mulhi},
        {"v_cvt_pk_u8_f32", k_cvtpku8},
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %.2f GHz\n", prop.name, cus, clk_ghz);
    for (int waves_per_simd : {8}) {
        // one workgroup of 256 threads = one wave per SIMD; `waves_per_simd` workgroups per CU
        const int grid = cus * waves_per_simd;
        printf("--- %d waves per SIMD ---\n", waves_per_simd);
        for (auto &k : ks) {
            float best = 1e9;
            for (int r = 0; r < 3; r++) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k.k, dim3(grid), dim3(256), 0, 0, sink, 12345u);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double inst_per_simd = (double)waves_per_simd * ITERS * 8;
            printf("%-20s %8.3f ms  -> %5.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", k.name, best,
                   best * 1e-3 * clk_ghz * 1e9 / inst_per_simd, clk_ghz);
        }
    }
    return 0;
}
