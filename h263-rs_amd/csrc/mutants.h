// mutants.h -- the ARITHMETIC MUTANTS of the kernels, all in one place (tests/test_gpu_mutation.py; never the product).
//
// A mutant build differs from the product in exactly one arithmetic detail that the reference fixes bit for bit.  The
// parity suite is run against each of them on the MI355X and must fail where a soft-float model of the mutation says it
// will: that is the evidence that "bit-exact" in those tests means something.  The product build defines none of the
// macros below; every switch is then a compile-time `false` and the kernels contain nothing of this file.
//   -DH263MI_MUTATE_PAIRWISE             idct_1d sums its eight products as a balanced tree (idct.rs:52-65 sums in order)
//   -DH263MI_MUTATE_DEQUANT_SATURATION   the dequantiser's saturated values keep their low bits: 2047.9375 instead of 2047
//   -DH263MI_MUTATE_DEQUANT_WRAP         the dequantiser of wide LEVELs saturates where the reference's i16 product wraps
//   (the fourth mutant, libh263mi_fma.so, is a compiler flag: -ffp-contract=fast fuses the IDCT's multiplies into its adds)
// Included by recon_kernel.inl behind the definitions it uses (f32x2, splat2, BasisPtr, basis_pair).
#pragma once

namespace h263mi {
namespace mutants {

#if defined(H263MI_MUTATE_PAIRWISE)
constexpr bool kPairwise = true;
#else
constexpr bool kPairwise = false;
#endif
#if defined(H263MI_MUTATE_DEQUANT_SATURATION)
constexpr bool kDequantSaturation = true;
#else
constexpr bool kDequantSaturation = false;
#endif
#if defined(H263MI_MUTATE_DEQUANT_WRAP)
constexpr bool kDequantWrap = true;
#else
constexpr bool kDequantWrap = false;
#endif

// kPairwise: the eight rounded products summed as a balanced tree instead of in the order of the frequency index
H263_DEV void idct_1d_pairwise(BasisPtr B, const float in[8], f32x2 out[4], f32x2 first)
{
#pragma unroll
    for (int ip = 0; ip < 4; ip++) {
        f32x2 pr[8];
        pr[0] = first;
#pragma unroll
        for (int f = 1; f < 8; f++) pr[f] = splat2(in[f]) * basis_pair(B, f, ip);
        out[ip] = ((pr[0] + pr[1]) + (pr[2] + pr[3])) + ((pr[4] + pr[5]) + (pr[6] + pr[7]));
    }
}

}  // namespace mutants
}  // namespace h263mi
