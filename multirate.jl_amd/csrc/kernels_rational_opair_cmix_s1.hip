// kernels_rational_opair_cmix_s1.hip -- instantiations of opair_kernel.inc: complex samples with Float64 arithmetic (ComplexF32 samples with
// Float64 taps, every component widened exactly; ComplexF64 samples), SMIN = 1, tapsPerPhi = 1..32, STRICT and FUSED.
#include "opair_kernel.inc"

namespace mrhip {

hipError_t launch_opair_cmix_s1(bool x_f64, bool fused, int T, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus)
{
    if (x_f64)
        return fused ? launch_opair_T<true, 2, 1, double, double>(T, block, lds, s, a, pa, num_cus)
                     : launch_opair_T<false, 2, 1, double, double>(T, block, lds, s, a, pa, num_cus);
    return fused ? launch_opair_T<true, 2, 1, float, double>(T, block, lds, s, a, pa, num_cus)
                 : launch_opair_T<false, 2, 1, float, double>(T, block, lds, s, a, pa, num_cus);
}

}  // namespace mrhip
