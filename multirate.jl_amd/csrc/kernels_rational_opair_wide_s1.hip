// kernels_rational_opair_wide_s1.hip -- instantiations of opair_kernel.inc: Float64 arithmetic on real samples (Float64 samples, or Float32
// samples widened exactly: the reference README's Float64-taps x Float32-samples case), SMIN = 1
// (M > L), tapsPerPhi = 1..32, STRICT and FUSED.
#include "opair_kernel.inc"

namespace mrhip {

hipError_t launch_opair_wide_s1(bool x_f64, bool fused, int T, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus)
{
    if (x_f64)
        return fused ? launch_opair_T<true, 1, 1, double, double>(T, block, lds, s, a, pa, num_cus)
                     : launch_opair_T<false, 1, 1, double, double>(T, block, lds, s, a, pa, num_cus);
    return fused ? launch_opair_T<true, 1, 1, float, double>(T, block, lds, s, a, pa, num_cus)
                 : launch_opair_T<false, 1, 1, float, double>(T, block, lds, s, a, pa, num_cus);
}

}  // namespace mrhip
