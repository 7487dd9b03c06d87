// kernels_rational_owave_wide_s1.hip -- instantiations of owave_kernel.inc: Float64 arithmetic on real samples (Float64 samples, or
// Float32 samples widened exactly: the reference README's Float64-taps x Float32-samples case), SMIN = 1, STRICT and FUSED.
#include "owave_kernel.inc"

namespace mrhip {

hipError_t launch_owave_wide_s1(bool x_f64, bool fused, int T, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus)
{
    if (x_f64)
        return fused ? launch_owave_T<true, 1, 1, double, double>(T, s, a, pa, num_cus)
                     : launch_owave_T<false, 1, 1, double, double>(T, s, a, pa, num_cus);
    return fused ? launch_owave_T<true, 1, 1, float, double>(T, s, a, pa, num_cus)
                 : launch_owave_T<false, 1, 1, float, double>(T, s, a, pa, num_cus);
}

}  // namespace mrhip
