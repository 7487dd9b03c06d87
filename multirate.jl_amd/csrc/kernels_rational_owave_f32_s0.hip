// kernels_rational_owave_f32_s0.hip -- instantiations of owave_kernel.inc: Float32 arithmetic (Float32 and ComplexF32 samples),
// SMIN = 0 (L > M), STRICT and FUSED.
#include "owave_kernel.inc"

namespace mrhip {

hipError_t launch_owave_f32_s0(int nc, bool fused, int T, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus)
{
    if (nc == 2)
        return fused ? launch_owave_T<true, 2, 0, float, float>(T, s, a, pa, num_cus)
                     : launch_owave_T<false, 2, 0, float, float>(T, s, a, pa, num_cus);
    return fused ? launch_owave_T<true, 1, 0, float, float>(T, s, a, pa, num_cus)
                 : launch_owave_T<false, 1, 0, float, float>(T, s, a, pa, num_cus);
}

}  // namespace mrhip
