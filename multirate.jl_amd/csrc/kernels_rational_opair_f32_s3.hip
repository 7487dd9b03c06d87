// kernels_rational_opair_f32_s3.hip -- instantiations of opair_kernel.inc: Float32 arithmetic (Float32 and ComplexF32 samples), SMIN = 3
// (3 <= M/L < 4), tapsPerPhi = 1..32, STRICT and FUSED.
#include "opair_kernel.inc"

namespace mrhip {

hipError_t launch_opair_f32_s3(int nc, bool fused, int T, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus)
{
    if (nc == 2)
        return fused ? launch_opair_T<true, 2, 3, float, float>(T, block, lds, s, a, pa, num_cus)
                     : launch_opair_T<false, 2, 3, float, float>(T, block, lds, s, a, pa, num_cus);
    return fused ? launch_opair_T<true, 1, 3, float, float>(T, block, lds, s, a, pa, num_cus)
                 : launch_opair_T<false, 1, 3, float, float>(T, block, lds, s, a, pa, num_cus);
}

}  // namespace mrhip
