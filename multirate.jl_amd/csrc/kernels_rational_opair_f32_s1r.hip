// kernels_rational_opair_f32_s1r.hip -- instantiations of opair_kernel.inc: Float32 arithmetic, Float32 samples, SMIN = 1
// (M > L, M < 2L), tapsPerPhi = 1..48, STRICT and FUSED.
#include "opair_kernel.inc"

namespace mrhip {

hipError_t launch_opair_f32_s1r(bool fused, int T, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus)
{
    return fused ? launch_opair_T<true, 1, 1, float, float>(T, block, lds, s, a, pa, num_cus)
                 : launch_opair_T<false, 1, 1, float, float>(T, block, lds, s, a, pa, num_cus);
}

}  // namespace mrhip
