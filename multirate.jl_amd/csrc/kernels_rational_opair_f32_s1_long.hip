// kernels_rational_opair_f32_s1_long.hip -- instantiations of opair_kernel.inc: Float32 arithmetic (Float32 and ComplexF32 samples), SMIN = 1,
// tapsPerPhi = 49..64 (two columns of up to 66 slots in registers: 140-168 VGPRs), STRICT and FUSED.
#define MRHIP_OPAIR_LONG
#include "opair_kernel.inc"

namespace mrhip {

hipError_t launch_opair_f32_s1_long(int nc, bool fused, int T, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus)
{
    if (nc == 2)
        return fused ? launch_opair_T<true, 2, 1, float, float>(T, block, lds, s, a, pa, num_cus)
                     : launch_opair_T<false, 2, 1, float, float>(T, block, lds, s, a, pa, num_cus);
    return fused ? launch_opair_T<true, 1, 1, float, float>(T, block, lds, s, a, pa, num_cus)
                 : launch_opair_T<false, 1, 1, float, float>(T, block, lds, s, a, pa, num_cus);
}

}  // namespace mrhip
