// kernels_fir_stream_f32.hip -- instantiations of fir_stream_kernel.inc: Float32 arithmetic (Float32 and ComplexF32 samples), M = 1..11, 13, 15, STRICT and FUSED.
#include "fir_stream_kernel.inc"

namespace mrhip {

hipError_t launch_fir_stream_f32(int nc, bool fused, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus)
{
    return nc == 2 ? launch_stream_m<float, float, 2>(fused, block, lds, s, a, pa, num_cus)
                   : launch_stream_m<float, float, 1>(fused, block, lds, s, a, pa, num_cus);
}

}  // namespace mrhip
