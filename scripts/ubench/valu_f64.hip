// Micro-benchmark: Float64 VALU issue rate on gfx950 (v_fma_f64, v_mul_f64 + v_add_f64, v_cvt_f64_f32) by waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)
template <int MODE>
__global__ void k(double *out, int iters, double a, double b)
{
    double acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 0.001 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if constexpr (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_fma(acc[i], a, b);
            } else if constexpr (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { double t = acc[i] * a; acc[i] = t + b; }
            } else if constexpr (MODE == 2) {   // add only
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = acc[i] + b;
            } else if constexpr (MODE == 3) {   // mul only
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = acc[i] * a;
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, int waves_per_simd, double ops_per_iter, int ncu, double *d)
{
    const int iters = 2000;
    dim3 block(256), grid(ncu * waves_per_simd);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, iters, 1.0000001, 0.5);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, iters, 1.0000001, 0.5);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double winstr = double(iters) * ops_per_iter * waves_per_simd;
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.3f wave-instr/us/SIMD  (at 2.4 GHz: %.2f cycles/instr)\n", name,
           waves_per_simd, ms, winstr / (ms * 1e3), (ms * 1e3 * 2400.0) / winstr);
}
int main()
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    printf("%s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    double *d; (void)hipMalloc(&d, 256 * 4 * 8 * 256 * 16);
    for (int w : {1, 2, 3, 4, 8}) {
        run<0>("v_fma_f64 x8 chains", w, 64, p.multiProcessorCount, d);
        run<1>("v_mul_f64+v_add_f64 x8", w, 128, p.multiProcessorCount, d);
        run<2>("v_add_f64 x8", w, 64, p.multiProcessorCount, d);
        run<3>("v_mul_f64 x8", w, 64, p.multiProcessorCount, d);
    }
    return 0;
}
