// VALU issue rate for the FIR dot-product dependency shape: K independent chains per lane, each step
// "p = t_i * x_i ; acc = acc + p" (mul independent, add dependent on the previous add and on its mul).
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)
template <int K>
__global__ void k(float *out, const float *in, int iters)
{
    float t[24], x[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) { t[i] = in[i]; x[i] = in[24 + i] + threadIdx.x; }
    float acc[K];
#pragma unroll
    for (int c = 0; c < K; ++c) acc[c] = c;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 24; ++i) {
#pragma unroll
            for (int c = 0; c < K; ++c) { float p = t[i] * x[(i + c) % 24]; acc[c] = acc[c] + p; }
        }
#pragma unroll
        for (int i = 0; i < 24; ++i) asm volatile("" : "+v"(x[i]));   // keep the products from being hoisted
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < K; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int K>
void run(int waves_per_simd, int ncu, float *d, float *in)
{
    const int iters = 2000;
    dim3 block(256), grid(ncu * waves_per_simd);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<K>, grid, block, 0, 0, d, in, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<K>, grid, block, 0, 0, d, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = double(iters) * 48.0 * K * waves_per_simd;
    printf("chains=%d waves/SIMD=%d  %.3f ms  cycles/instr/SIMD @2.4GHz = %.2f   per-wave cycles per MAC step = %.2f\n", K,
           waves_per_simd, ms, ms * 1e3 * 2400.0 / winstr, ms * 1e3 * 2400.0 / (double(iters) * 24.0 * K));
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    float *d, *in; hipMalloc(&d, 256 * 8 * 256 * 4 * 4); hipMalloc(&in, 64 * 4); hipMemset(in, 0, 64 * 4);
    for (int w : {1, 2, 4, 6, 8}) { run<1>(w, p.multiProcessorCount, d, in); run<2>(w, p.multiProcessorCount, d, in); run<4>(w, p.multiProcessorCount, d, in); }
    return 0;
}
