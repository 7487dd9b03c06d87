// Micro-benchmark: VALU issue rate on gfx950 for the instruction mixes the FIR kernels use.
// Each lane runs K independent accumulator chains; blocks of 256 threads, grid sized to put W waves
// on every SIMD.  Reports wave-instructions per cycle per SIMD (clock from hipDeviceProp / wall).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#pragma clang fp contract(off)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float *out, int iters, float a, float b)
{
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 0.001f + i;
    v2f p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = {acc[2 * i], acc[2 * i + 1]};
    v2f av = {a, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if constexpr (MODE == 0) {          // fma chains
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
            } else if constexpr (MODE == 1) {   // mul + add (strict)
#pragma unroll
                for (int i = 0; i < 8; ++i) { float t = acc[i] * a; acc[i] = t + b; }
            } else if constexpr (MODE == 2) {   // packed fma
#pragma unroll
                for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(p[i], av, av);
            } else if constexpr (MODE == 3) {   // packed mul + packed add
#pragma unroll
                for (int i = 0; i < 4; ++i) { v2f t = p[i] * av; p[i] = t + av; }
            } else if constexpr (MODE == 4) {   // single dependent chain mul,add (our dot product shape)
                float t0 = acc[1] * a; acc[0] = acc[0] + t0;
                float t1 = acc[2] * a; acc[0] = acc[0] + t1;
                float t2 = acc[3] * b; acc[0] = acc[0] + t2;
                float t3 = acc[4] * b; acc[0] = acc[0] + t3;
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int waves_per_simd, double ops_per_iter, int ncu, float *d)
{
    const int iters = 4000;
    dim3 block(256), grid(ncu * waves_per_simd);   // 256 threads = 4 waves = 1 per SIMD per block
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, iters, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = double(iters) * ops_per_iter * waves_per_simd;   // wave-instructions per SIMD
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.3f wave-instr/us/SIMD  (at 2.4 GHz: %.2f cycles/instr)\n", name,
           waves_per_simd, ms, winstr / (ms * 1e3), (ms * 1e3 * 2400.0) / winstr);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    float *d; hipMalloc(&d, 256 * 4 * 8 * 256 * 16);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32 x8 chains", w, 64, p.multiProcessorCount, d);
        run<1>("v_mul+v_add x8 chains", w, 128, p.multiProcessorCount, d);
        run<2>("v_pk_fma_f32 x4 chains", w, 32, p.multiProcessorCount, d);
        run<3>("v_pk_mul+v_pk_add x4", w, 64, p.multiProcessorCount, d);
        run<4>("dot-shape mul,add 1 chain", w, 64, p.multiProcessorCount, d);
    }
    return 0;
}
