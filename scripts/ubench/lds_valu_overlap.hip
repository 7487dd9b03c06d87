// Micro-benchmark: do hand-issued ds_read_b64 streams and Float64 VALU streams of the SAME wave overlap on gfx950?
// Per loop iteration a wave issues NR ds_read_b64 (conflict-free, 8 bytes per lane, results never used) and NV
// v_mul_f64 / v_add_f64 on 8 independent chains, in the arb_pipe_kernel pattern (reads, counted wait, arithmetic).
// Printed: cycles per iteration per SIMD (wall clock x nominal clock), for reads only, arithmetic only, both.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)
typedef unsigned v2u __attribute__((ext_vector_type(2)));

template <int NR, int NV, int WIDE>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a, double b)
{
    extern __shared__ unsigned char smem[];
    double *l = reinterpret_cast<double *>(smem);
    for (int i = threadIdx.x; i < 4096; i += 256) l[i] = i * 0.5;
    __syncthreads();
    unsigned addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem)) + (threadIdx.x & 63) * (WIDE ? 16 : 8);
    asm volatile("" : "+v"(addr));
    double acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 0.001 + i;
    double sink = 0;
    for (int it = 0; it < iters; ++it) {
        v2u r[NR > 0 ? NR : 1];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            if constexpr (WIDE) {
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                v4u t;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(addr), "n"((j * 1024) % 32768));
                r[j] = {t.x, t.w};
            } else {
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[j]) : "v"(addr), "n"((j * 512) % 32768));
            }
        }
#pragma unroll
        for (int v = 0; v < NV / 16; ++v) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { double t = acc[i] * a; acc[i] = t + b; }
        }
        if constexpr (NR > 0) {
            asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
            for (int j = 0; j < NR; ++j) asm volatile("" : "+v"(r[j]));
            if (it == iters - 1) sink += __builtin_bit_cast(double, r[0]);
        }
    }
    double s = sink;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NR, int NV, int WIDE>
double run(int wps, int ncu, double *d, double ghz)
{
    const int iters = 20000;
    dim3 block(256), grid(ncu * wps);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<NR, NV, WIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, 36864);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NR, NV, WIDE>), grid, block, 36864, 0, d, iters, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NR, NV, WIDE>), grid, block, 36864, 0, d, iters, 1.0000001, 1e-9);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * ghz * 1e9 / iters;    // cycles per iteration (all wps waves of a SIMD together)
    printf("  NR=%2d %s  NV=%2d  waves/SIMD=%d : %.3f ms, %.1f cycles per iteration-round at %.2f GHz (%.1f per wave-iteration)\n",
           NR, WIDE ? "b128" : "b64 ", NV, wps, ms, cyc, ghz, cyc / wps);
    return cyc;
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate * 1e-6;
    printf("%s CUs=%d clock=%d kHz (cycles below assume this clock; the f64 stream runs lower under the power cap)\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    double *d; hipMalloc(&d, sizeof(double) * 256 * 16 * 256);
    for (int w : {1, 2, 4}) {
        printf("waves per SIMD = %d\n", w);
        run<12, 0, 0>(w, p.multiProcessorCount, d, ghz);
        run<0, 32, 0>(w, p.multiProcessorCount, d, ghz);
        run<12, 32, 0>(w, p.multiProcessorCount, d, ghz);
        run<24, 64, 0>(w, p.multiProcessorCount, d, ghz);
        run<6, 32, 0>(w, p.multiProcessorCount, d, ghz);
        run<6, 0, 1>(w, p.multiProcessorCount, d, ghz);
        run<6, 32, 1>(w, p.multiProcessorCount, d, ghz);
        run<20, 64, 0>(w, p.multiProcessorCount, d, ghz);
    }
    return 0;
}
