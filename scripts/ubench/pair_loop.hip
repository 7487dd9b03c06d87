// Isolated inner loop of rational_pair_kernel: per step 13 aligned ds_read_b64 feed two 24-tap dot
// products.  Measures wave-steps per microsecond per CU for several loop shapes and occupancies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
#pragma clang fp contract(off)
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F &&f)
{ if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); } }
template <int OFF> __device__ __forceinline__ v2u_t lds_read_b64(unsigned a)
{ v2u_t v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF)); return v; }
template <int N, typename V> __device__ __forceinline__ void lgkm_wait(V &r)
{ asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r) : "n"(N < 15 ? N : 15)); }


typedef float v2f_t __attribute__((ext_vector_type(2)));
// packed f32: lo = x.half(SEL) * t.lo ; hi = x.half(SEL) * t.hi
template <int SEL> __device__ __forceinline__ v2f_t pk_mul_bcast(v2u_t x, v2f_t t)
{
    v2f_t d;
    if constexpr (SEL == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(x), "v"(t));
    else                    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(x), "v"(t));
    return d;
}
__device__ __forceinline__ v2f_t pk_add(v2f_t a, v2f_t b)
{
    v2f_t d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
template <int SEL> __device__ __forceinline__ v2f_t pk_fma_bcast(v2u_t x, v2f_t t, v2f_t c)
{
    v2f_t d;
    if constexpr (SEL == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(d) : "v"(x), "v"(t), "v"(c));
    else                    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(x), "v"(t), "v"(c));
    return d;
}

constexpr int T = 24, NPR = 13;
#ifndef TAPS_VGPR
#define TAPS_VGPR 0
#endif
// MODE 0: strict mul+add two chains; 1: fused; 2: LDS only; 3: math only (strict); 4: strict, chains kept scalar via asm fence
template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, const float *in, int steps, int cM)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *l = reinterpret_cast<float *>(smem);
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) l[i] = in[i & 1023] + i;
    __syncthreads();
    const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
    float taps[2][T];
#pragma unroll
    for (int i = 0; i < T; ++i) { taps[0][i] = in[i + (TAPS_VGPR ? (threadIdx.x & 63) : 0)]; taps[1][i] = in[32 + i + (TAPS_VGPR ? (threadIdx.x & 63) : 0)]; }
    float sum = 0.f;
    const unsigned wbase = lds_base + threadIdx.x * 8u;
    for (int j = 0; j < steps; ++j) {
        const unsigned waddr = wbase + static_cast<unsigned>(j & 3) * cM * 4u;
        v2u_t pr[NPR];
        float acc0 = 0.f, acc1 = 0.f;
        if constexpr (MODE != 3) {
            static_for<0, NPR>([&](auto I) { pr[decltype(I)::value] = lds_read_b64<decltype(I)::value * 8>(waddr); });
        } else {
            static_for<0, NPR>([&](auto I) { pr[decltype(I)::value] = v2u_t{__float_as_uint(sum + decltype(I)::value), __float_as_uint(sum)}; });
        }

        if constexpr (MODE == 5 || MODE == 6) {
            // skewed packed chains: step k uses sample w[k] for output-0 term k (lo) and output-1 term k-1 (hi)
            v2f_t tp[T + 1];
#pragma unroll
            for (int kk = 0; kk <= T; ++kk) tp[kk] = v2f_t{kk < T ? taps[0][kk] : 0.f, kk >= 1 ? taps[1][kk - 1] : 0.f};
            v2f_t acc;
            static_for<0, NPR>([&](auto I) {
                constexpr int r = decltype(I)::value;
                lgkm_wait<NPR - 1 - r>(pr[r]);
                if constexpr (r == 0) {
                    acc.x = tp[0].x * __uint_as_float(pr[0].x);
                    acc.y = -0.0f;
                    if constexpr (MODE == 5) acc = pk_add(acc, pk_mul_bcast<1>(pr[0], tp[1]));
                    else acc = pk_fma_bcast<1>(pr[0], tp[1], acc);
                } else {
                    if constexpr (2 * r < T) {
                        if constexpr (MODE == 5) acc = pk_add(acc, pk_mul_bcast<0>(pr[r], tp[2 * r]));
                        else acc = pk_fma_bcast<0>(pr[r], tp[2 * r], acc);
                    } else if constexpr (2 * r == T) {
                        float p = tp[T].y * __uint_as_float(pr[r].x); acc.y = acc.y + p;
                    }
                    if constexpr (2 * r + 1 < T) {
                        if constexpr (MODE == 5) acc = pk_add(acc, pk_mul_bcast<1>(pr[r], tp[2 * r + 1]));
                        else acc = pk_fma_bcast<1>(pr[r], tp[2 * r + 1], acc);
                    } else if constexpr (2 * r + 1 == T) {
                        float p = tp[T].y * __uint_as_float(pr[r].y); acc.y = acc.y + p;
                    }
                }
            });
            acc0 = acc.x; acc1 = acc.y;
        } else
        static_for<0, NPR>([&](auto I) {
            constexpr int r = decltype(I)::value;
            if constexpr (MODE != 3) lgkm_wait<NPR - 1 - r>(pr[r]);
            const float wlo = __uint_as_float(pr[r].x), whi = __uint_as_float(pr[r].y);
            if constexpr (MODE == 2) { acc0 += wlo; acc1 += whi; }
            else if constexpr (MODE == 1) {
                if constexpr (2 * r < T) acc0 = __builtin_fmaf(taps[0][2 * r], wlo, acc0);
                if constexpr (2 * r - 1 >= 0 && 2 * r - 1 < T) acc1 = __builtin_fmaf(taps[1][2 * r - 1], wlo, acc1);
                if constexpr (2 * r + 1 < T) acc0 = __builtin_fmaf(taps[0][2 * r + 1], whi, acc0);
                if constexpr (2 * r < T) acc1 = __builtin_fmaf(taps[1][2 * r], whi, acc1);
            } else {
                if constexpr (2 * r < T) { float p = taps[0][2 * r] * wlo; acc0 = acc0 + p; }
                if constexpr (MODE == 4) asm volatile("" : "+v"(acc0));
                if constexpr (2 * r - 1 >= 0 && 2 * r - 1 < T) { float p = taps[1][2 * r - 1] * wlo; acc1 = acc1 + p; }
                if constexpr (MODE == 4) asm volatile("" : "+v"(acc1));
                if constexpr (2 * r + 1 < T) { float p = taps[0][2 * r + 1] * whi; acc0 = acc0 + p; }
                if constexpr (MODE == 4) asm volatile("" : "+v"(acc0));
                if constexpr (2 * r < T) { float p = taps[1][2 * r] * whi; acc1 = acc1 + p; }
                if constexpr (MODE == 4) asm volatile("" : "+v"(acc1));
            }
        });
        sum += acc0 + acc1;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}
template <int MODE> void run(const char *name, int block, int blocks_per_cu, int ncu, float *d, float *in)
{
    const int steps = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 40000);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(ncu * blocks_per_cu), dim3(block), 36000, 0, d, in, steps, 960);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(ncu * blocks_per_cu), dim3(block), 36000, 0, d, in, steps, 960);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int waves = block / 64 * blocks_per_cu;
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k<MODE>));
    printf("%-22s regs=%3d waves/CU=%2d  %.3f ms  wave-steps/us/CU=%.2f  cycles(2.4GHz)/wave-step/CU=%.1f\n", name, fa.numRegs, waves, ms,
           double(steps) * waves / (ms * 1e3), ms * 1e3 * 2400.0 / (double(steps) * waves));
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    float *d, *in; hipMalloc(&d, 256 * 4 * 512 * 4 * 4); hipMalloc(&in, 1024 * 4); hipMemset(in, 0, 1024 * 4);
    for (int bpc : {1, 2, 3, 4}) {
        run<0>("strict (compiler)", 512, bpc, p.multiProcessorCount, d, in);
        run<4>("strict scalar chains", 512, bpc, p.multiProcessorCount, d, in);
        run<1>("fused (compiler)", 512, bpc, p.multiProcessorCount, d, in);
        run<2>("LDS reads only", 512, bpc, p.multiProcessorCount, d, in);
        run<3>("math only strict", 512, bpc, p.multiProcessorCount, d, in);
        run<5>("packed skewed strict", 512, bpc, p.multiProcessorCount, d, in);
        run<6>("packed skewed fused", 512, bpc, p.multiProcessorCount, d, in);
    }
    return 0;
}
