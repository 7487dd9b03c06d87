// How fast does one SIMD of gfx950 issue the packed-Float32 statement of interp_lane_kernel (interp_lane_quad.inc) when NOTHING else is in
// the way -- no global memory, no barriers?  Cycles per v_pk_* instruction for 1..4 waves per SIMD, three variants:
//   full   the generated statement as it stands (scalar tap loads double-buffered, two LDS writes at the end)
//   noload the same with the s_load / s_waitcnt lines taken out (what the arithmetic alone costs)
//   fused  the FUSED statement (v_pk_fma_f32)
// Build:  grep -v "s_load_dwordx4\|s_waitcnt" multirate.jl_amd/csrc/interp_lane_quad.inc > scripts/ubench/interp_lane_quad_noload.inc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -I multirate.jl_amd/csrc -I scripts/ubench scripts/ubench/valu_pk_rate.hip -o scripts/ubench/valu_pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f_t __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) float *cfloat_t;

template <bool FUSED, int T, int L>
__device__ __forceinline__ void quad_full(const v2f_t *w, cfloat_t taps, unsigned patch)
{
#include "interp_lane_quad.inc"
}
template <bool FUSED, int T, int L>
__device__ __forceinline__ void quad_noload(const v2f_t *w, cfloat_t taps, unsigned patch)
{
#include "interp_lane_quad_noload.inc"
}

template <int MODE>
__global__ __launch_bounds__(64, 4) void rate_kernel(const float *taps_g, const v2f_t *x, float *out, long long *cycles, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned char patch[64 * 144];
    v2f_t w[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) w[i] = x[threadIdx.x + 64 * i];
    const cfloat_t taps = (cfloat_t)taps_g;
    const unsigned pa = static_cast<unsigned>(reinterpret_cast<uintptr_t>(patch)) + threadIdx.x * 144u;
    const long long t0 = __builtin_readcyclecounter();
#pragma clang loop unroll(disable)
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) quad_full<false, 32, 4>(w, taps, pa);
        if constexpr (MODE == 1) quad_noload<false, 32, 4>(w, taps, pa);
        if constexpr (MODE == 2) quad_full<true, 32, 4>(w, taps, pa);
        if constexpr (MODE == 3) quad_noload<true, 32, 4>(w, taps, pa);
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + threadIdx.x] = *reinterpret_cast<float *>(patch + threadIdx.x * 144);
}

int main()
{
    const int iters = 2000;
    float *taps, *out; v2f_t *x; long long *cyc;
    hipMalloc(&taps, 4096); hipMemset(taps, 0, 4096);
    hipMalloc(&x, 64 * 32 * 8); hipMemset(x, 0, 64 * 32 * 8);
    const int maxg = 256 * 16;
    hipMalloc(&out, maxg * 64 * 4); hipMalloc(&cyc, maxg * 8);
    const char *names[4] = {"strict full", "strict noload", "fused full", "fused noload"};
    for (int mode = 0; mode < 4; ++mode)
        for (int wps = 1; wps <= 4; ++wps) {
            const int g = 256 * 4 * wps;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) rate_kernel<0><<<g, 64>>>(taps, x, out, cyc, iters);
                if (mode == 1) rate_kernel<1><<<g, 64>>>(taps, x, out, cyc, iters);
                if (mode == 2) rate_kernel<2><<<g, 64>>>(taps, x, out, cyc, iters);
                if (mode == 3) rate_kernel<3><<<g, 64>>>(taps, x, out, cyc, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> c(g); hipMemcpy(c.data(), cyc, g * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (auto v : c) avg += double(v); avg /= g;
            const double pk = (mode < 2 ? 2.0 : 1.0) * 32 * 4 * iters;       // packed instructions a wave issued
            std::printf("%-14s waves/SIMD=%d  %.3f ms  cycles/wave=%.0f (counter ticks)  ticks per pk instr and SIMD=%.3f  pk instr/s per SIMD=%.3e (x4 cycles = %.2f GHz equivalent)\n", names[mode], wps, ms, avg,
                        avg / (pk * wps), pk * wps / (ms * 1e-3), pk * wps / (ms * 1e-3) * 4 / 1e9);
        }
    return 0;
}
