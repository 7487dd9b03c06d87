// Micro-benchmark: LDS read throughput per CU on gfx950 for the access shapes the FIR kernels use
// (lane-linear, conflict-free): ds_read_b32 / b64 / read2_b64 / b128, 16 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *l = reinterpret_cast<unsigned *>(smem);
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) l[i] = i;
    __syncthreads();
    const unsigned base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned a = base + ((it & 7) << 9);
        if constexpr (MODE == 0) {          // 8 x ds_read_b32, lane stride 4 B
            unsigned r[8];
            const unsigned ad = a + threadIdx.x % 64 * 4u;
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[i]) : "v"(ad), "n"(0));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
#pragma unroll
            for (int i = 0; i < 8; ++i) acc ^= r[i];
        } else if constexpr (MODE == 1) {   // 8 x ds_read_b64, lane stride 8 B
            v2u r[8];
            const unsigned ad = a + threadIdx.x % 64 * 8u;
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[i]) : "v"(ad), "n"(0));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
#pragma unroll
            for (int i = 0; i < 8; ++i) acc ^= r[i].x ^ r[i].y;
        } else if constexpr (MODE == 2) {   // 8 x ds_read2_b64 (two adjacent pairs), lane stride 8 B
            v4u r[8];
            const unsigned ad = a + threadIdx.x % 64 * 8u;
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:1" : "=v"(r[i]) : "v"(ad));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
#pragma unroll
            for (int i = 0; i < 8; ++i) acc ^= r[i].x ^ r[i].w;
        } else if constexpr (MODE == 3) {   // 8 x ds_read_b128, lane stride 16 B
            v4u r[8];
            const unsigned ad = a + threadIdx.x % 64 * 16u;
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[i]) : "v"(ad), "n"(0));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
#pragma unroll
            for (int i = 0; i < 8; ++i) acc ^= r[i].x ^ r[i].w;
        } else if constexpr (MODE == 4) {   // 8 x ds_read_b128, lane stride 8 B (overlapping windows, 8-byte aligned only)
            v4u r[8];
            const unsigned ad = a + threadIdx.x % 64 * 8u;
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[i]) : "v"(ad), "n"(0));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
#pragma unroll
            for (int i = 0; i < 8; ++i) acc ^= r[i].x ^ r[i].w;
        } else if constexpr (MODE == 5) {   // 8 x ds_read_b64, lane stride 16 B (what a 4-positions-per-lane window read in pairs would do)
            v2u r[8];
            const unsigned ad = a + threadIdx.x % 64 * 16u;
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[i]) : "v"(ad), "n"(0));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
#pragma unroll
            for (int i = 0; i < 8; ++i) acc ^= r[i].x ^ r[i].y;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int MODE> void run(const char *name, int bytes_per_lane, int ncu, unsigned *d)
{
    const int iters = 20000, bpc = 4;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(ncu * bpc), dim3(256), 40960, 0, d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(ncu * bpc), dim3(256), 40960, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes_per_cu = double(iters) * 8 * 64 * bytes_per_lane * 4 * bpc;   // 4 waves/block
    printf("%-52s %.3f ms  %.1f GB/s per CU  (%.1f B/clk at 2.4 GHz; %.2f cycles per wave-instruction per CU)\n", name, ms,
           bytes_per_cu / (ms * 1e6), bytes_per_cu / (ms * 1e-3) / 2.4e9, ms * 1e-3 * 2.4e9 / (double(iters) * 8 * 4 * bpc));
}
int main()
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    unsigned *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("ds_read_b32  lane stride 4", 4, p.multiProcessorCount, d);
    run<1>("ds_read_b64  lane stride 8", 8, p.multiProcessorCount, d);
    run<2>("ds_read2_b64 lane stride 8 (adjacent pairs)", 16, p.multiProcessorCount, d);
    run<3>("ds_read_b128 lane stride 16", 16, p.multiProcessorCount, d);
    run<4>("ds_read_b128 lane stride 8 (8-byte aligned)", 16, p.multiProcessorCount, d);
    run<5>("ds_read_b64  lane stride 16", 8, p.multiProcessorCount, d);
    return 0;
}
