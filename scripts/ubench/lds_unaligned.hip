// Micro-benchmark: ds_read_b64 at 4-byte (not 8-byte) aligned addresses on gfx950: does it return the right data, and at what rate?
// Also the output-pair kernel's access pattern for M > L (lane stride 2.18 dwords, even-aligned) against a conflict-free one.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *l = reinterpret_cast<unsigned *>(smem);
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) l[i] = i;
    __syncthreads();
    const unsigned base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
    const unsigned lane = threadIdx.x % 64;
    unsigned acc = 0, first = 0;
    unsigned ad;
    if constexpr (MODE == 0) ad = lane * 8u;                                  // aligned, lane stride 8 B
    else if constexpr (MODE == 1) ad = lane * 8u + 4u;                        // 4-byte aligned only
    else if constexpr (MODE == 2) ad = ((lane * 2 * 160u / 147u) & ~1u) * 4u; // opair M > L: even-aligned, stride 2.18 dwords
    else if constexpr (MODE == 3) ad = (lane * 2 * 160u / 147u) * 4u;         // opair M > L without the even rounding (unaligned lanes)
    else ad = ((lane * 2 * 147u / 160u) & ~1u) * 4u;                          // opair L > M
    for (int it = 0; it < iters; ++it) {
        const unsigned a = base + ad + ((it & 7) << 9);
        v2u r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[i]) : "v"(a), "n"(0));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
        if (it == 0) first = r[0].x * 65536u + r[0].y;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc ^= r[i].x ^ r[i].y;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = first;
    if (acc == 0x12345678u) out[0] = acc;
}
template <int MODE> void run(const char *name, int ncu, unsigned *d)
{
    const int iters = 20000, bpc = 4;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(ncu * bpc), dim3(256), 40960, 0, d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(ncu * bpc), dim3(256), 40960, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> h(64);
    (void)hipMemcpy(h.data(), d, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (unsigned lane = 0; lane < 64; ++lane) {
        unsigned w;
        if (MODE == 0) w = lane * 2; else if (MODE == 1) w = lane * 2 + 1; else if (MODE == 2) w = (lane * 2 * 160u / 147u) & ~1u;
        else if (MODE == 3) w = lane * 2 * 160u / 147u; else w = (lane * 2 * 147u / 160u) & ~1u;
        if (h[lane] != w * 65536u + w + 1) ++bad;
    }
    printf("%-64s %.3f ms  %.2f cycles per wave-instruction per CU (2.4 GHz)  wrong lanes: %d\n", name, ms,
           ms * 1e-3 * 2.4e9 / (double(iters) * 8 * 4 * bpc), bad);
}
int main()
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    unsigned *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("ds_read_b64 aligned, lane stride 8 B", p.multiProcessorCount, d);
    run<1>("ds_read_b64 at 8k+4 (4-byte aligned), lane stride 8 B", p.multiProcessorCount, d);
    run<2>("ds_read_b64 opair 147//160 pattern (even-aligned, 2.18 dw/lane)", p.multiProcessorCount, d);
    run<3>("ds_read_b64 opair 147//160 pattern without even rounding", p.multiProcessorCount, d);
    run<4>("ds_read_b64 opair 160//147 pattern (even-aligned, 1.84 dw/lane)", p.multiProcessorCount, d);
    return 0;
}
