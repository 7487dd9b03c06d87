// Checks global_load_lds_dwordx4 semantics on gfx950: LDS destination layout and whether a source
// address that is only 4-byte aligned (the shifted copy B) is honoured.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float *src, float *outA, float *outB, int nwords)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *A = reinterpret_cast<float *>(smem);
    float *B = reinterpret_cast<float *>(smem + 8192 + 128);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
    for (int d = 0; d < 2; ++d) {
        const int ci = (d * nw + wave) * 64 + lane;               // 16-byte chunk index
        const float *ga = src + 4 * ci;
        const float *gb = src + 4 * ci + 1;                        // misaligned by 4 bytes
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)ga, (__attribute__((address_space(3))) void *)(A + 4 * 64 * (d * nw + wave)), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gb, (__attribute__((address_space(3))) void *)(B + 4 * 64 * (d * nw + wave)), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < nwords; i += blockDim.x) { outA[i] = A[i]; outB[i] = B[i]; }
}
int main()
{
    const int nthreads = 256, nwords = 2 * 4 * 64 * 4;   // 2 rounds x 4 waves x 64 lanes x 4 words = 2048
    std::vector<float> h(nwords + 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = float(i);
    float *d, *oa, *ob;
    hipMalloc(&d, h.size() * 4); hipMalloc(&oa, nwords * 4); hipMalloc(&ob, nwords * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(nthreads), 8192 * 2 + 256, 0, d, oa, ob, nwords);
    hipError_t e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    std::vector<float> a(nwords), b(nwords);
    hipMemcpy(a.data(), oa, nwords * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), ob, nwords * 4, hipMemcpyDeviceToHost);
    int badA = 0, badB = 0;
    for (int i = 0; i < nwords; ++i) { badA += a[i] != float(i); badB += b[i] != float(i + 1); }
    printf("copy A mismatches: %d   copy B (src+4B) mismatches: %d\n", badA, badB);
    printf("A[0..7]= %g %g %g %g %g %g %g %g\nB[0..7]= %g %g %g %g %g %g %g %g\n", a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7]);
    return 0;
}
