// Micro-benchmark: does the VGPR bank (register index mod 4) of the two sources of v_mul_f32 / v_add_f32
// change the issue rate on gfx950?  Hand-placed registers via inline asm; 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define I8(OP, D, A, B) \
    OP " v" #D ", v" #A ", v" #B "\n"
// 16 independent instructions; sources v[40+..] and v[60+..]; DELTA shifts the second source's bank
#define BLOCK16(OP, B0, B1, B2, B3) \
    OP " v20, v40, v" #B0 "\n" OP " v21, v41, v" #B1 "\n" OP " v22, v42, v" #B2 "\n" OP " v23, v43, v" #B3 "\n" \
    OP " v24, v40, v" #B0 "\n" OP " v25, v41, v" #B1 "\n" OP " v26, v42, v" #B2 "\n" OP " v27, v43, v" #B3 "\n" \
    OP " v28, v40, v" #B0 "\n" OP " v29, v41, v" #B1 "\n" OP " v30, v42, v" #B2 "\n" OP " v31, v43, v" #B3 "\n" \
    OP " v32, v40, v" #B0 "\n" OP " v33, v41, v" #B1 "\n" OP " v34, v42, v" #B2 "\n" OP " v35, v43, v" #B3 "\n"
#define CLOB "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v40","v41","v42","v43","v60","v61","v62","v63","v64","v65","v66","v67"
template <int MODE>
__global__ void k(float *out, int iters)
{
    asm volatile("v_mov_b32 v40, 1.0\nv_mov_b32 v41, 1.0\nv_mov_b32 v42, 1.0\nv_mov_b32 v43, 1.0\n"
                 "v_mov_b32 v60, 1.0\nv_mov_b32 v61, 1.0\nv_mov_b32 v62, 1.0\nv_mov_b32 v63, 1.0\n"
                 "v_mov_b32 v64, 1.0\nv_mov_b32 v65, 1.0\nv_mov_b32 v66, 1.0\nv_mov_b32 v67, 1.0\n" ::: CLOB);
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) asm volatile(BLOCK16("v_mul_f32", 60, 61, 62, 63) BLOCK16("v_mul_f32", 60, 61, 62, 63) BLOCK16("v_mul_f32", 60, 61, 62, 63) BLOCK16("v_mul_f32", 60, 61, 62, 63) ::: CLOB);   // same bank (delta 0 mod 4)
        if constexpr (MODE == 1) asm volatile(BLOCK16("v_mul_f32", 61, 62, 63, 64) BLOCK16("v_mul_f32", 61, 62, 63, 64) BLOCK16("v_mul_f32", 61, 62, 63, 64) BLOCK16("v_mul_f32", 61, 62, 63, 64) ::: CLOB);   // delta 1
        if constexpr (MODE == 2) asm volatile(BLOCK16("v_mul_f32", 62, 63, 64, 65) BLOCK16("v_mul_f32", 62, 63, 64, 65) BLOCK16("v_mul_f32", 62, 63, 64, 65) BLOCK16("v_mul_f32", 62, 63, 64, 65) ::: CLOB);   // delta 2
        if constexpr (MODE == 3) asm volatile(BLOCK16("v_mul_f32", 63, 64, 65, 66) BLOCK16("v_mul_f32", 63, 64, 65, 66) BLOCK16("v_mul_f32", 63, 64, 65, 66) BLOCK16("v_mul_f32", 63, 64, 65, 66) ::: CLOB);   // delta 3
        if constexpr (MODE == 4) asm volatile(BLOCK16("v_add_f32", 60, 61, 62, 63) BLOCK16("v_add_f32", 60, 61, 62, 63) BLOCK16("v_add_f32", 60, 61, 62, 63) BLOCK16("v_add_f32", 60, 61, 62, 63) ::: CLOB);
        if constexpr (MODE == 5) asm volatile(BLOCK16("v_add_f32", 61, 62, 63, 64) BLOCK16("v_add_f32", 61, 62, 63, 64) BLOCK16("v_add_f32", 61, 62, 63, 64) BLOCK16("v_add_f32", 61, 62, 63, 64) ::: CLOB);
        // same register twice (x*x): one fetch
        if constexpr (MODE == 6) asm volatile(BLOCK16("v_mul_f32", 40, 41, 42, 43) BLOCK16("v_mul_f32", 40, 41, 42, 43) BLOCK16("v_mul_f32", 40, 41, 42, 43) BLOCK16("v_mul_f32", 40, 41, 42, 43) ::: CLOB);
        // dependent pairs as in the FIR loop: mul then add on its result (two chains)
        if constexpr (MODE == 7) asm volatile(
            "v_mul_f32 v20, v40, v61\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v41, v62\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v42, v63\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v43, v64\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v40, v61\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v41, v62\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v42, v63\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v43, v64\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v40, v61\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v41, v62\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v42, v63\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v43, v64\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v40, v61\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v41, v62\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v42, v63\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v43, v64\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v40, v61\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v41, v62\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v42, v63\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v43, v64\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v40, v61\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v41, v62\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v42, v63\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v43, v64\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v40, v61\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v41, v62\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v42, v63\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v43, v64\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v40, v61\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v41, v62\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v42, v63\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v43, v64\nv_add_f32 v25, v25, v21\n" ::: CLOB);
        // same, but all sources bank-conflicting
        if constexpr (MODE == 8) asm volatile(
            "v_mul_f32 v20, v40, v60\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v41, v61\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v42, v62\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v43, v63\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v40, v60\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v41, v61\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v42, v62\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v43, v63\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v40, v60\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v41, v61\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v42, v62\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v43, v63\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v40, v60\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v41, v61\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v42, v62\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v43, v63\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v40, v60\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v41, v61\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v42, v62\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v43, v63\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v40, v60\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v41, v61\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v42, v62\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v43, v63\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v40, v60\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v41, v61\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v42, v62\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v43, v63\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v40, v60\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v41, v61\nv_add_f32 v25, v25, v21\n"
            "v_mul_f32 v20, v42, v62\nv_add_f32 v24, v24, v20\nv_mul_f32 v21, v43, v63\nv_add_f32 v25, v25, v21\n" ::: CLOB);
    }
    float r;
    asm volatile("v_add_f32 %0, v20, v24\n" : "=v"(r) :: CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE> void run(const char *name, int wps, int ncu, float *d)
{
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(ncu * wps), dim3(256), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(ncu * wps), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double winstr = double(iters) * 64 * wps;
    printf("%-44s waves/SIMD=%d  %.3f ms  cycles/instr @2.4GHz = %.2f\n", name, wps, ms, ms * 1e3 * 2400.0 / winstr);
}
int main()
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    float *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
    for (int w : {1, 4}) {
        run<0>("v_mul src banks equal (delta 0)", w, p.multiProcessorCount, d);
        run<1>("v_mul delta 1", w, p.multiProcessorCount, d);
        run<2>("v_mul delta 2", w, p.multiProcessorCount, d);
        run<3>("v_mul delta 3", w, p.multiProcessorCount, d);
        run<4>("v_add delta 0", w, p.multiProcessorCount, d);
        run<5>("v_add delta 1", w, p.multiProcessorCount, d);
        run<6>("v_mul x*x (same register)", w, p.multiProcessorCount, d);
        run<7>("mul->add dependent pairs, no src conflicts", w, p.multiProcessorCount, d);
        run<8>("mul->add dependent pairs, mul srcs conflict", w, p.multiProcessorCount, d);
    }
    return 0;
}
