// pair_device.h -- device-side helpers shared by the loader-wave kernels (opair_kernel.inc, kernels_fir_stream.hip): hand-issued LDS reads/writes with counted waits, LDS-DMA, compile-time bookkeeping
// of the register-ring pipeline.  gfx950 only.
#ifndef MRHIP_PAIR_DEVICE_H
#define MRHIP_PAIR_DEVICE_H

#include <hip/hip_runtime.h>
#include <type_traits>

#pragma clang fp contract(off)

namespace mrhip {
namespace dev {

typedef unsigned v2u_t __attribute__((ext_vector_type(2)));

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ds_read_b64 as inline asm (hipcc would fuse neighbours into the half-rate ds_read2_b64); the wait
// is issued by hand: LDS operations of a wave return in order, so with reads r_0..r_{n-1} issued back
// to back, waiting for lgkmcnt <= n-1-i guarantees r_i has landed.  "+v" pins uses after the wait.
template <int OFF>
__device__ __forceinline__ v2u_t lds_read_b64(unsigned byte_addr)
{
    v2u_t v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return v;
}
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
// two aligned pairs (pair P0 and P0+1 of the window at byte_addr) in one LDS instruction
template <int P0>
__device__ __forceinline__ v4u_t lds_read2_b64(unsigned byte_addr)
{
    v4u_t v;
    asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(byte_addr), "n"(P0), "n"(P0 + 1));
    return v;
}
template <int OFF>
__device__ __forceinline__ v4u_t lds_read_b128(unsigned byte_addr)   // byte_addr + OFF must be 16-byte aligned
{
    v4u_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return v;
}
template <int N, typename V>
__device__ __forceinline__ void lgkm_wait(V &reg)
{
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(reg) : "n"(N < 15 ? N : 15));
}

template <int N, typename V>
__device__ __forceinline__ void lgkm_wait2(V &reg_a, V &reg_b)   // one wait that releases two ring slots
{
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(reg_a), "+v"(reg_b) : "n"(N < 15 ? N : 15));
}

__device__ __forceinline__ unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }
template <typename V>
__device__ __forceinline__ void pin(V &reg)   // orders every later use of reg after the preceding volatile asm
{
    asm volatile("" : "+v"(reg));
}
__device__ __forceinline__ void lds_write_b32(unsigned byte_addr, float v)
{
    asm volatile("ds_write_b32 %0, %1" ::"v"(byte_addr), "v"(v));
}
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lds_write_b64(unsigned byte_addr, v2f_t v)   // 8-byte aligned
{
    asm volatile("ds_write_b64 %0, %1" ::"v"(byte_addr), "v"(v));
}

__device__ __forceinline__ void dma16(const void *gsrc, void *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void dma16_sc1(const void *gsrc, void *lds_wave_base)   // the same, bypassing this CU's L1 (aux 16 = sc1 on gfx950)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 16);
}

__device__ __forceinline__ void dma16_nt(const void *gsrc, void *lds_wave_base)    // ... streaming (aux 2 = nt): L2-served as well, no L1 allocation
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 2);
}

// shiftin! by the workgroup that leaves last (mrhip_internal.h: ShiftFold).  Called by EVERY thread of EVERY workgroup, at its very end.
// No fence: the barrier says this workgroup's waves have consumed every sample of `hist` they asked for; the last counter increment
// therefore follows every read of `hist` in the grid, and what the last workgroup writes is read by the next launch only.
template <typename TX, int NC>
__device__ __forceinline__ void shiftin_by_last_workgroup(const ShiftFold &sf, const void *x, const void *hist, long long x_stride, long long x_len, int H, int nch)
{
    if (!sf.hist_new) return;                                  // (kernel argument: uniform)
    __shared__ int s_last;
    if (!sf.done) {
        // into the OTHER history buffer (not a captured call): nobody in this launch reads what is written -- no counting, no waiting: the
        // last workgroup of the grid copies when it gets here
        if (blockIdx.x + 1u != gridDim.x || blockIdx.y != 0u || blockIdx.z != 0u) return;   // (one of them, on a grid of more dimensions too)
        s_last = 1;
    } else {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        const bool last = __hip_atomic_fetch_add(sf.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1u;
        if (last) __hip_atomic_store(sf.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last ? 1 : 0;
    }
    __syncthreads();
    }
    if (!s_last) return;
    const long long total = static_cast<long long>(nch) * H;
    for (long long t = threadIdx.x; t < total; t += blockDim.x) {
        const long long ch = t / H, i = t - ch * H;
        const long long e = i + x_len;                         // index into [hist ; x]
        const TX *src = e < H ? static_cast<const TX *>(hist) + (ch * H + e) * NC : static_cast<const TX *>(x) + (ch * x_stride + (e - H)) * NC;
        TX *dst = static_cast<TX *>(sf.hist_new) + t * NC;
#pragma unroll
        for (int c = 0; c < NC; ++c) dst[c] = src[c];
    }
}

// A 16-byte LDS read, complete when the statement is (read and wait are one asm statement: no register with a read in flight, and the
// compiler's own wait insertion -- which puts a vmcnt(0) in front of every LDS read it sees behind an LDS-DMA operation -- stays out of it)
__device__ __forceinline__ v4u_t lds_read16_now(unsigned byte_addr)
{
    v4u_t v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(byte_addr) : "memory");
    return v;
}

// ---- hand-offs between workgroups inside ONE launch (the resident ring consumer): L2-served loads and write-through stores ----
// (MI355X: a CU's L1 is never refreshed by another CU's stores and the per-XCD L2s are not coherent for plain write-back
//  stores; `sc1` loads bypass the L1, `sc1` stores are written through.  Producer: sc1 stores, s_waitcnt vmcnt(0), sc1 flag store;
//  consumer: sc1 poll of the flag, then sc1 loads of the bytes.)
__device__ __forceinline__ unsigned long long ld_sc1(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_sc1(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float *p) { return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ void st_sc1(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(float *p, float v) { __hip_atomic_store(reinterpret_cast<unsigned *>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// host-visible (pinned host memory, or device memory the host / another agent reads): system scope
__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void vm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Six aligned 16-byte units (96 consecutive bytes), L2-served, each by ONE instruction, AND the wait for them in one statement: the
// compiler never sees a register with a load in flight (it may park registers in scratch wherever it likes -- a parked destination
// would receive the data behind its back and whatever it parked there instead would be overwritten).  The wait is vmcnt(0): it also
// covers LDS-DMA transfers issued before it, which is what the ring's loader wants at that point anyway.
__device__ __forceinline__ void ld96_sc1(const void *p, v4u_t (&d)[6])
{
    asm volatile("global_load_dwordx4 %0, %6, off sc1\n\t"
                 "global_load_dwordx4 %1, %6, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %6, off offset:32 sc1\n\t"
                 "global_load_dwordx4 %3, %6, off offset:48 sc1\n\t"
                 "global_load_dwordx4 %4, %6, off offset:64 sc1\n\t"
                 "global_load_dwordx4 %5, %6, off offset:80 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5])
                 : "v"(p)
                 : "memory");
}
// two aligned 16-byte units from HOST (pinned) memory, system scope, and the wait for them in one statement (see ld96_sc1)
__device__ __forceinline__ void ld32_sys(const void *p0, const void *p1, v4u_t &d0, v4u_t &d1)
{
    asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\t"
                 "global_load_dwordx4 %1, %3, off sc0 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(d0), "=&v"(d1)
                 : "v"(p0), "v"(p1)
                 : "memory");
}
// write-through stores of N bytes of a lane's registers (N = 4, 8, 16, 32): visible beyond this XCD's L2 once vmcnt has counted them.
// AGENT scope (`sc1`): the readers are later kernels and copy engines of this device (what the host reads of a ring lives in pinned memory and
// is stored by st_sys).  Rounds' first form stored at system scope (`sc0 sc1`): 1-3 % slower in the ring (profiles/r05/experiments.md Q).
template <int N>
__device__ __forceinline__ void store_wt(void *dst, const void *regs)
{
    if constexpr (N == 4) {
        unsigned v; __builtin_memcpy(&v, regs, 4);
        asm volatile("global_store_dword %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
    } else if constexpr (N == 8) {
        v2u_t v; __builtin_memcpy(&v, regs, 8);
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
    } else if constexpr (N == 16) {
        v4u_t v; __builtin_memcpy(&v, regs, 16);
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
    } else {
        static_assert(N == 32, "store_wt");
        v4u_t v0, v1; __builtin_memcpy(&v0, regs, 16); __builtin_memcpy(&v1, static_cast<const char *>(regs) + 16, 16);
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1" ::"v"(dst), "v"(v0), "v"(v1) : "memory");
    }
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction only takes an immediate)
__device__ __forceinline__ void wait_vmcnt_le(int n)
{
    switch (n) {
#define MRHIP_W(K) case K: asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory"); break;
        MRHIP_W(0) MRHIP_W(1) MRHIP_W(2) MRHIP_W(3) MRHIP_W(4) MRHIP_W(5) MRHIP_W(6) MRHIP_W(7) MRHIP_W(8) MRHIP_W(9)
        MRHIP_W(10) MRHIP_W(11) MRHIP_W(12) MRHIP_W(13) MRHIP_W(14) MRHIP_W(15) MRHIP_W(16) MRHIP_W(17) MRHIP_W(18) MRHIP_W(19)
        MRHIP_W(20) MRHIP_W(21) MRHIP_W(22) MRHIP_W(23) MRHIP_W(24) MRHIP_W(25) MRHIP_W(26) MRHIP_W(27) MRHIP_W(28) MRHIP_W(29)
        MRHIP_W(30) MRHIP_W(31) MRHIP_W(32) MRHIP_W(33) MRHIP_W(34) MRHIP_W(35) MRHIP_W(36) MRHIP_W(37) MRHIP_W(38) MRHIP_W(39)
        MRHIP_W(40) MRHIP_W(41) MRHIP_W(42) MRHIP_W(43) MRHIP_W(44) MRHIP_W(45) MRHIP_W(46) MRHIP_W(47) MRHIP_W(48) MRHIP_W(49)
        MRHIP_W(50) MRHIP_W(51) MRHIP_W(52) MRHIP_W(53) MRHIP_W(54) MRHIP_W(55) MRHIP_W(56) MRHIP_W(57) MRHIP_W(58) MRHIP_W(59)
        MRHIP_W(60)
#undef MRHIP_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <bool FUSED>
__device__ __forceinline__ float macf(float t, float x, float acc)
{
    if constexpr (FUSED) return __builtin_fmaf(t, x, acc);
    else { const float p = t * x; return acc + p; }
}

// ---- compile-time bookkeeping of the ring pipeline (see run_steps in the kernel) ----
// virtual group p of a step issues a read iff p + K names a pair of this step (< NPR) or of the next one (>= NPRV)
constexpr int ring_nprv(int npr, int k) { return (npr + k - 1) / k * k; }
constexpr bool ring_issues(int p, int npr, int k) { return p + k < npr || p + k >= ring_nprv(npr, k); }
constexpr int ring_reads_upto(int p_last, int npr, int k)      // reads issued at virtual groups 0..p_last
{
    int n = 0;
    for (int p = 0; p <= p_last; ++p) n += ring_issues(p, npr, k) ? 1 : 0;
    return n;
}
// LDS operations a wave issues between the read of pair v and "use v" (steady state; the first step of a tile,
// whose pairs 0..K-1 come from the prologue, gives the same numbers), capped at 14
constexpr int ring_younger(int v, int npr, int k, int strip_ops = 3)
{
    const int nprv = ring_nprv(npr, k);
    int n = 0;
    if (v >= k) {                                   // issued at virtual group v-k of the same step
        for (int p = v - k + 1; p <= v - 1; ++p) n += ring_issues(p, npr, k) ? 1 : 0;
    } else {                                        // issued at virtual group v + nprv - k of the previous step
        for (int p = v + nprv - k + 1; p <= nprv - 1; ++p) n += ring_issues(p, npr, k) ? 1 : 0;
        n += strip_ops;                             // the strip operations between two steps (W W S)
        for (int p = 0; p <= v - 1; ++p) n += ring_issues(p, npr, k) ? 1 : 0;
    }
    return n < 14 ? n : 14;
}


// One wait per TWO ring units: at "use v" (v even) unit v+1 must be back as well.  Its read was issued one group
// later than unit v's, and the read of group v itself has not been issued yet at that point.
constexpr int ring_younger_pair(int v, int npr, int k, int strip_ops = 3)
{
    const int a = ring_younger(v, npr, k, strip_ops);
    int b = ring_younger(v + 1, npr, k, strip_ops) - (ring_issues(v, npr, k) ? 1 : 0);
    if (b < 0) b = 0;
    return a < b ? a : b;
}

template <int OFF>
__device__ __forceinline__ unsigned lds_read_b32(unsigned byte_addr)
{
    unsigned v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return v;
}

}  // namespace dev
}  // namespace mrhip
#endif
