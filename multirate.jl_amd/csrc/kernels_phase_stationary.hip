// kernels_phase_stationary.hip -- tuned gfx950 kernel for the polyphase (pfb) kernels:
// FIRRational, FIRInterpolator (and short FIRStandard / FIRDecimator), tapsPerPhi <= 32.
//
// Idea ("phase-stationary"): the phase of output k is (u0 + k*M) mod L, so outputs k and k + c*L
// use the same tap column.  A workgroup covers P = c*L outputs per step; a lane owns the outputs
// k = tile*P*J + j*P + t, j = 0..J-1, of every tile it is handed, and therefore ONE tap column for
// its whole life: the T taps sit in VGPRs, loaded once per workgroup (persistent grid).
//
// Staging.  Per tile the workgroup needs J*c*M + T input samples of one channel.  Interior tiles are
// brought in by LDS-DMA (global_load_lds_dwordx4: HBM -> LDS without touching VGPRs), double
// buffered: the DMA for tile i+1 is issued right after the barrier that opens tile i and lands while
// tile i is being computed, so there is ONE barrier per tile and no staging registers.  The first and
// last tile of a channel (history seam, end of input) take a checked register path into the same
// buffer.  4-byte samples are stored twice, the second copy shifted by one sample (DMA from
// source + 4 bytes) and placed 32 banks away, so every lane fetches its T-sample window with
// 8-byte-aligned ds_read_b64 (256 B/clk/CU; ds_read_b32 gives 128) whatever the parity of its start.
//
// Lane map.  For M <= L a lane is an output (consecutive lanes start 0 or 1 sample apart).  For M > L
// consecutive outputs skip input positions, 32 outputs would span more than 32 window starts and two
// lanes of a half-wave would hit one LDS bank pair; there a lane is an input POSITION (lanes whose
// position produces no output idle), which keeps every half-wave on 32 consecutive starts:
// conflict-free reads for the price of (M/L - 1) idle lanes.
//
// LDS traffic is T*sizeof(sample) per output (taps cost nothing); HBM traffic is the algorithmic
// sizeof(Tx) + (L/M)*sizeof(Tb) per input sample plus the (T-1)-sample halo per tile.
//
// Arithmetic: identical to the generic kernel (STRICT: separately rounded multiply and add, oldest
// sample first, first product initialises; FUSED: explicit fma), so results are bit-identical
// between the two kernels and the oracle.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "mrhip_internal.h"

#pragma clang fp contract(off)

namespace mrhip {
namespace {

constexpr int kMaxThreads = 512;
constexpr int kMaxDmaRounds = 4;   // LDS-DMA instructions per wave, per copy, per tile

inline bool debug_once()
{
    static int state = -1;   // -1 unknown, 0 off, 1 armed
    if (state < 0) { const char *v = std::getenv("MRHIP_DEBUG"); state = (v && v[0] == '1') ? 1 : 0; }
    if (state == 1) { state = 0; return true; }
    return false;
}

inline int env_int(const char *name, int dflt)
{
    const char *v = std::getenv(name);
    return v && *v ? std::atoi(v) : dflt;
}

template <typename R, bool FUSED>
__device__ __forceinline__ R mac(R t, R x, R acc)
{
    if constexpr (FUSED) {
        if constexpr (sizeof(R) == 4) return __builtin_fmaf(t, x, acc);
        else return __builtin_fma(t, x, acc);
    } else {
        R p = t * x;
        return acc + p;
    }
}

typedef unsigned v2u_t __attribute__((ext_vector_type(2)));

// compile-time loop (inline-asm immediates need template constants)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ds_read_b64 / ds_read_b32 as inline asm: hipcc fuses neighbouring 8-byte LDS loads into
// ds_read2_b64, which moves half the bytes per LDS cycle of ds_read_b64 (128 vs 256 B/clk/CU,
// MI355X_MICROARCH.md LDS table).  The compiler does not see loads inside asm, so the matching
// s_waitcnt is issued by hand through lgkm_wait below.
template <int OFF>
__device__ __forceinline__ v2u_t lds_read_b64(unsigned byte_addr)
{
    v2u_t v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return v;
}
// two consecutive 32-bit words from a 4-byte aligned address (the windows of 4-byte samples start at any sample)
template <int OFF>
__device__ __forceinline__ v2u_t lds_read2_b32(unsigned byte_addr)
{
    static_assert(OFF % 4 == 0 && OFF / 4 + 1 <= 255, "ds_read2_b32 offsets are 8-bit counts of dwords");
    v2u_t v;
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(byte_addr), "n"(OFF / 4), "n"(OFF / 4 + 1));
    return v;
}
template <int OFF>
__device__ __forceinline__ unsigned lds_read_b32(unsigned byte_addr)
{
    unsigned v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return v;
}
// Wait until at most N LDS operations of this wave are outstanding.  LDS operations return in
// order, so after issuing reads r_0..r_{n-1} back to back, lgkm_wait<n-1-i>(r_i) guarantees r_i has
// landed (extra outstanding operations only make the wait more conservative).  The "+v" ties the
// register to the wait so the compiler cannot move a use of it above the wait.
template <int N, typename V>
__device__ __forceinline__ void lgkm_wait(V &reg)
{
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(reg) : "n"(N < 15 ? N : 15));
}

// HBM -> LDS, 16 bytes per lane, no VGPR data: LDS destination = lds_wave_base + lane*16 (the base
// must be wave-uniform), global source is per lane and only needs 4-byte alignment (verified on
// gfx950 by scripts/ubench/dma_test.hip).
//
// OFF: the instruction's immediate offset, added to BOTH addresses: the DMAs a wave issues for one stage share ONE base (the LDS base
// travels in M0: one s_mov instead of one per round) and tell their rounds apart by OFF.
template <int OFF>
__device__ __forceinline__ void dma16(const void *gsrc, void *lds_wave_base)
{
    static_assert(OFF >= 0 && OFF < 4096, "13-bit signed immediate");
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, OFF, 0);
}

// One 16-byte chunk = VPS samples of SW 32-bit words each; checked element by element against the
// history seam (negative indices) and the end of the input.
template <int SW>
__device__ __forceinline__ uint4 load_chunk_checked(const unsigned *__restrict__ xc, const unsigned *__restrict__ hc,
                                                   int H, long long g, long long x_len)
{
    constexpr int VPS = 4 / SW;
    unsigned w[4];
#pragma unroll
    for (int e = 0; e < VPS; ++e) {
        const long long gi = g + e;
        const unsigned *src = nullptr;
        if (gi >= 0) { if (gi < x_len) src = xc + gi * SW; }
        else if (gi >= -static_cast<long long>(H)) src = hc + (static_cast<long long>(H) + gi) * SW;
#pragma unroll
        for (int s = 0; s < SW; ++s) w[e * SW + s] = src ? src[s] : 0u;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

template <int T, typename TX, typename R, int NC, bool FUSED>
__global__ __launch_bounds__(kMaxThreads) void poly_phase_stationary_kernel(PolyArgs a, TileArgs ta)
{
    constexpr int SW = static_cast<int>(sizeof(TX)) * NC / 4;   // 32-bit words per sample: 1, 2 or 4
    constexpr int VPS = 4 / SW;                                  // samples per 16-byte chunk
    // (rounds 3-5 kept a second copy of a 4-byte-sample stage, shifted by one sample, so that every window could be read by aligned 8-byte reads;
    //  it needed a second LDS base per stage -- see dma16 -- and is gone: ds_read2_b32 reads a window from any sample)
    constexpr bool TWO_COPIES = false;
    constexpr bool READ2 = SW == 1;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS byte offset of smem[0] (low 32 bits of the flat address of an LDS object are its LDS offset)
    const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));

    // ---- lane -> output map:  u = u0 + t*M ; phi = u mod L ; q = u div L (window offset, samples)
    int t, q, phi;
    bool active;
    if (ta.by_position) {
        q = threadIdx.x;
        const long long num = static_cast<long long>(q) * a.L - a.u0;
        const long long tt = num <= 0 ? 0 : (num + a.M - 1) / a.M;
        const long long u = a.u0 + tt * a.M;
        const long long qq = u / a.L;
        t = static_cast<int>(tt);
        phi = static_cast<int>(u - qq * a.L);
        active = tt < ta.P && qq == q;
    } else {
        t = threadIdx.x;
        const long long u = a.u0 + static_cast<long long>(t) * a.M;
        const long long qll = u / a.L;
        phi = static_cast<int>(u - qll * a.L);
        q = static_cast<int>(qll);
        active = t < ta.P;
    }
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;

    R taps[T];
    {
        const R *__restrict__ tp = static_cast<const R *>(a.taps) + static_cast<long long>(active ? phi : 0) * T;
#pragma unroll
        for (int i = 0; i < T; ++i) taps[i] = tp[i];
    }

    const int cM = ta.c * a.M;                 // input samples between a lane's consecutive outputs
    const long long tile_in = static_cast<long long>(ta.J) * cM;
    const long long tile_out = static_cast<long long>(ta.J) * ta.P;
    const int nchunks = ta.tile_len / VPS;

    // ---- staging of one tile into LDS stage `s` (asynchronous for interior tiles)
    auto stage_tile = [&](long long tile, int s) {
        const int ch = static_cast<int>(tile / ta.tiles_per_channel);
        const long long tau = tile - static_cast<long long>(ch) * ta.tiles_per_channel;
        const unsigned *__restrict__ xc = static_cast<const unsigned *>(a.x) + static_cast<long long>(ch) * a.x_stride * SW;
        // 0-based x index of the oldest sample any lane of this tile needs; chunks start at its
        // 16-byte-aligned floor
        const long long o = a.d0 - T + tau * tile_in;
        const long long g0 = o & ~static_cast<long long>(VPS - 1);
        unsigned char *stA = smem + static_cast<size_t>(s) * ta.stage_bytes;
        unsigned char *stB = stA + ta.copyB_offset_bytes;
        // wave-uniform: every byte the DMA touches (copy B reads one sample further) is inside x
        const bool interior = ta.x_aligned16 && g0 >= 0 && g0 + ta.tile_len + (TWO_COPIES ? 1 : 0) <= a.x_len;
        if (interior) {
            // this wave's chunks: slots wave * rounds ... + rounds - 1 (1 KiB each), i.e. chunks (wave * rounds * 64 + lane) + 64 d, d = 0 ... rounds - 1
            const int ci0 = wave * ta.dma_rounds * 64 + lane;
            const unsigned char *src = reinterpret_cast<const unsigned char *>(xc + g0 * SW) + static_cast<size_t>(ci0) * 16;
            unsigned char *const stw = stA + static_cast<size_t>(wave) * ta.dma_rounds * 1024;
            static_for<0, kMaxDmaRounds>([&](auto D) {
                constexpr int d = decltype(D)::value;
                if (d < ta.dma_rounds && ci0 + 64 * d < nchunks) dma16<d * 1024>(src, stw);     // (lanes beyond the tile: masked off)
            });
        } else {
            const unsigned *__restrict__ hc = static_cast<const unsigned *>(a.hist) + static_cast<long long>(ch) * a.H * SW;
            unsigned *lA = reinterpret_cast<unsigned *>(stA), *lB = reinterpret_cast<unsigned *>(stB);
            for (int ci = tid; ci < nchunks; ci += blockDim.x) {
                const uint4 v = load_chunk_checked<SW>(xc, hc, a.H, g0 + static_cast<long long>(ci) * VPS, a.x_len);
                *reinterpret_cast<uint4 *>(lA + ci * 4) = v;
                if constexpr (TWO_COPIES) {
                    // B[i] = A[i+1]  =>  B[4ci-1 .. 4ci+2] = v.x .. v.w ; the last word of the copy
                    // (index 4*nchunks-1) belongs to the next chunk and is never read
                    if (ci > 0) lB[ci * 4 - 1] = v.x;
                    *reinterpret_cast<uint2 *>(lB + ci * 4) = make_uint2(v.y, v.z);
                    lB[ci * 4 + 2] = v.w;
                }
            }
        }
    };

    long long tile = blockIdx.x;
    int s = 0;
    if (tile < ta.total_tiles) stage_tile(tile, 0);
    for (; tile < ta.total_tiles; tile += gridDim.x, s ^= 1) {
        const int ch = static_cast<int>(tile / ta.tiles_per_channel);
        const long long tau = tile - static_cast<long long>(ch) * ta.tiles_per_channel;
        R *__restrict__ yc = static_cast<R *>(a.y) + static_cast<long long>(ch) * a.y_stride * NC;
        const long long o = a.d0 - T + tau * tile_in;
        const int sh = static_cast<int>(o - (o & ~static_cast<long long>(VPS - 1)));   // origin offset inside chunk 0

        // One barrier per tile: it (a) publishes this tile's staged data -- every wave waits for ITS OWN DMA / LDS writes first, EXPLICITLY:
        // __syncthreads() alone does not (a workgroup-scope release needs no vmcnt wait for ordinary memory, and the LDS reads below are
        // assembly the compiler's own DMA-to-LDS bookkeeping cannot see).  Rounds 3-5 relied on it: with the signal just uploaded (every
        // line from HBM) the next tile's samples were still on their way when the barrier opened -- wrong outputs on the FIRST call after an
        // upload, found by tests/stress_random.py --seed 61 in round 6 (profiles/r06/experiments.md K) -- and (b) proves that every wave
        // has finished reading the other stage, which the next DMA is about to overwrite.
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (tile + gridDim.x < ta.total_tiles && !(ta.ablate & 1)) stage_tile(tile + gridDim.x, s ^ 1);

        if (active) {
            const unsigned stage_base = lds_base + static_cast<unsigned>(s) * ta.stage_bytes;
            const long long kbase = tau * tile_out + t;
            const long long nbase = a.d0 + q + tau * tile_in;     // 1-based newest-sample index, j = 0
#pragma unroll 1
            for (int j = 0; j < ta.J; ++j) {
                const long long k = kbase + static_cast<long long>(j) * ta.P;
                if (k >= a.n_out) break;
                const int start = sh + q + j * cM;                // LDS sample index of the oldest sample
                R acc[NC];
                const bool zero_start = nbase + static_cast<long long>(j) * cM < a.zero_start_below;   // support.jl:46
                auto step = [&](auto idx, const R (&xv)[NC]) {
                    constexpr int i = decltype(idx)::value;
                    if constexpr (i == 0) {
#pragma unroll
                        for (int c2 = 0; c2 < NC; ++c2) acc[c2] = taps[0] * xv[c2];
                        if (zero_start) {
#pragma unroll
                            for (int c2 = 0; c2 < NC; ++c2) acc[c2] = static_cast<R>(0) + acc[c2];
                        }
                    } else {
#pragma unroll
                        for (int c2 = 0; c2 < NC; ++c2) acc[c2] = mac<R, FUSED>(taps[i], xv[c2], acc[c2]);
                    }
                };
                if constexpr (READ2) {
                    // 4-byte samples: T/2 reads of two consecutive words each, from wherever the window starts
                    constexpr int NP = T / 2;
                    constexpr int NR = NP + (T & 1);
                    const unsigned waddr = stage_base + start * 4;
                    v2u_t pr[NP > 0 ? NP : 1];
                    unsigned last = 0;
                    static_for<0, NP>([&](auto I) { pr[decltype(I)::value] = lds_read2_b32<decltype(I)::value * 8>(waddr); });
                    if constexpr (T & 1) last = lds_read_b32<(T - 1) * 4>(waddr);
                    static_for<0, NP>([&](auto I) {
                        constexpr int i = decltype(I)::value;
                        lgkm_wait<NR - 1 - i>(pr[i]);
                        const R x0[1] = {static_cast<R>(__uint_as_float(pr[i].x))};
                        const R x1[1] = {static_cast<R>(__uint_as_float(pr[i].y))};
                        step(std::integral_constant<int, 2 * i>{}, x0);
                        step(std::integral_constant<int, 2 * i + 1>{}, x1);
                    });
                    if constexpr (T & 1) {
                        lgkm_wait<0>(last);
                        const R xl[1] = {static_cast<R>(__uint_as_float(last))};
                        step(std::integral_constant<int, T - 1>{}, xl);
                    }
                } else if constexpr (SW == 2) {
                    // 8-byte samples (ComplexF32 or Float64): one naturally aligned ds_read_b64 per tap
                    const unsigned waddr = stage_base + start * 8;
                    v2u_t pr[T];
                    static_for<0, T>([&](auto I) { pr[decltype(I)::value] = lds_read_b64<decltype(I)::value * 8>(waddr); });
                    static_for<0, T>([&](auto I) {
                        constexpr int i = decltype(I)::value;
                        lgkm_wait<T - 1 - i>(pr[i]);
                        TX tmp[NC];
                        __builtin_memcpy(tmp, &pr[i], 8);
                        R xv[NC];
#pragma unroll
                        for (int c2 = 0; c2 < NC; ++c2) xv[c2] = static_cast<R>(tmp[c2]);
                        step(I, xv);
                    });
                } else {
                    // 16-byte samples (ComplexF64): ds_read_b128, left to the compiler
                    const uint4 *wp = reinterpret_cast<const uint4 *>(smem + static_cast<size_t>(s) * ta.stage_bytes) + start;
                    static_for<0, T>([&](auto I) {
                        const uint4 v = wp[decltype(I)::value];
                        TX tmp[NC];
                        __builtin_memcpy(tmp, &v, 16);
                        R xv[NC];
#pragma unroll
                        for (int c2 = 0; c2 < NC; ++c2) xv[c2] = static_cast<R>(tmp[c2]);
                        step(I, xv);
                    });
                }
                if (ta.ablate & 2) {          // timing experiments only: keep the math live, drop the store
                    if (acc[0] == static_cast<R>(1.2345e30)) yc[k * NC] = acc[0];
                } else if constexpr (NC == 1) {
                    yc[k] = acc[0];
                } else {
                    using RV = typename std::conditional<sizeof(R) == 4, float2, double2>::type;
                    RV o2;
                    o2.x = acc[0];
                    o2.y = acc[1];
                    reinterpret_cast<RV *>(yc)[k] = o2;
                }
            }
        }
    }
}

template <typename TX, typename R, int NC, bool FUSED>
hipError_t launch_T(int T, dim3 grid, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const TileArgs &ta,
                    int num_cus, int blocks_per_cu_override)
{
#define MRHIP_CASE(TT)                                                                              \
    case TT: {                                                                                      \
        auto kfn = poly_phase_stationary_kernel<TT, TX, R, NC, FUSED>;                              \
        /* persistent grid = what is actually co-resident (registers, LDS and waves all count) */   \
        int per_cu = 0;                                                                             \
        hipError_t eo = occupancy_cached(reinterpret_cast<const void *>(kfn), block.x, lds, &per_cu); \
        if (eo != hipSuccess) return eo;                                                            \
        if (per_cu < 1) per_cu = 1;                                                                 \
        if (blocks_per_cu_override > 0) per_cu = blocks_per_cu_override;                            \
        long long g = static_cast<long long>(num_cus) * per_cu;                                     \
        if (g > ta.total_tiles) g = ta.total_tiles;                                                 \
        grid = dim3(static_cast<unsigned>(g < 1 ? 1 : g));                                          \
        if (debug_once()) {                                                                         \
            hipFuncAttributes fa;                                                                   \
            (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kfn));                   \
            std::fprintf(stderr, "[mrhip] phase_stationary T=%d grid=%u block=%u lds=%zu occ/CU=%d regs=%d c=%d P=%d J=%d " \
                         "bypos=%d tile_len=%d dma_rounds=%d tiles=%lld\n", TT, grid.x, block.x, lds, per_cu, fa.numRegs, \
                         ta.c, ta.P, ta.J, ta.by_position, ta.tile_len, ta.dma_rounds, ta.total_tiles); \
        }                                                                                           \
        launch_kernel(kfn, grid, block, lds, s, a, ta);                                        \
        return hipGetLastError();                                                                   \
    }
    switch (T) {
#ifdef MRHIP_PS_FAST_BUILD   /* developer builds: one tap count only (seconds instead of minutes) */
        MRHIP_CASE(24)
#else
        MRHIP_CASE(1) MRHIP_CASE(2) MRHIP_CASE(3) MRHIP_CASE(4) MRHIP_CASE(5) MRHIP_CASE(6) MRHIP_CASE(7) MRHIP_CASE(8)
        MRHIP_CASE(9) MRHIP_CASE(10) MRHIP_CASE(11) MRHIP_CASE(12) MRHIP_CASE(13) MRHIP_CASE(14) MRHIP_CASE(15) MRHIP_CASE(16)
        MRHIP_CASE(17) MRHIP_CASE(18) MRHIP_CASE(19) MRHIP_CASE(20) MRHIP_CASE(21) MRHIP_CASE(22) MRHIP_CASE(23) MRHIP_CASE(24)
        MRHIP_CASE(25) MRHIP_CASE(26) MRHIP_CASE(27) MRHIP_CASE(28) MRHIP_CASE(29) MRHIP_CASE(30) MRHIP_CASE(31) MRHIP_CASE(32)
#endif
    default: return hipErrorInvalidValue;
    }
#undef MRHIP_CASE
}

}  // namespace

// Host-side planning: pick the lane map, c (lanes per step), J (steps per tile) and the LDS layout.
// Returns false if the configuration is outside what this kernel covers (caller falls back).
bool plan_phase_stationary(const TypeKey &tk, const PolyArgs &a, int num_cus, TileArgs *out, dim3 *grid, dim3 *block, size_t *lds)
{
#ifdef MRHIP_PS_FAST_BUILD
    if (a.T != 24) return false;
#endif
    if (MRHIP_ENV_INT("MRHIP_PS", 1) == 0) return false;   // (tests and A/B runs: take this kernel out of the dispatcher)
    if (a.T < 1 || a.T > 32) return false;
    if (a.L > kMaxThreads) return false;
    const int sw = (tk.x_f64 ? 2 : 1) * (tk.complex_x ? 2 : 1);
    const int vps = 4 / sw;
    // tuning overrides for experiments (unset in production): MRHIP_PS_C / _J / _BYPOS / _BPC
    static const int env_c = env_int("MRHIP_PS_C", 0), env_j = env_int("MRHIP_PS_J", 0),
                     env_bypos = env_int("MRHIP_PS_BYPOS", -1);
    // lane map: by output (M <= L) or by input position (M > L, worthwhile while >= 70 % of lanes work)
    bool by_position = a.M > a.L && static_cast<double>(a.L) / a.M >= 0.70;
    if (env_bypos >= 0) by_position = env_bypos != 0 && a.M > a.L;
    // choose c: lanes per j-step (c*L outputs, or c*M positions) as close as possible to a multiple
    // of 64, at most kMaxThreads
    int best_c = 0;
    double best_score = -1.0;
    for (int c = 1; static_cast<long long>(c) * (by_position ? a.M : a.L) <= kMaxThreads; ++c) {
        const int P = c * a.L;
        const int lanes = by_position ? c * a.M : P;
        const int padded = (lanes + 63) / 64 * 64;
        double score = static_cast<double>(P) / padded;                  // lane utilisation
        if (padded < 192) score *= 0.5 + 0.5 * padded / 192.0;            // too few waves per block
        if (score > best_score + 1e-9) { best_score = score; best_c = c; }
    }
    if (!best_c) return false;
    if (env_c > 0 && static_cast<long long>(env_c) * (by_position ? a.M : a.L) <= kMaxThreads) best_c = env_c;
    const int c = best_c, P = c * a.L;
    const int lanes = by_position ? c * a.M : P;
    const int padded = (lanes + 63) / 64 * 64;
    const int nwaves = padded / 64;
    const long long cM = static_cast<long long>(c) * a.M;

    // J: as many steps as fit kDmaRoundsTarget KiB-slots per wave (each DMA round moves nwaves KiB per copy)
    const int rounds_target = sw == 1 ? 2 : 4;
    long long J = (static_cast<long long>(rounds_target) * nwaves * 64 * vps - a.T - 8) / (cM > 0 ? cM : 1);
    if (J > 16) J = 16;
    if (env_j > 0) J = env_j;
    if (J < 1) J = 1;
    // keep J*c*M a multiple of the chunk size so the offset of the tile origin inside its first
    // chunk (and with it the parity of every lane's window start) is the same for every tile
    while (J > 1 && (J * cM) % vps != 0) --J;
    long long tile_len = J * cM + a.T + 1 + (vps - 1);
    tile_len = (tile_len + vps - 1) / vps * vps;                          // whole 16-byte chunks
    const long long nchunks = tile_len / vps;
    const long long rounds = (nchunks + static_cast<long long>(nwaves) * 64 - 1) / (static_cast<long long>(nwaves) * 64);
    if (rounds > kMaxDmaRounds) return false;
    const size_t copy_bytes = static_cast<size_t>(rounds) * nwaves * 1024;        // multiple of 1 KiB (and of 256 B)
    // Copy B sits 32 banks (128 B) after copy A so that, of 32 consecutive window starts, the 16
    // even ones (copy A) and the 16 odd ones (copy B) together cover all 64 banks exactly once.  That
    // needs the first start of a half-wave to be even; when the tile origin makes it odd, two more
    // words of offset restore the disjoint cover.
    const long long o0 = a.d0 - a.T;
    const int sh0 = static_cast<int>(((o0 % vps) + vps) % vps);
    const bool sh_constant = (J * cM) % vps == 0;
    const size_t copyB_off = copy_bytes + 128 + ((sh_constant && (sh0 & 1)) ? 8 : 0);
    const size_t stage_bytes = copy_bytes;                                        // (one copy: see the kernel's READ2)
    const size_t total = 2 * stage_bytes;
    if (total > 156 * 1024) return false;

    TileArgs ta{};
    ta.c = c; ta.P = P; ta.J = static_cast<int>(J);
    ta.by_position = by_position ? 1 : 0;
    static const int env_ablate = env_int("MRHIP_PS_ABLATE", 0);   // timing experiments: 1 = no staging, 2 = no stores
    ta.ablate = env_ablate;
    ta.tile_len = static_cast<int>(tile_len);
    ta.dma_rounds = static_cast<int>(rounds);
    ta.copyB_offset_bytes = static_cast<int>(copyB_off);
    ta.stage_bytes = static_cast<int>(stage_bytes);
    const long long tile_out = J * P;
    ta.tiles_per_channel = (a.n_out + tile_out - 1) / tile_out;
    ta.total_tiles = ta.tiles_per_channel * a.nch;
    const uintptr_t xb = reinterpret_cast<uintptr_t>(a.x);
    ta.x_aligned16 = (xb % 16 == 0) && ((a.x_stride * sw * 4) % 16 == 0 || a.nch == 1);
    (void)num_cus;
    *out = ta;
    *grid = dim3(1);            // sized at launch from the occupancy of the chosen instantiation
    *block = dim3(static_cast<unsigned>(padded));
    *lds = total;
    return true;
}

hipError_t launch_poly_phase_stationary(const TypeKey &tk, bool fused, const PolyArgs &a, const TileArgs &ta, dim3 grid,
                                        dim3 block, size_t lds, hipStream_t s, const char **kname, int num_cus)
{
    *kname = "poly_phase_stationary_kernel";
    static const int bpc = env_int("MRHIP_PS_BPC", 0);
#define MRHIP_GO(TX, R, NC)                                                                          \
    return fused ? launch_T<TX, R, NC, true>(a.T, grid, block, lds, s, a, ta, num_cus, bpc)          \
                 : launch_T<TX, R, NC, false>(a.T, grid, block, lds, s, a, ta, num_cus, bpc)
    if (!tk.x_f64 && !tk.r_f64) { if (tk.complex_x) { MRHIP_GO(float, float, 2); } else { MRHIP_GO(float, float, 1); } }
    if (!tk.x_f64 && tk.r_f64) { if (tk.complex_x) { MRHIP_GO(float, double, 2); } else { MRHIP_GO(float, double, 1); } }
    if (tk.x_f64 && tk.r_f64) { if (tk.complex_x) { MRHIP_GO(double, double, 2); } else { MRHIP_GO(double, double, 1); } }
#undef MRHIP_GO
    return hipErrorInvalidValue;
}

}  // namespace mrhip
