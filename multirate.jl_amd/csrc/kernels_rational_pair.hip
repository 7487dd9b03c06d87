// kernels_rational_pair.hip -- the headline kernel: FIRRational with M > L (sample-rate reduction by a
// ratio close to one, e.g. 147//160), Float32 taps and samples, tapsPerPhi <= 32.
//
// Mapping ("pair of positions per lane, phase-stationary").  The phase of output k is (u0 + k*M) mod L,
// so outputs k and k + c*L share a tap column.  A workgroup covers one period of c*M input positions
// (c*L outputs) per step; lane l owns the two adjacent positions 2l and 2l+1 of that period, at every
// step j of every tile it is handed.  Each position produces zero or one output (M > L), always with
// the same phase, so the lane keeps TWO tap columns in VGPRs for its whole life and runs two
// independent accumulation chains (ILP hides the dependent-add latency of a single dot product).
// The two windows overlap in T-1 samples: one aligned run of T+1 (rounded to T+2) samples, fetched
// with (T+2)/2 ds_read_b64, feeds both outputs -- half the LDS read traffic of one-output-per-lane,
// and because every lane's run starts on an even sample index the reads are 8-byte aligned and the 32
// lanes of a half-wave cover 64 consecutive banks: conflict-free at 256 B/clk/CU with a single copy
// of the data in LDS.
//
// Staging.  A tile is J steps: J*c*M + T + 1 input samples of one channel, brought HBM -> LDS by
// LDS-DMA (global_load_lds_dwordx4, 16 B per lane, no VGPRs; the source only needs 4-byte alignment,
// scripts/ubench/dma_test.hip), double buffered: the DMA for tile i+1 is issued right after the one
// barrier that opens tile i and lands while tile i is computed.  The first/last tile of a channel
// (history seam, end of input) is staged through a checked register path into the same buffer.
//
// Arithmetic: exactly the generic kernel's (STRICT: separately rounded multiply and add, oldest sample
// first, first product initialises the accumulator; FUSED: explicit fma) => bit-identical results.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "mrhip_internal.h"

#pragma clang fp contract(off)

namespace mrhip {
namespace {

constexpr int kPairMaxThreads = 512;

inline bool pair_debug_once()
{
    static int state = -1;
    if (state < 0) { const char *v = std::getenv("MRHIP_DEBUG"); state = (v && v[0] == '1') ? 1 : 0; }
    if (state == 1) { state = 0; return true; }
    return false;
}
inline int pair_env_int(const char *name, int dflt)
{
    const char *v = std::getenv(name);
    return v && *v ? std::atoi(v) : dflt;
}

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
template <int N, typename V>
__device__ __forceinline__ void lgkm_wait(V &reg)
{
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(reg) : "n"(N < 15 ? N : 15));
}

__device__ __forceinline__ void dma16(const void *gsrc, void *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction only takes an immediate)
__device__ __forceinline__ void wait_vmcnt_le(int n)
{
    switch (n) {
#define MRHIP_W(K) case K: asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory"); break;
        MRHIP_W(0) MRHIP_W(1) MRHIP_W(2) MRHIP_W(3) MRHIP_W(4) MRHIP_W(5) MRHIP_W(6) MRHIP_W(7) MRHIP_W(8) MRHIP_W(9)
        MRHIP_W(10) MRHIP_W(11) MRHIP_W(12) MRHIP_W(13) MRHIP_W(14) MRHIP_W(15) MRHIP_W(16) MRHIP_W(17) MRHIP_W(18) MRHIP_W(19)
        MRHIP_W(20) MRHIP_W(21) MRHIP_W(22) MRHIP_W(23) MRHIP_W(24) MRHIP_W(25) MRHIP_W(26) MRHIP_W(27) MRHIP_W(28) MRHIP_W(29)
        MRHIP_W(30) MRHIP_W(31) MRHIP_W(32)
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

#ifdef MRHIP_PAIR_WPE   /* optional VGPR cap: waves per SIMD the register allocator must leave room for */
#define MRHIP_PAIR_BOUNDS __launch_bounds__(kPairMaxThreads + 64, MRHIP_PAIR_WPE)
#else
#define MRHIP_PAIR_BOUNDS __launch_bounds__(kPairMaxThreads + 64)
#endif
template <int T, bool FUSED>
__global__ MRHIP_PAIR_BOUNDS void rational_pair_kernel(PolyArgs a, PairArgs pa)
{
    constexpr int NPR = (T + 2) / 2;           // aligned 8-byte reads per lane per step: T+1 samples, rounded up
#ifdef MRHIP_PAIR_TWO_BATCH
    constexpr int NA = NPR > 8 ? (NPR + 1) / 2 : NPR;   // first read batch
#else
    constexpr int NA = NPR;                               // all reads up front
#endif
    constexpr int R1 = NA > 2 ? NA - 2 : NA;             // second batch is issued when pair R1 is about to be consumed

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int ncw = (blockDim.x >> 6) - 1;      // compute waves; the last wave of the workgroup is the loader

    // ---- tile walk shared by both roles: tile -> (channel, tile-in-channel) without per-tile divisions
    long long tile = blockIdx.x;
    int ch = static_cast<int>(tile / pa.tiles_per_channel);
    long long tau = tile - static_cast<long long>(ch) * pa.tiles_per_channel;

    if (wave == ncw) {
        // ================= loader wave: HBM -> LDS, one tile ahead of the compute waves =================
        // It is the only wave that waits on vmcnt, so the compute waves' output stores stay in flight
        // across tiles (their barrier carries no memory wait).
        const int nchunks = pa.tile_len / 4;
        const int nslots = (nchunks + 63) >> 6;          // 1 KiB LDS slots per stage
        // Stages one tile; returns the number of LDS-DMA operations it left in flight (0 for the
        // checked register path, which drains everything before returning).
        auto stage_tile = [&](int sch, long long stau, int stage) -> int {
            const float *__restrict__ xc = static_cast<const float *>(a.x) + static_cast<long long>(sch) * a.x_stride;
            const long long o = pa.o0 + stau * pa.tile_in;                   // x index of LDS sample 0 (may be < 0)
            unsigned char *st = smem + static_cast<size_t>(stage) * pa.stage_bytes;
            const bool interior = o >= 0 && o + pa.tile_len <= a.x_len;      // wave-uniform
            if (interior) {
                const unsigned char *src = reinterpret_cast<const unsigned char *>(xc + o);
                for (int slot = 0; slot < nslots; ++slot) {
                    const int ci = slot * 64 + lane;
                    const int cis = ci < nchunks ? ci : 0;                   // padding lanes re-read chunk 0 into LDS padding
                    dma16(src + static_cast<size_t>(cis) * 16, st + static_cast<size_t>(slot) * 1024);
                }
                return nslots;
            }
            // first / last tile of a channel: history seam and end of input, element-wise checked
            const float *__restrict__ hc = static_cast<const float *>(a.hist) + static_cast<long long>(sch) * a.H;
            float *l = reinterpret_cast<float *>(st);
            for (int ci = lane; ci < nchunks; ci += 64) {
                float4 v;
                float *pv = reinterpret_cast<float *>(&v);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const long long gi = o + 4LL * ci + e;
                    float val = 0.f;
                    if (gi >= 0) { if (gi < a.x_len) val = xc[gi]; }
                    else if (gi >= -static_cast<long long>(a.H)) val = hc[a.H + gi];
                    pv[e] = val;
                }
                *reinterpret_cast<float4 *>(l + ci * 4) = v;
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            return 0;
        };
        auto advance = [&](long long &tl, int &c2, long long &ta2) {
            tl += gridDim.x;
            c2 += pa.grid_div;
            ta2 += pa.grid_mod;
            if (ta2 >= pa.tiles_per_channel) { ta2 -= pa.tiles_per_channel; ++c2; }
        };
        // Three LDS stages, the DMA runs two tiles ahead: while the compute waves work on tile i the
        // loader has tile i+1 landing and tile i+2 being issued, so the HBM stream never pauses.
        // `ptile` walks two tiles ahead of `tile`.
        long long ptile = tile;
        int pch = ch;
        long long ptau = tau;
        int in_flight_newest = 0;
        if (ptile < pa.total_tiles) { (void)stage_tile(pch, ptau, 0); advance(ptile, pch, ptau); }
        if (ptile < pa.total_tiles && !(pa.ablate & 1)) { in_flight_newest = stage_tile(pch, ptau, 1); advance(ptile, pch, ptau); }
        wait_vmcnt_le(in_flight_newest);          // tile 0 has landed (only tile 1's operations may remain)
        int pstage = 2;
        for (; tile < pa.total_tiles; tile += gridDim.x) {
            if (!(pa.ablate & 4)) __builtin_amdgcn_s_barrier();   // tile `tile` is published; the stage of tile-1 is free again
            in_flight_newest = 0;
            if (ptile < pa.total_tiles && !(pa.ablate & 1)) {
                in_flight_newest = stage_tile(pch, ptau, pstage);
                advance(ptile, pch, ptau);
                pstage = pstage == 2 ? 0 : pstage + 1;
            }
            wait_vmcnt_le(in_flight_newest);      // everything older than the tile just issued has landed
        }
        return;
    }

    // ================= compute waves =================
    // the lane's two positions -> (output index within a step, phase, active)
    int t_out[2];
    bool act[2];
    float taps[2][T];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const long long p = 2LL * tid + s;                               // position inside the c*M period
        const long long num = p * a.L - a.u0;
        const long long tt = num <= 0 ? 0 : (num + a.M - 1) / a.M;       // first output at or after p
        const long long u = a.u0 + tt * a.M;
        const long long qq = u / a.L;
        act[s] = p < pa.cM && tt < pa.P && qq == p;
        t_out[s] = static_cast<int>(tt);
        const int phi = static_cast<int>(u - qq * a.L);
        const float *__restrict__ tp = static_cast<const float *>(a.taps) + static_cast<long long>(act[s] ? phi : 0) * T;
#pragma unroll
        for (int i = 0; i < T; ++i) taps[s][i] = tp[i];
    }
    // Output staging.  A wave's 128 positions produce one contiguous run of outputs [t_lo, t_hi) of the
    // step (about 128*L/M of them), scattered over its lanes' two accumulators with gaps where a position
    // has no output.  The lanes drop their results into a 512-byte LDS strip at (t - t_lo) and read the
    // strip back two per lane, so every step ends in ONE dense 8-byte-per-lane store instead of an
    // 8-byte store with holes plus a 4-byte store for the odd ones (half the write requests to L2).
    auto first_output_at = [&](long long p) -> long long {   // first output whose position is >= p
        const long long num = p * a.L - a.u0;
        return num <= 0 ? 0 : (num + a.M - 1) / a.M;
    };
    const long long t_lo_ll = first_output_at(128LL * wave);
    const long long t_hi_ll = first_output_at(128LL * (wave + 1));
    const unsigned t_lo = static_cast<unsigned>(t_lo_ll < pa.P ? t_lo_ll : pa.P);
    const unsigned n_w = static_cast<unsigned>((t_hi_ll < pa.P ? t_hi_ll : pa.P)) - t_lo;   // outputs of this wave per step
    float *const strip = reinterpret_cast<float *>(smem + 3 * static_cast<size_t>(pa.stage_bytes) + static_cast<size_t>(wave) * 512);
    const unsigned sidx0 = static_cast<unsigned>(t_out[0]) - t_lo, sidx1 = static_cast<unsigned>(t_out[1]) - t_lo;
    const unsigned my_pair = 2u * static_cast<unsigned>(lane);               // outputs my_pair, my_pair+1 of the strip
    const unsigned lane_win = static_cast<unsigned>(tid) * 8u;           // byte offset of sample 2*tid inside a stage

    int s = 0;
    for (; tile < pa.total_tiles; s = (s == 2 ? 0 : s + 1)) {
        const long long ntile = tile + gridDim.x;
        int nch = ch + pa.grid_div;
        long long ntau = tau + pa.grid_mod;
        if (ntau >= pa.tiles_per_channel) { ntau -= pa.tiles_per_channel; ++nch; }

        // One barrier per tile and no memory wait: the loader wave arrives only after this tile's
        // data has landed; all compute waves arriving proves the other stage is no longer read.
        if (!(pa.ablate & 4)) __builtin_amdgcn_s_barrier();   // (ablate bit 2: timing experiments without the barrier)
        asm volatile("" ::: "memory");

        float *__restrict__ yc = static_cast<float *>(a.y) + static_cast<long long>(ch) * a.y_stride + tau * pa.tile_out;
        const long long remaining = a.n_out - tau * pa.tile_out;          // outputs of this channel from this tile on
        const bool full = remaining >= pa.tile_out;                       // wave-uniform
        const unsigned wbase = lds_base + static_cast<unsigned>(s) * pa.stage_bytes + lane_win;

        auto run_steps = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll 1
            for (int j = 0; j < pa.J; ++j) {
                const unsigned waddr = wbase + static_cast<unsigned>(j) * pa.cM * 4u;
                // The T+2 samples are fetched in two batches (pairs [0,NA) up front, pairs [NA,NPR) once R1
                // pairs have been consumed) so that at most max(NA, NPR-R1) pairs are live at a time.
                v2u_t pr[NPR];
                static_for<0, NA>([&](auto I) { pr[decltype(I)::value] = lds_read_b64<decltype(I)::value * 8>(waddr); });
                float acc0 = 0.f, acc1 = 0.f;
                // sample w[i] = pr[i/2][i%2]; output 0 uses w[i], output 1 uses w[i+1], i = 0..T-1.
                // This file is compiled with -fno-slp-vectorize: hipcc would otherwise SLP-pack the two chains
                // into v_pk_* (no faster per flop) and serialise both outputs into ONE dependent chain.
                static_for<0, NPR>([&](auto I) {
                    constexpr int r = decltype(I)::value;
                    if constexpr (r == R1 && NA < NPR)
                        static_for<NA, NPR>([&](auto Q) { pr[decltype(Q)::value] = lds_read_b64<decltype(Q)::value * 8>(waddr); });
                    constexpr int issued = (r >= R1) ? NPR : NA;
                    lgkm_wait<issued - 1 - r>(pr[r]);
                    const float wlo = __uint_as_float(pr[r].x), whi = __uint_as_float(pr[r].y);
                    // w[2r] = wlo: output 0 tap 2r, output 1 tap 2r-1 ; w[2r+1] = whi: output 0 tap 2r+1, output 1 tap 2r
                    if constexpr (2 * r < T) { if constexpr (r == 0) acc0 = taps[0][0] * wlo; else acc0 = macf<FUSED>(taps[0][2 * r], wlo, acc0); }
                    if constexpr (2 * r - 1 >= 0 && 2 * r - 1 < T) acc1 = macf<FUSED>(taps[1][2 * r - 1], wlo, acc1);
                    if constexpr (2 * r + 1 < T) acc0 = macf<FUSED>(taps[0][2 * r + 1], whi, acc0);
                    if constexpr (2 * r < T) { if constexpr (r == 0) acc1 = taps[1][0] * whi; else acc1 = macf<FUSED>(taps[1][2 * r], whi, acc1); }
                });
                // byte offsets from the (wave-uniform) tile base stay 32-bit: scalar base + VGPR offset stores
                const unsigned kj = static_cast<unsigned>(j) * static_cast<unsigned>(pa.P) + t_lo;
                char *const ybytes = reinterpret_cast<char *>(yc);
                if (pa.ablate & 2) {   // timing experiments only: keep the arithmetic live, drop the stores
                    if (acc0 == 1.2345e30f || acc1 == 1.2345e30f) *reinterpret_cast<float *>(ybytes + (kj + my_pair) * 4u) = acc0 + acc1;
                } else {
                    if (act[0]) strip[sidx0] = acc0;
                    if (act[1]) strip[sidx1] = acc1;
                    const float2 v = reinterpret_cast<const float2 *>(strip)[lane];   // same wave: LDS ops are in order
                    const unsigned lim = FULL ? n_w
                                              : static_cast<unsigned>(remaining - kj > static_cast<long long>(n_w) ? n_w
                                                                      : (remaining > static_cast<long long>(kj) ? remaining - kj : 0));
                    if (my_pair + 1 < lim) {
                        __builtin_memcpy(ybytes + (kj + my_pair) * 4u, &v, 8);        // 4-byte aligned 8-byte store
                    } else if (my_pair < lim) {
                        *reinterpret_cast<float *>(ybytes + (kj + my_pair) * 4u) = v.x;
                    }
                }
            }
        };
        if (full) run_steps(std::true_type{});
        else run_steps(std::false_type{});

        tile = ntile; ch = nch; tau = ntau;
    }
}

template <bool FUSED>
hipError_t launch_pair_T(int T, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, PairArgs pa, int num_cus,
                         int blocks_per_cu_override)
{
#define MRHIP_CASE(TT)                                                                              \
    case TT: {                                                                                      \
        auto kfn = rational_pair_kernel<TT, FUSED>;                                                 \
        if (lds > 48 * 1024) {                                                                      \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),                 \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)); \
            if (e != hipSuccess) return e;                                                          \
        }                                                                                           \
        int per_cu = 0;                                                                             \
        hipError_t eo = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, static_cast<int>(block.x), lds); \
        if (eo != hipSuccess) return eo;                                                            \
        if (per_cu < 1) per_cu = 1;                                                                 \
        if (blocks_per_cu_override > 0) per_cu = blocks_per_cu_override;                            \
        long long g = static_cast<long long>(num_cus) * per_cu;                                     \
        if (g > pa.total_tiles) g = pa.total_tiles;                                                 \
        if (g < 1) g = 1;                                                                           \
        pa.grid_div = static_cast<int>(g / pa.tiles_per_channel);                                   \
        pa.grid_mod = static_cast<long long>(g % pa.tiles_per_channel);                             \
        if (pair_debug_once()) {                                                                    \
            hipFuncAttributes fa;                                                                   \
            (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kfn));                   \
            std::fprintf(stderr, "[mrhip] rational_pair T=%d grid=%lld block=%u lds=%zu occ/CU=%d regs=%d c=%d P=%d cM=%d J=%d " \
                         "tile_len=%d rounds=%d tiles=%lld\n", TT, g, block.x, lds, per_cu, fa.numRegs, pa.c, pa.P, pa.cM, \
                         pa.J, pa.tile_len, pa.dma_rounds, pa.total_tiles);                         \
        }                                                                                           \
        hipLaunchKernelGGL(kfn, dim3(static_cast<unsigned>(g)), block, lds, s, a, pa);              \
        return hipGetLastError();                                                                   \
    }
    switch (T) {
#ifdef MRHIP_PS_FAST_BUILD
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

// Covers: Float32 samples and taps (R = Float32), M > L with L/M >= 0.7, tapsPerPhi <= 32, no zero-start
// quirk (i.e. a pfb kernel: FIRRational).  Returns false otherwise (caller tries the next kernel).
bool plan_rational_pair(const TypeKey &tk, const PolyArgs &a, int num_cus, PairArgs *out, dim3 *block, size_t *lds)
{
    static const int enabled = pair_env_int("MRHIP_PAIR", 1);
    if (!enabled) return false;
    if (tk.x_f64 || tk.r_f64 || tk.complex_x) return false;
#ifdef MRHIP_PS_FAST_BUILD
    if (a.T != 24) return false;
#endif
    if (a.T < 1 || a.T > 32) return false;
    if (!(a.M > a.L) || static_cast<double>(a.L) / a.M < 0.70) return false;
    if (a.zero_start_below > 0) return false;
    static const int env_c = pair_env_int("MRHIP_PAIR_C", 0), env_r = pair_env_int("MRHIP_PAIR_ROUNDS", 0);
    // c: lanes = c*M/2 (c*M must be even), <= 512.  Measured on MI355X (147//160, 24 taps, scripts/
    // exp_ps_matrix.sh): small workgroups (3-4 compute waves + the loader) beat larger ones that fill
    // their last wave better -- more workgroups per CU smooth out the per-tile barrier -- so take the
    // smallest c that gives at least 3 compute waves, falling back to the best lane utilisation.
    int best_c = 0;
    double best = -1.0;
    for (int c = 1; static_cast<long long>(c) * a.M / 2 <= kPairMaxThreads; ++c) {
        if ((static_cast<long long>(c) * a.M) % 2) continue;
        const int lanes = static_cast<int>(static_cast<long long>(c) * a.M / 2);
        const int padded = (lanes + 63) / 64 * 64;
        if (padded >= 192 && static_cast<double>(lanes) / padded >= 0.75) { best_c = c; break; }
        const double score = static_cast<double>(lanes) / padded * (padded < 192 ? 0.5 + 0.5 * padded / 192.0 : 1.0);
        if (score > best + 1e-9) { best = score; best_c = c; }
    }
    if (env_c > 0 && (static_cast<long long>(env_c) * a.M) % 2 == 0 && static_cast<long long>(env_c) * a.M / 2 <= kPairMaxThreads)
        best_c = env_c;
    if (!best_c) return false;
    const int c = best_c;
    const long long cM = static_cast<long long>(c) * a.M;
    const int lanes = static_cast<int>(cM / 2);
    const int padded = (lanes + 63) / 64 * 64;
    const int nwaves = padded / 64;
    // tile: J steps; the stage is a whole number of 1 KiB DMA slots.  MRHIP_PAIR_ROUNDS (experiments) scales it.
    const int stage_kib = env_r > 0 ? env_r * nwaves : 4 * nwaves;
    long long J = (static_cast<long long>(stage_kib) * 256 - a.T - 2) / cM;   // 256 samples per KiB
    if (J < 1) {
        J = 1;
    }
    if (J > 64) J = 64;
    // small problems (few channels, short calls): shrink the tile until there are enough tiles to give
    // every CU a few workgroups -- a launch that occupies a third of the chip is latency-bound
    if (env_r <= 0) {
        const long long want_tiles = 4LL * num_cus;
        while (J > 2 && ((a.n_out + J * c * a.L - 1) / (J * c * a.L)) * a.nch < want_tiles) J = (J + 1) / 2;
    }
    long long tile_len = J * cM + a.T + 2;
    tile_len = (tile_len + 3) / 4 * 4;
    const long long nslots = (tile_len / 4 + 63) / 64;
    const size_t stage_bytes = static_cast<size_t>(nslots) * 1024;
    if (nslots > 32 || 3 * stage_bytes + static_cast<size_t>(nwaves) * 512 > 150 * 1024) return false;
    const long long need_rounds = nslots;
    PairArgs pa{};
    pa.c = c; pa.P = static_cast<int>(static_cast<long long>(c) * a.L); pa.cM = static_cast<int>(cM);
    pa.J = static_cast<int>(J);
    pa.tile_len = static_cast<int>(tile_len);
    pa.dma_rounds = static_cast<int>(need_rounds);
    pa.stage_bytes = static_cast<int>(stage_bytes);
    pa.o0 = a.d0 - a.T;
    static const int env_ablate = pair_env_int("MRHIP_PS_ABLATE", 0);   // timing experiments: 1 = no staging, 2 = no stores
    pa.ablate = env_ablate;
    pa.tile_in = J * cM;
    pa.tile_out = J * pa.P;
    pa.tiles_per_channel = (a.n_out + pa.tile_out - 1) / pa.tile_out;
    pa.total_tiles = pa.tiles_per_channel * a.nch;
    *out = pa;
    *block = dim3(static_cast<unsigned>(padded + 64));   // + the loader wave
    *lds = 3 * stage_bytes + static_cast<size_t>(nwaves) * 512;   // three pipeline stages + one output strip per compute wave
    return true;
}

hipError_t launch_rational_pair(bool fused, const PolyArgs &a, const PairArgs &pa, dim3 block, size_t lds, hipStream_t s,
                                const char **kname, int num_cus)
{
    *kname = "rational_pair_kernel";
    static const int bpc = pair_env_int("MRHIP_PAIR_BPC", 0);
    return fused ? launch_pair_T<true>(a.T, block, lds, s, a, pa, num_cus, bpc)
                 : launch_pair_T<false>(a.T, block, lds, s, a, pa, num_cus, bpc);
}

}  // namespace mrhip
