// kernels_arbitrary.hip -- tuned gfx950 kernel for FIRArbitrary (src/Filters.jl:693-742).
//
// Per output k the host-evaluated phase recurrence (src/Filters.jl:663-673) supplies the input index
// n_k and the accumulator value acc_k; phi = floor(acc), alpha = acc - phi.  The output is
//     y = dot(pfb[:,phi], window) + dot(dpfb[:,phi], window) * alpha        (:724-730)
// with the combine done in Float64 and rounded once to the output type.
//
// A persistent workgroup keeps both polyphase banks in LDS (column pitch T+1 elements, so lanes with
// different phases land on different banks) and, per tile of consecutive outputs of one channel,
// stages the contiguous run of input samples those outputs touch, for CPL channels at once.  One lane = one
// output index of CPL channels: the phase schedule is shared by all channels, so the two taps read per
// step feed 2*CPL dot products (LDS reads per output and tap: 1 + 2/CPL instead of 3 -- the kernel was
// LDS-bound at three reads per tap); the sample is read once per tap and channel and feeds both dots.
//
// Arithmetic: identical to arb_generic_kernel (STRICT / FUSED) => bit-identical results.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "mrhip_internal.h"
#include "pair_device.h"

#pragma clang fp contract(off)

namespace mrhip {
namespace {

constexpr int kArbThreads = 256;
constexpr int kArbPrefetch = 6;      // samples a thread holds in registers for the next tile (arb_tiled_kernel)

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

// LDS reads are what bounded round 2's form of this kernel (PMC: the LDS busy 74 % of the launch at 8.5 cycles per read
// instruction -- the compiler merged the 8-byte reads of a lane into ds_read2_b64, 128 B/clk -- while the Float64 VALU
// was 39 % busy; profiles/r03/c4_arb_tiled_r02kernel_pmc_summary.json).  This form reads PAIRS with one aligned access
// at the full 256 B/clk (ds_read_b128 / ds_read_b64):
//   * taps: a lane's column depends on its phase, so the banks are laid out with an ODD column pitch -- 32 lanes with 32
//     different phases then hit 32 different 8-byte bank pairs, and equal phases broadcast -- and read 8 bytes at a time
//     through two base registers the compiler cannot prove adjacent (it would merge t[i], t[i+1] into ds_read2_b64, which
//     moves 128 B/clk; a 16-byte read of a phase-gathered column conflicts two ways: 16 lanes, 16 bank quads, 32 phases);
//   * samples: a lane's window starts at an arbitrary sample, so the tile is kept TWICE in LDS, the second copy shifted
//     by one sample: a lane whose window starts at an even offset reads copy A, an odd one copy B, both aligned
//     (16-byte samples -- ComplexF64 -- are one read each and need no second copy).
// The next tile's samples (and the lanes' schedule entries) are loaded into registers BEFORE the current tile is
// computed and written to LDS after it, so no wave waits for HBM inside a tile.
// Tiles handed out in runs (ArbTileArgs::counters; kernels_arb_pipe.hip has the story): the workgroups of a CU do not advance evenly,
// with tile += gridDim the kernel's tail runs under-occupied.  Two runs are always in hand; lane 0 asks for another when one is
// taken into use and publishes the answer before the barrier at the top of the next tile.
struct TileHandout {
    unsigned *ctr;
    long long G, q0, q1;
    int run;
    static constexpr long long kNone = -1;
    __device__ __forceinline__ long long first(unsigned *counters, int run_tiles, unsigned *s_grab, int tid)
    {
        ctr = counters; G = gridDim.x; run = ctr ? run_tiles : 1; q0 = q1 = kNone;
        if (ctr) {
            if (tid == 0) { const unsigned b = atomicAdd(ctr, 2u); s_grab[0] = b; s_grab[1] = b + 1u; }
            __syncthreads();
            q0 = (G + s_grab[0]) * run; q1 = (G + s_grab[1]) * run;
            __syncthreads();
        }
        return static_cast<long long>(blockIdx.x) * run;
    }
    __device__ __forceinline__ long long after(long long t)
    {
        if (!ctr) return t + G;
        if (((t + 1) & (run - 1)) != 0) return t + 1;          // (run is a power of two)
        const long long r = q0;
        q0 = q1; q1 = kNone;
        return r;
    }
    __device__ __forceinline__ bool wants() const { return ctr && q1 == kNone; }
    __device__ __forceinline__ void take(const unsigned *s_grab, unsigned it) { q1 = (G + s_grab[it & 1]) * run; }
    __device__ __forceinline__ void leave(int tid) const
    {
        if (ctr && tid == 0) {
            __threadfence();
            if (atomicAdd(ctr + 64, 1u) == static_cast<unsigned>(G) - 1u) {
                __threadfence();
                ctr[0] = 0u; ctr[64] = 0u;                        // re-armed for the next launch
            }
        }
    }
};

// hand-outs of tiles only where a workgroup has tiles to balance (per_wg tiles each): run length, or 0 for tile += gridDim
static int handout_run(long long per_wg, long long min_tiles)
{
    if (per_wg < min_tiles) return 0;
    int r = 2;
    while (r < 32 && per_wg / (2 * r) >= 40) r *= 2;
    return r;
}

template <typename TX, typename R, int NC, bool FUSED, int CPL, bool PREFETCH>
__global__ __launch_bounds__(kArbThreads, 4) void arb_tiled_kernel(ArbArgs a, ArbTileArgs ta)
{
    struct alignas(sizeof(TX) * NC) Sample { TX c[NC]; };
    constexpr int SPR = sizeof(Sample) >= 16 ? 1 : 2;                    // samples per LDS read
    struct alignas(sizeof(Sample) * SPR) SampleRd { Sample s[SPR]; };
    constexpr int PF = PREFETCH ? kArbPrefetch : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    R *const lpfb = reinterpret_cast<R *>(smem);
    R *const ldpfb = lpfb + ta.bank_elems;
    Sample *const lxA = reinterpret_cast<Sample *>(smem + ta.x_offset_bytes);      // [CPL][MS]
    const int MS = ta.max_span;                                                    // even
    // the same, one sample later -- and 128 bytes further round the banks, so that the lanes of one read that use copy B
    // (odd window offsets) do not land on the banks of their neighbours in copy A
    Sample *const lxB = lxA + (SPR == 2 ? CPL * MS + ta.copyb_pad : 0);

    const int tid = threadIdx.x;
    const int T = a.T, TP = ta.tap_pitch;
    {   // both tap banks -> LDS once per workgroup: element (phi, i) at phi*TP + i
        const R *__restrict__ g0 = static_cast<const R *>(a.taps);
        const R *__restrict__ g1 = static_cast<const R *>(a.dtaps);
        const int total = a.Nphi * T;
        for (int e = tid; e < total; e += kArbThreads) {
            const int phi = e / T, i = e - phi * T;
            lpfb[phi * TP + i] = g0[e];
            ldpfb[phi * TP + i] = g1[e];
        }
    }
    long long ngroups;
    tiles_take_dyn(a.n_out, ta, ngroups, a.dyn);                    // (a device-planned call: the count from the call record)

    // tile -> (first output, last output, channel group, first sample, span)
    struct TileInfo { long long k0, klast, o; int ch0, nchl, span; long long n_lo; };
    auto tile_info = [&](long long tile) {
        TileInfo ti;
        // time-major: the workgroups that run together work on the same stretch of the (shared) phase schedule for
        // different channel groups, so its entries are read from HBM once
        const long long tau = tile / ngroups;
        const int cg = static_cast<int>(tile - tau * ngroups);
        ti.ch0 = cg * CPL;
        ti.nchl = a.nch - ti.ch0 < CPL ? a.nch - ti.ch0 : CPL;
        ti.k0 = tau * ta.tile_out;
        ti.klast = (ti.k0 + ta.tile_out < a.n_out ? ti.k0 + ta.tile_out : a.n_out) - 1;
        ti.n_lo = a.n_idx[ti.k0];
        const long long n_hi = a.n_idx[ti.klast];
        ti.o = ti.n_lo - T;                                 // x[n_lo - T .. n_hi - 1] (0-based), n = 1-based newest sample
        ti.span = static_cast<int>(n_hi - ti.n_lo) + T;
        return ti;
    };
    // element e of the tile: channel e / MS, sample e % MS.  Branch-free: the load is unconditional (from a harmless address
    // where the element lies outside the signal) and the zero is selected after it has arrived -- conditional loads
    // made the compiler wait for all earlier loads (s_waitcnt vmcnt(0)) in front of every one of them
    auto sample_src = [&](const TileInfo &ti, int e, bool *valid) -> const Sample * {
        const int cc = e / MS, sidx = e - cc * MS;
        const long long gi = ti.o + sidx;
        const Sample *px = static_cast<const Sample *>(a.x) + static_cast<long long>(ti.ch0 + cc) * a.x_stride + gi;
        const Sample *ph = static_cast<const Sample *>(a.hist) + static_cast<long long>(ti.ch0 + cc) * a.H + (a.H + gi);
        *valid = cc < ti.nchl && sidx < ti.span && gi < a.x_len && gi >= -static_cast<long long>(a.H);
        const Sample *p = gi >= 0 ? px : ph;
        return *valid ? p : static_cast<const Sample *>(a.taps);
    };
    auto zero_unless = [&](Sample v, bool valid) -> Sample {
#pragma unroll
        for (int c = 0; c < NC; ++c) v.c[c] = valid ? v.c[c] : static_cast<TX>(0);
        return v;
    };
    auto store_sample = [&](int e, const Sample &v) {
        lxA[e] = v;
        if constexpr (SPR == 2) {
            const int cc = e / MS, sidx = e - cc * MS;
            if (sidx > 0) lxB[e - 1] = v;                    // B[s] = sample s + 1
        }
    };
    const int telems = CPL * MS;

    __shared__ unsigned s_grab[2];
    TileHandout th;
    long long tile = th.first(ta.counters, ta.run_tiles, s_grab, tid);
    if (tile >= ta.total_tiles) { th.leave(tid); dev::shiftin_by_last_workgroup<TX, NC>(a.fold, a.x, a.hist, a.x_stride, a.x_len, a.H, a.nch); return; }
    unsigned it = 0;
    bool asked_prev = false;
    TileInfo cur = tile_info(tile);
    Sample pv[PF];
    unsigned pvalid = 0;                                     // bit j: pv[j] lies inside the signal
    if constexpr (PREFETCH) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            bool ok;
            pv[j] = *sample_src(cur, tid + j * kArbThreads, &ok);
            pvalid |= ok ? 1u << j : 0u;
        }
    }
    long long n_pre = 0;
    double acc_pre = 0.0;
    if (PREFETCH && cur.k0 + tid <= cur.klast) { n_pre = a.n_idx[cur.k0 + tid]; acc_pre = a.acc[cur.k0 + tid]; }

    for (;;) {
        __syncthreads();   // the previous tile's reads are done (and, first time, the tap banks are written)
        if constexpr (PREFETCH) {
#pragma unroll
            for (int j = 0; j < PF; ++j) { const int e = tid + j * kArbThreads; if (e < telems) store_sample(e, zero_unless(pv[j], (pvalid >> j) & 1u)); }
        } else {
            for (int e = tid; e < telems; e += kArbThreads) {
                bool ok;
                const Sample v = *sample_src(cur, e, &ok);
                store_sample(e, zero_unless(v, ok));
            }
        }
        __syncthreads();
        // the next tile's loads go out now and land while this tile is computed
        if (asked_prev) th.take(s_grab, it - 1);               // the answer to the previous tile's request
        const long long next = th.after(tile);
        const bool asks = th.wants();                           // (uniform) a run was taken into use: ask for another
        unsigned grabbed = 0u;
        if (asks && tid == 0) grabbed = atomicAdd(th.ctr, 1u);
        const bool have_next = next < ta.total_tiles;
        TileInfo nxt = cur;
        const long long n_mine = n_pre;
        const double acc_mine = acc_pre;
        if constexpr (PREFETCH) {
            if (have_next) {
                nxt = tile_info(next);
                pvalid = 0;
#pragma unroll
                for (int j = 0; j < PF; ++j) {
                    bool ok;
                    pv[j] = *sample_src(nxt, tid + j * kArbThreads, &ok);
                    pvalid |= ok ? 1u << j : 0u;
                }
                if (nxt.k0 + tid <= nxt.klast) { n_pre = a.n_idx[nxt.k0 + tid]; acc_pre = a.acc[nxt.k0 + tid]; }
            }
        }

        for (long long k = cur.k0 + tid; k <= cur.klast; k += kArbThreads) {
            const bool first_round = k == cur.k0 + tid;
            const long long n = PREFETCH && first_round ? n_mine : a.n_idx[k];
            const double pacc = PREFETCH && first_round ? acc_mine : a.acc[k];
            const double phif = __builtin_floor(pacc);
            const double alpha = pacc - phif;               // src/Filters.jl:671-672
            const int phi = static_cast<int>(phif) - 1;     // 0-based column
            const R *tp = lpfb + phi * TP;
            const R *dp = ldpfb + phi * TP;
            int one = 1;
            asm volatile("" : "+v"(one));                   // opaque: keeps t[i] and t[i + 1] two ds_read_b64 (see the kernel's header)
            const R *tp1 = tp + one, *dp1 = dp + one;
            const int w = static_cast<int>(n - cur.n_lo);   // oldest sample of this output's window, within the tile
            const Sample *wp = (SPR == 2 && (w & 1)) ? lxB + (w - 1) : lxA + w;
            R lo[CPL][NC], up[CPL][NC];
            const Sample *wpc[CPL];                         // one base per channel: the tap index is an immediate offset
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) wpc[cc] = wp + cc * MS;
            auto pair_step = [&](int i, bool first) {       // taps i and i + 1
                const R t0 = tp[i], t1 = tp1[i], d0 = dp[i], d1 = dp1[i];
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) {
                    Sample v0, v1;
                    if constexpr (SPR == 2) {
                        const SampleRd v = *reinterpret_cast<const SampleRd *>(wpc[cc] + i);
                        v0 = v.s[0]; v1 = v.s[SPR - 1];
                    } else {
                        v0 = wpc[cc][i]; v1 = wpc[cc][i + 1];
                    }
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        if (first) { lo[cc][c] = t0 * static_cast<R>(v0.c[c]); up[cc][c] = d0 * static_cast<R>(v0.c[c]); }
                        else { lo[cc][c] = mac<R, FUSED>(t0, static_cast<R>(v0.c[c]), lo[cc][c]); up[cc][c] = mac<R, FUSED>(d0, static_cast<R>(v0.c[c]), up[cc][c]); }
                        lo[cc][c] = mac<R, FUSED>(t1, static_cast<R>(v1.c[c]), lo[cc][c]);
                        up[cc][c] = mac<R, FUSED>(d1, static_cast<R>(v1.c[c]), up[cc][c]);
                    }
                }
            };
            auto single_step = [&](int i, bool first) {     // tap i alone (odd tapsPerPhi: the last one; tapsPerPhi == 1)
                const R t = tp[i], d = dp[i];
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) {
                    const Sample v = wpc[cc][i];
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        if (first) { lo[cc][c] = t * static_cast<R>(v.c[c]); up[cc][c] = d * static_cast<R>(v.c[c]); }
                        else { lo[cc][c] = mac<R, FUSED>(t, static_cast<R>(v.c[c]), lo[cc][c]); up[cc][c] = mac<R, FUSED>(d, static_cast<R>(v.c[c]), up[cc][c]); }
                    }
                }
            };
            const int Tp = T & ~1;
            if (Tp >= 2) {
                pair_step(0, true);
#pragma unroll 2
                for (int i = 2; i < Tp; i += 2) pair_step(i, false);
                if (T & 1) single_step(T - 1, false);
            } else {
                single_step(0, true);
            }
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) {
                if (cc < cur.nchl) {
                    R *__restrict__ yc = static_cast<R *>(a.y) + static_cast<long long>(cur.ch0 + cc) * a.y_stride * NC;
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const double prod = static_cast<double>(up[cc][c]) * alpha;      // Filters.jl:730, Float64 combine
                        const double sum = static_cast<double>(lo[cc][c]) + prod;
                        yc[k * NC + c] = static_cast<R>(sum);
                    }
                }
            }
        }
        if (!have_next) break;
        if (asks && tid == 0) s_grab[it & 1] = grabbed;        // (read behind the barrier at the top of the next tile)
        asked_prev = asks;
        ++it;
        tile = next;
        cur = PREFETCH ? nxt : tile_info(next);
    }
    th.leave(tid);
    dev::shiftin_by_last_workgroup<TX, NC>(a.fold, a.x, a.hist, a.x_stride, a.x_len, a.H, a.nch);
}

template <typename TX, typename R, int NC>
hipError_t launch_arb(bool fused, const ArbArgs &a, const ArbTileArgs &ta, size_t lds, hipStream_t s, int num_cus)
{
    auto go = [&](auto kfn) -> hipError_t {
        int per_cu = 0;
        hipError_t eo = occupancy_cached(reinterpret_cast<const void *>(kfn), kArbThreads, lds, &per_cu);
        if (eo != hipSuccess) return eo;
        if (per_cu < 1) per_cu = 1;
        long long g = static_cast<long long>(num_cus) * per_cu;
        if (g > ta.total_tiles) g = ta.total_tiles;
        if (g < 1) g = 1;
        static int dbg = -1;
        if (dbg < 0) { const char *v = std::getenv("MRHIP_DEBUG"); dbg = (v && v[0] == '1') ? 1 : 0; }
        if (dbg == 1) {
            dbg = 0;
            hipFuncAttributes fa;
            (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kfn));
            std::fprintf(stderr, "[mrhip] arb_tiled T=%d Nphi=%d grid=%lld lds=%zu occ/CU=%d regs=%d tile_out=%lld max_span=%d tiles=%lld\n",
                         a.T, a.Nphi, g, lds, per_cu, fa.numRegs, ta.tile_out, ta.max_span, ta.total_tiles);
        }
        ArbTileArgs tq = ta;
        tq.run_tiles = tq.counters ? handout_run(ta.total_tiles / g, 64) : 0;
        if (tq.run_tiles == 0) tq.counters = nullptr;
        launch_kernel(kfn, dim3(static_cast<unsigned>(g)), dim3(kArbThreads), lds, s, a, tq);
        return hipGetLastError();
    };
    const bool pf = ta.prefetch != 0;
#define MRHIP_ARB_GO(C)                                                                                          \
    return fused ? (pf ? go(arb_tiled_kernel<TX, R, NC, true, C, true>) : go(arb_tiled_kernel<TX, R, NC, true, C, false>))   \
                 : (pf ? go(arb_tiled_kernel<TX, R, NC, false, C, true>) : go(arb_tiled_kernel<TX, R, NC, false, C, false>));
    switch (ta.cpl) {
    case 8: MRHIP_ARB_GO(8)
    case 4: MRHIP_ARB_GO(4)
    case 2: MRHIP_ARB_GO(2)
    default: MRHIP_ARB_GO(1)
    }
#undef MRHIP_ARB_GO
}


// ---- FIRFarrow, tiled --------------------------------------------------------------------------------------
// Same staging as arb_tiled_kernel, for the Farrow filter (src/Filters.jl:764-839): a workgroup takes a tile of 256
// consecutive output indices.  Every thread evaluates the T polynomial taps of ITS output once (Float64 Horner,
// rounded to the tap type: Polynomials.jl polyval + the store into currentTaps::Vector{Th}) into an LDS column,
// then the workgroup walks over ALL channels in groups of CPL: stage the contiguous sample run of the group,
// one Vector dot per channel (start from zero on the seam, support.jl:46) with the taps read back from LDS.
// The taps depend on the output index only, so they are evaluated once per 256 outputs x all channels.
// TREG > 0: T <= TREG and the lane keeps its T taps in registers (no tap columns in LDS: 64 KB less per workgroup for 32
// Float64 taps, twice the resident waves); TREG == 0: tap columns in LDS, any T that fits.
template <typename TX, typename R, int NC, bool FUSED, int CPL, int TREG>
__global__ __launch_bounds__(kArbThreads, 3) void farrow_tiled_kernel(FarrowArgs a, ArbTileArgs ta)
{
    struct alignas(sizeof(TX) * NC) Sample { TX c[NC]; };
    constexpr int SPR = sizeof(Sample) >= 16 ? 1 : 2;                        // samples per LDS read (see arb_tiled_kernel)
    struct alignas(sizeof(Sample) * SPR) SampleRd { Sample s[SPR]; };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    R *const tl = reinterpret_cast<R *>(smem);                                  // taps: [T][256] (TREG == 0)
    R treg[TREG > 0 ? TREG : 1];
    const int MS = ta.max_span;                                                 // even
    Sample *const lxA = reinterpret_cast<Sample *>(smem + ta.x_offset_bytes);   // samples: [CPL][MS]
    Sample *const lxB = lxA + (SPR == 2 ? CPL * MS + ta.copyb_pad : 0);         // the same, one sample later, 128 B round the banks
    const int tid = threadIdx.x;
    const int T = a.T, P = a.polyorder;
    {
        long long ngroups;
        tiles_take_dyn(a.n_out, ta, ngroups, a.dyn);                // (a device-planned call: the count from the call record)
    }

    __shared__ unsigned s_grab[2];
    TileHandout th;
    unsigned it = 0;
    bool asked_prev = false;
    for (long long tile = th.first(ta.counters, ta.run_tiles, s_grab, tid); tile < ta.total_tiles; ++it) {
        // (a tile is a stretch of outputs x ALL channel groups here: there are barriers between this request and the next tile)
        if (asked_prev) th.take(s_grab, it - 1);
        const long long tile_after = th.after(tile);
        const bool asks = th.wants();
        if (asks && tid == 0) s_grab[it & 1] = atomicAdd(th.ctr, 1u);
        asked_prev = asks;
        const long long k0 = tile * ta.tile_out;
        const long long klast = (k0 + ta.tile_out < a.n_out ? k0 + ta.tile_out : a.n_out) - 1;
        const long long n_lo = a.n_idx[k0], n_hi = a.n_idx[klast];
        const long long o = n_lo - T;
        const int span = static_cast<int>(n_hi - n_lo) + T;
        const long long k = k0 + tid;
        const bool have = k <= klast;
        long long n = 0;
        if (have) {
            n = a.n_idx[k];
            const double phase = a.acc[k];
            auto tap_of = [&](int i) -> R {          // Horner in Float64, separately rounded multiply and add
                const double *__restrict__ c = a.pnfb + static_cast<long long>(i) * (P + 1);
                double yv = c[P];
                for (int j = P - 1; j >= 0; --j) { const double t = phase * yv; yv = c[j] + t; }
                return a.tap_f32 ? static_cast<R>(static_cast<float>(yv)) : static_cast<R>(yv);
            };
            if constexpr (TREG > 0) {
                // (the degree in the outer loop, the taps unrolled inside: TREG independent Horner chains whose coefficient loads
                //  go out together -- see kernels_farrow_pipe.hip)
                double yv[TREG];
#pragma unroll
                for (int i = 0; i < TREG; ++i) yv[i] = i < T ? a.pnfb[static_cast<long long>(i) * (P + 1) + P] : 0.0;
                for (int j = P - 1; j >= 0; --j) {
#pragma unroll
                    for (int i = 0; i < TREG; ++i) {
                        if (i < T) {
                            const double t = phase * yv[i];
                            yv[i] = a.pnfb[static_cast<long long>(i) * (P + 1) + j] + t;
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < TREG; ++i)
                    if (i < T) treg[i] = a.tap_f32 ? static_cast<R>(static_cast<float>(yv[i])) : static_cast<R>(yv[i]);
            } else {
                for (int i = 0; i < T; ++i) tl[i * kArbThreads + tid] = tap_of(i);
            }
        }
        const bool seam = n < a.seam_below;           // kernel.xIdx < kernel.tapsPer𝜙, Filters.jl:818 (never in a piece that continues a call)
        const int w = have ? static_cast<int>(n - n_lo) : 0;
        const bool oddw = SPR == 2 && (w & 1);
        const Sample *const wp = oddw ? lxB + (w - 1) : lxA + w;
        for (int ch0 = 0; ch0 < a.nch; ch0 += CPL) {
            const int nchl = a.nch - ch0 < CPL ? a.nch - ch0 : CPL;
            __syncthreads();   // the previous group's reads are done
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) {
                if (cc < nchl) {
                    const Sample *__restrict__ xc = static_cast<const Sample *>(a.x) + static_cast<long long>(ch0 + cc) * a.x_stride;
                    const Sample *__restrict__ hc = static_cast<const Sample *>(a.hist) + static_cast<long long>(ch0 + cc) * a.H;
                    for (int s = tid; s < span; s += kArbThreads) {
                        const long long gi = o + s;
                        Sample v;
#pragma unroll
                        for (int c = 0; c < NC; ++c) v.c[c] = static_cast<TX>(0);
                        if (gi >= 0) { if (gi < a.x_len) v = xc[gi]; }
                        else if (gi >= -static_cast<long long>(a.H)) v = hc[a.H + gi];
                        lxA[cc * MS + s] = v;
                        if (SPR == 2 && s > 0) lxB[cc * MS + s - 1] = v;
                    }
                }
            }
            __syncthreads();
            if (have) {
                R acc[CPL][NC];
                const Sample *wpc[CPL];                         // one base per channel: the tap index is an immediate offset
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) wpc[cc] = wp + cc * MS;
                auto tap_at = [&](int i) -> R { if constexpr (TREG > 0) return treg[i]; else return tl[i * kArbThreads + tid]; };
                auto single_step = [&](int i, R t, bool first) {
#pragma unroll
                    for (int cc = 0; cc < CPL; ++cc) {
                        const Sample v = wpc[cc][i];
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            if (first) {
                                acc[cc][c] = t * static_cast<R>(v.c[c]);
                                if (seam) acc[cc][c] = static_cast<R>(0) + acc[cc][c];       // support.jl:46: the seam dot starts from zero
                            } else {
                                acc[cc][c] = mac<R, FUSED>(t, static_cast<R>(v.c[c]), acc[cc][c]);
                            }
                        }
                    }
                };
                auto pair_step = [&](int i, R t0, R t1, bool first) {
#pragma unroll
                    for (int cc = 0; cc < CPL; ++cc) {
                        Sample v0, v1;
                        if constexpr (SPR == 2) {
                            const SampleRd v = *reinterpret_cast<const SampleRd *>(wpc[cc] + i);
                            v0 = v.s[0]; v1 = v.s[SPR - 1];
                        } else {
                            v0 = wpc[cc][i]; v1 = wpc[cc][i + 1];
                        }
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            if (first) {
                                acc[cc][c] = t0 * static_cast<R>(v0.c[c]);
                                if (seam) acc[cc][c] = static_cast<R>(0) + acc[cc][c];
                            } else {
                                acc[cc][c] = mac<R, FUSED>(t0, static_cast<R>(v0.c[c]), acc[cc][c]);
                            }
                            acc[cc][c] = mac<R, FUSED>(t1, static_cast<R>(v1.c[c]), acc[cc][c]);
                        }
                    }
                };
                if constexpr (TREG > 0) {
#pragma unroll
                    for (int i = 0; i < TREG; i += 2) {          // (the conditions are wave-uniform)
                        if (i + 1 < T) pair_step(i, treg[i], treg[i + 1 < TREG ? i + 1 : i], i == 0);
                        else if (i < T) single_step(i, treg[i], i == 0);
                        // keep the scheduler from hoisting every LDS read of the unrolled loop to its top (252 VGPRs, or spills)
                        if ((i & 3) == 2) __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                    const int Tp = T & ~1;
                    if (Tp >= 2) {
                        pair_step(0, tap_at(0), tap_at(1), true);
#pragma unroll 2
                        for (int i = 2; i < Tp; i += 2) pair_step(i, tap_at(i), tap_at(i + 1), false);
                        if (T & 1) single_step(T - 1, tap_at(T - 1), false);
                    } else {
                        single_step(0, tap_at(0), true);
                    }
                }
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) {
                    if (cc < nchl) {
                        R *__restrict__ yc = static_cast<R *>(a.y) + static_cast<long long>(ch0 + cc) * a.y_stride * NC;
#pragma unroll
                        for (int c = 0; c < NC; ++c) yc[k * NC + c] = acc[cc][c];
                    }
                }
            }
        }
        if constexpr (TREG == 0) __syncthreads();   // the tap columns are rewritten by the next tile
        tile = tile_after;
    }
    th.leave(tid);
    dev::shiftin_by_last_workgroup<TX, NC>(a.fold, a.x, a.hist, a.x_stride, a.x_len, a.H, a.nch);
}

template <typename TX, typename R, int NC>
hipError_t launch_farrow_t(bool fused, const FarrowArgs &a, const ArbTileArgs &ta, size_t lds, hipStream_t s, int num_cus)
{
    auto go = [&](auto kfn) -> hipError_t {
        int per_cu = 0;
        hipError_t eo = occupancy_cached(reinterpret_cast<const void *>(kfn), kArbThreads, lds, &per_cu);
        if (eo != hipSuccess) return eo;
        if (per_cu < 1) per_cu = 1;
        long long g = static_cast<long long>(num_cus) * per_cu;
        if (g > ta.total_tiles) g = ta.total_tiles;
        if (g < 1) g = 1;
        // (a tile is a stretch of outputs x all channel groups: few, long tiles that the workgroups finish evenly -- handing them
        //  out, one per request, measured 4.22 against 4.18 ms at config 4's shape in ComplexF64: off unless MRHIP_FARROW_TILED_DYNAMIC=1)
        ArbTileArgs tq = ta;
        const long long per_wg = ta.total_tiles / g;
        tq.run_tiles = !tq.counters || per_wg < 8 || MRHIP_ENV_INT("MRHIP_FARROW_TILED_DYNAMIC", 0) == 0 ? 0 : (per_wg >= 128 ? handout_run(per_wg, 128) : 1);
        if (tq.run_tiles == 0) tq.counters = nullptr;
        launch_kernel(kfn, dim3(static_cast<unsigned>(g)), dim3(kArbThreads), lds, s, a, tq);
        return hipGetLastError();
    };
    if (ta.tap_pitch == 1) {     // taps in registers (T <= 32)
        if (ta.cpl == 4) return fused ? go(farrow_tiled_kernel<TX, R, NC, true, 4, 32>) : go(farrow_tiled_kernel<TX, R, NC, false, 4, 32>);
        return fused ? go(farrow_tiled_kernel<TX, R, NC, true, 1, 32>) : go(farrow_tiled_kernel<TX, R, NC, false, 1, 32>);
    }
    if (ta.cpl == 4) return fused ? go(farrow_tiled_kernel<TX, R, NC, true, 4, 0>) : go(farrow_tiled_kernel<TX, R, NC, false, 4, 0>);
    return fused ? go(farrow_tiled_kernel<TX, R, NC, true, 1, 0>) : go(farrow_tiled_kernel<TX, R, NC, false, 1, 0>);
}

}  // namespace

// `n_idx_host` is the host copy of the per-output input indices (non-decreasing); when the schedule was evaluated on
// the device it is NULL and `spans[z]` holds the largest n[last] - n[first] over the aligned tiles of 256 << z outputs
// (kernels_schedule.hip).  Returns false when the tap banks plus a useful sample tile do not fit LDS (caller uses the
// generic kernel).
bool plan_arb_tiled(const TypeKey &tk, const ArbArgs &a, const int32_t *n_idx_host, const int *spans, int num_cus, ArbTileArgs *out, size_t *lds)
{
    static const int enabled = [] { const char *v = std::getenv("MRHIP_ARB_TILED"); return !(v && v[0] == '0'); }();
    if (!enabled || a.n_out < 1) return false;
    const size_t rs = tk.r_f64 ? 8 : 4;
    const size_t sb = (tk.x_f64 ? 8 : 4) * (tk.complex_x ? 2 : 1);
    if (sb <= 8) {                                       // Float32 and 8-byte samples (Float64, ComplexF32): the hand-pipelined kernel
        long long span256 = -1;
        if (n_idx_host) {
            for (long long k0 = 0; k0 < a.n_out; k0 += 256) {
                const long long kl = std::min<long long>(k0 + 256, a.n_out) - 1;
                span256 = std::max<long long>(span256, static_cast<long long>(n_idx_host[kl]) - n_idx_host[k0]);
            }
        } else if (spans) {
            span256 = spans[sched_span_index(256)];
        }
        if (span256 >= 0 && plan_arb_pipe(tk, a, span256, out, lds)) return true;
    }
    const int copies = sb >= 16 ? 1 : 2;                 // the sample tile is kept twice, one sample apart (aligned pair reads)
    const int TP = a.T | 1;                              // odd column pitch: lanes with different phases read different banks
    const size_t bank_elems = static_cast<size_t>(a.Nphi) * TP;
    const size_t banks_bytes = (2 * bank_elems * rs + 15) / 16 * 16;
    if (banks_bytes > 96 * 1024) return false;
    // channels per lane: the tap reads are shared by CPL channels (needs enough channels to keep the machine busy)
    static const int env_cpl = [] { const char *v = std::getenv("MRHIP_ARB_CPL"); return v && *v ? std::atoi(v) : 0; }();
    int cpl = a.nch >= 32 ? 4 : (a.nch >= 8 ? 2 : 1);
    if (tk.complex_x && tk.r_f64 && cpl > 2) cpl = 2;     // (four complex Float64 channels per lane do not fit 128 VGPRs)
    if (env_cpl == 1 || env_cpl == 2 || env_cpl == 4 || env_cpl == 8) cpl = env_cpl;
    static const int env_tile = [] { const char *v = std::getenv("MRHIP_ARB_TILE"); return v && *v ? std::atoi(v) : 0; }();
    static const int env_pf = [] { const char *v = std::getenv("MRHIP_ARB_PREFETCH"); return v && *v ? std::atoi(v) : 1; }();
    static const int arb_cap_kib = [] { const char *v = std::getenv("MRHIP_ARB_CAP_KIB"); return v && *v ? std::atoi(v) : 48; }();
    // tile: a multiple of 256 outputs whose sample span fits the remaining budget
    long long tile_out = env_tile >= 256 ? env_tile / 256 * 256 : (cpl >= 4 && tk.r_f64 ? 256 : 1024);
    if (!n_idx_host) {                       // device schedule: spans are known for 256, 512, 1024 only
        if (!spans) return false;
        tile_out = tile_out >= 1024 ? 1024 : (tile_out >= 512 ? 512 : 256);
    }
    const long long want_tiles = 4LL * num_cus;
    for (;;) {
        const long long groups = (a.nch + cpl - 1) / cpl;
        long long to = tile_out;
        while (to > 256 && ((a.n_out + to - 1) / to) * groups < want_tiles) to /= 2;
        for (;;) {
            long long max_span = 0;
            if (n_idx_host) {
                for (long long k0 = 0; k0 < a.n_out; k0 += to) {
                    const long long kl = std::min<long long>(k0 + to, a.n_out) - 1;
                    max_span = std::max<long long>(max_span, static_cast<long long>(n_idx_host[kl]) - n_idx_host[k0] + a.T);
                }
            } else {
                max_span = static_cast<long long>(spans[sched_span_index(to)]) + a.T;
            }
            max_span = (max_span + 2) / 2 * 2;           // even (+ room for the pair read of an odd window)
            const int copyb_pad = copies == 2 ? static_cast<int>((128 + 256 - (static_cast<size_t>(max_span) * sb * cpl) % 256) % 256 / sb) : 0;   // samples: copy B starts 128 B (mod 256) after copy A
            const size_t total = banks_bytes + (static_cast<size_t>(max_span) * cpl * copies + copyb_pad) * sb;
            if (total <= static_cast<size_t>(tk.r_f64 ? 64 : arb_cap_kib) * 1024 || to == 256) {
                if (total > 150 * 1024) break;           // does not fit with this many channels per lane
                ArbTileArgs ta{};
                ta.tap_pitch = TP;
                ta.bank_elems = static_cast<int>(bank_elems);
                ta.x_offset_bytes = static_cast<int>(banks_bytes);
                ta.max_span = static_cast<int>(max_span);
                ta.copyb_pad = copyb_pad;
                ta.tile_out = to;
                ta.tiles_per_channel = (a.n_out + to - 1) / to;
                ta.total_tiles = ta.tiles_per_channel * groups;
                ta.cpl = cpl;
                // the next tile's samples wait in registers while this one is computed, if they fit kArbPrefetch per thread
                ta.prefetch = env_pf && to == 256 && static_cast<long long>(cpl) * max_span <= static_cast<long long>(kArbPrefetch) * kArbThreads;
                *out = ta;
                *lds = total;
                return true;
            }
            to /= 2;
        }
        if (cpl == 1) return false;
        cpl /= 2;                                         // a decimating rate stretches the span: fewer channels per lane
    }
}

hipError_t launch_arb_tiled(const TypeKey &tk, bool fused, const ArbArgs &a, const ArbTileArgs &ta, size_t lds, hipStream_t s,
                            const char **kname, int num_cus)
{
    if (ta.pipe) return launch_arb_pipe(tk, fused, a, ta, lds, s, kname, num_cus);
    *kname = "arb_tiled_kernel";
    if (!tk.x_f64 && !tk.r_f64) return tk.complex_x ? launch_arb<float, float, 2>(fused, a, ta, lds, s, num_cus) : launch_arb<float, float, 1>(fused, a, ta, lds, s, num_cus);
    if (!tk.x_f64 && tk.r_f64) return tk.complex_x ? launch_arb<float, double, 2>(fused, a, ta, lds, s, num_cus) : launch_arb<float, double, 1>(fused, a, ta, lds, s, num_cus);
    if (tk.x_f64 && tk.r_f64) return tk.complex_x ? launch_arb<double, double, 2>(fused, a, ta, lds, s, num_cus) : launch_arb<double, double, 1>(fused, a, ta, lds, s, num_cus);
    return hipErrorInvalidValue;
}

// (kernels_farrow_pipe.hip)
bool plan_farrow_pipe(const TypeKey &tk, const FarrowArgs &a, long long span256, ArbTileArgs *out, size_t *lds);
hipError_t launch_farrow_pipe(const TypeKey &tk, bool fused, const FarrowArgs &a, const ArbTileArgs &ta, size_t lds, hipStream_t s,
                              const char **kname, int num_cus);

// FIRFarrow: tap columns of 256 outputs (T*256 elements of R) plus the sample runs of CPL channels must fit LDS.
bool plan_farrow_tiled(const TypeKey &tk, const FarrowArgs &a, const int32_t *n_idx_host, const int *spans, int num_cus, ArbTileArgs *out, size_t *lds)
{
    static const int enabled = [] { const char *v = std::getenv("MRHIP_FARROW_TILED"); return !(v && v[0] == '0'); }();
    (void)num_cus;
    if (!enabled || a.n_out < 1) return false;
    const size_t rs = tk.r_f64 ? 8 : 4;
    const size_t sb = (tk.x_f64 ? 8 : 4) * (tk.complex_x ? 2 : 1);
    if (sb <= 8 && a.T <= 32) {                          // Float32 and 8-byte samples (Float64, ComplexF32): the hand-pipelined kernel
        long long span256 = -1;
        if (n_idx_host) {
            for (long long k0 = 0; k0 < a.n_out; k0 += 256) {
                const long long kl = std::min<long long>(k0 + 256, a.n_out) - 1;
                span256 = std::max<long long>(span256, static_cast<long long>(n_idx_host[kl]) - n_idx_host[k0]);
            }
        } else if (spans) {
            span256 = spans[sched_span_index(256)];
        }
        if (span256 >= 0 && plan_farrow_pipe(tk, a, span256, out, lds)) return true;
    }
    static const int regs_ok = [] { const char *v = std::getenv("MRHIP_FARROW_REGS"); return !(v && v[0] == '0'); }();
    const bool in_regs = a.T <= 32 && regs_ok;          // the lane keeps its taps in registers: no tap columns in LDS
    const size_t taps_bytes = in_regs ? 0 : (static_cast<size_t>(a.T) * kArbThreads * rs + 15) / 16 * 16;
    if (taps_bytes > 96 * 1024) return false;
    const int cpl = a.nch >= 4 ? 4 : 1;
    const int copies = sb >= 16 ? 1 : 2;              // the sample tile twice, one sample apart: aligned pair reads
    const long long tile_out = kArbThreads;
    long long max_span = 0;
    if (n_idx_host) {
        for (long long k0 = 0; k0 < a.n_out; k0 += tile_out) {
            const long long kl = std::min<long long>(k0 + tile_out, a.n_out) - 1;
            max_span = std::max<long long>(max_span, static_cast<long long>(n_idx_host[kl]) - n_idx_host[k0] + a.T);
        }
    } else {
        if (!spans) return false;
        max_span = static_cast<long long>(spans[sched_span_index(tile_out)]) + a.T;
    }
    max_span = (max_span + 2) / 2 * 2;
    const int copyb_pad = copies == 2 ? static_cast<int>((128 + 256 - (static_cast<size_t>(max_span) * sb * cpl) % 256) % 256 / sb) : 0;
    const size_t total = taps_bytes + (static_cast<size_t>(max_span) * cpl * copies + copyb_pad) * sb;
    if (total > 150 * 1024) return false;
    ArbTileArgs ta{};
    ta.cpl = cpl;
    ta.tap_pitch = in_regs ? 1 : 0;                   // (FIRFarrow has no tap bank: the field says where the taps live)
    ta.x_offset_bytes = static_cast<int>(taps_bytes);
    ta.max_span = static_cast<int>(max_span);
    ta.copyb_pad = copyb_pad;
    ta.tile_out = tile_out;
    ta.tiles_per_channel = (a.n_out + tile_out - 1) / tile_out;
    ta.total_tiles = ta.tiles_per_channel;            // a tile covers all channels
    *out = ta;
    *lds = total;
    return true;
}

hipError_t launch_farrow_tiled(const TypeKey &tk, bool fused, const FarrowArgs &a, const ArbTileArgs &ta, size_t lds, hipStream_t s,
                               const char **kname, int num_cus)
{
    if (ta.pipe) return launch_farrow_pipe(tk, fused, a, ta, lds, s, kname, num_cus);
    *kname = "farrow_tiled_kernel";
    if (!tk.x_f64 && !tk.r_f64) return tk.complex_x ? launch_farrow_t<float, float, 2>(fused, a, ta, lds, s, num_cus) : launch_farrow_t<float, float, 1>(fused, a, ta, lds, s, num_cus);
    if (!tk.x_f64 && tk.r_f64) return tk.complex_x ? launch_farrow_t<float, double, 2>(fused, a, ta, lds, s, num_cus) : launch_farrow_t<float, double, 1>(fused, a, ta, lds, s, num_cus);
    if (tk.x_f64 && tk.r_f64) return tk.complex_x ? launch_farrow_t<double, double, 2>(fused, a, ta, lds, s, num_cus) : launch_farrow_t<double, double, 1>(fused, a, ta, lds, s, num_cus);
    return hipErrorInvalidValue;
}

}  // namespace mrhip
