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

#pragma clang fp contract(off)

namespace mrhip {
namespace {

constexpr int kArbThreads = 256;

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

template <typename TX, typename R, int NC, bool FUSED, int CPL>
__global__ __launch_bounds__(kArbThreads) void arb_tiled_kernel(ArbArgs a, ArbTileArgs ta)
{
    struct alignas(sizeof(TX) * NC) Sample { TX c[NC]; };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    R *const lpfb = reinterpret_cast<R *>(smem);
    R *const ldpfb = lpfb + ta.bank_elems;
    Sample *const lx = reinterpret_cast<Sample *>(smem + ta.x_offset_bytes);

    const int tid = threadIdx.x;
    const int T = a.T, TP = ta.tap_pitch;
    // both tap banks -> LDS once per workgroup: element (phi, i) at phi*TP + i
    {
        const R *__restrict__ g0 = static_cast<const R *>(a.taps);
        const R *__restrict__ g1 = static_cast<const R *>(a.dtaps);
        const int total = a.Nphi * T;
        for (int e = tid; e < total; e += kArbThreads) {
            const int phi = e / T, i = e - phi * T;
            lpfb[phi * TP + i] = g0[e];
            ldpfb[phi * TP + i] = g1[e];
        }
    }

    for (long long tile = blockIdx.x; tile < ta.total_tiles; tile += gridDim.x) {
        // time-major: the workgroups that run together work on the same stretch of the (shared) phase schedule for
        // different channel groups, so its entries are read from HBM once
        const long long ngroups = ta.total_tiles / ta.tiles_per_channel;
        const long long tau = tile / ngroups;
        const int cg = static_cast<int>(tile - tau * ngroups);                          // channel group
        const int ch0 = cg * CPL;
        const int nchl = a.nch - ch0 < CPL ? a.nch - ch0 : CPL;                         // channels of this group
        const long long k0 = tau * ta.tile_out;
        const long long klast = (k0 + ta.tile_out < a.n_out ? k0 + ta.tile_out : a.n_out) - 1;
        // samples this tile touches: x[n_lo - T .. n_hi - 1] (0-based), n = 1-based newest-sample index
        const long long n_lo = a.n_idx[k0], n_hi = a.n_idx[klast];
        const long long o = n_lo - T;
        const int span = static_cast<int>(n_hi - n_lo) + T;

        __syncthreads();   // previous tile's reads are done (and, first time, the tap banks are written)
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) {
            if (cc < nchl) {
                const Sample *__restrict__ xc = static_cast<const Sample *>(a.x) + static_cast<long long>(ch0 + cc) * a.x_stride;
                const Sample *__restrict__ hc = static_cast<const Sample *>(a.hist) + static_cast<long long>(ch0 + cc) * a.H;
                Sample *const lxc = lx + static_cast<size_t>(cc) * ta.max_span;
                for (int s = tid; s < span; s += kArbThreads) {
                    const long long gi = o + s;
                    Sample v;
#pragma unroll
                    for (int c = 0; c < NC; ++c) v.c[c] = static_cast<TX>(0);
                    if (gi >= 0) { if (gi < a.x_len) v = xc[gi]; }
                    else if (gi >= -static_cast<long long>(a.H)) v = hc[a.H + gi];
                    lxc[s] = v;
                }
            }
        }
        __syncthreads();

        for (long long k = k0 + tid; k <= klast; k += kArbThreads) {
            const long long n = a.n_idx[k];
            const double pacc = a.acc[k];
            const double phif = __builtin_floor(pacc);
            const double alpha = pacc - phif;               // src/Filters.jl:671-672
            const int phi = static_cast<int>(phif) - 1;     // 0-based column
            const R *tp = lpfb + phi * TP;
            const R *dp = ldpfb + phi * TP;
            const Sample *wp = lx + (n - n_lo);             // oldest sample of this output's window (channel 0 of the group)
            R lo[CPL][NC], up[CPL][NC];
            {
                const R t = tp[0], d = dp[0];
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) {
                    const Sample v = wp[static_cast<size_t>(cc) * ta.max_span];
#pragma unroll
                    for (int c = 0; c < NC; ++c) { lo[cc][c] = t * static_cast<R>(v.c[c]); up[cc][c] = d * static_cast<R>(v.c[c]); }
                }
            }
#pragma unroll 4
            for (int i = 1; i < T; ++i) {
                const R t = tp[i], d = dp[i];
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) {
                    const Sample v = wp[static_cast<size_t>(cc) * ta.max_span + i];
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        lo[cc][c] = mac<R, FUSED>(t, static_cast<R>(v.c[c]), lo[cc][c]);
                        up[cc][c] = mac<R, FUSED>(d, static_cast<R>(v.c[c]), up[cc][c]);
                    }
                }
            }
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) {
                if (cc < nchl) {
                    R *__restrict__ yc = static_cast<R *>(a.y) + static_cast<long long>(ch0 + cc) * a.y_stride * NC;
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const double prod = static_cast<double>(up[cc][c]) * alpha;      // Filters.jl:730, Float64 combine
                        const double sum = static_cast<double>(lo[cc][c]) + prod;
                        yc[k * NC + c] = static_cast<R>(sum);
                    }
                }
            }
        }
    }
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
        launch_kernel(kfn, dim3(static_cast<unsigned>(g)), dim3(kArbThreads), lds, s, a, ta);
        return hipGetLastError();
    };
    switch (ta.cpl) {
    case 8: return fused ? go(arb_tiled_kernel<TX, R, NC, true, 8>) : go(arb_tiled_kernel<TX, R, NC, false, 8>);
    case 4: return fused ? go(arb_tiled_kernel<TX, R, NC, true, 4>) : go(arb_tiled_kernel<TX, R, NC, false, 4>);
    case 2: return fused ? go(arb_tiled_kernel<TX, R, NC, true, 2>) : go(arb_tiled_kernel<TX, R, NC, false, 2>);
    default: return fused ? go(arb_tiled_kernel<TX, R, NC, true, 1>) : go(arb_tiled_kernel<TX, R, NC, false, 1>);
    }
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
__global__ __launch_bounds__(kArbThreads) void farrow_tiled_kernel(FarrowArgs a, ArbTileArgs ta)
{
    struct alignas(sizeof(TX) * NC) Sample { TX c[NC]; };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    R *const tl = reinterpret_cast<R *>(smem);                                  // taps: [T][256] (TREG == 0)
    R treg[TREG > 0 ? TREG : 1];
    Sample *const lx = reinterpret_cast<Sample *>(smem + ta.x_offset_bytes);    // samples: [CPL][max_span]
    const int tid = threadIdx.x;
    const int T = a.T, P = a.polyorder;

    for (long long tile = blockIdx.x; tile < ta.total_tiles; tile += gridDim.x) {
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
#pragma unroll
                for (int i = 0; i < TREG; ++i)
                    if (i < T) treg[i] = tap_of(i);
            } else {
                for (int i = 0; i < T; ++i) tl[i * kArbThreads + tid] = tap_of(i);
            }
        }
        const bool seam = n < T;                      // kernel.xIdx < kernel.tapsPer𝜙, Filters.jl:818
        for (int ch0 = 0; ch0 < a.nch; ch0 += CPL) {
            const int nchl = a.nch - ch0 < CPL ? a.nch - ch0 : CPL;
            __syncthreads();   // the previous group's reads are done
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) {
                if (cc < nchl) {
                    const Sample *__restrict__ xc = static_cast<const Sample *>(a.x) + static_cast<long long>(ch0 + cc) * a.x_stride;
                    const Sample *__restrict__ hc = static_cast<const Sample *>(a.hist) + static_cast<long long>(ch0 + cc) * a.H;
                    Sample *const lxc = lx + static_cast<size_t>(cc) * ta.max_span;
                    for (int s = tid; s < span; s += kArbThreads) {
                        const long long gi = o + s;
                        Sample v;
#pragma unroll
                        for (int c = 0; c < NC; ++c) v.c[c] = static_cast<TX>(0);
                        if (gi >= 0) { if (gi < a.x_len) v = xc[gi]; }
                        else if (gi >= -static_cast<long long>(a.H)) v = hc[a.H + gi];
                        lxc[s] = v;
                    }
                }
            }
            __syncthreads();
            if (have) {
                const Sample *wp = lx + (n - n_lo);
                R acc[CPL][NC];
                {
                    R t;
                    if constexpr (TREG > 0) t = treg[0]; else t = tl[tid];
#pragma unroll
                    for (int cc = 0; cc < CPL; ++cc) {
                        const Sample v = wp[static_cast<size_t>(cc) * ta.max_span];
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            acc[cc][c] = t * static_cast<R>(v.c[c]);
                            if (seam) acc[cc][c] = static_cast<R>(0) + acc[cc][c];
                        }
                    }
                }
                auto tap_step = [&](int i, R t) {
#pragma unroll
                    for (int cc = 0; cc < CPL; ++cc) {
                        const Sample v = wp[static_cast<size_t>(cc) * ta.max_span + i];
#pragma unroll
                        for (int c = 0; c < NC; ++c) acc[cc][c] = mac<R, FUSED>(t, static_cast<R>(v.c[c]), acc[cc][c]);
                    }
                };
                if constexpr (TREG > 0) {
#pragma unroll
                    for (int i = 1; i < TREG; ++i)
                        if (i < T) tap_step(i, treg[i]);        // (wave-uniform)
                } else {
#pragma unroll 4
                    for (int i = 1; i < T; ++i) tap_step(i, tl[i * kArbThreads + tid]);
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
    }
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
        launch_kernel(kfn, dim3(static_cast<unsigned>(g)), dim3(kArbThreads), lds, s, a, ta);
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
    const int TP = a.T + 1;
    const size_t bank_elems = static_cast<size_t>(a.Nphi) * TP;
    const size_t banks_bytes = (2 * bank_elems * rs + 15) / 16 * 16;
    if (banks_bytes > 96 * 1024) return false;
    // channels per lane: the tap reads are shared by CPL channels (needs enough channels to keep the machine busy)
    static const int env_cpl = [] { const char *v = std::getenv("MRHIP_ARB_CPL"); return v && *v ? std::atoi(v) : 0; }();
    int cpl = a.nch >= 32 ? 4 : (a.nch >= 8 ? 2 : 1);
    if (env_cpl == 1 || env_cpl == 2 || env_cpl == 4 || env_cpl == 8) cpl = env_cpl;
    const long long groups = (a.nch + cpl - 1) / cpl;
    // tile: a multiple of 256 outputs whose sample span fits the remaining budget
    static const int env_tile = [] { const char *v = std::getenv("MRHIP_ARB_TILE"); return v && *v ? std::atoi(v) : 0; }();
    // (measured, 256 ch x 2e6, rate pi/3, `scripts/exp_arb_knobs.py`: Float64 arithmetic -- config 4 -- with several channels per
    //  lane: 256-output tiles 20.4 % of HBM, 1024-output tiles 18.2 %; Float32 arithmetic: 12.9 % against 19.6 %)
    long long tile_out = env_tile >= 256 ? env_tile / 256 * 256 : (cpl >= 4 && tk.r_f64 ? 256 : 1024);
    if (!n_idx_host) {                       // device schedule: spans are known for 256, 512, 1024 only
        if (!spans) return false;
        tile_out = tile_out >= 1024 ? 1024 : (tile_out >= 512 ? 512 : 256);
    }
    const long long want_tiles = 4LL * num_cus;
    static const int arb_cap_kib = [] { const char *v = std::getenv("MRHIP_ARB_CAP_KIB"); return v && *v ? std::atoi(v) : 36; }();
    while (tile_out > 256 && ((a.n_out + tile_out - 1) / tile_out) * groups < want_tiles) tile_out /= 2;
    for (;;) {
        long long max_span = 0;
        if (n_idx_host) {
            for (long long k0 = 0; k0 < a.n_out; k0 += tile_out) {
                const long long kl = std::min<long long>(k0 + tile_out, a.n_out) - 1;
                max_span = std::max<long long>(max_span, static_cast<long long>(n_idx_host[kl]) - n_idx_host[k0] + a.T);
            }
        } else {
            max_span = static_cast<long long>(spans[tile_out == 1024 ? 2 : (tile_out == 512 ? 1 : 0)]) + a.T;
        }
        const size_t total = banks_bytes + static_cast<size_t>(max_span) * sb * cpl;
        // Float32 arithmetic: large tiles as long as five workgroups still fit a CU (a decimating rate stretches the span)
        if (total <= (tk.r_f64 ? 64 : arb_cap_kib) * 1024 || tile_out == 256) {
            if (total > 150 * 1024) return false;
            ArbTileArgs ta{};
            ta.tap_pitch = TP;
            ta.bank_elems = static_cast<int>(bank_elems);
            ta.x_offset_bytes = static_cast<int>(banks_bytes);
            ta.max_span = static_cast<int>(max_span);
            ta.tile_out = tile_out;
            ta.tiles_per_channel = (a.n_out + tile_out - 1) / tile_out;
            ta.total_tiles = ta.tiles_per_channel * groups;
            ta.cpl = cpl;
            *out = ta;
            *lds = total;
            return true;
        }
        tile_out /= 2;
    }
}

hipError_t launch_arb_tiled(const TypeKey &tk, bool fused, const ArbArgs &a, const ArbTileArgs &ta, size_t lds, hipStream_t s,
                            const char **kname, int num_cus)
{
    *kname = "arb_tiled_kernel";
    if (!tk.x_f64 && !tk.r_f64) return tk.complex_x ? launch_arb<float, float, 2>(fused, a, ta, lds, s, num_cus) : launch_arb<float, float, 1>(fused, a, ta, lds, s, num_cus);
    if (!tk.x_f64 && tk.r_f64) return tk.complex_x ? launch_arb<float, double, 2>(fused, a, ta, lds, s, num_cus) : launch_arb<float, double, 1>(fused, a, ta, lds, s, num_cus);
    if (tk.x_f64 && tk.r_f64) return tk.complex_x ? launch_arb<double, double, 2>(fused, a, ta, lds, s, num_cus) : launch_arb<double, double, 1>(fused, a, ta, lds, s, num_cus);
    return hipErrorInvalidValue;
}

// FIRFarrow: tap columns of 256 outputs (T*256 elements of R) plus the sample runs of CPL channels must fit LDS.
bool plan_farrow_tiled(const TypeKey &tk, const FarrowArgs &a, const int32_t *n_idx_host, const int *spans, int num_cus, ArbTileArgs *out, size_t *lds)
{
    static const int enabled = [] { const char *v = std::getenv("MRHIP_FARROW_TILED"); return !(v && v[0] == '0'); }();
    (void)num_cus;
    if (!enabled || a.n_out < 1) return false;
    const size_t rs = tk.r_f64 ? 8 : 4;
    const size_t sb = (tk.x_f64 ? 8 : 4) * (tk.complex_x ? 2 : 1);
    static const int regs_ok = [] { const char *v = std::getenv("MRHIP_FARROW_REGS"); return !(v && v[0] == '0'); }();
    const bool in_regs = a.T <= 32 && regs_ok;          // the lane keeps its taps in registers: no tap columns in LDS
    const size_t taps_bytes = in_regs ? 0 : (static_cast<size_t>(a.T) * kArbThreads * rs + 15) / 16 * 16;
    if (taps_bytes > 96 * 1024) return false;
    const int cpl = a.nch >= 4 ? 4 : 1;
    const long long tile_out = kArbThreads;
    long long max_span = 0;
    if (n_idx_host) {
        for (long long k0 = 0; k0 < a.n_out; k0 += tile_out) {
            const long long kl = std::min<long long>(k0 + tile_out, a.n_out) - 1;
            max_span = std::max<long long>(max_span, static_cast<long long>(n_idx_host[kl]) - n_idx_host[k0] + a.T);
        }
    } else {
        if (!spans) return false;
        max_span = static_cast<long long>(spans[0]) + a.T;     // tile_out == 256
    }
    const size_t total = taps_bytes + static_cast<size_t>(max_span) * sb * cpl;
    if (total > 150 * 1024) return false;
    ArbTileArgs ta{};
    ta.cpl = cpl;
    ta.tap_pitch = in_regs ? 1 : 0;                   // (FIRFarrow has no tap bank: the field says where the taps live)
    ta.x_offset_bytes = static_cast<int>(taps_bytes);
    ta.max_span = static_cast<int>(max_span);
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
    *kname = "farrow_tiled_kernel";
    if (!tk.x_f64 && !tk.r_f64) return tk.complex_x ? launch_farrow_t<float, float, 2>(fused, a, ta, lds, s, num_cus) : launch_farrow_t<float, float, 1>(fused, a, ta, lds, s, num_cus);
    if (!tk.x_f64 && tk.r_f64) return tk.complex_x ? launch_farrow_t<float, double, 2>(fused, a, ta, lds, s, num_cus) : launch_farrow_t<float, double, 1>(fused, a, ta, lds, s, num_cus);
    if (tk.x_f64 && tk.r_f64) return tk.complex_x ? launch_farrow_t<double, double, 2>(fused, a, ta, lds, s, num_cus) : launch_farrow_t<double, double, 1>(fused, a, ta, lds, s, num_cus);
    return hipErrorInvalidValue;
}

}  // namespace mrhip
