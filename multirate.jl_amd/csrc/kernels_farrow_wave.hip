// kernels_farrow_wave.hip -- FIRFarrow (src/Filters.jl:764-846) for FEW channels: the shape of the reference's own
// FIRArbitrary-vs-FIRFarrow benchmark (examples/Arb-Farrow Speed Comparison.jl:38-54: ONE channel of 1e7 samples).
//
// farrow_pipe_kernel stages the samples of 256 outputs in LDS behind a barrier, one tile ahead: with 64 channels the
// ~3 us a tile waits for its staging hide behind 16 channel groups of dot products; with one channel they are the tile
// (0.21 ms per 1e7 samples, three times FIRArbitrary on the same shape, measured per polynomial degree and per channel
// count in scripts/exp_farrow_1ch.py: the time follows the number of TILES, not the arithmetic).  With few channels there is
// nothing to share between lanes but the input itself, and that the caches do: here a lane owns one output index, evaluates
// its taps (Float64 Horner, the degree in the outer loop: tapsPerPhi independent chains; coefficients by scalar loads: they
// are wave-uniform) and reads its window straight from global memory -- for a fixed tap the 64 lanes of a wave read 64
// (nearly) consecutive samples: one coalesced load, served by the L1 after the first tap.  No LDS, no barrier, no tile: a
// wave is its own pipeline and the CU hides latency with 6-8 of them per SIMD.
//
// Arithmetic: exactly farrow_kernel's (kernels_generic.hip) -- oldest sample first, first product initialises the
// accumulator (from +0 on the seam, support.jl:46), separately rounded multiply and add unless FUSED; bit-identical.
#include <algorithm>
#include <cstdio>

#include "mrhip_internal.h"
#include "pair_device.h"

#pragma clang fp contract(off)

namespace mrhip {
namespace {

constexpr int kFwThreads = 256;

template <typename R, bool FUSED>
__device__ __forceinline__ R fw_mac(R t, R x, R acc)
{
    if constexpr (FUSED) {
        if constexpr (sizeof(R) == 4) return __builtin_fmaf(t, x, acc);
        else return __builtin_fma(t, x, acc);
    } else {
        const R p = t * x;
        return acc + p;
    }
}

// TREG = tap slots (tapsPerPhi rounded up to 4, 8, 12, 16, 24 or 32).  One straight-line instruction stream whatever
// tapsPerPhi is: a slot past the last tap works on a zero row of the (padded, degree-major: a.pnfb is [polyorder+1][32] here)
// coefficient table and on the sample behind the window, and its product is replaced by -0.0 -- x + (-0.0) == x for every x,
// signed zeros, NaN and infinities included -- so the lanes perform exactly the reference's operations in the reference's
// order (a zero TAP alone would not do: 0 * Inf = NaN, and -0.0 + +0.0 = +0.0).  All addresses are one scalar base plus an
// immediate.  (The first form guarded every slot with `if (i < T)`: a branch, a vmcnt(0) and SGPR spills per tap.)
template <typename TX, typename R, int NC, bool FUSED, int TREG>
__global__ __launch_bounds__(kFwThreads) void farrow_wave_kernel(FarrowArgs a)
{
    // wave-uniform reads through the constant address space: scalar loads (their own queue: they do not wait behind the
    // wave's outstanding vector loads, which return in order) with immediate offsets
    typedef const __attribute__((address_space(4))) double *coef_t;
    const coef_t coef = (coef_t)(a.pnfb);
    const long long n_out = a.dyn ? a.dyn->n_out : a.n_out;              // (a device-planned call: the count from the call record)
    const int T = a.T, P = a.polyorder;
    const long long step = static_cast<long long>(gridDim.x) * kFwThreads;
    const int lane = threadIdx.x & 63;
    long long kw = static_cast<long long>(blockIdx.x) * kFwThreads + (threadIdx.x & ~63);   // this wave's first output
    const bool work = kw < n_out;                                        // (a wave without outputs still takes part in the epilogue below)
    constexpr bool EARLY = TREG <= 16;                                   // channel 0's samples are requested before the Horner steps
    constexpr int DEAD = TREG <= 16 ? 3 : 7;                             // slot classes 4, 8, 12, 16 | 24, 32 (launch_fw_t)
    // (the entry's index stays the 32-bit value the load delivers until the top of the iteration that uses it: widened right
    //  behind the prefetch it made the wave wait there -- vmcnt is in order -- for the twelve sample loads issued before it,
    //  i.e. BEFORE the Horner steps those loads are meant to hide behind)
    auto entry = [&](long long kw_, int *n_, double *ph_) {              // schedule entry of this lane's output (idle lanes of the
        long long kk = kw_ + lane;                                       // last wave repeat its last output)
        if (kk >= n_out) kk = n_out - 1;
        *n_ = a.n_idx[kk];
        *ph_ = a.acc[kk];
    };
    int n32 = 0;
    double phase = 0.0;
    if (work) entry(kw, &n32, &phase);
    while (work) {                                                       // (wave-uniform trip count)
        const long long n = n32;
        const long long k = kw + lane;
        const bool have = k < n_out;
        const long long kw_next = kw + step;
        const bool more = kw_next < n_out;
        int n_next = n32;
        double phase_next = phase;
        const bool seam = n < a.seam_below;                              // kernel.xIdx < kernel.tapsPer𝜙, Filters.jl:818
        const long long base = n - T;                                    // 0-based index of the oldest sample (negative: history)
        // the whole wave inside the signal, slots past the window included: no per-sample checks (everything but the first
        // and the last outputs of a call)
        const bool inside = __all(base >= 0 && base + TREG <= a.x_len) != 0;
        TX v[TREG][NC];
        auto load_samples = [&](int ch) {
            const TX *__restrict__ xc = static_cast<const TX *>(a.x) + static_cast<long long>(ch) * a.x_stride * NC;
            const TX *__restrict__ hc = static_cast<const TX *>(a.hist) + static_cast<long long>(ch) * a.H * NC;
            if (inside) {
                const TX *__restrict__ p = xc + base * NC;
#pragma unroll
                for (int i = 0; i < TREG; ++i) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) v[i][c] = p[i * NC + c];
                }
            } else {
#pragma unroll
                for (int i = 0; i < TREG; ++i) {
                    long long xi = base + i;
                    if (xi > n - 1) xi = n - 1;                          // a slot past the window: the window's last sample again
                    const TX *p = xi >= 0 ? xc + xi * NC : hc + (static_cast<long long>(a.H) + xi) * NC;
#pragma unroll
                    for (int c = 0; c < NC; ++c) v[i][c] = p[c];
                }
                // (keeps the two paths' loads apart: merged at the join, the common path computed a 64-bit address per load for
                //  them instead of one base with immediate offsets)
                asm volatile("; window at the signal's ends");
            }
        };
        if constexpr (EARLY) load_samples(0);                            // in flight during the Horner steps
        if (more) entry(kw_next, &n_next, &phase_next);                  // ... and the next output's schedule entry during everything
        // currentTaps[i] = polyval(pnfb[i], phase) stored into Vector{Th} (Filters.jl:790-792; Polynomials.jl polyval:
        // y = p[end]; y = p[i] + x*y): the degree in the outer loop, TREG independent chains inside
        double yv[TREG];
        {
            const coef_t cj = coef + P * 32;
#pragma unroll
            for (int i = 0; i < TREG; ++i) yv[i] = cj[i];
        }
        for (int j = P - 1; j >= 0; --j) {
            const coef_t cj = coef + j * 32;
#pragma unroll
            for (int i = 0; i < TREG; ++i) {
                const double t = phase * yv[i];
                yv[i] = cj[i] + t;
            }
        }
        R treg[TREG];
        if (a.tap_f32) {                                                 // (a wave-uniform BRANCH: as a select per tap it was two conversions
#pragma unroll                                                           //  and two selects per tap for every filter)
            for (int i = 0; i < TREG; ++i) treg[i] = static_cast<R>(static_cast<float>(yv[i]));
        } else {
#pragma unroll
            for (int i = 0; i < TREG; ++i) treg[i] = static_cast<R>(yv[i]);
        }
        for (int ch = 0; ch < a.nch; ++ch) {
            if (!EARLY || ch > 0) load_samples(ch);
            R acc[NC];
#pragma unroll
            for (int i = 0; i < TREG; ++i) {
                // (wave-uniform; tapsPerPhi lies in the slot class: only its last DEAD slots can be past the window)
                const bool live = i < TREG - DEAD ? true : i < T;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const R x = static_cast<R>(v[i][c]);
                    if (i == 0) {
                        acc[c] = treg[0] * x;                            // the first product initialises the accumulator ...
                        if (seam) acc[c] = static_cast<R>(0) + acc[c];   // ... from zero on the seam (support.jl:46)
                    } else if constexpr (FUSED) {
                        const R r = fw_mac<R, true>(treg[i], x, acc[c]);
                        acc[c] = live ? r : acc[c];
                    } else {
                        R pr = treg[i] * x;
                        pr = live ? pr : static_cast<R>(-0.0);
                        acc[c] = acc[c] + pr;
                    }
                }
            }
            if (have) {
                R *__restrict__ yc = static_cast<R *>(a.y) + (static_cast<long long>(ch) * a.y_stride + k) * NC;
#pragma unroll
                for (int c = 0; c < NC; ++c) yc[c] = acc[c];
            }
        }
        if (!more) break;
        kw = kw_next; n32 = n_next; phase = phase_next;
    }
    dev::shiftin_by_last_workgroup<TX, NC>(a.fold, a.x, a.hist, a.x_stride, a.x_len, a.H, a.nch);
}

template <typename TX, typename R, int NC>
hipError_t launch_fw_t(bool fused, const FarrowArgs &a, hipStream_t s, int num_cus)
{
    auto go = [&](auto kfn) -> hipError_t {
        long long g = (std::max<long long>(a.n_out, 1) + kFwThreads - 1) / kFwThreads;
        int per_cu = 0;                                                  // exactly the workgroups the chip holds at once, grid-stride beyond:
        hipError_t eo = occupancy_cached(reinterpret_cast<const void *>(kfn), kFwThreads, 0, &per_cu);   // a second, part-filled round costs a fifth
        if (eo != hipSuccess) return eo;
        const long long cap = static_cast<long long>(num_cus) * std::max(per_cu, 1);
        if (g > cap) g = cap;
        if (MRHIP_ENV_INT("MRHIP_DEBUG", 0) == 1) {
            hipFuncAttributes fa;
            (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kfn));
            std::fprintf(stderr, "[mrhip] farrow_wave T=%d P=%d nch=%d grid=%lld regs=%d scratch=%zu\n", a.T, a.polyorder, a.nch, g, fa.numRegs, fa.localSizeBytes);
        }
        launch_kernel(kfn, dim3(static_cast<unsigned>(g)), dim3(kFwThreads), 0, s, a);
        return hipGetLastError();
    };
#define MRHIP_FW_GO(TR) return fused ? go(farrow_wave_kernel<TX, R, NC, true, TR>) : go(farrow_wave_kernel<TX, R, NC, false, TR>);
    if (a.T <= 4) { MRHIP_FW_GO(4) }
    if (a.T <= 8) { MRHIP_FW_GO(8) }
    if (a.T <= 12) { MRHIP_FW_GO(12) }
    if (a.T <= 16) { MRHIP_FW_GO(16) }
    if (a.T <= 24) { MRHIP_FW_GO(24) }
    MRHIP_FW_GO(32)
#undef MRHIP_FW_GO
}

}  // namespace

// Eligible: at most 16 taps per phase and at most 8 channels (measured, profiles/r04/farrow_wave_vs_pipe.txt, 1e7 samples in all:
// 10 taps, 1 / 4 / 8 channels 0.072 / 0.047 / 0.039 ms against farrow_pipe_kernel's 0.210 / 0.069 / 0.055; at 16 channels level, at 64
// the LDS tiles win 0.22 vs 0.34; with 32 taps -- samples requested after the Horner steps, 64 registers of taps -- the pipe kernel is
// ahead from one channel on: 0.283 vs 0.316).  MRHIP_FARROW_WAVE=0: off; MRHIP_FARROW_WAVE_MAXCH=n: at most n channels, any tap count
// up to 32 (tests run every slot class with it).
bool plan_farrow_wave(const FarrowArgs &a)
{
    if (MRHIP_ENV_INT("MRHIP_FARROW_WAVE", 1) == 0) return false;
    const int forced = MRHIP_ENV_INT("MRHIP_FARROW_WAVE_MAXCH", -1);
    const int maxch = forced >= 0 ? forced : (a.T <= 16 ? 8 : 0);
    return a.nch >= 1 && a.nch <= maxch && a.T >= 1 && a.T <= 32 && (a.n_out >= 1 || a.dyn);
}

hipError_t launch_farrow_wave(const TypeKey &tk, bool fused, const FarrowArgs &a, hipStream_t s, const char **kname, int num_cus)
{
    *kname = "farrow_wave_kernel";
    if (!tk.x_f64 && !tk.r_f64) return tk.complex_x ? launch_fw_t<float, float, 2>(fused, a, s, num_cus) : launch_fw_t<float, float, 1>(fused, a, s, num_cus);
    if (!tk.x_f64 && tk.r_f64) return tk.complex_x ? launch_fw_t<float, double, 2>(fused, a, s, num_cus) : launch_fw_t<float, double, 1>(fused, a, s, num_cus);
    if (tk.x_f64 && tk.r_f64) return tk.complex_x ? launch_fw_t<double, double, 2>(fused, a, s, num_cus) : launch_fw_t<double, double, 1>(fused, a, s, num_cus);
    return hipErrorInvalidValue;
}

}  // namespace mrhip
