// kernels_generic.hip -- universal gfx950 kernels for the five FIR kernel kinds.
//
// One thread per output sample, every output independent (closed-form output -> (phase, input
// index) map).  These kernels accept ANY (L, M, tapsPerPhi, hLen) and every
// dtype combination; the tuned kernels in kernels_rational_tiled.hip cover the throughput
// configurations and fall back to these.
//
// Arithmetic contract (include/multirate_hip.h, mrhip_numerics): the dot product visits the
// logical window [history ; x] oldest sample first, the first product initialises the
// accumulator, and in STRICT mode every multiply and add is separately rounded in
// R = promote_type(Th, Tx) -- the order the reference source states (src/support.jl:5-55).
// This file is compiled with -ffp-contract=off; FUSED mode calls fma explicitly.
#include "mrhip_internal.h"
#include "pair_device.h"

#pragma clang fp contract(off)

namespace mrhip {
namespace {

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

// sample `xi` (0-based, may be negative => history) of channel-local pointers, component c
template <typename TX, int NC>
__device__ __forceinline__ void load_sample(const TX *__restrict__ xc, const TX *__restrict__ hc, int H,
                                            long long xi, TX (&v)[NC])
{
    const TX *p = xi >= 0 ? xc + xi * NC : hc + (static_cast<long long>(H) + xi) * NC;
#pragma unroll
    for (int c = 0; c < NC; ++c) v[c] = p[c];
}

// Rational family: y_k = sum_{i=0}^{T-1} taps[phi][i] * ext[n - T + i],  ext = [history ; x]
//   u = u0 + k*M ; phi = u mod L ; n = d0 + u div L (1-based index of the newest sample)
// reference loops: src/Filters.jl:462-468 (Standard), :505-512 (Interpolator),
// :558-569 (Rational), :613-625 (Decimator); dot: src/support.jl:5-55.
template <typename TX, typename R, int NC, bool FUSED>
__global__ __launch_bounds__(256) void poly_generic_kernel(PolyArgs a)
{
    const long long k = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (a.dyn) { a.n_out = a.dyn->n_out; a.u0 = a.dyn->u0; a.d0 = a.dyn->d0; }       // a device-planned call (mrhip_internal.h: DevCall)
    else if (a.rec && k == 0 && blockIdx.y == 0) {                                    // the host planned it: file the end state
        a.rec->phiIdx = a.phi_end; a.rec->inputDeficit = a.d_end; a.rec->n_written = a.n_out; a.rec->calls += 1;
    }
    if (k >= a.n_out) return;
    const R *__restrict__ taps = static_cast<const R *>(a.taps);
    for (int ch = blockIdx.y; ch < a.nch; ch += gridDim.y) {
        const TX *__restrict__ xc = static_cast<const TX *>(a.x) + static_cast<long long>(ch) * a.x_stride * NC;
        const TX *__restrict__ hc = static_cast<const TX *>(a.hist) + static_cast<long long>(ch) * a.H * NC;
        R *__restrict__ yc = static_cast<R *>(a.y) + static_cast<long long>(ch) * a.y_stride * NC;

        const long long u = a.u0 + k * a.M;
        const long long q = u / a.L;
        const int phi = static_cast<int>(u - q * a.L);
        const long long n = a.d0 + q;              // 1-based newest-sample index
        const R *__restrict__ tp = taps + static_cast<long long>(phi) * a.T;
        const long long base = n - a.T;            // 0-based index of the oldest sample

        TX v[NC];
        R acc[NC];
        load_sample<TX, NC>(xc, hc, a.H, base, v);
        const R t0 = tp[0];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = t0 * static_cast<R>(v[c]);
        if (n < a.zero_start_below) {
            // src/support.jl:46: the Vector seam variant starts from zero(...) and adds the first
            // product; differs from the other three only for a -0.0 first product.
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = static_cast<R>(0) + acc[c];
        }
        for (int i = 1; i < a.T; ++i) {
            load_sample<TX, NC>(xc, hc, a.H, base + i, v);
            const R t = tp[i];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = mac<R, FUSED>(t, static_cast<R>(v[c]), acc[c]);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) yc[k * NC + c] = acc[c];
    }
}

// FIRArbitrary: two dots over one window, then yLower + yUpper*alpha in Float64, rounded once to
// the output type (src/Filters.jl:717-732; alpha is Float64 in the reference, so the combine
// promotes).  (n, acc) per output come from the host-evaluated phase recurrence.
template <typename TX, typename R, int NC, bool FUSED>
__global__ __launch_bounds__(256) void arb_generic_kernel(ArbArgs a)
{
    const long long k = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (a.dyn) a.n_out = a.dyn->n_out;              // a device-planned call: the count the schedule's FINISH kernel left
    if (k < a.n_out) {                               // (no early return: every thread takes part in the history epilogue below)
    const long long n = a.n_idx[k];
    const double pacc = a.acc[k];
    const double phif = __builtin_floor(pacc);
    const double alpha = pacc - phif;               // src/Filters.jl:671-672
    int phi = static_cast<int>(phif) - 1;           // 0-based column
    const R *__restrict__ tp = static_cast<const R *>(a.taps) + static_cast<long long>(phi) * a.T;
    const R *__restrict__ dp = static_cast<const R *>(a.dtaps) + static_cast<long long>(phi) * a.T;
    const long long base = n - a.T;
    for (int ch = blockIdx.y; ch < a.nch; ch += gridDim.y) {
        const TX *__restrict__ xc = static_cast<const TX *>(a.x) + static_cast<long long>(ch) * a.x_stride * NC;
        const TX *__restrict__ hc = static_cast<const TX *>(a.hist) + static_cast<long long>(ch) * a.H * NC;
        R *__restrict__ yc = static_cast<R *>(a.y) + static_cast<long long>(ch) * a.y_stride * NC;
        TX v[NC];
        R lo[NC], up[NC];
        load_sample<TX, NC>(xc, hc, a.H, base, v);
        {
            const R t = tp[0], d = dp[0];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                lo[c] = t * static_cast<R>(v[c]);
                up[c] = d * static_cast<R>(v[c]);
            }
        }
        for (int i = 1; i < a.T; ++i) {
            load_sample<TX, NC>(xc, hc, a.H, base + i, v);
            const R t = tp[i], d = dp[i];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                lo[c] = mac<R, FUSED>(t, static_cast<R>(v[c]), lo[c]);
                up[c] = mac<R, FUSED>(d, static_cast<R>(v[c]), up[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double prod = static_cast<double>(up[c]) * alpha;
            const double sum = static_cast<double>(lo[c]) + prod;
            yc[k * NC + c] = static_cast<R>(sum);
        }
    }
    }
    dev::shiftin_by_last_workgroup<TX, NC>(a.fold, a.x, a.hist, a.x_stride, a.x_len, a.H, a.nch);
}

// FIRFarrow (src/Filters.jl:123-147, 764-839): the taps of every output are polynomials in the Float64
// phase 𝜙Idx evaluated by Horner in Float64 (Polynomials.jl polyval: y = p[end]; y = p[i] + x*y), stored
// into currentTaps::Vector{Th} (rounded to Th), then one Vector unsafedot (support.jl:33-55, start from
// zero on the seam, :46).  One thread per output: it evaluates its tapsPerPhi taps ONCE into an LDS
// column (the taps depend on the output, not on the channel) and reuses them for every channel it
// visits; CACHE = false recomputes them per channel when the bank does not fit LDS.  (n, phase) per
// output come from the host-evaluated recurrence of update() (:780-788), shared with FIRArbitrary.
template <typename TX, typename R, int NC, bool FUSED, bool CACHE>
__global__ __launch_bounds__(256) void farrow_kernel(FarrowArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char farrow_smem[];
    R *const tl = reinterpret_cast<R *>(farrow_smem);
    const int tid = threadIdx.x, bs = blockDim.x;
    const long long k = static_cast<long long>(blockIdx.x) * bs + tid;
    if (a.dyn) a.n_out = a.dyn->n_out;             // a device-planned call
    if (k >= a.n_out) return;                      // no barriers below: each thread only reads its own LDS column
    const long long n = a.n_idx[k];
    const double phase = a.acc[k];
    const int P = a.polyorder;
    auto tap = [&](int i) -> R {
        const double *__restrict__ c = a.pnfb + static_cast<long long>(i) * (P + 1);
        double yv = c[P];
        for (int j = P - 1; j >= 0; --j) { const double t = phase * yv; yv = c[j] + t; }
        return a.tap_f32 ? static_cast<R>(static_cast<float>(yv)) : static_cast<R>(yv);
    };
    if constexpr (CACHE)
        for (int i = 0; i < a.T; ++i) tl[i * bs + tid] = tap(i);
    const long long base = n - a.T;
    const bool seam = n < a.seam_below;            // kernel.xIdx < kernel.tapsPer𝜙, Filters.jl:818 (never in a piece that continues a call)
    for (int ch = blockIdx.y; ch < a.nch; ch += gridDim.y) {
        const TX *__restrict__ xc = static_cast<const TX *>(a.x) + static_cast<long long>(ch) * a.x_stride * NC;
        const TX *__restrict__ hc = static_cast<const TX *>(a.hist) + static_cast<long long>(ch) * a.H * NC;
        R *__restrict__ yc = static_cast<R *>(a.y) + static_cast<long long>(ch) * a.y_stride * NC;
        TX v[NC];
        R acc[NC];
        load_sample<TX, NC>(xc, hc, a.H, base, v);
        const R t0 = CACHE ? tl[tid] : tap(0);
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = t0 * static_cast<R>(v[c]);
        if (seam) {
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = static_cast<R>(0) + acc[c];
        }
        for (int i = 1; i < a.T; ++i) {
            load_sample<TX, NC>(xc, hc, a.H, base + i, v);
            const R t = CACHE ? tl[i * bs + tid] : tap(i);
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = mac<R, FUSED>(t, static_cast<R>(v[c]), acc[c]);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) yc[k * NC + c] = acc[c];
    }
}

// shiftin!: hist_new <- last H samples of [hist_old ; x]   (src/support.jl:61-80)
template <typename TX, int NC>
__global__ __launch_bounds__(256) void shiftin_kernel(HistArgs a)
{
    const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    const long long total = static_cast<long long>(a.nch) * a.H;
    if (t >= total) return;
    const int ch = static_cast<int>(t / a.H);
    const int i = static_cast<int>(t - static_cast<long long>(ch) * a.H);
    const long long x_len = a.dyn ? a.dyn->x_len : a.x_len;
    const long long e = static_cast<long long>(i) + x_len;     // index into [hist_old ; x]
    const TX *src = e < a.H
        ? static_cast<const TX *>(a.hist_old) + (static_cast<long long>(ch) * a.H + e) * NC
        : static_cast<const TX *>(a.x) + (static_cast<long long>(ch) * a.x_stride + (e - a.H)) * NC;
    TX *dst = static_cast<TX *>(a.hist_new) + t * NC;
#pragma unroll
    for (int c = 0; c < NC; ++c) dst[c] = src[c];
}

template <typename F>
hipError_t dispatch_types(const TypeKey &tk, F &&f)
{
    // (Tx scalar, R) combinations that promote_type can produce: (f32,f32) (f32,f64) (f64,f64)
    if (!tk.x_f64 && !tk.r_f64) return tk.complex_x ? f.template operator()<float, float, 2>() : f.template operator()<float, float, 1>();
    if (!tk.x_f64 && tk.r_f64) return tk.complex_x ? f.template operator()<float, double, 2>() : f.template operator()<float, double, 1>();
    if (tk.x_f64 && tk.r_f64) return tk.complex_x ? f.template operator()<double, double, 2>() : f.template operator()<double, double, 1>();
    return hipErrorInvalidValue;
}

inline dim3 grid_for(long long n_out, int nch)
{
    const long long bx = n_out > 0 ? (n_out + 255) / 256 : 1;
    return dim3(static_cast<unsigned>(bx), static_cast<unsigned>(nch < 65535 ? nch : 65535), 1);
}

}  // namespace

hipError_t launch_poly_generic(const TypeKey &tk, bool fused, const PolyArgs &a, hipStream_t s, const char **kname)
{
    if (a.n_out <= 0 && !a.dyn) return hipSuccess;
    if ((a.n_out + 255) / 256 > 0x7fffffffLL) return hipErrorInvalidValue;
    *kname = "poly_generic_kernel";
    return dispatch_types(tk, [&]<typename TX, typename R, int NC>() -> hipError_t {
        if (fused) launch_kernel(poly_generic_kernel<TX, R, NC, true>, grid_for(a.n_out, a.nch), dim3(256), 0, s, a);
        else launch_kernel(poly_generic_kernel<TX, R, NC, false>, grid_for(a.n_out, a.nch), dim3(256), 0, s, a);
        return hipGetLastError();
    });
}

hipError_t launch_arb_generic(const TypeKey &tk, bool fused, const ArbArgs &a, hipStream_t s, const char **kname)
{
    if (a.n_out <= 0 && !a.dyn) return hipSuccess;
    if ((a.n_out + 255) / 256 > 0x7fffffffLL) return hipErrorInvalidValue;
    *kname = "arb_generic_kernel";
    return dispatch_types(tk, [&]<typename TX, typename R, int NC>() -> hipError_t {
        if (fused) launch_kernel(arb_generic_kernel<TX, R, NC, true>, grid_for(a.n_out, a.nch), dim3(256), 0, s, a);
        else launch_kernel(arb_generic_kernel<TX, R, NC, false>, grid_for(a.n_out, a.nch), dim3(256), 0, s, a);
        return hipGetLastError();
    });
}

hipError_t launch_farrow(const TypeKey &tk, bool fused, const FarrowArgs &a, hipStream_t s, const char **kname)
{
    if (a.n_out <= 0 && !a.dyn) return hipSuccess;
    *kname = "farrow_kernel";
    return dispatch_types(tk, [&]<typename TX, typename R, int NC>() -> hipError_t {
        // block size: as many outputs as keep one LDS column of T taps per thread within 64 KiB
        const long long per_thread = static_cast<long long>(a.T) * static_cast<long long>(sizeof(R));
        int bs = static_cast<int>(65536 / per_thread) / 64 * 64;
        const bool cache = bs >= 64;
        if (bs > 256) bs = 256;
        if (!cache) bs = 256;
        const size_t lds = cache ? static_cast<size_t>(per_thread) * bs : 0;
        const long long bx = std::max<long long>((a.n_out + bs - 1) / bs, 1);
        if (bx > 0x7fffffffLL) return hipErrorInvalidValue;
        // channels are split over blockIdx.y only as far as needed to fill the machine: the taps are
        // evaluated once per (output, blockIdx.y)
        long long by = 1;
        while (by < a.nch && bx * by < 2048) by *= 2;
        if (by > a.nch) by = a.nch;
        const dim3 grid(static_cast<unsigned>(bx), static_cast<unsigned>(by), 1);
#define MRHIP_FARROW_LAUNCH(F, C)                                                                         \
        {                                                                                                 \
            auto kfn = farrow_kernel<TX, R, NC, F, C>;                                                    \
            if (lds > 48 * 1024) {                                                                        \
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),                   \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)); \
                if (e != hipSuccess) return e;                                                            \
            }                                                                                             \
            launch_kernel(kfn, grid, dim3(bs), lds, s, a);                                           \
        }
        if (fused) { if (cache) MRHIP_FARROW_LAUNCH(true, true) else MRHIP_FARROW_LAUNCH(true, false) }
        else { if (cache) MRHIP_FARROW_LAUNCH(false, true) else MRHIP_FARROW_LAUNCH(false, false) }
#undef MRHIP_FARROW_LAUNCH
        return hipGetLastError();
    });
}

hipError_t launch_shiftin(const TypeKey &tk, const HistArgs &a, hipStream_t s)
{
    const long long total = static_cast<long long>(a.nch) * a.H;
    if (total <= 0) return hipSuccess;
    const dim3 grid(static_cast<unsigned>((total + 255) / 256));
    if (tk.x_f64) {
        if (tk.complex_x) hipLaunchKernelGGL((shiftin_kernel<double, 2>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((shiftin_kernel<double, 1>), grid, dim3(256), 0, s, a);
    } else {
        if (tk.complex_x) hipLaunchKernelGGL((shiftin_kernel<float, 2>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((shiftin_kernel<float, 1>), grid, dim3(256), 0, s, a);
    }
    return hipGetLastError();
}

}  // namespace mrhip
