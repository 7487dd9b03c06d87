// kernels_poly_tiled.hip -- LDS-tiled kernel for the rational family with ANY tapsPerPhi: what the
// register-resident kernels (tapsPerPhi <= 32; L = 1: hLen <= 512) do not take -- long FIRRational /
// FIRInterpolator filters, L > 512 phases, very long FIRStandard / FIRDecimator filters -- used to fall to the
// one-thread-per-output kernel that fetches every tap and every sample through L1 (0.3 TB/s on
// 2//3 with 72 taps).
//
//     y_k = sum_{i=0}^{T-1} pfb[i, phi_k] * ext[n_k - T + i],   u = u0 + k*M, phi_k = u mod L, n_k = d0 + u div L
// (src/Filters.jl:462-468, :505-512, :558-569, :613-625; dot: src/support.jl:5-55).
//
// Same structure as arb_tiled_kernel: a persistent workgroup keeps the polyphase bank in LDS (column pitch
// T+1 elements, so lanes with different phases land on different banks) and, per tile of consecutive outputs,
// stages the contiguous run of samples those outputs touch ([history ; x] seam included) for CPL channels;
// one lane = one output index of CPL channels, so a tap read feeds CPL dot products (1 + 1/CPL LDS reads
// per multiply-add instead of 2).
//
// Arithmetic: identical to poly_generic_kernel (STRICT / FUSED, zero-start quirk of support.jl:46 included)
// => bit-identical results.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "mrhip_internal.h"

#pragma clang fp contract(off)

namespace mrhip {
namespace {

constexpr int kTiledThreads = 256;

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
__global__ __launch_bounds__(kTiledThreads) void poly_tiled_kernel(PolyArgs a, ArbTileArgs ta)
{
    struct alignas(sizeof(TX) * NC) Sample { TX c[NC]; };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    R *const lpfb = reinterpret_cast<R *>(smem);
    Sample *const lx = reinterpret_cast<Sample *>(smem + ta.x_offset_bytes);

    const int tid = threadIdx.x;
    const int T = a.T, TP = ta.tap_pitch;
    {   // the tap bank -> LDS once per workgroup: element (phi, i) at phi*TP + i
        const R *__restrict__ g0 = static_cast<const R *>(a.taps);
        const int total = a.L * T;
        for (int e = tid; e < total; e += kTiledThreads) {
            const int phi = e / T, i = e - phi * T;
            lpfb[phi * TP + i] = g0[e];
        }
    }
    auto newest_of = [&](long long k, int *phi) -> long long {     // 1-based index of the newest sample of output k
        const long long u = a.u0 + k * a.M;
        const long long q = u / a.L;
        *phi = static_cast<int>(u - q * a.L);
        return a.d0 + q;
    };

    for (long long tile = blockIdx.x; tile < ta.total_tiles; tile += gridDim.x) {
        const int cg = static_cast<int>(tile / ta.tiles_per_channel);                  // channel group
        const long long tau = tile - static_cast<long long>(cg) * ta.tiles_per_channel;
        const int ch0 = cg * CPL;
        const int nchl = a.nch - ch0 < CPL ? a.nch - ch0 : CPL;
        const long long k0 = tau * ta.tile_out;
        const long long klast = (k0 + ta.tile_out < a.n_out ? k0 + ta.tile_out : a.n_out) - 1;
        int phi_unused;
        const long long n_lo = newest_of(k0, &phi_unused), n_hi = newest_of(klast, &phi_unused);
        const long long o = n_lo - T;                                                   // 0-based x index of LDS sample 0 (may be < 0)
        const int span = static_cast<int>(n_hi - n_lo) + T;

        __syncthreads();   // previous tile's reads are done (and, first time, the tap bank is written)
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) {
            if (cc < nchl) {
                const Sample *__restrict__ xc = static_cast<const Sample *>(a.x) + static_cast<long long>(ch0 + cc) * a.x_stride;
                const Sample *__restrict__ hc = static_cast<const Sample *>(a.hist) + static_cast<long long>(ch0 + cc) * a.H;
                Sample *const lxc = lx + static_cast<size_t>(cc) * ta.max_span;
                for (int s = tid; s < span; s += kTiledThreads) {
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

        for (long long k = k0 + tid; k <= klast; k += kTiledThreads) {
            int phi;
            const long long n = newest_of(k, &phi);
            const R *tp = lpfb + phi * TP;
            const Sample *wp = lx + (n - n_lo);             // oldest sample of this output's window (channel 0 of the group)
            R acc[CPL][NC];
            {
                const R t = tp[0];
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) {
                    const Sample v = wp[static_cast<size_t>(cc) * ta.max_span];
#pragma unroll
                    for (int c = 0; c < NC; ++c) acc[cc][c] = t * static_cast<R>(v.c[c]);
                }
            }
            if (n < a.zero_start_below) {                   // support.jl:46 (see poly_generic_kernel)
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc)
#pragma unroll
                    for (int c = 0; c < NC; ++c) acc[cc][c] = static_cast<R>(0) + acc[cc][c];
            }
#pragma unroll 4
            for (int i = 1; i < T; ++i) {
                const R t = tp[i];
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) {
                    const Sample v = wp[static_cast<size_t>(cc) * ta.max_span + i];
#pragma unroll
                    for (int c = 0; c < NC; ++c) acc[cc][c] = mac<R, FUSED>(t, static_cast<R>(v.c[c]), acc[cc][c]);
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
}

template <typename TX, typename R, int NC>
hipError_t launch_tiled(bool fused, const PolyArgs &a, const ArbTileArgs &ta, size_t lds, hipStream_t s, int num_cus)
{
    auto go = [&](auto kfn) -> hipError_t {
        int per_cu = 0;
        hipError_t eo = occupancy_cached(reinterpret_cast<const void *>(kfn), kTiledThreads, lds, &per_cu);
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
            std::fprintf(stderr, "[mrhip] poly_tiled T=%d L=%d M=%d grid=%lld lds=%zu occ/CU=%d regs=%d cpl=%d tile_out=%lld max_span=%d tiles=%lld\n",
                         a.T, a.L, a.M, g, lds, per_cu, fa.numRegs, ta.cpl, ta.tile_out, ta.max_span, ta.total_tiles);
        }
        launch_kernel(kfn, dim3(static_cast<unsigned>(g)), dim3(kTiledThreads), lds, s, a, ta);
        return hipGetLastError();
    };
    switch (ta.cpl) {
    case 4: return fused ? go(poly_tiled_kernel<TX, R, NC, true, 4>) : go(poly_tiled_kernel<TX, R, NC, false, 4>);
    case 2: return fused ? go(poly_tiled_kernel<TX, R, NC, true, 2>) : go(poly_tiled_kernel<TX, R, NC, false, 2>);
    default: return fused ? go(poly_tiled_kernel<TX, R, NC, true, 1>) : go(poly_tiled_kernel<TX, R, NC, false, 1>);
    }
}

}  // namespace

// Returns false when the tap bank plus a useful sample tile do not fit LDS (caller uses the generic kernel).
bool plan_poly_tiled(const TypeKey &tk, const PolyArgs &a, int num_cus, ArbTileArgs *out, size_t *lds)
{
    static const int enabled = [] { const char *v = std::getenv("MRHIP_POLY_TILED"); return !(v && v[0] == '0'); }();
    if (!enabled || a.n_out < 1 || a.T < 1) return false;
    const size_t rs = tk.r_f64 ? 8 : 4;
    const size_t sb = (tk.x_f64 ? 8 : 4) * (tk.complex_x ? 2 : 1);
    const int TP = a.T + 1;
    const size_t bank_elems = static_cast<size_t>(a.L) * TP;
    const size_t bank_bytes = (bank_elems * rs + 15) / 16 * 16;
    if (bank_bytes > 96 * 1024) return false;
    static const int env_cpl = [] { const char *v = std::getenv("MRHIP_TILED_CPL"); return v && *v ? std::atoi(v) : 0; }();
    static const int env_tile = [] { const char *v = std::getenv("MRHIP_TILED_TILE"); return v && *v ? std::atoi(v) : 0; }();
    int cpl = a.nch >= 32 ? 4 : (a.nch >= 8 ? 2 : 1);
    if (env_cpl == 1 || env_cpl == 2 || env_cpl == 4) cpl = env_cpl;
    long long tile_out = env_tile >= 64 ? env_tile / 64 * 64 : 256;
    // samples a tile of `t` outputs can touch: floor((u_first + (t-1)*M)/L) - floor(u_first/L) + T
    auto span_of = [&](long long t) { return ((t - 1) * a.M + a.L - 1) / a.L + a.T + 1; };
    // two or more workgroups per CU when the tap bank leaves room for it; a bank that takes most of that keeps full tiles
    // and one workgroup per CU instead of shrinking the tile to a quarter of the workgroup (147//160 with 128 taps per
    // phase: 76 KB of taps)
    const size_t budget = std::max<size_t>(64 * 1024, std::min<size_t>(bank_bytes + 40 * 1024, 150 * 1024));
    for (;;) {
        const long long max_span = span_of(tile_out);
        const size_t total = bank_bytes + static_cast<size_t>(max_span) * sb * cpl;
        if (total <= budget || (cpl == 1 && tile_out == 64)) {
            if (total > 150 * 1024 || max_span > (1 << 30)) return false;
            const long long groups = (a.nch + cpl - 1) / cpl;
            ArbTileArgs ta{};
            ta.tap_pitch = TP;
            ta.bank_elems = static_cast<int>(bank_elems);
            ta.x_offset_bytes = static_cast<int>(bank_bytes);
            ta.max_span = static_cast<int>(max_span);
            ta.tile_out = tile_out;
            ta.tiles_per_channel = (a.n_out + tile_out - 1) / tile_out;
            ta.total_tiles = ta.tiles_per_channel * groups;
            ta.cpl = cpl;
            (void)num_cus;
            *out = ta;
            *lds = total;
            return true;
        }
        if (cpl > 1) cpl /= 2;
        else tile_out /= 2;
    }
}

hipError_t launch_poly_tiled(const TypeKey &tk, bool fused, const PolyArgs &a, const ArbTileArgs &ta, size_t lds, hipStream_t s,
                             const char **kname, int num_cus)
{
    *kname = "poly_tiled_kernel";
    if (!tk.x_f64 && !tk.r_f64) return tk.complex_x ? launch_tiled<float, float, 2>(fused, a, ta, lds, s, num_cus) : launch_tiled<float, float, 1>(fused, a, ta, lds, s, num_cus);
    if (!tk.x_f64 && tk.r_f64) return tk.complex_x ? launch_tiled<float, double, 2>(fused, a, ta, lds, s, num_cus) : launch_tiled<float, double, 1>(fused, a, ta, lds, s, num_cus);
    if (tk.x_f64 && tk.r_f64) return tk.complex_x ? launch_tiled<double, double, 2>(fused, a, ta, lds, s, num_cus) : launch_tiled<double, double, 1>(fused, a, ta, lds, s, num_cus);
    return hipErrorInvalidValue;
}

}  // namespace mrhip
