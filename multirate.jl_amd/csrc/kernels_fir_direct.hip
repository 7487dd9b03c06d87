// kernels_fir_direct.hip -- gfx950 kernel for the single-column kernels FIRStandard (M = 1) and FIRDecimator (L = 1),
// tap counts up to 512, every dtype and every M: what kernels_fir_stream.hip (Float32 arithmetic, M in {1,2,4,8},
// 32..512 taps) does not take.  (Round 1's two-outputs-per-lane variant of this kernel is superseded by the streaming
// kernel and gone.)
//
// reference: src/Filters.jl:450-473 (Standard), :598-631 (Decimator); dot: src/support.jl:33-55.
//
// All outputs use the same (flipped) tap vector, so taps are wave-uniform.  They are kept PACKED in
// VGPRs -- lane l of register c holds tap 64c + l, loaded once per workgroup -- and tap i is broadcast
// with v_readlane_b32 into an SGPR that feeds the VALU directly: no memory latency in the tap loop,
// no LDS traffic for taps, ceil(T/64) registers for any T.  A lane owns
// one output per step; consecutive lanes own consecutive outputs, whose windows start M samples
// apart.  Reading sample i of every lane's window straight out of a linear LDS tile would be a
// stride-M access (a 4-way bank conflict for M = 4, ComplexF32), so the tile is stored TRANSPOSED by
// residue: sample s lives at row s mod M, column s div M.  Lane k then reads tap i at row i mod M,
// column k + i div M: consecutive lanes -> consecutive addresses, conflict-free for every M.
//
// Arithmetic: identical to the generic kernel (STRICT / FUSED as in multirate_hip.h), including the
// start-from-zero quirk of the Vector seam variant (support.jl:46) => bit-identical results.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "mrhip_internal.h"

#pragma clang fp contract(off)

namespace mrhip {
namespace {

constexpr int kDirectThreads = 256;

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

template <typename R>
__device__ __forceinline__ R bcast_lane(R v, int lane)
{
    if constexpr (sizeof(R) == 4) {
        return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane));
    } else {
        const unsigned long long u = __double_as_longlong(v);
        const unsigned lo = __builtin_amdgcn_readlane(static_cast<unsigned>(u), lane);
        const unsigned hi = __builtin_amdgcn_readlane(static_cast<unsigned>(u >> 32), lane);
        return __longlong_as_double((static_cast<unsigned long long>(hi) << 32) | lo);
    }
}

template <typename TX, typename R, int NC, bool FUSED, int NCH>
__global__ __launch_bounds__(kDirectThreads) void fir_direct_kernel(PolyArgs a, DirectArgs da)
{
    struct alignas(sizeof(TX) * NC) Sample { TX c[NC]; };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Sample *const lds = reinterpret_cast<Sample *>(smem);

    const int tid = threadIdx.x;
    const int T = a.T, M = a.M, QP = da.row_pitch;
    // Tap 0 initialises the accumulator and is kept as a scalar; taps 1..T-1 are packed: lane l of tv[c]
    // holds tap 1 + 64c + l, and lane l of ov[c] the LDS element offset of that tap's sample relative to
    // the lane's window origin in the transposed tile: (i mod M) * pitch + i div M.
    const R *__restrict__ taps_g = static_cast<const R *>(a.taps);
    const R tap0 = taps_g[0];
    R tv[NCH];
    int ov[NCH];
    {
        const int l = tid & 63;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int i = 1 + 64 * c + l;
            tv[c] = i < T ? taps_g[i] : static_cast<R>(0);
            ov[c] = (i % M) * QP + i / M;
        }
    }
    const int TR = T - 1;                                       // taps handled by the packed loops
    const long long tile_out = static_cast<long long>(da.J) * kDirectThreads;
    const long long tile_in = tile_out * M;

    for (long long tile = blockIdx.x; tile < da.total_tiles; tile += gridDim.x) {
        const int ch = static_cast<int>(tile / da.tiles_per_channel);
        const long long tau = tile - static_cast<long long>(ch) * da.tiles_per_channel;
        const Sample *__restrict__ xc = static_cast<const Sample *>(a.x) + static_cast<long long>(ch) * a.x_stride;
        const Sample *__restrict__ hc = static_cast<const Sample *>(a.hist) + static_cast<long long>(ch) * a.H;
        R *__restrict__ yc = static_cast<R *>(a.y) + (static_cast<long long>(ch) * a.y_stride + tau * tile_out) * NC;
        // x index (0-based) of tile sample 0 = oldest sample of the tile's first output
        const long long o = a.d0 - T + tau * tile_in;

        __syncthreads();   // previous tile's reads are done
        // stage tile samples [0, tile_len) transposed by residue mod M
        for (int s = tid; s < da.tile_len; s += kDirectThreads) {
            const long long gi = o + s;
            Sample v;
#pragma unroll
            for (int c = 0; c < NC; ++c) v.c[c] = static_cast<TX>(0);
            if (gi >= 0) { if (gi < a.x_len) v = xc[gi]; }
            else if (gi >= -static_cast<long long>(a.H)) v = hc[a.H + gi];
            const int q = s / M, r = s - q * M;
            lds[r * QP + q] = v;
        }
        __syncthreads();

        const long long remaining = a.n_out - tau * tile_out;
#pragma unroll 1
        for (int j = 0; j < da.J; ++j) {
            const int kl = j * kDirectThreads + tid;               // output index inside the tile
            if (kl >= remaining) break;
            const Sample *wp = lds + kl;
            R acc[NC];
            {
                const Sample v = wp[0];
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] = tap0 * static_cast<R>(v.c[c]);
                // 1-based newest-sample index of this output: n = d0 + (tau*tile_out + kl)*M
                if (a.d0 + (tau * tile_out + kl) * M < a.zero_start_below) {   // support.jl:46
#pragma unroll
                    for (int c = 0; c < NC; ++c) acc[c] = static_cast<R>(0) + acc[c];
                }
            }
            auto one_tap = [&](R tvec, int ovec, int ii) {
                const R t = bcast_lane<R>(tvec, ii);
                const int off = __builtin_amdgcn_readlane(ovec, ii);
                const Sample v = wp[off];
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] = mac<R, FUSED>(t, static_cast<R>(v.c[c]), acc[c]);
            };
#pragma unroll
            for (int cch = 0; cch < NCH; ++cch) {
                const int base = 64 * cch;
                if (base < TR) {                                 // wave-uniform
                    const int cnt = TR - base < 64 ? TR - base : 64;
                    const int ngroups = cnt >> 3;
                    for (int g = 0; g < ngroups; ++g) {
                        // 8 taps per trip: the eight LDS reads are independent of the arithmetic, so the
                        // compiler issues them ahead and overlaps their latency with the previous group
                        R t8[8];
                        Sample v8[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            t8[u] = bcast_lane<R>(tv[cch], 8 * g + u);
                            v8[u] = wp[__builtin_amdgcn_readlane(ov[cch], 8 * g + u)];
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
#pragma unroll
                            for (int c = 0; c < NC; ++c) acc[c] = mac<R, FUSED>(t8[u], static_cast<R>(v8[u].c[c]), acc[c]);
                        }
                    }
                    for (int ii = ngroups * 8; ii < cnt; ++ii) one_tap(tv[cch], ov[cch], ii);
                }
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) yc[static_cast<long long>(kl) * NC + c] = acc[c];
        }
    }
}


template <typename TX, typename R, int NC>
hipError_t launch_direct(bool fused, const PolyArgs &a, DirectArgs da, size_t lds, hipStream_t s, int num_cus)
{
    auto go = [&](auto kfn) -> hipError_t {
        int per_cu = 0;
        hipError_t eo = occupancy_cached(reinterpret_cast<const void *>(kfn), kDirectThreads, lds, &per_cu);
        if (eo != hipSuccess) return eo;
        if (per_cu < 1) per_cu = 1;
        long long g = static_cast<long long>(num_cus) * per_cu;
        if (g > da.total_tiles) g = da.total_tiles;
        if (g < 1) g = 1;
        static int dbg = -1;
        if (dbg < 0) { const char *v = std::getenv("MRHIP_DEBUG"); dbg = (v && v[0] == '1') ? 1 : 0; }
        if (dbg == 1) {
            dbg = 0;
            hipFuncAttributes fa;
            (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kfn));
            std::fprintf(stderr, "[mrhip] fir_direct T=%d M=%d grid=%lld lds=%zu occ/CU=%d regs=%d J=%d tile_len=%d pitch=%d tiles=%lld\n",
                         a.T, a.M, g, lds, per_cu, fa.numRegs, da.J, da.tile_len, da.row_pitch, da.total_tiles);
        }
        launch_kernel(kfn, dim3(static_cast<unsigned>(g)), dim3(kDirectThreads), lds, s, a, da);
        return hipGetLastError();
    };
    const int nch = (a.T + 63) / 64;
#define MRHIP_GO(N) (fused ? go(fir_direct_kernel<TX, R, NC, true, N>) : go(fir_direct_kernel<TX, R, NC, false, N>))
    if (nch <= 1) return MRHIP_GO(1);
    if (nch <= 2) return MRHIP_GO(2);
    if (nch <= 4) return MRHIP_GO(4);
    if (nch <= 8) return MRHIP_GO(8);
#undef MRHIP_GO
    return hipErrorInvalidValue;
}

}  // namespace

// Covers L == 1 (FIRStandard, FIRDecimator).  Returns false when the tile does not fit LDS.
bool plan_fir_direct(const TypeKey &tk, const PolyArgs &a, int num_cus, DirectArgs *out, size_t *lds)
{
    static const int enabled = [] { const char *v = std::getenv("MRHIP_DIRECT"); return !(v && v[0] == '0'); }();
    if (!enabled || a.L != 1 || a.T > 512) return false;   // packed taps: up to 8 registers
    const int sb = (tk.x_f64 ? 8 : 4) * (tk.complex_x ? 2 : 1);
    const long long opt = 1;                             // outputs per thread per step
    const long long rows = a.M;                          // residue rows of the transposed tile
    // J steps of 256 threads; keep the tile near 24 KiB
    static const int tile_kib = [] { const char *v = std::getenv("MRHIP_DIRECT_TILE_KIB"); return v && *v ? std::atoi(v) : 24; }();
    long long J = (static_cast<long long>(tile_kib) * 1024 / sb - a.T) / (static_cast<long long>(kDirectThreads) * opt * a.M);
    if (J > 16) J = 16;
    if (J < 1) J = 1;
    const long long want_tiles = 4LL * num_cus;
    while (J > 1 && ((a.n_out + J * kDirectThreads * opt - 1) / (J * kDirectThreads * opt)) * a.nch < want_tiles) J = (J + 1) / 2;
    const long long tile_len = J * kDirectThreads * opt * a.M + a.T;       // samples
    const long long Q = (tile_len + rows - 1) / rows;
    // row pitch: rows of one residue start 8 bytes apart in bank space so the transposed stores of a
    // wave (consecutive samples -> consecutive rows) spread over the banks
    long long pitch = Q + 1;
    const long long bytes = pitch * rows * sb;
    if (bytes > 64 * 1024) return false;
    DirectArgs da{};
    da.J = static_cast<int>(J);
    da.tile_len = static_cast<int>(tile_len);
    da.row_pitch = static_cast<int>(pitch);
    da.tiles_per_channel = (a.n_out + J * kDirectThreads * opt - 1) / (J * kDirectThreads * opt);
    da.total_tiles = da.tiles_per_channel * a.nch;
    *out = da;
    *lds = static_cast<size_t>(bytes);
    return true;
}

hipError_t launch_fir_direct(const TypeKey &tk, bool fused, const PolyArgs &a, const DirectArgs &da, size_t lds, hipStream_t s,
                             const char **kname, int num_cus)
{
    *kname = "fir_direct_kernel";
    if (!tk.x_f64 && !tk.r_f64) return tk.complex_x ? launch_direct<float, float, 2>(fused, a, da, lds, s, num_cus) : launch_direct<float, float, 1>(fused, a, da, lds, s, num_cus);
    if (!tk.x_f64 && tk.r_f64) return tk.complex_x ? launch_direct<float, double, 2>(fused, a, da, lds, s, num_cus) : launch_direct<float, double, 1>(fused, a, da, lds, s, num_cus);
    if (tk.x_f64 && tk.r_f64) return tk.complex_x ? launch_direct<double, double, 2>(fused, a, da, lds, s, num_cus) : launch_direct<double, double, 1>(fused, a, da, lds, s, num_cus);
    return hipErrorInvalidValue;
}

}  // namespace mrhip
