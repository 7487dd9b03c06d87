// kernels_fir_stream_rt.hip -- fir_stream_kernel with the decimation M as a RUN-TIME value: FIRStandard / FIRDecimator
// (src/Filters.jl:450-473, :598-631; dot: src/support.jl:33-55) at ANY decimation whose step fits the LDS (M * bytes-per-sample
// <= ~900 with one output per lane, half that with two), one kernel per (sample type, arithmetic type, components, bytes per
// LDS read).
//
// fir_stream_kernel.inc is instantiated per M (the position of the second output's first sample inside the first block of
// reads, the pad period of the staged tile and every LDS offset are compile-time constants there): rounds 2-3 had 40
// decimations x 3 arithmetic families x 2 numerics modes, 12.6 MB of code objects, and any M outside the list fell to
// fir_direct_kernel at a third of the rate.  Here the same machinery (taps by scalar loads; tiles staged by the loader wave of
// pair_loader.h with the pad chunks of the bank rule of kernels_fir_stream.hip) takes M, the lane stride and the pad period as
// values, in one of two lane maps (pa.rt; kernels_fir_stream.hip plans which, from the measured table):
//   * TWO adjacent outputs per lane: ONE run of T + M samples feeds both dots; the run is walked in blocks of BS samples that
//     are sorted, wave-uniformly, into
//       B  the first output alone   (blocks wholly below sample M),
//       A  both outputs             (blocks wholly inside [M, T)),
//       D  the second output alone  (blocks wholly inside [T, T + M)),
//       C  anything else, sample by sample under wave-uniform masks (the block that holds sample M, the ends of the two
//          windows when T or M is not a whole number of blocks; in FUSED mode every block of a tile that touches the
//          start-from-zero seam of support.jl:46);
//   * ONE output per lane (large decimations: half the LDS per lane, twice the waves per CU): a run of T samples, every block
//     of class B but the last, which is of class C when T is not a whole number of blocks.
// All of it is branch-free: the accumulators start at -0.0 (at +0.0 on the seam) and EVERY tap is one multiply and one add --
// x + (-0.0) == x for every x (signed zeros, NaN and infinities included), so the first product "initialises" the accumulator
// exactly as the reference's `dotprod = h[1] * x[..]` does, without a special first step at a run-time position.  The mixed
// block reads both tap runs whole (they may reach into the zero pads either side of the device tap vector, api.hip
// upload_taps) and keeps, per tap, the old accumulator where a bit mask says the sample is outside that window.
//
// Arithmetic: exactly fir_stream_kernel's (STRICT: separately rounded multiply and add, oldest sample first; FUSED: fma) =>
// bit-identical results.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "mrhip_internal.h"
#include "pair_device.h"
#include "pair_loader.h"

#pragma clang fp contract(off)

namespace mrhip {
namespace {

using namespace dev;

constexpr int kRtMaxThreads = 512;
constexpr int kRtGroups = 32;

template <typename TXS, typename R, int NC, int RD, bool FUSED>
__global__ __launch_bounds__(kRtMaxThreads + 64)
void fir_stream_rt_kernel(PolyArgs a, PairArgs pa)
{
    constexpr int ES = static_cast<int>(sizeof(TXS)) * NC;       // bytes per sample
    constexpr int NW = ES / 4;                                   // 32-bit words per sample (the loader's unit)
    constexpr int SPR = RD / ES;                                 // samples per LDS read
    constexpr int BS = ES == 16 ? 8 : 16;                        // samples per block
    constexpr int CPB = BS / SPR;                                // reads per block
    static_assert(SPR >= 1 && CPB >= 1, "read geometry");
    using read_t = std::conditional_t<RD == 8, v2u_t, v4u_t>;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int ncw = (blockDim.x >> 6) - 1;      // compute waves; the last wave is the loader
    pair_take_dyn(a, pa);                       // a device-planned call: n_out, d0 and the step walk from the call record

    if (wave == ncw) {
        pair_loader_wave<NW>(a, pa, smem, lane);
        return;
    }
    volatile unsigned *const tile_flag = reinterpret_cast<volatile unsigned *>(smem + pa.flags_off);
    typedef const __attribute__((address_space(4))) R *const_taps_t;   // scalar tap loads (see fir_stream_kernel.inc)
    const const_taps_t tc = (const_taps_t)(static_cast<const R *>(__builtin_assume_aligned(a.taps, 64)));
    const int T = a.T;
    const int M = static_cast<int>(a.M);
    const int n_out = static_cast<int>(a.n_out);
    // pa.rt == 2: ONE output per lane (large decimations: a lane's run is M samples shorter and the lanes lie M instead of 2M
    // samples apart -- half the LDS per lane, twice the waves per CU; every block is then of the first class or mixed)
    const bool single = pa.rt == 2;
    const int opl = single ? 1 : 2;             // outputs per lane
    const int lanes = single ? pa.P : pa.P >> 1;
    // the staged tile: CD data chunks of 16 bytes, then one pad chunk (pad_every = 0: linear)
    const int CD = pa.pad_every > 0 ? pa.pad_every : 0x7fffffff;
    const int lane_bytes = pa.pad_every > 0 ? 16 * (pa.pad_every + 1) : opl * M * ES;
    // the block walk of a lane's run (wave-uniform; see the header)
    const int NBLK = (T + (single ? 0 : M) + BS - 1) / BS;
    const int fB1 = single ? T / BS : std::min(M / BS, T / BS);                   // B: [0, fB1)
    const int fA0 = single ? fB1 : std::max(fB1, (M + BS - 1) / BS), fA1 = single ? fB1 : std::max(fA0, T / BS);   // A: [fA0, fA1)
    const int fD0 = single ? fB1 : std::max(fA1, std::max((T + BS - 1) / BS, (M + BS - 1) / BS));
    const int fD1 = single ? fB1 : std::max(fD0, (T + M) / BS);                   // D: [fD0, fD1)

    for (int s = 0;; s = (s + 1 == pa.ns ? 0 : s + 1)) {
        __builtin_amdgcn_s_barrier();             // one barrier per tile, no memory wait (opair_kernel.inc)
        asm volatile("" ::: "memory");
        const unsigned tg = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(tile_flag[2 * s])));
        const unsigned tj = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(tile_flag[2 * s + 1])));
        if (tj == 0u) break;                      // end marker
        const TileAt ta = pair_tile_at(pa, tg, tj);
        const int J = ta.jt;
        R *__restrict__ yc = static_cast<R *>(a.y) + (static_cast<long long>(ta.ch) * a.y_stride + static_cast<long long>(ta.st) * pa.P) * NC;
        const int first_out = ta.st * pa.P;                               // channel-relative index of the tile's first output
        const int remaining = n_out - first_out;
        // start-from-zero quirk (support.jl:46): outputs whose newest-sample index n = d0 + k*M is below the threshold
        const bool tile_has_zs = a.d0 + static_cast<long long>(first_out) * M < a.zero_start_below;   // wave-uniform
        // ... FUSED: such a tile (the first of a channel, at most) goes sample by sample (block_c: its first products are
        // rounded before the zero is added, as the reference's are; in STRICT every product is)
        const bool slow = FUSED && tile_has_zs;
        const int bB1 = slow ? 0 : fB1, bA0 = slow ? 0 : fA0, bA1 = slow ? 0 : fA1;
        const int bD0 = slow ? 0 : fD0, bD1 = slow ? 0 : fD1;
        const unsigned char *const stage = smem + static_cast<size_t>(s) * pa.stage_bytes;
        if (tid < lanes) {
#pragma unroll 1
            for (int j = 0; j < J; ++j) {
                const int k0 = j * pa.P + opl * tid;                      // tile-relative index of this lane's first output
                if (k0 >= remaining) break;
                const unsigned char *const run = stage + (static_cast<size_t>(j) * lanes + tid) * lane_bytes;
                // the accumulators start at -0.0 (the first product then initialises them exactly), at +0.0 where the
                // reference's seam variant starts from zero(Ty) (support.jl:46)
                bool zs0 = false, zs1 = false;
                if (tile_has_zs) {
                    const long long n0 = a.d0 + static_cast<long long>(first_out + k0) * M;
                    zs0 = n0 < a.zero_start_below; zs1 = n0 + M < a.zero_start_below;
                }
                R acc0[NC], acc1[NC];
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) { acc0[cc] = zs0 ? static_cast<R>(0.0) : static_cast<R>(-0.0); acc1[cc] = zs1 ? static_cast<R>(0.0) : static_cast<R>(-0.0); }
                auto unpack = [&](const unsigned (&u)[4], int e, R (&we)[NC]) {
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) {
                        if constexpr (sizeof(TXS) == 8) we[cc] = __hiloint2double(static_cast<int>(u[2 * (e * NC + cc) + 1]), static_cast<int>(u[2 * (e * NC + cc)]));
                        else we[cc] = static_cast<R>(__uint_as_float(u[e * NC + cc]));
                    }
                };
                // byte offset of the run's next read: pads after every CD chunks (wave-uniform running state)
                int roff = 0, rcnt = 0;
                auto next_off = [&]() {
                    const int o = roff;
                    roff += RD;
                    if (++rcnt == CD) { rcnt = 0; roff += 16; }
                    return o;
                };
                auto load_block = [&](R (&w)[BS][NC]) {
                    int off[CPB];
#pragma unroll
                    for (int ii = 0; ii < CPB; ++ii) off[ii] = next_off();
#pragma unroll
                    for (int ii = 0; ii < CPB; ++ii) {
                        const read_t v = *reinterpret_cast<const read_t *>(run + off[ii]);
                        unsigned u[4];
                        if constexpr (RD == 8) { u[0] = v.x; u[1] = v.y; u[2] = u[3] = 0u; }
                        else { u[0] = v.x; u[1] = v.y; u[2] = v.z; u[3] = v.w; }
#pragma unroll
                        for (int e = 0; e < SPR; ++e) unpack(u, e, w[ii * SPR + e]);
                    }
                };
                auto mac = [&](R (&acc)[NC], R t, const R (&w)[NC]) {   // (ComplexF32: one packed multiply, one packed add)
                    if constexpr (NC == 2 && sizeof(R) == 4) {
                        v2f_t av = {acc[0], acc[1]};
                        const v2f_t wv = {w[0], w[1]}, tv = {t, t};
                        if constexpr (FUSED) av = __builtin_elementwise_fma(tv, wv, av);
                        else { const v2f_t p = tv * wv; av = av + p; }
                        acc[0] = av.x; acc[1] = av.y;
                    } else {
#pragma unroll
                        for (int cc = 0; cc < NC; ++cc) {
                            if constexpr (FUSED) {
                                if constexpr (sizeof(R) == 4) acc[cc] = __builtin_fmaf(t, w[cc], acc[cc]);
                                else acc[cc] = __builtin_fma(t, w[cc], acc[cc]);
                            } else { const R p = t * w[cc]; acc[cc] = acc[cc] + p; }
                        }
                    }
                };
                R w[BS][NC];
                // Sample by sample, branch-free.  The two tap runs of the block are read whole (they may reach into the zero
                // pads either side of the tap vector, api.hip upload_taps) and a sample outside a window leaves that
                // accumulator as it is: masks of the block's live samples, one select per tap.  FIRST (FUSED tiles on the
                // seam): the first product of a window is rounded before it is added (the reference's `h[1] * x[..]`, then
                // `+ zero`, support.jl:35,46), as every product is in STRICT.
                auto block_c = [&](int b, auto first_tag) {
                    constexpr bool FIRST = decltype(first_tag)::value;
                    load_block(w);
                    const int j0 = b * BS;
                    // bit e: sample j0 + e is inside [0, T) / [M, T + M)
                    auto below = [](int n) -> unsigned { return n <= 0 ? 0u : (n >= BS ? (1u << BS) - 1u : (1u << n) - 1u); };
                    const unsigned m0 = below(T - j0);
                    const unsigned m1 = single ? 0u : below(T + M - j0) & ~below(M - j0);
#pragma unroll
                    for (int e = 0; e < BS; ++e) {
                        const R t0 = tc[j0 + e], t1 = tc[j0 + e - M];
                        R r0[NC], r1[NC];
#pragma unroll
                        for (int cc = 0; cc < NC; ++cc) { r0[cc] = acc0[cc]; r1[cc] = acc1[cc]; }
                        mac(r0, t0, w[e]); mac(r1, t1, w[e]);
                        if constexpr (FIRST) {
                            const bool f0 = j0 + e == 0, f1 = j0 + e == M;
#pragma unroll
                            for (int cc = 0; cc < NC; ++cc) {
                                const R p0 = t0 * w[e][cc], p1 = t1 * w[e][cc];
                                const R s0 = acc0[cc] + p0, s1 = acc1[cc] + p1;
                                r0[cc] = f0 ? s0 : r0[cc]; r1[cc] = f1 ? s1 : r1[cc];
                            }
                        }
                        const bool live0 = (m0 >> e) & 1u, live1 = (m1 >> e) & 1u;
#pragma unroll
                        for (int cc = 0; cc < NC; ++cc) { acc0[cc] = live0 ? r0[cc] : acc0[cc]; acc1[cc] = live1 ? r1[cc] : acc1[cc]; }
                    }
                };
                auto blocks_c = [&](int &b, int end) {
                    if (FUSED && slow) { for (; b < end; ++b) block_c(b, std::true_type{}); }
                    else { for (; b < end; ++b) block_c(b, std::false_type{}); }
                };
                int b = 0;
                for (; b < bB1; ++b) {                                    // B: the first output alone
                    load_block(w);
                    const int j0 = b * BS;
#pragma unroll
                    for (int e = 0; e < BS; ++e) mac(acc0, tc[j0 + e], w[e]);
                }
                blocks_c(b, bA0);
#ifndef MRHIP_STREAM_RT_UNROLL
#define MRHIP_STREAM_RT_UNROLL 2
#endif
#pragma unroll MRHIP_STREAM_RT_UNROLL
                for (; b < bA1; ++b) {                                    // A: both outputs over all BS samples
                    load_block(w);
                    const int j0 = b * BS;
#pragma unroll
                    for (int e = 0; e < BS; ++e) {
                        mac(acc0, tc[j0 + e], w[e]);
                        mac(acc1, tc[j0 + e - M], w[e]);
                    }
                }
                blocks_c(b, bD0);
                for (; b < bD1; ++b) {                                    // D: the second output alone
                    load_block(w);
                    const int j0 = b * BS - M;
#pragma unroll
                    for (int e = 0; e < BS; ++e) mac(acc1, tc[j0 + e], w[e]);
                }
                blocks_c(b, NBLK);

                R *const dst = yc + static_cast<long long>(k0) * NC;
                if (!single && k0 + 1 < remaining) {
                    R o2[2 * NC];
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) { o2[cc] = acc0[cc]; o2[NC + cc] = acc1[cc]; }
                    __builtin_memcpy(dst, o2, sizeof(o2));
                } else {
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) dst[cc] = acc0[cc];
                }
            }
        }
    }
}

template <typename TXS, typename R, int NC, int RD>
hipError_t launch_rt_t(bool fused, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, PairArgs pa, int num_cus)
{
    auto go = [&](auto kfn) -> hipError_t {
        int per_cu = 0;
        hipError_t eo = occupancy_cached(reinterpret_cast<const void *>(kfn), block.x, lds, &per_cu);
        if (eo != hipSuccess) return eo;
        if (per_cu < 1) per_cu = 1;
        const int bpc = MRHIP_ENV_INT("MRHIP_STREAM_BPC", 0);
        if (bpc > 0) per_cu = bpc;
        long long g = static_cast<long long>(num_cus) * per_cu;
        if (g > static_cast<long long>(pa.total_steps)) g = pa.total_steps;
        if (g < 1) g = 1;
        pa.ngroups = static_cast<int>(g < kRtGroups ? g : kRtGroups);
        pa.steps_per_group = static_cast<unsigned>((pa.total_steps + pa.ngroups - 1) / pa.ngroups);
        pa.static_grabs = (static_cast<long long>(pa.total_steps) + pa.J - 1) / pa.J <= 3 * g;
        if (a.multi) {   // independent streams: group = stream, its workgroups deal its tiles round-robin (pa.total_steps: the longest stream's)
            long long w = static_cast<long long>(num_cus) * per_cu / a.multi_n;
            const long long tiles = (static_cast<long long>(pa.total_steps) + pa.J - 1) / pa.J;
            if (w > tiles) w = tiles;
            if (w < 1) w = 1;
            g = w * a.multi_n;
            pa.ngroups = a.multi_n;
            pa.static_grabs = 1;
        }
        static int dbg = MRHIP_ENV_INT("MRHIP_DEBUG", 0);
        if (dbg == 1) {
            dbg = 0;
            hipFuncAttributes fa;
            (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kfn));
            std::fprintf(stderr, "[mrhip] fir_stream_rt in=%zuB arith=%zuB T=%d M=%lld nc=%d rd=%d grid=%lld block=%u lds=%zu occ/CU=%d regs=%d P=%d cM=%d J=%d ns=%d pad_every=%d\n",
                         sizeof(TXS) * NC, sizeof(R), a.T, static_cast<long long>(a.M), NC, RD, g, block.x, lds, per_cu, fa.numRegs, pa.P, pa.cM, pa.J, pa.ns, pa.pad_every);
        }
        launch_kernel(kfn, dim3(static_cast<unsigned>(g)), block, lds, s, a, pa);
        return hipGetLastError();
    };
    return fused ? go(fir_stream_rt_kernel<TXS, R, NC, RD, true>) : go(fir_stream_rt_kernel<TXS, R, NC, RD, false>);
}

}  // namespace

hipError_t launch_fir_stream_rt(bool fused, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus)
{
    const bool rd8 = pa.rt_rd == 8;             // (lane strides that are an odd multiple of 8 bytes)
    if (!pa.r_f64) {
        if (pa.nc == 2) return rd8 ? launch_rt_t<float, float, 2, 8>(fused, block, lds, s, a, pa, num_cus) : launch_rt_t<float, float, 2, 16>(fused, block, lds, s, a, pa, num_cus);
        return rd8 ? launch_rt_t<float, float, 1, 8>(fused, block, lds, s, a, pa, num_cus) : launch_rt_t<float, float, 1, 16>(fused, block, lds, s, a, pa, num_cus);
    }
    if (pa.x_f64) {
        if (pa.nc == 2) return launch_rt_t<double, double, 2, 16>(fused, block, lds, s, a, pa, num_cus);
        return rd8 ? launch_rt_t<double, double, 1, 8>(fused, block, lds, s, a, pa, num_cus) : launch_rt_t<double, double, 1, 16>(fused, block, lds, s, a, pa, num_cus);
    }
    if (pa.nc == 2) return rd8 ? launch_rt_t<float, double, 2, 8>(fused, block, lds, s, a, pa, num_cus) : launch_rt_t<float, double, 2, 16>(fused, block, lds, s, a, pa, num_cus);
    return rd8 ? launch_rt_t<float, double, 1, 8>(fused, block, lds, s, a, pa, num_cus) : launch_rt_t<float, double, 1, 16>(fused, block, lds, s, a, pa, num_cus);
}

}  // namespace mrhip
