// kernels_decim_lane.hip -- FIRDecimator (src/Filters.jl:598-631) 1//4 with 128 taps, ComplexF32 samples x Float32 taps (BASELINE config 3b's
// shape), A LANE PER CHANNEL in TRANSPOSED form, ONE WAVE PER STRETCH.
//
// fir_stream_kernel gives a lane an OUTPUT and reads its window of 128 samples from LDS (four 16-byte reads per 32 packed instructions; config
// 3b: 3.35e8 vector instructions for 2.56e8 of arithmetic, the vector ALU 75 % busy: profiles/r06/experiments.md C).  interp_lane_kernel's
// design -- the 64 lanes of a wave are 64 channels, the taps wave-uniform scalar operands, nothing shared between waves -- cannot keep a
// window of 128 samples in registers.  It can keep the 32 OUTPUTS whose windows contain the current input sample: every input sample x[m]
// is multiplied by 32 taps and added to 32 accumulators (output j takes it with tap m - (4 j - 127)); an output is complete after its newest
// sample and leaves, its accumulator starts the output 32 further on.  Every accumulator still meets its 128 products oldest sample first,
// each product rounded, each sum rounded (FUSED: one fma): the reference's dot (src/support.jl:33-42), bit for bit -- started from -0.0,
// which is "the first product initialises" ((-0.0) + p == p for every p), or from +0.0 where the reference's loop does (support.jl:46: outputs
// whose window reaches into the history).
//   * one generated statement (decim_lane_group.inc, scripts/gen_decim_lane_asm.py) per GROUP of four samples: 256 packed instructions, the
//     taps by eight s_load_dwordx16 into a double buffer of fixed SGPRs; no LDS read, no barrier;
//   * within an iteration of two groups the accumulators stay in their registers: what changes from group to group is which taps a slot needs
//     -- a rotation of four fixed columns, i.e. a pointer into the doubled columns the host prepared (api.hip: decim_tab) --; after the
//     iteration they move two registers down, so that the completing slots are always the same two;
//   * samples arrive as in interp_lane_kernel: units of eight samples x 64 channels, 64-byte pieces of eight channel rows per load
//     instruction, transposed through the wave's own LDS patch, a unit requested a unit ahead; 16 outputs per channel leave as 128-byte lines;
//   * the price: a stretch of outputs [j0, j1) starts 33 groups early (the accumulators of j0 ... j0 + 31 need the samples of groups
//     j0 - 31 ...): 33 of (j1 - j0) + 33 groups are warm-up.  The stretch length is chosen so that the stretches just fill the chip's waves.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "mrhip_internal.h"
#include "pair_device.h"

#pragma clang fp contract(off)

namespace mrhip {
namespace {

using dev::v2f_t;
using dev::v4u_t;

constexpr int kDlUnit = 8;                       // samples per unit and channel (64 bytes: eight lanes)
constexpr int kDlPitchIn = 72;                   // bytes between channel rows of a unit (18 dwords: 32 lanes' 8-byte reads, 32 bank pairs)
constexpr int kDlPitchOut = 136;                 // bytes between channel rows of 16 outputs (34 dwords)
constexpr int kDlIn = 64 * kDlPitchIn, kDlOut = 64 * kDlPitchOut;
constexpr int kDlM = 4, kDlTaps = 128, kDlSlots = kDlTaps / kDlM;

typedef const __attribute__((address_space(4))) float *cfloat_t;     // wave-uniform reads: scalar loads

struct DecimLaneArgs {
    int stretch;             // outputs per stretch (a multiple of 16: whole 128-byte lines)
    int ngroups;             // groups of 64 channels
    int y16;                 // 16-byte stores into y are aligned
    unsigned *counters;      // [0] stretches handed out beyond the first grid-full, [64] workgroups through (zero between launches)
    const float *tab;        // D_i[k] = h[124 + i - 4 (k mod 32)], i = 0 .. 3, k = 0 .. 63 (h: oldest-sample tap first)
};

template <typename P>
using gptr_t = __attribute__((address_space(1))) P *;
template <typename P>
__device__ __forceinline__ gptr_t<P> dl_uniform_ptr(P *p)      // (kernels_arb_lane.hip: lane_uniform_ptr)
{
    unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>(p)));
    unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>(p) >> 32));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return reinterpret_cast<gptr_t<P>>((static_cast<unsigned long long>(hi) << 32) | lo);
}

#ifdef MRHIP_DL_TRACE
__device__ unsigned long long g_dl_prof[4];     // shader-clock sums over all waves: [0] inside the statement [1] whole stretches [2] groups
#endif
// the four samples of a group added to the 32 outputs in flight: ONE hand-scheduled statement
template <bool FUSED, bool ROTATE = false>
__device__ __forceinline__ void decim_lane_group(v2f_t (&acc)[kDlSlots], const v2f_t (&x)[kDlM], cfloat_t tp)
{
    v2f_t t[4];
#include "decim_lane_group.inc"
    (void)t; (void)x; (void)tp;
}

template <bool FUSED, int O>
__device__ __forceinline__ void decim_stretch(const PolyArgs &a, const DecimLaneArgs &la, unsigned char *lds, int lane, int ch0, long long j0, long long j1)
{
    constexpr int U = kDlUnit, S = kDlSlots, M = kDlM;
    const int H = a.H;
    const int kr = static_cast<int>(j1 - j0);                   // outputs of this stretch
    const int lrow = lane >> 3, sq = lane & 7;                  // loads and stores: instruction i covers channel rows 8 i + lrow; this lane's piece sq
    const bool grp_full = ch0 + 64 <= a.nch;                    // (uniform)
    const unsigned long long *const xg = static_cast<const unsigned long long *>(a.x);
    const unsigned long long *const hg = static_cast<const unsigned long long *>(a.hist);
    const unsigned xoff = (static_cast<unsigned>(lrow) * static_cast<unsigned>(a.x_stride) + static_cast<unsigned>(sq)) * 8u;
    const unsigned yoff = (static_cast<unsigned>(lrow) * static_cast<unsigned>(a.y_stride) + 2u * static_cast<unsigned>(sq)) * 8u;
    unsigned char *const in_w = lds + lrow * kDlPitchIn + sq * 8;                    // (+ i * 8 rows) where this lane's loaded sample goes
    const unsigned char *const in_r = lds + lane * kDlPitchIn;                       // this lane's channel: the unit's eight samples
    unsigned char *const out_w = lds + kDlIn + lane * kDlPitchOut;                   // this lane's channel: 16 outputs
    const unsigned char *const out_r = lds + kDlIn + lrow * kDlPitchOut + sq * 16;   // (+ i * 8 rows) the two outputs this lane stores

    // unit u = samples 8 u ... 8 u + 7 (0-based indices into x; negative: the history) of every channel row; v[i]: this lane's sample of rows 8 i + lrow
    auto load_unit = [&](long long u, unsigned long long (&v)[8]) {
        const long long s0 = u * U;
        if (grp_full && s0 >= 0 && s0 + U <= a.x_len) {          // (uniform) inside the signal, a full group: no per-lane checks
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const gptr_t<const unsigned char> b = dl_uniform_ptr(reinterpret_cast<const unsigned char *>(xg + static_cast<long long>(ch0 + 8 * i) * a.x_stride + s0));
                v[i] = *reinterpret_cast<gptr_t<const unsigned long long>>(b + xoff);
            }
        } else {                                                 // history, the end of the signal, a partial channel group
            int lr = lrow;
            asm volatile("" : "+v"(lr));                         // (nothing of this path is worth a register across the groups)
            const long long sidx = s0 + sq;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = ch0 + 8 * i + lr;
                const bool ok = row < a.nch && sidx < a.x_len && sidx >= -static_cast<long long>(H);
                const unsigned long long *const p = sidx >= 0 ? xg + static_cast<long long>(row) * a.x_stride + sidx : hg + static_cast<long long>(row) * H + (H + sidx);
                const unsigned long long t = *(ok ? p : static_cast<const unsigned long long *>(a.taps));
                v[i] = ok ? t : 0ull;
            }
        }
    };
    auto unit_put = [&](const unsigned long long (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<unsigned long long *>(in_w + i * (8 * kDlPitchIn)) = v[i];
    };
    auto unit_get = [&](int j) { return __builtin_bit_cast(v2f_t, *reinterpret_cast<const unsigned long long *>(in_r + j * 8)); };   // (read as the type it was written as)
    // outputs o0 ... o0 + 15 of the stretch out of the patch: instruction i = 128-byte lines of channel rows 8 i ... 8 i + 7
    auto flush = [&](int o0) {
        const int jo = o0 + 2 * sq;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned long long d0 = *reinterpret_cast<const unsigned long long *>(out_r + i * (8 * kDlPitchOut)), d1 = *reinterpret_cast<const unsigned long long *>(out_r + i * (8 * kDlPitchOut) + 8);
            if (jo >= kr || ch0 + 8 * i + lrow >= a.nch) continue;
            const gptr_t<unsigned char> yb = dl_uniform_ptr(reinterpret_cast<unsigned char *>(static_cast<unsigned long long *>(a.y) + static_cast<long long>(ch0 + 8 * i) * a.y_stride + (j0 + o0)));
            if (la.y16 && jo + 1 < kr) {
                *reinterpret_cast<gptr_t<v4u_t>>(yb + yoff) = v4u_t{static_cast<unsigned>(d0), static_cast<unsigned>(d0 >> 32), static_cast<unsigned>(d1), static_cast<unsigned>(d1 >> 32)};
            } else {
                *reinterpret_cast<gptr_t<unsigned long long>>(yb + yoff) = d0;
                if (jo + 1 < kr) *reinterpret_cast<gptr_t<unsigned long long>>(yb + yoff + 8u) = d1;
            }
        }
    };

    // Output j's newest sample is x[d0 - 1 + 4 j] (0-based; d0 = inputDeficit, 1-based: Filters.jl:613-625); group g = the samples
    // B + 4 g ... B + 4 g + 3, B = d0 - 4: the last of them is output g's newest.  The stretch starts 33 groups early (31 are needed; with 33
    // the first sample sits at O = d0 in its unit of eight -- j0 is a multiple of 16 -- and stays there: two groups are one unit).
    const long long B = a.d0 - M;
    const long long g_start = j0 - (S + 1);
    const long long m_first = B + M * g_start;                  // (may be negative: the history)
    long long u = m_first >= 0 ? m_first / U : -((-m_first + U - 1) / U);      // floor(m_first / 8): X holds unit u and the first four samples of unit u + 1
    // (O == m_first - 8 u: try_launch_decim_lane instantiates the stretch for the call's d0)
    v2f_t X[U + 4];
    {
        unsigned long long pv[2][8];
        load_unit(u, pv[0]);
        load_unit(u + 1, pv[1]);
        unit_put(pv[0]);
#pragma unroll
        for (int j = 0; j < U; ++j) X[j] = unit_get(j);
        unit_put(pv[1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) X[U + j] = unit_get(j);
    }
    unsigned long long nxa[8], nxb[8];                          // TWO units in flight: a unit lasts two groups (~2 us of a wave's time at three
    load_unit(u + 2, nxa);                                      // waves per SIMD), a trip to HBM under load is no shorter
    load_unit(u + 3, nxb);
    // The 32 outputs in flight.  At the top of the loop below slot k holds output g + k; inside an iteration of four groups the slots stay where
    // they are (the statement of the iteration's k-th group is handed the tap columns k places on: slot s then has age (s - k) mod 32, slot k
    // completes and starts output g + k + 32 in place); after the fourth group the accumulators rotate by four registers (36 moves per 1 024
    // packed instructions).  No per-group choice of registers is left: a switch over the completing slot cost more than the arithmetic.
    const float neg0 = __uint_as_float(0x80000000u);
    auto fresh_for = [&](long long j) {                         // support.jl:46: an output whose window reaches into the history starts from +0
        const float ini = a.d0 + M * j < a.zero_start_below ? 0.0f : neg0;
        return v2f_t{ini, ini};
    };
    v2f_t acc[S];
#pragma unroll
    for (int s_ = 0; s_ < S; ++s_) acc[s_] = fresh_for(g_start + s_);
    const cfloat_t tab = (cfloat_t)(la.tab);
    auto unit_shift = [&](unsigned long long (&nx)[8]) {       // the registers move on by one unit: `nx` (the older unit in flight) goes in, and is requested anew
#pragma unroll
        for (int i = 0; i < 4; ++i) X[i] = X[U + i];
#pragma unroll
        for (int i = 4; i < U; ++i) X[i] = unit_get(i);         // (the patch still holds the unit that becomes current)
        unit_put(nx);
#pragma unroll
        for (int i = 0; i < 4; ++i) X[U + i] = unit_get(i);
        ++u;
        load_unit(u + 3, nx);
    };
    // output g is complete.  g - j0 = 4 t - 33 + k (k: the group's place in the iteration): a line of 16 outputs is full only behind a k = 0 group
    auto emit = [&](long long g, v2f_t done, bool may_flush) {
        if (g < j0 || g >= j1) return;                          // (uniform) warm-up, or the ragged end of the last stretch
        const int oi = static_cast<int>(g - j0);
        *reinterpret_cast<unsigned long long *>(out_w + (oi & 15) * 8) = __builtin_bit_cast(unsigned long long, done);
        if (may_flush && (oi & 15) == 15) flush(oi - 15);
    };
#ifdef MRHIP_DL_TRACE
    unsigned long long p_stmt = 0;
    const unsigned long long tl0 = __builtin_amdgcn_s_memtime();
#endif
#pragma clang loop unroll(disable)
    for (long long g = g_start; g < j1; g += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const v2f_t x[M] = {X[O + 4 * (k & 1)], X[O + 4 * (k & 1) + 1], X[O + 4 * (k & 1) + 2], X[O + 4 * (k & 1) + 3]};
#ifdef MRHIP_DL_TRACE
            const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
            decim_lane_group<FUSED>(acc, x, tab + (S - k));
#ifdef MRHIP_DL_TRACE
            p_stmt += __builtin_amdgcn_s_memtime() - ts0;
#endif
            const v2f_t done = acc[k];
            acc[k] = fresh_for(g + k + S);
            emit(g + k, done, k == 0);
            if (k == 1) unit_shift(nxa);
            if (k == 3) unit_shift(nxb);
        }
        {
            const v2f_t none[M] = {};
            decim_lane_group<FUSED, true>(acc, none, tab);      // the accumulators move four registers down
        }
    }
    if (kr & 15) flush(kr & ~15);
#ifdef MRHIP_DL_TRACE
    if (lane == 0) {
        atomicAdd(&g_dl_prof[0], p_stmt);
        atomicAdd(&g_dl_prof[1], __builtin_amdgcn_s_memtime() - tl0);
        atomicAdd(&g_dl_prof[2], static_cast<unsigned long long>(j1 - g_start));
    }
#endif
}

// Three waves per SIMD (168 VGPRs): 64 of accumulators, 24 of samples, 16 for the unit in flight.
template <bool FUSED, int O>
__global__ __launch_bounds__(64, 3) void decim_lane_kernel(PolyArgs a, DecimLaneArgs la)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[kDlIn + kDlOut];
    const int lane = threadIdx.x;
    const long long SO = la.stretch;
    const long long NS = (a.n_out + SO - 1) / SO;
    const long long items = NS * la.ngroups;                    // item i: stretch i / ngroups of channel group i % ngroups
    unsigned *const ctr = la.counters;
    const long long G = gridDim.x;
#ifdef MRHIP_DL_TRACE
    const unsigned t_start = static_cast<unsigned>(__builtin_amdgcn_s_memrealtime());
#endif
    for (long long item = blockIdx.x; item < items;) {
        const long long sigma = item / la.ngroups;
        const int grp = static_cast<int>(item - sigma * la.ngroups);
        const long long j0 = sigma * SO;
        const long long j1 = j0 + SO < a.n_out ? j0 + SO : a.n_out;
        decim_stretch<FUSED, O>(a, la, lds, lane, grp * 64, j0, j1);
        unsigned nxt = 0;
        if (lane == 0) nxt = atomicAdd(ctr, 1u);
        item = G + static_cast<long long>(__builtin_amdgcn_readfirstlane(nxt));
    }
#ifdef MRHIP_DL_TRACE
    if (lane == 0) {                                            // (100 MHz ticks: first / last start, last end, sum of lifetimes)
        const unsigned t_end = static_cast<unsigned>(__builtin_amdgcn_s_memrealtime());
        atomicMin(ctr + 1, t_start); atomicMax(ctr + 2, t_start); atomicMax(ctr + 3, t_end); atomicAdd(ctr + 4, t_end - t_start); atomicMin(ctr + 5, t_end);
    }
#endif
    // every workgroup counts itself off; the last one re-arms the counters for the next launch (stream order makes it visible)
    if (lane == 0) {
        __threadfence();
        if (atomicAdd(ctr + 64, 1u) == static_cast<unsigned>(G) - 1u) {
            __threadfence();
            ctr[0] = 0u;
            ctr[64] = 0u;
        }
    }
}

}  // namespace

// The doubled, age-ordered tap columns of the transposed form (see the header): 4 x 64 floats from the 128 taps (oldest-sample tap first).
void decim_lane_table(const float *taps_oldest_first, float *tab)
{
    for (int i = 0; i < kDlM; ++i)
        for (int k = 0; k < 2 * kDlSlots; ++k) tab[i * 2 * kDlSlots + k] = taps_oldest_first[kDlTaps - kDlM + i - kDlM * (k % kDlSlots)];
}

// NOT THE DEFAULT: measured on config 3b it is slower than fir_stream_kernel -- 0.95-0.99 against 0.86 ms on the same box (FUSED 0.80 / 0.66; the
// second form, with one unit in flight and a rotation every two groups: 1.07).
// The statement itself runs as interp_lane_kernel's does, but a group is only four samples: per group the wave also retires an output, and
// per two groups it moves the accumulators on (34 moves), takes a unit of samples through its LDS patch (two dependent LDS round trips) and
// requests the next one -- in-kernel clocks (-DMRHIP_DL_TRACE): 28 % of a wave's time inside the statement; with 11 waves per CU (LDS) the
// vector ALU is ~50 % busy, and 33 of 369 groups of every stretch are warm-up.  (A first form chose the completing slot and the samples'
// registers by `switch`: the compiler's branch trees and register copies cost more than the arithmetic, 1.25 ms.)  Kept behind a switch, with
// its parity test, as the record of the experiment: MRHIP_DECIM_LANE = 1: long calls (>= 2e7 channel-outputs), 2: whatever the length.
// Eligible: FIRDecimator 1//4 with 128 taps (BASELINE config 3b's shape), ComplexF32 samples x Float32 taps, a call the host planned for one
// filter, enough channels to fill most of a wave's lanes.  Returns false when the call is not this kernel's; otherwise launches and leaves
// the launch status in *err.
bool try_launch_decim_lane(const TypeKey &tk, bool fused, const PolyArgs &a, const float *tab, unsigned *counters, hipStream_t s, const char **kname, int num_cus, hipError_t *err)
{
    const int mode = MRHIP_ENV_INT("MRHIP_DECIM_LANE", 0);
    if (mode == 0 || !counters || !tab) return false;
    if (tk.x_f64 || tk.r_f64 || !tk.complex_x) return false;
    if (a.L != 1 || a.M != kDlM || a.T != kDlTaps || a.H != kDlTaps - 1) return false;
    if (a.dyn || a.multi || a.ring_dev || a.n_out < 1 || a.x_len < 64 || a.d0 < 1 || a.d0 > kDlM) return false;
    if (mode != 2 && static_cast<double>(a.n_out) * a.nch < static_cast<double>(MRHIP_ENV_INT("MRHIP_DECIM_LANE_MIN", 20000000))) return false;
    const int min_ch = MRHIP_ENV_INT("MRHIP_LANE_MIN_CH", 48);
    if (a.nch < min_ch || (a.nch % 64 != 0 && a.nch % 64 < min_ch)) return false;
    if (static_cast<double>(a.x_stride) * 8.0 * 8.0 >= 4294967296.0 || static_cast<double>(a.y_stride) * 8.0 * 8.0 >= 4294967296.0) return false;
    DecimLaneArgs la{};
    la.ngroups = (a.nch + 63) / 64;
    la.y16 = (reinterpret_cast<uintptr_t>(a.y) % 16 == 0) && (a.y_stride % 2 == 0);
    la.counters = counters;
    la.tab = tab;
    *kname = "decim_lane_kernel";
    auto go = [&](auto kfn) -> hipError_t {
        int per_cu = 0;
        hipError_t eo = occupancy_cached(reinterpret_cast<const void *>(kfn), 64, 0, &per_cu);
        if (eo != hipSuccess) return eo;
        if (per_cu < 1) per_cu = 1;
        // every stretch pays 31 groups of warm-up: as few stretches as fill the chip's waves once (at most MRHIP_DECIM_STRETCH outputs each)
        if (per_cu > 11) per_cu = 11;                           // (measured: with 13 KB of LDS a twelfth wave per CU starts only when another has left)
        const long long slots = static_cast<long long>(num_cus) * per_cu;
        long long so = (a.n_out * la.ngroups + slots - 1) / slots;
        so = (so + 15) / 16 * 16;
        const long long so_max = std::max(16, MRHIP_ENV_INT("MRHIP_DECIM_STRETCH", 1024) / 16 * 16);
        if (so > so_max) so = so_max;
        if (so < 64) so = 64;
        la.stretch = static_cast<int>(so);
        const long long items = (a.n_out + so - 1) / so * la.ngroups;
        long long g = slots;
        if (g > items) g = items;
        if (g < 1) g = 1;
        if (MRHIP_ENV_INT("MRHIP_DEBUG", 0) == 1) {
            hipFuncAttributes fa;
            (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kfn));
            std::fprintf(stderr, "[mrhip] decim_lane grid=%lld occ/CU=%d regs=%d stretch=%d items=%lld\n", g, per_cu, fa.numRegs, la.stretch, items);
        }
#ifdef MRHIP_DL_TRACE
        {
            unsigned init[5] = {0xffffffffu, 0u, 0u, 0u, 0xffffffffu};
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(counters + 1, init, sizeof init, hipMemcpyHostToDevice);
        }
#endif
        launch_kernel(kfn, dim3(static_cast<unsigned>(g)), dim3(64), 0, s, a, la);
#ifdef MRHIP_DL_TRACE
        {
            unsigned t[5];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(t, counters + 1, sizeof t, hipMemcpyDeviceToHost);
            std::fprintf(stderr, "[dl_trace] waves=%lld first start 0, last start %.1f us, first end %.1f us, last end %.1f us, mean lifetime %.1f us\n", g, (t[1] - t[0]) / 100.0,
                         (t[4] - t[0]) / 100.0, (t[2] - t[0]) / 100.0, t[3] / 100.0 / static_cast<double>(g));
            unsigned long long pr[4] = {0, 0, 0, 0}, zz[4] = {0, 0, 0, 0};
            (void)hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_dl_prof), sizeof pr);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dl_prof), zz, sizeof zz);
            if (pr[2]) std::fprintf(stderr, "[dl_trace] memtime ticks per group and wave: statement %.1f, everything %.1f (groups %llu)\n", double(pr[0]) / pr[2], double(pr[1]) / pr[2], pr[2]);
            unsigned z[5] = {0, 0, 0, 0, 0};
            (void)hipMemcpy(counters + 1, z, sizeof z, hipMemcpyHostToDevice);
        }
#endif
        return hipGetLastError();
    };
    switch (static_cast<int>(a.d0)) {                           // (where a group's first sample sits in its unit of eight: the call's inputDeficit)
    case 1: *err = fused ? go(decim_lane_kernel<true, 1>) : go(decim_lane_kernel<false, 1>); break;
    case 2: *err = fused ? go(decim_lane_kernel<true, 2>) : go(decim_lane_kernel<false, 2>); break;
    case 3: *err = fused ? go(decim_lane_kernel<true, 3>) : go(decim_lane_kernel<false, 3>); break;
    default: *err = fused ? go(decim_lane_kernel<true, 4>) : go(decim_lane_kernel<false, 4>); break;
    }
    return true;
}

}  // namespace mrhip
