// kernels_arb_lane.hip -- FIRArbitrary (src/Filters.jl:693-742), Float64 taps x Float64 samples, A LANE PER CHANNEL.
//
// arb_pipe_kernel gives every lane an OUTPUT (its own phase, its own window) and four channels: per tap pair a lane reads 4 taps
// and 8 samples from LDS for 32 Float64 instructions, and that LDS time does not hide behind the arithmetic (C4 =
// VALU + LDS).  Here the 64 lanes of a wave are 64 CHANNELS at the SAME output index:
//   * phase, alpha and window position are wave-uniform: the taps of the output's two PFB columns (src/Filters.jl:724-727) are
//     SCALAR loads through the constant address space and feed the VALU as SGPR operands -- no LDS read and no VGPR for a tap;
//   * a lane computes TWO consecutive outputs from ONE window of T + 1 samples read once into registers: at a rate >= 1 the
//     second window starts 0 or 1 samples after the first (update(), src/Filters.jl:663-673: xIdx advances by
//     floor((acc + delta - 1) / Nphi) <= 1 when delta = Nphi / rate <= Nphi); the offset is wave-uniform, two code variants;
//   * LDS reads per 32 Float64 instructions: 33 / 8 = 4.1 (arb_pipe_kernel: 12), all conflict-free 8-byte reads (odd row pitch).
// Same arithmetic in the same order as arb_pipe_kernel / arb_generic_kernel / the oracle: oldest sample first, the first product
// initialises (-0.0 start), separately rounded multiply and add (FUSED: one fma per tap), y = yLower + yUpper * alpha combined in
// Float64 (Filters.jl:730).  Bit-identical results.
//
// A workgroup (8 waves) walks a STRETCH of consecutive outputs of one group of 64 channels.  The samples live in an LDS RING
// [channel][ring + mirror] (a window that wraps reads on into the mirror of the ring's first T samples: immediate offsets, no
// per-read address arithmetic); a step = 16 outputs (a pair per wave) and at most one new block of 16 samples per channel, loaded
// a step ahead into registers by all 512 threads (8 lanes x 16 B... 128 contiguous bytes per channel) and written to the ring
// after the step's arithmetic; ONE barrier per step.  Results go through an LDS tile so that every store instruction writes whole
// 128-byte lines (a lane's own outputs are 80 MB apart from its neighbour's).  Stretches are handed out from a counter.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "mrhip_internal.h"
#include "pair_device.h"

#pragma clang fp contract(off)

namespace mrhip {
namespace {

using dev::v2u_t;

constexpr int kLaneWaves = 8;
constexpr int kLaneThreads = 64 * (kLaneWaves + 1);   // eight compute waves and the loader wave
constexpr int kLaneStep = 2 * kLaneWaves;        // outputs per step: one pair per wave
constexpr int kLaneBlock = 16;                   // samples per staging block and channel (128 bytes)
constexpr int kLaneOutRow = 144;                 // bytes per channel row of an output tile: 16 outputs + 16 bytes (bank stagger of the 16-byte writes)
constexpr int kLaneMirror = 32;                  // ring positions repeated behind the ring (T <= 32: a window reads T + 1 samples)

typedef const __attribute__((address_space(4))) double *cdouble_t;   // wave-uniform reads: scalar loads
typedef const __attribute__((address_space(4))) int *cint_t;
typedef double v2d_t __attribute__((ext_vector_type(2)));

template <bool FUSED>
__device__ __forceinline__ double mac_lane(double t, double x, double acc)
{
    if constexpr (FUSED) return __builtin_fma(t, x, acc);
    else {
        const double p = t * x;
        return acc + p;
    }
}

__device__ __forceinline__ long long uniform_ll(long long v)
{
    const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(static_cast<unsigned long long>(v)));
    const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(static_cast<unsigned long long>(v) >> 32));
    return static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo);
}

// A wave-uniform GLOBAL pointer the compiler can no longer fold into vector address arithmetic: base (SGPR pair) + 32-bit lane offset
// then selects the scalar-base form of global_load / global_store (kernels_arb_pipe.hip: opaque_uniform).
template <typename P>
using gptr_t = __attribute__((address_space(1))) P *;
template <typename P>
__device__ __forceinline__ gptr_t<P> lane_uniform_ptr(P *p)
{
    unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>(p)));
    unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>(p) >> 32));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return reinterpret_cast<gptr_t<P>>((static_cast<unsigned long long>(hi) << 32) | lo);
}
// the lane's number, derived afresh (opaque to the compiler: a lane constant kept live across the loader's loop is spilled to scratch, and
// a scratch reload waits for EVERY vector memory operation -- the blocks in flight included)
__device__ __forceinline__ unsigned lane_id_fresh()
{
    unsigned l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// The workgroup's barrier WITHOUT __syncthreads()'s wait for the vector memory counter: the loader wave keeps global loads in flight
// across steps (two blocks ahead), and a vmcnt(0) in front of every barrier would make every step as long as a trip to HBM.  LDS
// operations are complete (lgkmcnt) before the barrier; "memory": the compiler moves no access across it.
__device__ __forceinline__ void lane_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// A pair of outputs whose windows start D samples apart, for this lane's channel: the window is read in two halves of T/2 + 1
// samples (half the registers; the second output's taps i use the samples i + D).  ONE hand-scheduled statement per pair
// (arb_lane_pair.inc, generated by scripts/gen_arb_lane_asm.py): window reads, the scalar tap loads a block of eight taps ahead of
// the arithmetic that uses them -- in a double buffer of fixed scalar registers --, and the arithmetic itself.
template <bool FUSED, int T, int D>
__device__ __forceinline__ void lane_pair(unsigned addr, cdouble_t tl0, cdouble_t tu0, cdouble_t tl1, cdouble_t tu1, double &lo0, double &up0, double &lo1, double &up1)
{
    static_assert(T == 32 || T == 16, "the generated statements");
    double t0, t1;
#include "arb_lane_pair.inc"
    (void)t0; (void)t1;
}

// the eight compute waves of a stretch: wave wv owns outputs kb + 2 wv, + 1 of every step
template <bool FUSED, int T>
__device__ __forceinline__ void lane_compute(const ArbArgs &a, const ArbLaneArgs &la, unsigned char *smem, int lane, int wv, long long k0, long long k1, int nsteps)
{
    const cint_t n_idx = (cint_t)(a.n_idx);
    const cdouble_t acc_tab = (cdouble_t)(a.acc);
    const cdouble_t pfb = (cdouble_t)(a.taps), dpfb = (cdouble_t)(a.dtaps);
    const unsigned RING = static_cast<unsigned>(la.ring);
    const unsigned ring_row = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem)) + static_cast<unsigned>(lane) * static_cast<unsigned>(la.pitch8) * 8u;
    unsigned char *const out_lane = smem + static_cast<size_t>(64) * la.pitch8 * 8 + lane * kLaneOutRow + wv * 16;
#pragma clang loop unroll(disable)
    for (int s = 0; s < nsteps; ++s) {
        const long long kp = k0 + static_cast<long long>(s) * kLaneStep + 2 * wv;
        if (kp < k1) {                                          // (uniform)
            const long long kq = kp + 1 < k1 ? kp + 1 : kp;
            const int n0 = n_idx[kp], n1 = n_idx[kq];
            const double acc0 = acc_tab[kp], acc1 = acc_tab[kq];
            const double phif0 = __builtin_floor(acc0), phif1 = __builtin_floor(acc1);
            const double alpha0 = acc0 - phif0, alpha1 = acc1 - phif1;                          // src/Filters.jl:671-672
            const int phi0 = __builtin_amdgcn_readfirstlane(static_cast<int>(phif0) - 1);       // 0-based column
            const int phi1 = __builtin_amdgcn_readfirstlane(static_cast<int>(phif1) - 1);
            const unsigned r0 = (static_cast<unsigned>(n0 - T) + RING) % RING;                // (ring coordinate = sample index + RING)
            const unsigned addr = ring_row + r0 * 8u;
            double lo0, up0, lo1, up1;
            if (n1 != n0) lane_pair<FUSED, T, 1>(addr, pfb + phi0 * T, dpfb + phi0 * T, pfb + phi1 * T, dpfb + phi1 * T, lo0, up0, lo1, up1);
            else lane_pair<FUSED, T, 0>(addr, pfb + phi0 * T, dpfb + phi0 * T, pfb + phi1 * T, dpfb + phi1 * T, lo0, up0, lo1, up1);
            const double prod0 = up0 * alpha0, prod1 = up1 * alpha1;                            // Filters.jl:730, rounded once each
            const v2d_t res = {lo0 + prod0, lo1 + prod1};
            *reinterpret_cast<v2d_t *>(out_lane + static_cast<size_t>(s & 1) * (64 * kLaneOutRow)) = res;
        }
        asm volatile("" ::: "memory");
        lane_barrier();
    }
}

// The loader wave of a stretch: stages the samples (global -> registers -> ring, one block of 16 per channel and step, requested TWO
// steps before the step that needs it and written to the ring one step before: the wave never waits for HBM) and stores the outputs of
// the step before (LDS tile -> whole 128-byte lines of y).  Lane l serves piece l % 8 of the rows of channels l / 8 + 8 j.
__device__ __forceinline__ void lane_loader(const ArbArgs &a, const ArbLaneArgs &la, unsigned char *smem, int lane, int ch0, long long k0, long long k1, int nsteps, int T)
{
    const cint_t n_idx = (cint_t)(a.n_idx);
    const int RING = la.ring, P8 = la.pitch8, H = a.H;
    const int sq = lane & 7, scl = lane >> 3;
    unsigned long long *const ring64 = reinterpret_cast<unsigned long long *>(smem);
    const unsigned char *const out0 = smem + static_cast<size_t>(64) * P8 * 8;
    const unsigned long long *const xg = static_cast<const unsigned long long *>(a.x);
    const unsigned long long *const hg = static_cast<const unsigned long long *>(a.hist);
    const bool grp_full = ch0 + 64 <= a.nch;                    // (uniform)
    // Addresses are a wave-uniform 64-bit base per row group j plus ONE 32-bit lane offset (the lane's row within the group of 8, its
    // piece): the scalar-base form of global_load / global_store, no 64-bit vector arithmetic.  (8 rows fit 32 bits: plan_arb_lane)

    // block u of the ring coordinate = samples 16 u - RING ... + 15; this lane's two of channel row j: sq and sq + 8
    auto interior = [&](int u) {                                // (uniform) inside the signal, a full group: no per-lane checks
        const long long s0 = static_cast<long long>(u) * kLaneBlock - RING;
        return grp_full && s0 >= 0 && s0 + kLaneBlock <= a.x_len;
    };
    // 16 loads, nothing waited for -- ALWAYS (an edge block or no block at all: the same loads from a place inside the signal, their
    // values unused): with the loads behind a condition the compiler's wait for the OLDER block in flight must also cover the path
    // without them, on which that block's loads are the youngest: s_waitcnt vmcnt(0), i.e. a wait for HBM in every step
    auto issue = [&](int u, unsigned long long (&v)[16]) {
        long long s0 = static_cast<long long>(u) * kLaneBlock - RING;
        const bool in = interior(u);
        if (!in) s0 = s0 < 0 ? 0 : (s0 + kLaneBlock <= a.x_len ? s0 : a.x_len - kLaneBlock);     // (x_len >= 16: plan_arb_lane)
        const unsigned char *const b0 = reinterpret_cast<const unsigned char *>(xg + static_cast<long long>(ch0) * a.x_stride + s0);
        const long long rs = in ? a.x_stride * 64 : 0;          // bytes between the row groups (an edge block: rows ch0 ... ch0 + 7 eight times)
        const unsigned l = lane_id_fresh();
        const unsigned xo = ((l >> 3) * static_cast<unsigned>(a.x_stride) + (l & 7u)) * 8u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const gptr_t<const unsigned char> bj = lane_uniform_ptr(b0 + j * rs);
            v[2 * j] = *reinterpret_cast<gptr_t<const unsigned long long>>(bj + xo);
            v[2 * j + 1] = *reinterpret_cast<gptr_t<const unsigned long long>>(bj + xo + 64u);
        }
    };
    auto ring_at = [&](int u, int j) -> unsigned long long * {  // this lane's place for row j of block u
        const unsigned pos = static_cast<unsigned>(u % (RING / kLaneBlock)) * kLaneBlock;       // (uniform)
        const unsigned l = lane_id_fresh();
        return ring64 + ((l >> 3) + static_cast<unsigned>(8 * j)) * static_cast<unsigned>(P8) + (l & 7u) + pos;
    };
    auto put = [&](int u, const unsigned long long (&v)[16]) {
        const bool mirror = static_cast<unsigned>(u % (RING / kLaneBlock)) * kLaneBlock < static_cast<unsigned>(kLaneMirror);   // (uniform)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned long long *const p = ring_at(u, j);
            p[0] = v[2 * j];
            p[8] = v[2 * j + 1];
            if (mirror) {                                       // the ring's first samples once more behind its end
                p[RING] = v[2 * j];
                p[RING + 8] = v[2 * j + 1];
            }
        }
    };
    auto put_edge = [&](int u) {                                // history, the end of the signal, a partial channel group: row by row, waited for
        const long long s0 = static_cast<long long>(u) * kLaneBlock - RING;
        const bool mirror = static_cast<unsigned>(u % (RING / kLaneBlock)) * kLaneBlock < static_cast<unsigned>(kLaneMirror);
#pragma unroll 1
        for (int j = 0; j < 8; ++j) {
            const int c = ch0 + scl + 8 * j;
            const bool c_ok = c < a.nch;
            const unsigned long long *const xrow = xg + static_cast<long long>(c_ok ? c : ch0) * a.x_stride;
            const unsigned long long *const hrow = hg + static_cast<long long>(c_ok ? c : ch0) * H;
            unsigned long long t2[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const long long sidx = s0 + sq + 8 * e;
                const bool ok = c_ok && sidx < a.x_len && sidx >= -static_cast<long long>(H);
                const unsigned long long *p = sidx >= 0 ? xrow + sidx : hrow + (H + sidx);
                const unsigned long long t = *(ok ? p : static_cast<const unsigned long long *>(a.taps));
                t2[e] = ok ? t : 0ull;
            }
            unsigned long long *const p = ring_at(u, j);
            p[0] = t2[0];
            p[8] = t2[1];
            if (mirror) {
                p[RING] = t2[0];
                p[RING + 8] = t2[1];
            }
        }
    };
    // (32-bit lane arithmetic only: a 64-bit lane value spilled to scratch is reloaded with a wait for EVERY vector memory operation --
    //  the blocks in flight included)
    const int kr = static_cast<int>(k1 - k0);                  // outputs of this stretch
    auto flush = [&](int step_, int tb) {                       // outputs 16 step_ + 2 sq, + 1 (of the stretch) of the channels scl + 8 j
        const unsigned l = lane_id_fresh();
        const int sq = static_cast<int>(l & 7u), scl = static_cast<int>(l >> 3);
        const unsigned yoff = (static_cast<unsigned>(scl) * static_cast<unsigned>(a.y_stride) + 2u * static_cast<unsigned>(sq)) * 8u;
        const int ko = step_ * kLaneStep + 2 * sq;
        if (ko >= kr) return;
        const unsigned char *const tile = out0 + static_cast<size_t>(tb) * (64 * kLaneOutRow) + scl * kLaneOutRow + sq * 16;
        unsigned char *const y0 = reinterpret_cast<unsigned char *>(static_cast<double *>(a.y) + static_cast<long long>(ch0) * a.y_stride + k0 + static_cast<long long>(step_) * kLaneStep);   // (uniform)
        const bool both = ko + 1 < kr;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (grp_full || ch0 + scl + 8 * j < a.nch) {
                const v2d_t v = *reinterpret_cast<const v2d_t *>(tile + 8 * j * kLaneOutRow);
                const gptr_t<unsigned char> yj = lane_uniform_ptr(y0 + static_cast<long long>(8 * j) * a.y_stride * 8);
                if (la.y16 && both) *reinterpret_cast<gptr_t<v2d_t>>(yj + yoff) = v;
                else {
                    *reinterpret_cast<gptr_t<double>>(yj + yoff) = v.x;
                    if (both) *reinterpret_cast<gptr_t<double>>(yj + yoff + 8u) = v.y;
                }
            }
        }
    };
    // E(t): the ring must hold every block below E(t) before step t's windows are read
    auto E = [&](int t) {
        const long long kl = k0 + static_cast<long long>(t + 1) * kLaneStep - 1;
        return (n_idx[kl < k1 ? kl : k1 - 1] + RING + kLaneBlock - 1) / kLaneBlock;
    };
    // the stretch's first windows: every block from the first output's oldest sample to the last sample of step 0, waited for
    int e_iss = E(0);                                           // blocks below e_iss are in the ring or requested
    unsigned long long va[16], vb[16];                          // the two blocks in flight: va for odd steps, vb for even ones
    for (int u = (n_idx[k0] - T + RING) / kLaneBlock; u < e_iss; ++u) {
        if (interior(u)) {
            issue(u, va);
            put(u, va);
        } else {
            put_edge(u);
        }
    }
    // At a rate >= 1 sixteen outputs advance by at most sixteen samples: a step needs at most ONE new block.  The block step t needs is
    // requested during step t - 2 (here: step 1's) and goes into the ring during step t - 1, while the compute waves read step t - 1's
    // windows: its places hold samples older than any of them (plan_arb_lane: the ring's length).
    int ua = -1, ub = -1;                                       // the block va / vb holds (-1: none; edge blocks are not requested: staged at their turn)
    if (nsteps > 1 && E(1) > e_iss) ua = e_iss++;
    issue(ua >= 0 ? ua : e_iss - 1, va);
    lane_barrier();
    auto step = [&](int s, unsigned long long (&vnew)[16], int &unew, unsigned long long (&vold)[16], int &uold) {
        // vnew: free (its block went into the ring during the step before); vold: the block step s + 1 needs, if any
        if (s + 2 < nsteps && E(s + 2) > e_iss) unew = e_iss++;
        issue(unew >= 0 ? unew : e_iss - 1, vnew);
        if (s > 0) flush(s - 1, (s - 1) & 1);
        if (uold >= 0) {
            if (interior(uold)) put(uold, vold);
            else put_edge(uold);
            uold = -1;
        }
        lane_barrier();
    };
    int s = 0;
#pragma clang loop unroll(disable)
    for (; s + 1 < nsteps; s += 2) {
        step(s, vb, ub, va, ua);
        step(s + 1, va, ua, vb, ub);
    }
    if (s < nsteps) step(s, vb, ub, va, ua);
    flush(nsteps - 1, (nsteps - 1) & 1);
}

template <bool FUSED, int T>
__global__ __launch_bounds__(kLaneThreads, 6) void arb_lane_kernel(ArbArgs a, ArbLaneArgs la)
{
    static_assert(T >= 2 && T <= kLaneMirror, "a window of T + 1 samples lies inside ring + mirror");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned s_item;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long n_out = a.dyn ? a.dyn->n_out : a.n_out;     // (a device-planned call: the count from the call record)
    const long long STRETCH = la.stretch;
    const long long items = (n_out + STRETCH - 1) / STRETCH * la.ngroups;
    unsigned *const ctr = la.counters;
    const long long G = gridDim.x;

    // (the two roles run the item loop apart: what only the loader needs -- the signal's and the outputs' addresses and strides -- is then
    //  not live across the compute waves' statement, whose fixed scalar registers leave the compiler 36 SGPRs)
    if (wv == kLaneWaves) {
        for (long long item = blockIdx.x; item < items;) {
            const long long sigma = item / la.ngroups;
            const int grp = static_cast<int>(item - sigma * la.ngroups);
            const long long k0 = sigma * STRETCH;
            const long long k1 = k0 + STRETCH < n_out ? k0 + STRETCH : n_out;
            lane_loader(a, la, smem, lane, grp * 64, k0, k1, static_cast<int>((k1 - k0 + kLaneStep - 1) / kLaneStep), T);
            if (lane == 0) s_item = atomicAdd(ctr, 1u);
            lane_barrier();                                    // every window of this stretch is read, its last tile stored; the next item
            item = G + uniform_ll(static_cast<long long>(s_item));
        }
    } else {
        for (long long item = blockIdx.x; item < items;) {
            const long long sigma = item / la.ngroups;
            const long long k0 = sigma * STRETCH;
            const long long k1 = k0 + STRETCH < n_out ? k0 + STRETCH : n_out;
            lane_barrier();                                    // (the loader's first blocks are in the ring)
            lane_compute<FUSED, T>(a, la, smem, lane, wv, k0, k1, static_cast<int>((k1 - k0 + kLaneStep - 1) / kLaneStep));
            lane_barrier();
            item = G + uniform_ll(static_cast<long long>(s_item));
        }
    }
    // every workgroup counts itself off; the last one re-arms the counters for the next launch (stream order makes it visible)
    if (tid == 0) {
        __threadfence();
        if (atomicAdd(ctr + 64, 1u) == static_cast<unsigned>(G) - 1u) {
            __threadfence();
            ctr[0] = 0u;
            ctr[64] = 0u;
        }
    }
    dev::shiftin_by_last_workgroup<double, 1>(a.fold, a.x, a.hist, a.x_stride, a.x_len, a.H, a.nch);
}

template <int T>
hipError_t launch_lane_t(bool fused, const ArbArgs &a, const ArbLaneArgs &la, size_t lds, hipStream_t s, int num_cus)
{
    auto go = [&](auto kfn) -> hipError_t {
        int per_cu = 0;
        hipError_t eo = occupancy_cached(reinterpret_cast<const void *>(kfn), kLaneThreads, lds, &per_cu);
        if (eo != hipSuccess) return eo;
        if (per_cu < 1) per_cu = 1;
        const int bpc = MRHIP_ENV_INT("MRHIP_LANE_BPC", 0);
        if (bpc > 0 && bpc < per_cu) per_cu = bpc;
        const long long items = (a.n_out + la.stretch - 1) / la.stretch * la.ngroups;
        long long g = static_cast<long long>(num_cus) * per_cu;
        if (g > items) g = items;
        if (g < 1) g = 1;
        if (MRHIP_ENV_INT("MRHIP_DEBUG", 0) == 1) {
            hipFuncAttributes fa;
            (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kfn));
            std::fprintf(stderr, "[mrhip] arb_lane T=%d Nphi=%d grid=%lld lds=%zu occ/CU=%d regs=%d ring=%d stretch=%d items=%lld\n",
                         a.T, a.Nphi, g, lds, per_cu, fa.numRegs, la.ring, la.stretch, items);
        }
        launch_kernel(kfn, dim3(static_cast<unsigned>(g)), dim3(kLaneThreads), lds, s, a, la);
        return hipGetLastError();
    };
    return fused ? go(arb_lane_kernel<true, T>) : go(arb_lane_kernel<false, T>);
}

}  // namespace

// Eligible: Float64 taps and real Float64 samples, tapsPerPhi in {16, 24, 32}, a rate >= 1 (consecutive outputs then start 0 or 1
// samples apart and 16 outputs need at most 16 new samples), enough channels to fill most of a wave's lanes, and the tiles'
// hand-out counters (the filter's).
bool plan_arb_lane(const TypeKey &tk, const ArbArgs &a, double rate, ArbLaneArgs *out, size_t *lds)
{
    if (MRHIP_ENV_INT("MRHIP_ARB_LANE", 1) == 0) return false;
    if (!tk.x_f64 || !tk.r_f64 || tk.complex_x) return false;     // (the tap banks are kept in the arithmetic type: Float64)
    if (a.T != 32 && a.T != 16) return false;
    if (!(rate >= 1.0) || a.n_out < 1 || a.H > kLaneMirror || a.x_len < 64) return false;
    if (static_cast<double>(a.x_stride) * 8.0 * 8.0 >= 4294967296.0 || static_cast<double>(a.y_stride) * 8.0 * 8.0 >= 4294967296.0) return false;   // (the loader's 32-bit lane offsets span 8 rows)
    const int min_ch = MRHIP_ENV_INT("MRHIP_LANE_MIN_CH", 48);
    if (a.nch < min_ch || (a.nch % 64 != 0 && a.nch % 64 < min_ch)) return false;          // (a last group with few channels wastes its lanes)
    ArbLaneArgs la{};
    // the ring: a step reads [first window's oldest, last output's newest) = at most 15 + T samples and the block written behind it
    // ends at most 16 + 15 samples later: ring >= T + 46, a multiple of 16
    la.ring = ((a.T + 46 + 15) / 16) * 16;
    la.pitch8 = la.ring + kLaneMirror + 1;
    if ((la.pitch8 & 1) == 0) ++la.pitch8;
    int stretch = MRHIP_ENV_INT("MRHIP_LANE_STRETCH", 512);
    if (stretch < kLaneStep) stretch = kLaneStep;
    la.stretch = stretch / kLaneStep * kLaneStep;
    la.ngroups = (a.nch + 63) / 64;
    la.y16 = (reinterpret_cast<uintptr_t>(a.y) % 16 == 0) && (a.y_stride % 2 == 0);
    *out = la;
    *lds = static_cast<size_t>(64) * la.pitch8 * 8 + 2 * 64 * kLaneOutRow;
    return true;
}

hipError_t launch_arb_lane(bool fused, const ArbArgs &a, const ArbLaneArgs &la, size_t lds, hipStream_t s, const char **kname, int num_cus)
{
    *kname = "arb_lane_kernel";
    switch (a.T) {
    case 32: return launch_lane_t<32>(fused, a, la, lds, s, num_cus);
    case 16: return launch_lane_t<16>(fused, a, la, lds, s, num_cus);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace mrhip
