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
constexpr int kLaneThreads = 64 * kLaneWaves;
constexpr int kLaneStep = 2 * kLaneWaves;        // outputs per step: one pair per wave
constexpr int kLaneBlock = 16;                   // samples per staging block and channel (128 bytes)
constexpr int kLaneOutRow = 144;                 // bytes per channel row of an output tile: 16 outputs + 16 bytes (bank stagger of the 16-byte writes)
constexpr int kLaneMirror = 32;                  // ring positions repeated behind the ring (T <= 32: a window reads T + 1 samples)

#ifdef MRHIP_LANE_TRACE
// debug builds (make EXP=1 EXPFLAGS=-DMRHIP_LANE_TRACE): clock sums per wave of the LAST launch, printed at process exit:
// [0] pair statement [1] barrier [2] everything else [3] steps
__device__ unsigned long long g_lane_prof[8192][8];
#define LANE_T0() unsigned long long lt_ = __builtin_amdgcn_s_memtime()
#define LANE_TICK(acc) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); (acc) += n_ - lt_; lt_ = n_; } while (0)
#else
#define LANE_T0() do {} while (0)
#define LANE_TICK(acc) do {} while (0)
#endif

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

// The workgroup's barrier WITHOUT __syncthreads()'s wait for the vector memory counter: every wave keeps the global loads of its rows in flight
// across steps (two blocks ahead), and a vmcnt(0) in front of every barrier would make every step as long as a trip to HBM.  LDS
// operations are complete (lgkmcnt) before the barrier; "memory": the compiler moves no access across it.
__device__ __forceinline__ void lane_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The ring: a step reads [first window's oldest, last output's newest) = at most 15 + T samples, and the block written behind it ends
// at most 16 + 15 samples later: ring >= T + 46, a multiple of 16; rows ring + mirror + 1 eight-byte units apart, odd (the 64 lanes of a
// window read then fall into 64 different bank pairs).
constexpr int lane_ring(int T) { return (T + 46 + 15) / 16 * 16; }
constexpr int lane_pitch8(int T) { return (lane_ring(T) + kLaneMirror + 1) | 1; }

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

// One stretch of outputs [k0, k1) of one group of 64 channels.  Every wave does three things per step:
//   * its pair of outputs (kb + 2 wv, + 1) for the 64 channels of its lanes: lane_pair;
//   * staging for EIGHT channel rows (8 wv ... + 7; lane l: row l / 8, samples l % 8 and l % 8 + 8 of the block): the block step s + 2
//     needs is requested at step s (global -> registers), goes into the ring during step s + 1 -- behind the wave's own arithmetic, while
//     other waves still read step s + 1's windows: its places hold samples older than any of them (plan_arb_lane: the ring's length) --
//     and is read from step s + 2 on: nobody waits for HBM;
//   * the outputs of the step before for the same eight rows: LDS tile -> whole 128-byte lines of y.
// (An earlier form gave staging and stores to a ninth wave: its dependent waits -- schedule entry, LDS tile, vector memory -- made it
//  the last at every barrier, 5 400 cycles a step against the compute waves' 2 400; profiles/r06/experiments.md.)
template <bool FUSED, int T>
__device__ __forceinline__ void lane_stretch(const ArbArgs &a, const ArbLaneArgs &la, unsigned char *smem, int lane, int wv, int ch0, long long k0, long long k1)
{
    const cint_t n_idx = (cint_t)(a.n_idx);
    // (the ring's length and pitch follow from T alone -- plan_arb_lane computes the same values for the launch: compile-time
    //  divisors, and two scalars less to keep across the pair's statement, which leaves the compiler 36 SGPRs)
    constexpr int RING = lane_ring(T), P8 = lane_pitch8(T);
    const int H = a.H;
    const int nsteps = static_cast<int>((k1 - k0 + kLaneStep - 1) / kLaneStep);
    const int kr = static_cast<int>(k1 - k0);                  // outputs of this stretch
    constexpr unsigned URING = static_cast<unsigned>(RING);
    const unsigned ring_row = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem)) + static_cast<unsigned>(lane) * static_cast<unsigned>(P8) * 8u;
    unsigned char *const out0 = smem + static_cast<size_t>(64) * P8 * 8;
    unsigned char *const out_lane = out0 + lane * kLaneOutRow + wv * 16;                       // this lane's pair in a tile
    // staging and stores: this lane's row of the group and its piece of the row
    const int srow = 8 * wv + (lane >> 3), sq = lane & 7;
    const bool grp_full = ch0 + 64 <= a.nch;                    // (uniform)
    const bool row_ok = ch0 + srow < a.nch;
    unsigned long long *const ring_lane = reinterpret_cast<unsigned long long *>(smem) + static_cast<unsigned>(srow) * static_cast<unsigned>(P8) + sq;
    const unsigned char *const tile_lane = out0 + srow * kLaneOutRow + sq * 16;
    const unsigned long long *const xg = static_cast<const unsigned long long *>(a.x);
    const unsigned long long *const hg = static_cast<const unsigned long long *>(a.hist);
    // addresses = a wave-uniform base (the wave's first row) + ONE 32-bit lane offset: the scalar-base form of global_load / global_store
    const unsigned xoff = (static_cast<unsigned>(lane >> 3) * static_cast<unsigned>(a.x_stride) + static_cast<unsigned>(sq)) * 8u;
    const unsigned yoff = (static_cast<unsigned>(lane >> 3) * static_cast<unsigned>(a.y_stride) + 2u * static_cast<unsigned>(sq)) * 8u;
    const unsigned long long *const xw = xg + static_cast<long long>(ch0 + 8 * wv) * a.x_stride;      // (uniform) the wave's first row of x
    double *const yw = static_cast<double *>(a.y) + static_cast<long long>(ch0 + 8 * wv) * a.y_stride + k0;

    // block u of the ring coordinate = samples 16 u - RING ... + 15 of every row; this lane's two: sq and sq + 8.  Blocks u_lo <= u < u_hi
    // lie inside the signal for every row of a full group: no per-lane checks.
    const int u_lo = RING / kLaneBlock;
    const int u_hi = grp_full ? static_cast<int>((a.x_len + RING) / kLaneBlock) : u_lo;
    auto interior = [&](int u) { return u >= u_lo && u < u_hi; };   // (uniform)
    // what the dummy loads read (below): a block inside the signal, rows that exist
    const int u_safe_hi = static_cast<int>((a.x_len + RING) / kLaneBlock) - 1;     // (x_len >= 64: plan_arb_lane)
    const unsigned long long *const xw_eff = grp_full || ch0 + 8 * wv + 8 <= a.nch ? xw : xg + static_cast<long long>(ch0) * a.x_stride;
    // two loads, nothing waited for -- ALWAYS (an edge block or no block at all: the same loads from a place inside the signal, their
    // values unused): with the loads behind a condition the compiler's wait for the OLDER block in flight must also cover the path
    // without them, on which that block's loads are the youngest: s_waitcnt vmcnt(0), i.e. a wait for HBM in every step
    auto issue = [&](int u, unsigned long long &v0, unsigned long long &v1) {
        const int uc = u < u_lo ? u_lo : (u > u_safe_hi ? u_safe_hi : u);
        const gptr_t<const unsigned char> b = lane_uniform_ptr(reinterpret_cast<const unsigned char *>(xw_eff + (static_cast<long long>(uc) * kLaneBlock - RING)));
        v0 = *reinterpret_cast<gptr_t<const unsigned long long>>(b + xoff);
        v1 = *reinterpret_cast<gptr_t<const unsigned long long>>(b + xoff + 64u);
    };
    auto ring_put = [&](int u, unsigned long long v0, unsigned long long v1) {
        const unsigned pos = static_cast<unsigned>(u % (RING / kLaneBlock)) * kLaneBlock;       // (uniform) the block's place in the ring
        unsigned long long *const p = ring_lane + pos;
        p[0] = v0;
        p[8] = v1;
        if (pos < static_cast<unsigned>(kLaneMirror)) {         // (uniform) the ring's first samples once more behind its end
            p[RING] = v0;
            p[RING + 8] = v1;
        }
    };
    auto put_edge = [&](int u) {                                // history, the end of the signal, a partial channel group: loaded here, waited for
        const long long s0 = static_cast<long long>(u) * kLaneBlock - RING;
        const unsigned long long *const xrow = xg + static_cast<long long>(row_ok ? ch0 + srow : ch0) * a.x_stride;
        const unsigned long long *const hrow = hg + static_cast<long long>(row_ok ? ch0 + srow : ch0) * H;
        unsigned long long t2[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const long long sidx = s0 + sq + 8 * e;
            const bool ok = row_ok && sidx < a.x_len && sidx >= -static_cast<long long>(H);
            const unsigned long long *p = sidx >= 0 ? xrow + sidx : hrow + (H + sidx);
            const unsigned long long t = *(ok ? p : static_cast<const unsigned long long *>(a.taps));
            t2[e] = ok ? t : 0ull;
        }
        ring_put(u, t2[0], t2[1]);
    };
    // the outputs 16 step_ + 2 sq, + 1 (of the stretch) of this lane's row: read from the tile in front of the pair's statement (whose first
    // wait covers the read), stored behind it
    auto tile_take = [&](int tb) { return *reinterpret_cast<const v2d_t *>(tile_lane + static_cast<size_t>(tb) * (64 * kLaneOutRow)); };
    auto store = [&](int step_, v2d_t v) {
        const int ko = step_ * kLaneStep + 2 * sq;
        if (ko >= kr || !row_ok) return;
        const gptr_t<unsigned char> yb = lane_uniform_ptr(reinterpret_cast<unsigned char *>(yw + static_cast<long long>(step_) * kLaneStep));
        if (la.y16 && ko + 1 < kr) *reinterpret_cast<gptr_t<v2d_t>>(yb + yoff) = v;
        else {
            *reinterpret_cast<gptr_t<double>>(yb + yoff) = v.x;
            if (ko + 1 < kr) *reinterpret_cast<gptr_t<double>>(yb + yoff + 8u) = v.y;
        }
    };
    // The schedule entries of the stretch, ONE vector load per wave and stretch: lane l holds (n_k, acc_k) of output 16 (l / 2) + 2 wv + l % 2
    // -- this wave's pair of step l / 2 -- and the input index of step l's LAST output (which blocks the step needs); a step takes them
    // with v_readlane.  (Read per step by scalar loads they cost a trip to HBM every step: the schedule streams, every step's entries are a
    // new cache line -- 3 100 cycles a step outside the pair's statement; profiles/r06/experiments.md.)
    // A stretch may be a CHUNK of several consecutive 512-output stretches (the hand-out gives out runs of them: the ring carries on, no
    // window is staged twice, and the next run's entries are requested 28 steps before they are needed -- starting a stretch from
    // nothing costs ~4 us of dependent trips to HBM, 2.7 % of a 512-output stretch): entries of the run of 32 steps in hand (`ent_*`)
    // and of the next one (`nxt_*`).
    struct Entries { int n, last, phi; double alpha; };
    auto load_entries = [&](int run, int &n_, double &acc_, int &last_) {       // (plain vector loads: waited for at their first use)
        long long kk = k0 + static_cast<long long>(run) * (32 * kLaneStep) + kLaneStep * (lane >> 1) + 2 * wv + (lane & 1);
        if (kk >= k1) kk = k1 - 1;
        long long kl = k0 + static_cast<long long>(run) * (32 * kLaneStep) + static_cast<long long>(kLaneStep) * lane + kLaneStep - 1;
        if (kl >= k1) kl = k1 - 1;
        n_ = a.n_idx[kk];
        acc_ = a.acc[kk];
        last_ = a.n_idx[kl];
    };
    auto finish_entries = [](int n_, double acc_, int last_) {                 // phase (0-based PFB column) and alpha, src/Filters.jl:671-672
        Entries e;
        const double phif = __builtin_floor(acc_);
        e.n = n_; e.last = last_; e.phi = static_cast<int>(phif) - 1; e.alpha = acc_ - phif;
        return e;
    };
    Entries ent, nxt;
    int raw_n = 0, raw_last = 0;
    double raw_acc = 0.0;
    load_entries(0, raw_n, raw_acc, raw_last);
    ent = finish_entries(raw_n, raw_acc, raw_last);
    nxt = ent;
    const int nruns = (nsteps + 31) / 32;
    if (nruns > 1) load_entries(1, raw_n, raw_acc, raw_last);
    int run_of_ent = 0;                                         // `ent` holds the entries of steps 32 run_of_ent ... + 31, `nxt` (once finished) the run after
    bool nxt_ready = false;
    auto lane_i = [](int v, int l) { return __builtin_amdgcn_readlane(v, l); };
    auto lane_d = [](double v, int l) {
        const v2u_t b = __builtin_bit_cast(v2u_t, v);
        const v2u_t r = {static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(b.x), l)), static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(b.y), l))};
        return __builtin_bit_cast(double, r);
    };
    // entries of (global) step t: t lies in the run in hand or in the next one
    auto take_next = [&]() { if (!nxt_ready) { nxt = finish_entries(raw_n, raw_acc, raw_last); nxt_ready = true; } };
    // E(t): the ring must hold every block below E(t) before step t's windows are read
    auto E = [&](int t) {
        const int l = t - 32 * run_of_ent;
        int last;
        if (l < 32) last = lane_i(ent.last, l);
        else { take_next(); last = lane_i(nxt.last, l - 32); }
        return (last + RING + kLaneBlock - 1) / kLaneBlock;
    };
    // what the pair's statement of step s_ takes: the window's LDS address, the four tap columns, alpha of both outputs, and whether the
    // second window starts a sample later.  Prepared BEHIND the statement of the step before, in front of the barrier: a wave that
    // leaves the barrier goes straight into its statement.
    const cdouble_t pfb = (cdouble_t)(a.taps), dpfb = (cdouble_t)(a.dtaps);
    struct PairIn { unsigned addr; cdouble_t tl0, tu0, tl1, tu1; double alpha0, alpha1; bool apart; };
    auto prepare = [&](int s_) {
        PairIn q;
        int l = s_ - 32 * run_of_ent;
        if (l >= 32) {                                          // (uniform) the next run begins: its entries become the ones in hand, the run after it is requested
            take_next();
            ent = nxt;
            ++run_of_ent;
            nxt_ready = false;
            if (run_of_ent + 1 < nruns) load_entries(run_of_ent + 1, raw_n, raw_acc, raw_last);
            l -= 32;
        }
        const int n0 = lane_i(ent.n, 2 * l), n1 = lane_i(ent.n, 2 * l + 1);
        const int phi0 = lane_i(ent.phi, 2 * l), phi1 = lane_i(ent.phi, 2 * l + 1);
        q.alpha0 = lane_d(ent.alpha, 2 * l); q.alpha1 = lane_d(ent.alpha, 2 * l + 1);
        const unsigned r0 = (static_cast<unsigned>(n0 - T) + URING) % URING;                  // (ring coordinate = sample index + RING)
        q.addr = ring_row + r0 * 8u;
        q.tl0 = pfb + phi0 * T; q.tu0 = dpfb + phi0 * T; q.tl1 = pfb + phi1 * T; q.tu1 = dpfb + phi1 * T;
        q.apart = n1 != n0;
        return q;
    };

    // the stretch's first windows: every block from the first output's oldest sample to the last sample of step 0, waited for
    int e_iss = E(0);                                           // blocks below e_iss are in the ring or requested
    unsigned long long va0 = 0, va1 = 0, vb0 = 0, vb1 = 0;      // the two blocks in flight: va for odd steps, vb for even ones
    for (int u = (n_idx[k0] - T + RING) / kLaneBlock; u < e_iss; ++u) {     // (n_idx[k0]: one scalar load a stretch)
        if (interior(u)) {
            issue(u, va0, va1);
            ring_put(u, va0, va1);
        } else {
            put_edge(u);
        }
    }
    // At a rate >= 1 sixteen outputs advance by at most sixteen samples: a step needs at most ONE new block.
    int ua = -1, ub = -1;                                       // the block va / vb holds (-1: none; an edge block is loaded at its turn)
    if (nsteps > 1 && E(1) > e_iss) ua = e_iss++;
    issue(ua >= 0 ? ua : e_iss - 1, va0, va1);
    PairIn cur = prepare(0);
    lane_barrier();
#ifdef MRHIP_LANE_TRACE
    unsigned long long p_asm = 0, p_bar = 0, p_rest = 0;
#endif
    LANE_T0();
    auto step = [&](int s, unsigned long long &vn0, unsigned long long &vn1, int &unew, unsigned long long &vo0, unsigned long long &vo1, int &uold) {
        // vn: free (its block went into the ring during the step before); vo: the block step s + 1 needs, if any
        if (s + 2 < nsteps && E(s + 2) > e_iss) unew = e_iss++;
        issue(unew >= 0 ? unew : e_iss - 1, vn0, vn1);
        const v2d_t done = tile_take((s + 1) & 1);              // the step before's outputs (step 0: whatever the tile holds, not stored)
        const int kp = s * kLaneStep + 2 * wv;
        if (kp < kr) {                                          // (uniform)
            double lo0, up0, lo1, up1;
            LANE_TICK(p_rest);
            if (cur.apart) lane_pair<FUSED, T, 1>(cur.addr, cur.tl0, cur.tu0, cur.tl1, cur.tu1, lo0, up0, lo1, up1);
            else lane_pair<FUSED, T, 0>(cur.addr, cur.tl0, cur.tu0, cur.tl1, cur.tu1, lo0, up0, lo1, up1);
            LANE_TICK(p_asm);
            const double prod0 = up0 * cur.alpha0, prod1 = up1 * cur.alpha1;                    // Filters.jl:730, rounded once each
            const v2d_t res = {lo0 + prod0, lo1 + prod1};
            *reinterpret_cast<v2d_t *>(out_lane + static_cast<size_t>(s & 1) * (64 * kLaneOutRow)) = res;
        }
        if (s + 1 < nsteps) cur = prepare(s + 1);
        asm volatile("" ::: "memory");                          // (the ring writes below stay below the statement's reads above)
        // (the ring write first: its wait lets the two loads requested at the top of this step stay in flight -- with the store in front of
        //  it the compiler's count, the minimum over the paths with and without a store, makes it wait for the first of them)
        if (uold >= 0) {
            if (interior(uold)) ring_put(uold, vo0, vo1);
            else put_edge(uold);
            uold = -1;
        }
        if (s > 0) store(s - 1, done);
        LANE_TICK(p_rest);
        lane_barrier();
        LANE_TICK(p_bar);
    };
    int s = 0;
#pragma clang loop unroll(disable)
    for (; s + 1 < nsteps; s += 2) {
        step(s, vb0, vb1, ub, va0, va1, ua);
        step(s + 1, va0, va1, ua, vb0, vb1, ub);
    }
    if (s < nsteps) step(s, vb0, vb1, ub, va0, va1, ua);
    store(nsteps - 1, tile_take((nsteps - 1) & 1));
#ifdef MRHIP_LANE_TRACE
    if (lane == 0) {
        unsigned long long *const g = g_lane_prof[(blockIdx.x * kLaneWaves + wv) & 8191];
        g[0] += p_asm; g[1] += p_bar; g[2] += p_rest; g[3] += nsteps;
    }
#endif
}

template <bool FUSED, int T>
__global__ __launch_bounds__(kLaneThreads, 4) void arb_lane_kernel(ArbArgs a, ArbLaneArgs la)
{
    static_assert(T >= 2 && T <= kLaneMirror, "a window of T + 1 samples lies inside ring + mirror");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned s_item[2];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long n_out = a.dyn ? a.dyn->n_out : a.n_out;     // (a device-planned call: the count from the call record)
    const long long STRETCH = la.stretch;
    const long long NS = (n_out + STRETCH - 1) / STRETCH;       // stretches per channel group
    const long long total = NS * la.ngroups;                    // ... in all, group by group: index i = group i / NS, stretch i % NS
    unsigned *const ctr = la.counters;
    const long long G = gridDim.x;
    const long long CMAX = la.chunk_max;

    // Chunks of consecutive stretches are handed out from the counter, large first and smaller towards the end (each workgroup asks for
    // 1/(2 G) of what it believes is left, at most chunk_max, at least one): inside a chunk the ring carries on from stretch to stretch.
    long long seen = 0;                                         // the counter as this workgroup last saw it
    for (;;) {
        if (tid == 0) {
            long long c = (total - seen) / (2 * G);
            c = c < 1 ? 1 : (c > CMAX ? CMAX : c);
            s_item[0] = atomicAdd(ctr, static_cast<unsigned>(c));
            s_item[1] = static_cast<unsigned>(c);
        }
        lane_barrier();                                        // (also: every window of the chunk before is read, every tile stored from)
        long long first = uniform_ll(static_cast<long long>(s_item[0]));
        long long count = uniform_ll(static_cast<long long>(s_item[1]));
        lane_barrier();                                        // (s_item is read by everyone before it is written again)
        if (first >= total) break;
        seen = first + count;
        if (first + count > total) count = total - first;
        while (count > 0) {                                     // (a chunk that crosses into the next channel group: two pieces)
            const long long grp = first / NS, sigma = first - grp * NS;
            const long long here = sigma + count <= NS ? count : NS - sigma;
            const long long k0 = sigma * STRETCH;
            const long long k1 = k0 + here * STRETCH < n_out ? k0 + here * STRETCH : n_out;
            lane_stretch<FUSED, T>(a, la, smem, lane, wv, static_cast<int>(grp) * 64, k0, k1);
            first += here; count -= here;
            if (count > 0) lane_barrier();
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
        if (g > items) g = items;                               // (workgroups beyond the stretches would only ask the counter and leave)
        if (g < 1) g = 1;
        if (MRHIP_ENV_INT("MRHIP_DEBUG", 0) == 1) {
            hipFuncAttributes fa;
            (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kfn));
            std::fprintf(stderr, "[mrhip] arb_lane T=%d Nphi=%d grid=%lld lds=%zu occ/CU=%d regs=%d ring=%d stretch=%d items=%lld\n",
                         a.T, a.Nphi, g, lds, per_cu, fa.numRegs, la.ring, la.stretch, items);
        }
#ifdef MRHIP_LANE_TRACE
        {
            static bool armed = false;
            static long long last_grid = 0;
            last_grid = g;
            (void)hipStreamSynchronize(s);
            static unsigned long long zeros[8192][8];
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lane_prof), zeros, sizeof(zeros));
            if (!armed) {
                armed = true;
                std::atexit([] {
                    (void)hipDeviceSynchronize();
                    static unsigned long long h[8192][8];
                    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lane_prof), sizeof(h)) != hipSuccess) return;
                    double c[4] = {0, 0, 0, 0};
                    long long nc = 0;
                    for (long long b = 0; b < last_grid && (b + 1) * kLaneWaves <= 8192; ++b)
                        for (int w = 0; w < kLaneWaves; ++w) { for (int i = 0; i < 4; ++i) c[i] += h[b * kLaneWaves + w][i]; ++nc; }
                    if (c[3] > 0)
                        std::fprintf(stderr, "[lane_trace] grid=%lld  shader cycles per wave and step: pair statement %.1f barrier %.1f everything else %.1f (steps per wave %.0f)\n",
                                     last_grid, c[0] / c[3], c[1] / c[3], c[2] / c[3], c[3] / nc);
                });
            }
        }
#endif
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
    if (static_cast<double>(a.x_stride) * 8.0 * 8.0 >= 4294967296.0 || static_cast<double>(a.y_stride) * 8.0 * 8.0 >= 4294967296.0) return false;   // (the 32-bit lane offsets of staging and stores span 8 rows)
    const int min_ch = MRHIP_ENV_INT("MRHIP_LANE_MIN_CH", 48);
    if (a.nch < min_ch || (a.nch % 64 != 0 && a.nch % 64 < min_ch)) return false;          // (a last group with few channels wastes its lanes)
    ArbLaneArgs la{};
    la.ring = lane_ring(a.T);
    la.pitch8 = lane_pitch8(a.T);
    int stretch = MRHIP_ENV_INT("MRHIP_LANE_STRETCH", 512);
    if (stretch < kLaneStep) stretch = kLaneStep;
    if (stretch > 32 * kLaneStep) stretch = 32 * kLaneStep;    // (a wave holds its 64 schedule entries of a stretch in one register)
    la.stretch = stretch / kLaneStep * kLaneStep;
    la.ngroups = (a.nch + 63) / 64;
    la.chunk_max = std::max(1, std::min(8, MRHIP_ENV_INT("MRHIP_LANE_CHUNK", 8)));       // (8 x 512 outputs: step counts and offsets stay small)
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
