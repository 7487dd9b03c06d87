// kernels_schedule.hip -- the FIRArbitrary / FIRFarrow phase schedule evaluated ON THE DEVICE, exactly.
//
// Reference: update(kernel::FIRArbitrary), src/Filters.jl:663-673, and update(kernel::FIRFarrow), :780-792 -- a serial
// Float64 recurrence   a1 = fl(acc + delta);  if a1 > N: xIdx += floor(fl((a1-1)/N)); acc = mod(a1-1, N) + 1
// whose roundings the outputs depend on (phiIdx = floor(acc), alpha = acc - phiIdx, Filters.jl:671-672).  The host
// loop that evaluates it (host_logic.cpp, 1.3 ns per output) bounded BASELINE config 4's wall time in rounds 1-2.
//
// The algorithm is modelled, with the argument for its exactness, in scripts/sched_model.py (and run on the CPU
// against the plain recurrence by tests/test_sched_model.py); in short:
//   K1  sched_tables_kernel   a piece of the schedule is cut into segments of 64 steps; every segment is run in real
//                             Float64 arithmetic from each of the NWIN grid values around a predicted start (anchor);
//                             the end of candidate c is (candidate c', shift) of the next segment, so a segment is a
//                             map on a finite set; the workgroup composes the maps of its 64 segments (a "group").
//   K2  sched_chain_kernel    one wave walks the groups' maps from the piece's TRUE start state.
//   K3  sched_emit_kernel     every segment is re-run from the start the chain gave it, writes the schedule entries
//                             (1-based input index, accumulator), and must end exactly where the next segment starts.
// A piece whose every segment verifies IS the serial recurrence (induction from the true start).  Any doubt -- a start
// off the value grid, a table entry that could not be located, a failed end check -- raises `fail_piece`; the kernels
// of later pieces then do nothing and the host redoes that piece with its serial loop.  Nothing unverified is used.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "mrhip_internal.h"

#pragma clang fp contract(off)

namespace mrhip {
namespace {

constexpr int kSeg = kSchedSeg;          // steps per segment
constexpr int kGroupSegs = 64;           // segments per group (= per workgroup of K1 / K3)
constexpr int kTabThreads = 1024;        // at most: the tables kernel runs 64 segments x nwin candidates, one thread each (nwin <= 16: one round)

// One update() of the reference, in a form whose every operation is exact except the one the reference rounds too
// (the sum): mod(a1-1, N) == (a1-1) - k*N with k = floor(fl((a1-1)/N)), minus one more N when the rounded quotient
// overshot an integer (possible for N that is not a power of two; the reference's xIdx step keeps that overshoot,
// Filters.jl:667, and so does `x` here).  k*N <= a1 < 2^53 is an integer: exact; the difference is a multiple of
// ulp(a1) no larger than a1: exact.
__device__ __forceinline__ void sched_step(double &acc, long long &x, const SchedPlan &c)
{
    const double a1 = acc + c.delta;                                  // :664
    if (a1 > c.N) {                                                   // :666
        const double am1 = a1 - 1.0;
        const double q = c.pow2 ? am1 * c.invN : am1 / c.N;           // :667 (exact scaling for a power of two)
        const double k = __builtin_floor(q);
        double r = am1 - k * c.N;
        if (r < 0.0) r += c.N;
        acc = r + 1.0;                                                // :668
        x += static_cast<long long>(k);
    } else {
        acc = a1;
    }
}

// The same step without a branch, for the 64-step runs of the tables and emit kernels (a lone wave there pays ~170 cycles a step for the
// form above: four branches, a run-time test of `pow2` per step and a Float64 -> Int64 conversion of k in five instructions).  The same
// operations on the same values in the same order -- the results of the path not taken are discarded -- and k, an integer below 2^18 + 1
// (make_sched_plan), goes through Int32.
template <bool POW2>
__device__ __forceinline__ void sched_step_nb(double &acc, long long &x, const SchedPlan &c)
{
    const double a1 = acc + c.delta;                                  // :664
    const double am1 = a1 - 1.0;
    const double q = POW2 ? am1 * c.invN : am1 / c.N;                 // :667
    const double k = __builtin_floor(q);
    double r = am1 - k * c.N;
    // (a power of two: the scaling, the floor, k * N and the difference are all exact -- r is am1 mod N, never negative; only a ROUNDED
    //  quotient can overshoot an integer: the comment above sched_step)
    if constexpr (!POW2) r = r < 0.0 ? r + c.N : r;
    const bool wrap = a1 > c.N;                                       // :666
    acc = wrap ? r + 1.0 : a1;                                        // :668
    x += wrap ? static_cast<long long>(__double2int_rz(k)) : 0LL;
}

// Predicted phase after k steps from the piece's start: the un-rounded recurrence (double-double product) plus the
// drift per step measured so far.  Accuracy only decides how often a start falls outside the candidate window.
__device__ __forceinline__ double sched_anchor(double acc_p, double k, double slope, const SchedPlan &c)
{
    const double hi = k * c.delta;
    const double lo = __builtin_fma(k, c.delta, -hi);
    const double w = __builtin_floor(((acc_p - 1.0) + hi) / c.N);
    double r = ((acc_p - 1.0) + (hi - w * c.N)) + (lo + slope * k);
    if (r < 0.0) r += c.N;
    else if (r >= c.N) r -= c.N;
    return r + 1.0;
}

__device__ __forceinline__ double sched_base(double anchor, const SchedPlan &c)
{
    return __builtin_floor((anchor - c.halfwin) * c.inv_umin) * c.umin;   // window start, on the value grid
}

// candidate ci of the window: base + ci*umin folded onto a legal state in [1, N+1).  Fold FIRST, then add the offset:
// base - N is exact and keeps the fine grid next to 1.0.
__device__ __forceinline__ double sched_cand(double base, int ci, const SchedPlan &c)
{
    const double off = static_cast<double>(ci) * c.umin;
    const double v = base + off;
    if (v >= c.N + 1.0) return (base - c.N) + off;
    if (v < 1.0) return (base + c.N) + off;
    return v;
}

// T == candidate(ci) + shift, shift a multiple of G; false when T is off the grid or the identity does not hold in
// Float64 (the shift would cross a binade or the wrap: outside the equivariance argument).
__device__ __forceinline__ bool sched_locate(double T, double base, const SchedPlan &c, int *ci, double *shift)
{
    double d = T - base;
    if (d > 0.5 * c.N) d = (T - c.N) - base;            // the same point of the phase circle, every step exact
    else if (d < -0.5 * c.N) d = T - (base - c.N);
    const double cu = d * c.inv_umin;
    if (cu != __builtin_floor(cu)) return false;
    if (cu >= 0.0 && cu < static_cast<double>(c.nwin)) {
        *ci = static_cast<int>(cu);
        *shift = 0.0;
    } else {
        const double m = __builtin_floor(d * c.inv_G);
        *ci = static_cast<int>(cu - m * static_cast<double>(c.ncand));
        *shift = m * c.G;
    }
    return sched_cand(base, *ci, c) + *shift == T;
}

// A kernel of piece `piece` has nothing to do when an EARLIER piece failed verification or found the end of the call.
// Within its own piece it must not stop on `done`: the workgroup that holds the call's end can finish before a workgroup
// of an earlier group (another XCD, running behind) has started, and that group's entries are part of the call.
// (Round 3's first form returned on any `done`: now and then whole groups of the last piece kept what the buffer held
// before -- the failure profiles/r03/experiments.md D first took for a cache effect.)
__device__ __forceinline__ bool sched_stop(const SchedStatus *st, int piece)
{
    const int fp = __hip_atomic_load(&st->fail_piece, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int dn = __hip_atomic_load(&st->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // 1 + the piece that holds the end
    return fp != kSchedNoFail || (dn != 0 && dn - 1 < piece);
}

// MRHIP_SCHED_TRACE (a build switch, diagnostics): the 100 MHz clock at stations of the tables and emit kernels, per workgroup of the LAST
// launch of each; printed at exit (per station: the earliest and the latest workgroup, relative to the kernel's first stamp).
#ifdef MRHIP_SCHED_TRACE
__device__ unsigned long long g_sched_trace[2][64][16];
#define SCHED_STAMP(k, i) do { if (threadIdx.x == 0 && blockIdx.x < 64) g_sched_trace[k][blockIdx.x][i] = wall_clock64(); } while (0)
#else
#define SCHED_STAMP(k, i) do { } while (0)
#endif

#ifndef MRHIP_SCHED_WIDE_MIN
#define MRHIP_SCHED_WIDE_MIN 48     // groups of a piece from which the emit kernel stores through LDS (whole sectors) instead of straight from registers
#endif

// write-through stores / L2-served loads of 4- and 8-byte fields (the hand-off inside the emit kernel: see sched_emit_kernel)
__device__ __forceinline__ void st_wt(long long *p, long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_wt(double *p, double v) { __hip_atomic_store(reinterpret_cast<long long *>(p), __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ long long ld_wt(const long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_wt(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_wt(const double *p) { return __longlong_as_double(__hip_atomic_load(reinterpret_cast<const long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }

__device__ __forceinline__ SchedPieceState sched_begin_body(const SchedBeginArgs &a, long long x_len, long long k_first, bool file);
__device__ __noinline__ void sched_finish_body(const SchedPlan &c, SchedFinishArgs a);

// ---- K1: candidate tables of one group (64 segments) per workgroup, composed --------------------------------
// fu.begin (piece 0 of a call): the call's BEGIN rides here -- every thread derives the call-start state from the record
// itself (the status word is still the previous call's: nothing here reads it), one thread files it for the kernels behind.
__global__ __launch_bounds__(kTabThreads) void sched_tables_kernel(SchedPlan c, SchedPieceArgs a, SchedFuseArgs fu)
{
    SCHED_STAMP(0, 0);
    if (!fu.begin && sched_stop(a.status, a.piece)) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nwin = c.nwin;
    double *const eC = reinterpret_cast<double *>(smem);                 // [64][nwin] candidate start value
    double *const eSh = eC + kGroupSegs * nwin;                          // shift into the next segment's window
    int *const eW = reinterpret_cast<int *>(eSh + kGroupSegs * nwin);    // xIdx advance over the segment
    int *const eCn = eW + kGroupSegs * nwin;                             // candidate of the next segment (-1: none)

    const SchedPieceState ps = fu.begin ? sched_begin_body(fu.b, fu.b_x_len, fu.b_k_first, blockIdx.x == 0 && threadIdx.x == 0) : a.state[a.piece];
    const double slope = ps.ksteps > 0.0 ? ps.drift / ps.ksteps : 0.0;
    const int g = blockIdx.x;
    const long long seg0 = static_cast<long long>(g) * kGroupSegs;
    const int tasks = kGroupSegs * nwin;
    SCHED_STAMP(0, 1);
    for (int t = threadIdx.x; t < tasks; t += static_cast<int>(blockDim.x)) {
        const int sl = t / nwin, ci = t - sl * nwin;
        const double k0 = static_cast<double>((seg0 + sl) * kSeg);
        const double base = sched_base(sched_anchor(ps.acc, k0, slope, c), c);
        const double base_next = sched_base(sched_anchor(ps.acc, k0 + kSeg, slope, c), c);
        const double start = sched_cand(base, ci, c);
        double acc = start;
        long long x = 0;
        if (c.pow2) {
#pragma unroll 4
            for (int i = 0; i < kSeg; ++i) sched_step_nb<true>(acc, x, c);
        } else {
#pragma unroll 4
            for (int i = 0; i < kSeg; ++i) sched_step_nb<false>(acc, x, c);
        }
        if (a.corrupt_group == g && sl == 7) acc += c.G;                 // test hook (MRHIP_SCHED_CORRUPT): a wrong table must be caught
        int cn = -1;
        double sh = 0.0;
        if (!sched_locate(acc, base_next, c, &cn, &sh)) cn = -1;
        eC[t] = start;
        eSh[t] = sh;
        eW[t] = static_cast<int>(x);
        eCn[t] = cn;
    }
    __syncthreads();
    SCHED_STAMP(0, 2);
    if (tasks == static_cast<int>(blockDim.x)) {
        // compose by a parallel scan (one thread per (segment, candidate): nwin <= 16): after round d the entry (sl, c) is the map of
        // segments sl-2d+1 .. sl applied to candidate c -- six rounds instead of 64 dependent look-ups per candidate.  Shifts are
        // small multiples of G and advances integers: the sums are exact in any order, the tables are the serial walk's.
        const int t = threadIdx.x, sl = t / nwin, ci0 = t - sl * nwin;
        for (int d = 1; d < kGroupSegs; d *= 2) {
            int cn = eCn[t], w = eW[t];
            double sh = eSh[t];
            if (sl >= d) {
                // first the older half (segments sl-2d+1 .. sl-d, candidate ci0), then this entry's half from where that one ends
                const int tb = (sl - d) * nwin + ci0;
                const int c1 = eCn[tb];
                if (c1 < 0) { cn = -1; }
                else {
                    const int tt = sl * nwin + c1;
                    cn = eCn[tt];
                    sh = eSh[tb] + eSh[tt];
                    w = eW[tb] + eW[tt];
                }
            }
            __syncthreads();
            eCn[t] = cn; eSh[t] = sh; eW[t] = w;
            __syncthreads();
        }
        // (sl, c0): the state in front of segment sl when the group starts at candidate c0 = the map of segments 0 .. sl-1
        SCHED_STAMP(0, 3);
        int pc = ci0, pw = 0;
        double ps_ = 0.0;
        if (sl > 0) { const int tb = (sl - 1) * nwin + ci0; pc = eCn[tb]; ps_ = eSh[tb]; pw = eW[tb]; }
        const size_t pi = static_cast<size_t>(seg0 + sl) * nwin + ci0;
        a.pathT[pi] = pc >= 0 ? eC[sl * nwin + pc] + ps_ : __builtin_nan("");
        a.pathW[pi] = pc >= 0 ? pw : 0;
        if (sl == kGroupSegs - 1) {
            SchedGroupEntry ge;
            ge.shift = eSh[t];
            ge.advance = eW[t];
            ge.next = eCn[t];
            a.gtab[static_cast<size_t>(g) * nwin + ci0] = ge;
        }
        SCHED_STAMP(0, 4);
        return;
    }
    // compose: candidate c0 of the group's first segment walked through the 64 maps; the path is kept for K3
    for (int c0 = threadIdx.x; c0 < nwin; c0 += static_cast<int>(blockDim.x)) {
        int ci = c0;
        double S = 0.0;
        long long W = 0;
        bool ok = true;
        for (int sl = 0; sl < kGroupSegs; ++sl) {
            const size_t pi = static_cast<size_t>(seg0 + sl) * nwin + c0;
            a.pathT[pi] = ok ? eC[sl * nwin + ci] + S : __builtin_nan("");
            a.pathW[pi] = static_cast<int>(W);
            if (ok) {
                const int e = sl * nwin + ci;
                S += eSh[e];
                W += eW[e];
                ci = eCn[e];
                ok = ci >= 0;
            }
        }
        SchedGroupEntry ge;
        ge.shift = S;
        ge.advance = static_cast<int>(W);
        ge.next = ok ? ci : -1;
        a.gtab[static_cast<size_t>(g) * nwin + c0] = ge;
    }
}

// ---- K2: the group maps chained from the piece's true start ---------------------------------------------------
// One workgroup of 16 waves; lane = candidate (nwin <= 64).  Wave w owns a contiguous run of groups:
//   phase 1  it composes their maps into one (every lane carries its candidate through the run: one cross-lane read
//            per group, all candidates at once);
//   phase 2  wave 0 walks the 16 run maps from the true start: every run's start;
//   phase 3  every wave walks its own groups again from its run's start and writes each group's start.
// Dependent steps: 2 * ceil(groups / 16) + 16 instead of `groups`.
constexpr int kChainWaves = 16;

struct ChainState { int ci; double S; long long W; bool ok; };

__device__ __forceinline__ void chain_apply(ChainState &st, const SchedGroupEntry &ge)   // ge = this lane's entry of the group's map
{
    const int src = st.ok ? st.ci : 0;
    const double gsh = __shfl(ge.shift, src);
    const int gadv = __shfl(ge.advance, src);
    const int gnext = __shfl(ge.next, src);
    if (st.ok) {
        st.S += gsh;
        st.W += gadv;
        st.ci = gnext;
        st.ok = gnext >= 0;
    }
}

// The same when EVERY lane carries the same state (the walk from the piece's true start): the source lane is wave-uniform, so the three
// values come by v_readlane instead of a shuffle through the LDS crossbar (a dependent ~100 cycles each, sixteen groups deep).
__device__ __forceinline__ void chain_apply_uniform(ChainState &st, const SchedGroupEntry &ge)
{
    const int src = __builtin_amdgcn_readfirstlane(st.ok ? st.ci : 0);
    const long long shb = __double_as_longlong(ge.shift);
    const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(shb & 0xffffffffLL), src));
    const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(shb >> 32), src));
    const double gsh = __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo));
    const int gadv = __builtin_amdgcn_readlane(ge.advance, src);
    const int gnext = __builtin_amdgcn_readlane(ge.next, src);
    if (st.ok) {
        st.S += gsh;
        st.W += gadv;
        st.ci = gnext;
        st.ok = gnext >= 0;
    }
}

__global__ __launch_bounds__(kChainWaves * 64) void sched_chain_kernel(SchedPlan c, SchedPieceArgs a)
{
    SCHED_STAMP(0, 8);
    if (sched_stop(a.status, a.piece)) return;
    __shared__ SchedGroupEntry s_run[kChainWaves][64];      // phase 1 result: the map of every wave's run
    __shared__ int s_ci[kChainWaves];
    __shared__ double s_S[kChainWaves];
    __shared__ long long s_W[kChainWaves];
    __shared__ int s_ok[kChainWaves];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwin = c.nwin;
    const int per = (a.ngroups + kChainWaves - 1) / kChainWaves;
    const int g0 = wave * per, g1 = g0 + per < a.ngroups ? g0 + per : a.ngroups;
    auto entry = [&](int g) {
        SchedGroupEntry ge{};
        ge.next = -1;
        if (lane < nwin) ge = a.gtab[static_cast<size_t>(g) * nwin + lane];
        return ge;
    };
    {   // phase 1: lane's candidate carried through the run
        ChainState st{lane, 0.0, 0, lane < nwin};
        int g = g0;
        // (sixteen maps in flight per round of loads, then four: a run of 72 groups -- a piece of 1 150 -- took 18 dependent round trips
        //  with four, here and again in phase 3: 36 of this kernel's 48 us, profiles/r05/experiments.md O)
        for (; g + 16 <= g1; g += 16) {
            SchedGroupEntry e[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) e[q] = entry(g + q);
#pragma unroll
            for (int q = 0; q < 16; ++q) chain_apply(st, e[q]);
        }
        for (; g + 4 <= g1; g += 4) {                       // four maps in flight per round of loads
            const SchedGroupEntry e0 = entry(g), e1 = entry(g + 1), e2 = entry(g + 2), e3 = entry(g + 3);
            chain_apply(st, e0); chain_apply(st, e1); chain_apply(st, e2); chain_apply(st, e3);
        }
        for (; g < g1; ++g) chain_apply(st, entry(g));
        SchedGroupEntry r;
        r.shift = st.S;
        r.advance = static_cast<int>(st.W);
        r.next = st.ok ? st.ci : -1;
        s_run[wave][lane] = r;
    }
    SCHED_STAMP(0, 9);
    __syncthreads();
    SCHED_STAMP(0, 10);
    if (wave == 0) {   // phase 2: the runs chained from the piece's true start
        const SchedPieceState ps = a.state[a.piece];
        const double slope = ps.ksteps > 0.0 ? ps.drift / ps.ksteps : 0.0;
        ChainState st{0, 0.0, 0, false};
        st.ok = sched_locate(ps.acc, sched_base(sched_anchor(ps.acc, 0.0, slope, c), c), c, &st.ci, &st.S);
        for (int w = 0; w < kChainWaves; ++w) {
            if (lane == 0) { s_ci[w] = st.ci; s_S[w] = st.S; s_W[w] = st.W; s_ok[w] = st.ok; }
            if (w * per < a.ngroups) chain_apply_uniform(st, s_run[w][lane]);
        }
        if (lane == 0 && !st.ok) atomicMin(&a.status->fail_piece, a.piece);
    }
    SCHED_STAMP(0, 11);
    __syncthreads();
    SCHED_STAMP(0, 12);
    {   // phase 3: every group's start
        ChainState st{s_ci[wave], s_S[wave], s_W[wave], s_ok[wave] != 0};
        auto put = [&](int g) {
            if (lane == 0) {
                SchedGroupStart gs;
                gs.cand = st.ok ? st.ci : -1;
                gs.shift = st.S;
                gs.advance = st.W;
                gs.pad = 0;
                a.gstart[g] = gs;
            }
        };
        int g = g0;
        for (; g + 16 <= g1; g += 16) {
            SchedGroupEntry e[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) e[q] = entry(g + q);
#pragma unroll
            for (int q = 0; q < 16; ++q) { put(g + q); chain_apply_uniform(st, e[q]); }
        }
        for (; g + 4 <= g1; g += 4) {
            const SchedGroupEntry e0 = entry(g), e1 = entry(g + 1), e2 = entry(g + 2), e3 = entry(g + 3);
            put(g); chain_apply_uniform(st, e0);
            put(g + 1); chain_apply_uniform(st, e1);
            put(g + 2); chain_apply_uniform(st, e2);
            put(g + 3); chain_apply_uniform(st, e3);
        }
        for (; g < g1; ++g) { put(g); chain_apply_uniform(st, entry(g)); }
    }
    SCHED_STAMP(0, 13);
}

// ---- K3: run every segment from its true start, emit, verify -------------------------------------------------
// fu.finish (the last piece of a call): the workgroups count themselves off when they are through -- the ones that had
// nothing to do too -- and the last one is the call's FINISH.
__device__ __forceinline__ void sched_emit_body(const SchedPlan &c, const SchedPieceArgs &a, bool fold_chain);
__global__ __launch_bounds__(kGroupSegs) void sched_emit_kernel(SchedPlan c, SchedPieceArgs a, SchedFuseArgs fu)
{
    SCHED_STAMP(1, 0);
    sched_emit_body(c, a, fu.fold_chain != 0);
    SCHED_STAMP(1, 6);
    if (!fu.finish) return;
    static_assert(kGroupSegs == 64, "one wave per workgroup: the count below is a wave's");
    // What the FINISH reads of this kernel's work -- the status word's fields and the state behind the piece -- was stored WRITE-THROUGH
    // (st_wt below) and is drained here, before this workgroup counts itself off; the FINISH reads it past its L1 (ld_wt).  The two
    // __threadfence() of rounds 3-4 wrote back every dirty line of the XCD's L2 -- 48 KB of schedule entries per workgroup, which only
    // the NEXT kernel reads -- in front of a count: most of this kernel's 18 us in a one-piece call (profiles/r05/experiments.md O).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SCHED_STAMP(1, 7);
    int last = 0;
    if (threadIdx.x == 0) last = __hip_atomic_fetch_add(&a.status->groups_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.ngroups - 1 ? 1 : 0;
    last = __shfl(last, 0);
    SCHED_STAMP(1, 8);
    if (!last) return;
    if (threadIdx.x == 0) sched_finish_body(c, fu.f);
    SCHED_STAMP(1, 9);
}

__device__ __forceinline__ void sched_emit_body(const SchedPlan &c, const SchedPieceArgs &a, bool fold_chain)
{
    // test hook (MRHIP_SCHED_CORRUPT = -2 - g): group g starts ~0.5 ms late, i.e. after the group that holds the call's end
    // has set `done` -- it must still write its entries (sched_stop)
    if (a.corrupt_group <= -2 && static_cast<int>(blockIdx.x) == -a.corrupt_group - 2)
        for (int i = 0; i < 128; ++i) __builtin_amdgcn_s_sleep(127);
    static_assert(kGroupSegs == 64 && kSeg % 4 == 0, "one wave per workgroup; four steps per 16-byte store");
    __shared__ double s_T[kGroupSegs + 1];
    __shared__ long long s_X[kGroupSegs + 1];
    const int g = blockIdx.x, sl = threadIdx.x, nwin = c.nwin;
    // Everything this workgroup needs from memory before its run is requested in ONE round -- the status word, the piece's state, the call's
    // length, the maps of the groups in front of it (a piece of few groups) -- and used only behind the last request: three dependent round
    // trips in front of the run were 2.8 of this kernel's 10.9 us in a one-piece call (profiles/r05/experiments.md O).
    constexpr int kFoldMax = 16;
    const bool fold_wide = fold_chain && a.ngroups <= kFoldMax + 1;
    const int st_fail = __hip_atomic_load(&a.status->fail_piece, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (sched_stop)
    const int st_done = __hip_atomic_load(&a.status->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const SchedPieceState ps = a.state[a.piece];
    const long long x_len = a.status->x_len;               // (written by the call's BEGIN kernel, which ran before every piece)
    SchedGroupEntry e[kFoldMax + 1];
#pragma unroll
    for (int q = 0; q <= kFoldMax; ++q) {
        e[q] = SchedGroupEntry{};
        e[q].next = -1;
        if (fold_wide && sl < nwin) e[q] = a.gtab[static_cast<size_t>(q <= g && q < a.ngroups ? q : 0) * nwin + sl];
    }
    if (__any(st_fail != kSchedNoFail || (st_done != 0 && st_done - 1 < a.piece))) return;   // (one decision for the wave: the lanes shuffle below)
    SCHED_STAMP(1, 1);
    SchedGroupStart gs, gnx;                                // this group's true start and the next group's
    if (fold_chain) {
        // a piece of few groups has no chain kernel: the wave (lane = candidate) walks the groups' maps from the piece's true start
        // to its own group -- what sched_chain_kernel's phases 2 and 3 do, for one group
        const int lane = sl;
        const double slope0 = ps.ksteps > 0.0 ? ps.drift / ps.ksteps : 0.0;
        ChainState st{0, 0.0, 0, false};
        st.ok = sched_locate(ps.acc, sched_base(sched_anchor(ps.acc, 0.0, slope0, c), c), c, &st.ci, &st.S);
        auto put = [&](SchedGroupStart &o) { o.cand = st.ok ? st.ci : -1; o.shift = st.S; o.advance = st.W; o.pad = 0; };
        auto entry = [&](int gg) {
            SchedGroupEntry ge{};
            ge.next = -1;
            if (lane < nwin) ge = a.gtab[static_cast<size_t>(gg) * nwin + lane];
            return ge;
        };
        // (at most 17 maps: all of them came with the first round of loads -- the last group, which is the kernel's critical path, took one
        //  round trip per four maps)
        if (fold_wide) {
#pragma unroll
            for (int q = 0; q <= kFoldMax; ++q) {
                if (q == g) put(gs);
                if (q < g || (q == g && g + 1 < a.ngroups)) chain_apply_uniform(st, e[q]);
            }
            put(gnx);
        } else {
            int gg = 0;
            for (; gg + 4 <= g; gg += 4) {                      // four maps in flight per round of loads
                const SchedGroupEntry e0 = entry(gg), e1 = entry(gg + 1), e2 = entry(gg + 2), e3 = entry(gg + 3);
                chain_apply_uniform(st, e0); chain_apply_uniform(st, e1); chain_apply_uniform(st, e2); chain_apply_uniform(st, e3);
            }
            for (; gg < g; ++gg) chain_apply_uniform(st, entry(gg));
            put(gs);
            if (g + 1 < a.ngroups) chain_apply_uniform(st, entry(g));
            put(gnx);
        }
    } else {
        gs = a.gstart[g];
        gnx = g + 1 < a.ngroups ? a.gstart[g + 1] : gs;
    }
    SCHED_STAMP(1, 2);
    const long long seg = static_cast<long long>(g) * kGroupSegs + sl;
    bool bad = gs.cand < 0;
    double T = ps.acc;
    long long X = ps.xIdx;
    if (!bad) {
        const size_t pi = static_cast<size_t>(seg) * nwin + gs.cand;
        T = a.pathT[pi] + gs.shift;
        X = ps.xIdx + gs.advance + a.pathW[pi];
        bad = !(T == T);                                  // NaN: the walk inside the group had no valid entry
    }
    s_T[sl] = T;
    s_X[sl] = X;
    SCHED_STAMP(1, 3);
    double acc = T;
    long long x = X;
    long long end_k = -1;
    double end_acc = 0.0;
    long long end_x = 0;
    const long long kbase = a.k0 + seg * kSeg;            // step number of this segment's first output within the call
    int *__restrict__ gn = a.sched_n + a.k0 + static_cast<long long>(g) * kGroupSegs * kSeg;
    double *__restrict__ ga = a.sched_acc + a.k0 + static_cast<long long>(g) * kGroupSegs * kSeg;
    // A lane owns 64 CONSECUTIVE entries: it keeps four steps in registers and stores them as 16-byte units (one for the indices, two for the
    // phases) straight from there -- 48 store instructions per lane that nobody waits for.  (Rounds 3-4 transposed 16 steps at a time
    // through LDS for coalesced rows: two barriers and 64 LDS operations per 16 steps, all in the one wave that also walks the dependent
    // chain -- 10.7 of this kernel's 17.8 us in a one-piece call, profiles/r05/experiments.md O.)
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(gn) | reinterpret_cast<uintptr_t>(ga)) & 15u) == 0;
    int *__restrict__ ln = gn + sl * kSeg;
    double *__restrict__ la = ga + sl * kSeg;
    auto run = [&](auto pow2_tag) {
        constexpr bool P2 = decltype(pow2_tag)::value;
        for (int i0 = 0; i0 < kSeg; i0 += 4) {
            int nn[4];
            double aa[4];
            const long long xb = x;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                nn[ii] = static_cast<int>(x);
                aa[ii] = acc;
                sched_step_nb<P2>(acc, x, c);
            }
            // the call's end -- the one step whose xIdx is the first beyond x_len (xIdx never decreases) -- is looked for once per four
            // steps: steps i0 + 1 .. i0 + 4 of the segment (step 64 is the next segment's step 0: "the call ends between two segments")
            if (xb <= x_len && x > x_len) {
#pragma unroll
                for (int j = 1; j <= 4; ++j) {
                    const long long xj = j < 4 ? static_cast<long long>(nn[j < 4 ? j : 0]) : x, xp = nn[j - 1];
                    if (xj > x_len && xp <= x_len) { end_k = kbase + i0 + j; end_acc = j < 4 ? aa[j < 4 ? j : 0] : acc; end_x = xj; }
                }
            }
            if (vec_ok) {
                *reinterpret_cast<int4 *>(ln + i0) = make_int4(nn[0], nn[1], nn[2], nn[3]);
                *reinterpret_cast<double2 *>(la + i0) = make_double2(aa[0], aa[1]);
                *reinterpret_cast<double2 *>(la + i0 + 2) = make_double2(aa[2], aa[3]);
            } else {
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) { ln[i0 + ii] = nn[ii]; la[i0 + ii] = aa[ii]; }
            }
        }
    };
    // A long piece (hundreds of groups: its entries together are far more than the L2 holds) goes through LDS all the same, in the 16-byte
    // units: sixteen steps per lane, then four (eight) neighbouring lanes store one segment's 64 (128) contiguous bytes -- whole sectors for
    // the HBM, where the direct form above leaves 16-byte pieces of lines that are evicted before their neighbours arrive (emit kernel of a
    // 1 150-group piece: 39 us with the direct stores).  One wave per workgroup: the barriers only order the LDS traffic.
    __shared__ int4 s_nu[kGroupSegs * 5];                  // [segment][4 units + 1 pad]
    __shared__ double2 s_au[kGroupSegs * 9];               // [segment][8 units + 1 pad]
    auto run_wide = [&](auto pow2_tag) {
        constexpr bool P2 = decltype(pow2_tag)::value;
        for (int i16 = 0; i16 < kSeg; i16 += 16) {
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                const int i0 = i16 + 4 * blk;
                int nn[4];
                double aa[4];
                const long long xb = x;
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    nn[ii] = static_cast<int>(x);
                    aa[ii] = acc;
                    sched_step_nb<P2>(acc, x, c);
                }
                if (xb <= x_len && x > x_len) {
#pragma unroll
                    for (int j = 1; j <= 4; ++j) {
                        const long long xj = j < 4 ? static_cast<long long>(nn[j < 4 ? j : 0]) : x, xp = nn[j - 1];
                        if (xj > x_len && xp <= x_len) { end_k = kbase + i0 + j; end_acc = j < 4 ? aa[j < 4 ? j : 0] : acc; end_x = xj; }
                    }
                }
                s_nu[sl * 5 + blk] = make_int4(nn[0], nn[1], nn[2], nn[3]);
                s_au[sl * 9 + 2 * blk] = make_double2(aa[0], aa[1]);
                s_au[sl * 9 + 2 * blk + 1] = make_double2(aa[2], aa[3]);
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sg = (sl >> 2) + 16 * j, u = sl & 3;
                *reinterpret_cast<int4 *>(gn + sg * kSeg + i16 + 4 * u) = s_nu[sg * 5 + u];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int sg = (sl >> 3) + 8 * j, u = sl & 7;
                *reinterpret_cast<double2 *>(ga + sg * kSeg + i16 + 2 * u) = s_au[sg * 9 + u];
            }
            __syncthreads();
        }
    };
    const bool wide = vec_ok && a.ngroups > MRHIP_SCHED_WIDE_MIN;
    if (wide) { if (c.pow2) run_wide(std::true_type{}); else run_wide(std::false_type{}); }
    else if (c.pow2) run(std::true_type{}); else run(std::false_type{});
    SCHED_STAMP(1, 4);
    // the next segment's start: the next lane's, or the next group's first segment
    if (sl == kGroupSegs - 1) {
        if (g + 1 < a.ngroups) {
            const SchedGroupStart gn = gnx;
            if (gn.cand < 0) bad = true;
            else {
                const size_t pn = static_cast<size_t>(seg + 1) * nwin + gn.cand;
                s_T[kGroupSegs] = a.pathT[pn] + gn.shift;
                s_X[kGroupSegs] = ps.xIdx + gn.advance + a.pathW[pn];
            }
        } else {                                          // last segment of the piece: its end IS the next piece's start
            s_T[kGroupSegs] = acc;
            s_X[kGroupSegs] = x;
        }
    }
    __syncthreads();
    if (!bad && !(acc == s_T[sl + 1] && x == s_X[sl + 1])) bad = true;
    SCHED_STAMP(1, 5);
    if (bad) atomicMin(&a.status->fail_piece, a.piece);
    if (end_k >= 0) {                                     // unique: xIdx never decreases
        st_wt(&a.status->end_k, end_k);
        st_wt(&a.status->end_acc, end_acc);
        st_wt(&a.status->end_xIdx, end_x);
        // the drift baseline at the call's end (the next call starts there): this piece's, plus what its first end_k - k0 steps show
        const double steps = static_cast<double>(end_k - a.k0);
        double de = end_acc - sched_anchor(ps.acc, steps, 0.0, c);
        if (de > 0.5 * c.N) de -= c.N;
        else if (de < -0.5 * c.N) de += c.N;
        st_wt(&a.status->end_drift, ps.drift + de);
        st_wt(&a.status->end_ksteps, ps.ksteps + steps);
        __hip_atomic_store(&a.status->done, a.piece + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (sl == kGroupSegs - 1 && g + 1 == a.ngroups) {     // state of the next piece (used only if this piece verified)
        SchedPieceState ns;
        ns.acc = acc;
        ns.xIdx = x;
        double d = acc - sched_anchor(ps.acc, static_cast<double>(static_cast<long long>(a.ngroups) * kGroupSegs * kSeg), 0.0, c);
        if (d > 0.5 * c.N) d -= c.N;
        else if (d < -0.5 * c.N) d += c.N;
        ns.drift = ps.drift + d;
        ns.ksteps = ps.ksteps + static_cast<double>(static_cast<long long>(a.ngroups) * kGroupSegs * kSeg);
        SchedPieceState *o = &a.state[a.piece + 1];          // (the FINISH of this very kernel may read it: the serial tail of a call the pieces did not reach the end of)
        st_wt(&o->acc, ns.acc); st_wt(&o->xIdx, ns.xIdx); st_wt(&o->drift, ns.drift); st_wt(&o->ksteps, ns.ksteps);
    }
    // (rounds 1-3 reported the largest input span of the aligned tiles here, for the filter kernels' planners; they size their
    //  tiles from an a-priori bound now -- api.hip: span_bounds -- because the filter kernel is enqueued before this one has run)
}

// ---- BEGIN / FINISH: one lane each, round the pieces of a call (mrhip_internal.h: SchedBeginArgs) ---------------------
// The call-start state of piece 0 and the armed status word; `file`: this lane writes them for the kernels behind it.
__device__ __forceinline__ SchedPieceState sched_begin_body(const SchedBeginArgs &a, long long x_len, long long k_first, bool file)
{
    if (a.x_from) {                   // a chained call: what the stage before it wrote (never more than the bound the launch was sized for)
        const long long n = a.x_from->n_out;
        x_len = n < 0 ? 0 : (n < x_len ? n : x_len);
    }
    SchedPieceState ps;
    if (a.use_host) { ps.acc = a.acc; ps.xIdx = a.xIdx; ps.drift = a.drift; ps.ksteps = a.ksteps; }
    else { const DevStream r = *a.rec; ps.acc = r.acc; ps.xIdx = r.inputDeficit; ps.drift = r.drift; ps.ksteps = r.ksteps; }   // xIdx starts at inputDeficit, Filters.jl:715
    if (!file) return ps;
    SchedStatus st{};
    st.fail_piece = kSchedNoFail;
    if (ps.xIdx > x_len) {            // not one output (Filters.jl:705-709): the call ends before its first entry
        st.done = 1;
        st.end_k = k_first;
        st.end_acc = ps.acc;
        st.end_xIdx = ps.xIdx;
        st.end_drift = ps.drift;
        st.end_ksteps = ps.ksteps;
    }
    st.x_len = x_len;                 // the pieces and the FINISH kernel read it from here
    *a.status = st;
    a.state[0] = ps;
    return ps;
}

__global__ __launch_bounds__(64) void sched_begin_kernel(SchedBeginArgs a, long long x_len, long long k_first)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    (void)sched_begin_body(a, x_len, k_first, true);
}

__device__ __noinline__ void sched_finish_body(const SchedPlan &c, SchedFinishArgs a)
{
    DevStream r = *a.rec;
    SchedStatus st{};                 // (fields other workgroups of this kernel wrote: L2-served loads, see sched_emit_kernel)
    st.fail_piece = ld_wt(&a.status->fail_piece); st.done = ld_wt(&a.status->done);
    st.end_k = ld_wt(&a.status->end_k); st.end_acc = ld_wt(&a.status->end_acc); st.end_xIdx = ld_wt(&a.status->end_xIdx);
    st.x_len = ld_wt(&a.status->x_len); st.end_drift = ld_wt(&a.status->end_drift); st.end_ksteps = ld_wt(&a.status->end_ksteps);
    a.x_len = st.x_len;               // (as the BEGIN kernel resolved it: a chained call's comes from the stage before)
    const int fail = st.fail_piece;
    DevCall call{};
    call.x_len = st.x_len;
    r.sched_fail = kSchedNoFail;
    if (fail != kSchedNoFail && !a.serial_fallback) {
        // the host waits for this call anyway: it redoes the piece with its own serial loop and continues from there
        // (arb_schedule.hip: sched_collect); nothing of the stream state moves, the filter kernel finds no outputs
        r.sched_fail = fail;
        {
            SchedPieceState fs;
            fs.acc = ld_wt(&a.state[fail].acc); fs.xIdx = ld_wt(&a.state[fail].xIdx); fs.drift = ld_wt(&a.state[fail].drift); fs.ksteps = ld_wt(&a.state[fail].ksteps);
            *a.fail_state = fs;
        }
        call.n_out = 0;
        *a.call = call;
        *a.rec = r;
        *a.mirror = r;
        if (a.count_out) *a.count_out = 0;
        return;
    }
    long long n_out, end_x;
    double end_acc, drift, ksteps;
    if (fail == kSchedNoFail && st.done) {
        n_out = st.end_k; end_acc = st.end_acc; end_x = st.end_xIdx;
        // the drift baseline at the call's end, as the lane that found the end measured it.  (Rounds 3-4 took the state behind
        // the last piece that lay wholly before the end: a stream of calls of one piece each never got a baseline, and its
        // pieces stayed at 16 x nothing.)
        drift = st.end_drift; ksteps = st.end_ksteps;
    } else {
        // the serial recurrence, from the verified start of the piece that failed (or from behind the last piece when the
        // pieces did not reach the call's end) to the end of the call: exact, and slow -- one lane
        const int p0 = fail != kSchedNoFail ? fail : a.np;
        long long k = a.k_first;
        double ks = a.ks_first;
        for (int p = 0; p < p0; ++p) { const long long P = sched_piece_steps(ks, a.pmax); k += P; ks += static_cast<double>(P); }
        SchedPieceState ps;
        ps.acc = ld_wt(&a.state[p0].acc); ps.xIdx = ld_wt(&a.state[p0].xIdx); ps.drift = ld_wt(&a.state[p0].drift); ps.ksteps = ld_wt(&a.state[p0].ksteps);
        double acc = ps.acc;
        long long x = ps.xIdx;
        const long long k_start = k;
        while (x <= a.x_len && k < a.est) {
            a.sched_n[k] = static_cast<int>(x);
            a.sched_acc[k] = acc;
            ++k;
            sched_step(acc, x, c);
        }
        if (x <= a.x_len) r.error = MRHIP_ERR_INVALID_ARG;        // est is an upper bound of the count: cannot happen
        n_out = k; end_acc = acc; end_x = x;
        drift = ps.drift; ksteps = ps.ksteps;
        r.fallback_steps += k - k_start;
    }
    call.n_out = n_out;
    if (n_out > a.y_capacity) { call.n_out = a.y_capacity; r.error = MRHIP_ERR_BUFFER_TOO_SMALL; }
    call.k_done = n_out;
    r.acc = end_acc;
    r.inputDeficit = end_x - a.x_len;                            // Filters.jl:734
    r.drift = drift; r.ksteps = ksteps;
    r.n_written = call.n_out;
    r.calls += 1;
    *a.call = call;
    *a.rec = r;
    *a.mirror = r;
    if (a.count_out) *a.count_out = call.n_out;
}

__global__ __launch_bounds__(64) void sched_finish_kernel(SchedPlan c, SchedFinishArgs a)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    sched_finish_body(c, a);
}

}  // namespace

hipError_t launch_sched_begin(const SchedBeginArgs &a, long long x_len, long long k_first, hipStream_t s)
{
    hipLaunchKernelGGL(sched_begin_kernel, dim3(1), dim3(64), 0, s, a, x_len, k_first);
    return hipGetLastError();
}

hipError_t launch_sched_finish(const SchedPlan &c, const SchedFinishArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(sched_finish_kernel, dim3(1), dim3(64), 0, s, c, a);
    return hipGetLastError();
}

size_t sched_tables_lds(const SchedPlan &c) { return static_cast<size_t>(kGroupSegs) * c.nwin * 24; }

// Enqueue the three kernels of one piece (ngroups * 4096 steps from step a.k0 of the call) on `s`.
hipError_t launch_schedule_piece(const SchedPlan &c, const SchedPieceArgs &a, hipStream_t s, const SchedFuseArgs *fu_)
{
    SchedFuseArgs fu{};
    if (fu_) fu = *fu_;
    const size_t lds = sched_tables_lds(c);
    {   // the dynamic-LDS attribute is per device: remembered per (kernel, device) like every other launch's
        int per_cu = 0;
        hipError_t e = occupancy_cached(reinterpret_cast<const void *>(sched_tables_kernel), kTabThreads, lds, &per_cu);
        if (e != hipSuccess) return e;
    }
#ifdef MRHIP_SCHED_TRACE
    {
        static bool armed = false;
        static int last_groups = 0;
        last_groups = a.ngroups;
        if (!armed) {
            armed = true;
            std::atexit([] {
                (void)hipDeviceSynchronize();
                static unsigned long long h[2][64][16];
                if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sched_trace), sizeof(h)) != hipSuccess) return;
                for (int k = 0; k < 2; ++k) {
                    const int ng = std::min(last_groups, 64);
                    unsigned long long t0 = ~0ull;
                    for (int g = 0; g < ng; ++g) if (h[k][g][0]) t0 = std::min(t0, h[k][g][0]);
                    std::fprintf(stderr, "[sched_trace] %s kernel, %d workgroups; station: earliest .. latest workgroup (us from the first stamp)\n", k ? "emit" : "tables", ng);
                    for (int i = 0; i < 16; ++i) {
                        unsigned long long lo = ~0ull, hi = 0;
                        for (int g = 0; g < ng; ++g) if (h[k][g][i]) { lo = std::min(lo, h[k][g][i]); hi = std::max(hi, h[k][g][i]); }
                        if (hi) std::fprintf(stderr, "[sched_trace]   %d: %.2f .. %.2f\n", i, (lo - t0) / 100.0, (hi - t0) / 100.0);
                    }
                }
            });
        }
    }
#endif
    const int tab_threads = std::min(kTabThreads, kGroupSegs * c.nwin);      // (64 * nwin: whole waves)
    hipLaunchKernelGGL(sched_tables_kernel, dim3(static_cast<unsigned>(a.ngroups)), dim3(static_cast<unsigned>(tab_threads)), lds, s, c, a, fu);
    // few groups: no chain kernel, the emit kernel's workgroups walk the maps themselves (MRHIP_SCHED_FOLD: at most that many groups)
    static const int fold_max = [] { const char *v = std::getenv("MRHIP_SCHED_FOLD"); return v && *v ? std::atoi(v) : 16; }();
    fu.fold_chain = a.ngroups <= fold_max ? 1 : 0;
    if (!fu.fold_chain) hipLaunchKernelGGL(sched_chain_kernel, dim3(1), dim3(kChainWaves * 64), 0, s, c, a);
    hipLaunchKernelGGL(sched_emit_kernel, dim3(static_cast<unsigned>(a.ngroups)), dim3(kGroupSegs), 0, s, c, a, fu);
    return hipGetLastError();
}

// Constants of the parallel evaluation for one (delta, Nphi); plan.ok == 0: the host loop evaluates the schedule.
SchedPlan make_sched_plan(double delta, int64_t Nphi, int win_mult, int win_min)
{
    SchedPlan c{};
    c.delta = delta;
    c.N = static_cast<double>(Nphi);
    c.invN = 1.0 / c.N;
    c.pow2 = (Nphi & (Nphi - 1)) == 0;
    // (the xIdx advance of a 4096-step group is kept in 32 bits: delta / N < 2^18)
    if (!(delta > 0.0) || !(delta < c.N * 0x1p18) || Nphi > (1 << 24)) return c;
    const double lo = 1.0 + delta, hi = c.N + 1.0 + delta;
    c.umin = std::nextafter(lo, INFINITY) - lo;          // ulp(fl(1 + delta)): every value of the recurrence is a multiple
    const double utop = std::nextafter(hi, INFINITY) - hi;
    c.G = 2.0 * utop;                                    // shifts by multiples of G commute with every rounding (ties to even)
    c.inv_umin = 1.0 / c.umin;
    c.inv_G = 1.0 / c.G;
    c.ncand = static_cast<int>(c.G / c.umin);
    // One residue system modulo G is all the chain needs: a start outside the window is the candidate congruent to it plus a
    // shift, and every segment is verified (K3).  (Rounds 3-4 ran max(4 * ncand, 16) candidates -- the tables kernel, four to
    // eight times the work, was 0.8 ms of BASELINE config 4's 5.4 ms per call on a stream that does not repeat:
    // profiles/r04/experiments.md P; scripts/sched_model.py: as many pieces verify with the small window.)
    c.nwin = std::max(std::max(win_mult, 1) * c.ncand, std::max(win_min, 2));      // (MRHIP_SCHED_WIN_MULT, MRHIP_SCHED_WIN_MIN)
    c.halfwin = static_cast<double>(c.nwin / 2) * c.umin;
    c.ok = c.nwin <= kSchedMaxWin && c.umin <= 0x1p-20;
    return c;
}

}  // namespace mrhip
