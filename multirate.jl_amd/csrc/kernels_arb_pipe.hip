// kernels_arb_pipe.hip -- FIRArbitrary (src/Filters.jl:693-742) with Float64 arithmetic over 8-byte samples (Float64,
// ComplexF32): the hand-scheduled form of arb_tiled_kernel (kernels_arbitrary.hip), same arithmetic, bit-identical results.
//
// What bounded arb_tiled_kernel on BASELINE config 4 (PMC, profiles/r03/experiments.md A): the LDS array 67 % active with 48 %
// of those cycles bank conflicts, and only 4 of 10 VALU instructions outside the dot products... both from leaving the
// inner loop to the compiler:
//   * it merges two 8-byte reads of one lane (t[i], t[i + 2] of the unrolled loop) into ds_read2_b64: half rate, and banked
//     modulo 32 instead of 64;
//   * the 16-byte sample reads (two copies of the tile, one sample apart) are serviced in lane groups {0-3, 12-15, 20-27}...
//     (MI355X_MICROARCH.md, LDS): 16 lanes that span 28 outputs, so copy A and copy B, 128 bytes apart, collide;
//   * every read is waited for right in front of its use.
// Here every LDS read is an 8-byte ds_read_b64 issued by hand (lane groups {0-31}, {32-63}, 64 banks: the odd tap pitch
// makes 32 phases 32 different bank pairs, and 32 consecutive outputs at a rate <= 1 touch 32 consecutive samples), ONE copy
// of the tile, the reads of tap pair i + 1 in flight while pair i is computed (two register sets, counted s_waitcnt
// lgkmcnt(n): LDS returns in order), two sample buffers in LDS so a tile costs one barrier, and the staging addresses are
// scalar base + lane offset (no per-element integer division, no 64-bit vector address arithmetic).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#include "mrhip_internal.h"
#include "pair_device.h"

#pragma clang fp contract(off)

namespace mrhip {
namespace {

using dev::v2u_t;

#ifdef MRHIP_AP_TRACE
// debug builds (-DMRHIP_AP_TRACE): start / end time of every workgroup of the last 64 launches, dumped at process exit
__device__ unsigned long long g_ap_trace[64][2048][2];
__device__ unsigned g_ap_slot;
#endif
constexpr int kPipeThreads = 256;
constexpr int kPipeElems = 8;        // 8-byte samples a thread stages per tile: CPL channels x ROWS rows of 256 samples

template <bool FUSED, typename R>
__device__ __forceinline__ R macd(R t, R x, R acc)
{
    if constexpr (FUSED) {
        if constexpr (sizeof(R) == 4) return __builtin_fmaf(t, x, acc);
        else return __builtin_fma(t, x, acc);
    } else {
        const R p = t * x;
        return acc + p;
    }
}

// a tap as the LDS read delivers it: 8 bytes (Float64 arithmetic) or 4 (Float32)
template <typename R> struct TapBits { using type = v2u_t; };
template <> struct TapBits<float> { using type = unsigned; };
template <typename R, int OFF>
__device__ __forceinline__ typename TapBits<R>::type lds_read_tap(unsigned byte_addr)
{
    if constexpr (sizeof(R) == 8) return dev::lds_read_b64<OFF>(byte_addr);
    else return dev::lds_read_b32<OFF>(byte_addr);
}

// A wave-uniform GLOBAL pointer the compiler can no longer fold into vector address arithmetic: base (SGPR pair) + 32-bit
// lane offset then selects the scalar-base form of global_load / global_store (no 64-bit vector adds per access).  The result
// is typed as an address-space-1 pointer: rebuilt from integers as a generic pointer it is accessed with flat_load /
// flat_store, which also count in lgkmcnt -- the counter the hand-issued LDS pipeline waits on.
template <typename P>
using global_ptr = __attribute__((address_space(1))) P *;
template <typename P>
__device__ __forceinline__ global_ptr<P> opaque_uniform(P *p)
{
    unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>(p)));
    unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>(p) >> 32));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return reinterpret_cast<global_ptr<P>>((static_cast<unsigned long long>(hi) << 32) | lo);
}

template <typename TX, typename R, int NC>
__device__ __forceinline__ R sample_part(v2u_t v, int c)          // 8-byte sample: component c
{
    if constexpr (NC == 1) {
        static_assert(sizeof(TX) == 8 && sizeof(R) == 8, "one 8-byte real sample");
        return __builtin_bit_cast(double, v);
    } else {
        static_assert(sizeof(TX) == 4 && NC == 2, "one ComplexF32 sample");
        return static_cast<R>(__builtin_bit_cast(float, c == 0 ? v.x : v.y));
    }
}
template <typename R>
__device__ __forceinline__ R sample_of_pair(v2u_t v, int which)   // two Float32 samples in one 8-byte read
{
    return static_cast<R>(__builtin_bit_cast(float, which == 0 ? v.x : v.y));
}

// DMA: the sample rows go HBM -> LDS by LDS-DMA (global_load_lds, 16 bytes per lane: no staging registers, no ds_write); the
// rows then have a pitch of whole 16-byte chunks.  First / last tiles and partial channel groups are staged synchronously
// through registers.
// (experiment, profiles/r04/experiments.md: C4 4.97 ms with the 16-byte tap reads vs 4.83-4.9 without -- the LDS cost follows the
//  bytes, not the instruction count)
#ifndef MRHIP_AP_TD128
#define MRHIP_AP_TD128 0
#endif
// TEX > 0: tapsPerPhi == TEX exactly (a multiple of 4): the tap-pair pipeline is unrolled whole with immediate LDS offsets -- no
// running bases to advance (6 vector adds per two pairs), no loop counter.
template <typename TX, typename R, int NC, bool FUSED, int CPL, bool DMA, int TEX = 0>
__global__ __launch_bounds__(kPipeThreads, 4) void arb_pipe_kernel(ArbArgs a, ArbTileArgs ta)
{
    constexpr int ROWS = kPipeElems / CPL;
    constexpr unsigned SB = sizeof(TX) * NC;                          // bytes per sample: 8, or 4 (Float32)
    constexpr bool PAIR = SB == 4;                                    // one 8-byte read = the two samples of a tap pair; the tile is
                                                                      // kept twice, one sample apart (copy B: odd window starts)
    static_assert(SB == 8 || (SB == 4 && NC == 1), "8-byte samples, or Float32");
    using StageT = std::conditional_t<SB == 8, unsigned long long, unsigned>;
    // TD: Float64 taps -- tap i of a phase and its difference-bank partner sit side by side in LDS (16 bytes) and come with ONE
    // ds_read_b128: two tap reads per tap pair instead of four
    constexpr bool TD = sizeof(R) == 8 && MRHIP_AP_TD128 != 0;
    constexpr int NR = (TD ? 2 : 4) + (PAIR ? 1 : 2) * CPL;           // LDS reads per tap pair
    static_assert(NR <= 15, "lgkmcnt is a 4-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
    const int tid = threadIdx.x;
    const int T = a.T, TP = ta.tap_pitch, MS = ta.max_span;
    constexpr unsigned RS = sizeof(R);                                // bytes per tap
    using TapReg = typename TapBits<R>::type;
    R *const lpfb = reinterpret_cast<R *>(smem);
    R *const ldpfb = lpfb + ta.bank_elems;
    {   // both tap banks -> LDS once per workgroup: element (phi, i) at phi*TP + i
        const R *__restrict__ g0 = static_cast<const R *>(a.taps);
        const R *__restrict__ g1 = static_cast<const R *>(a.dtaps);
        const int total = a.Nphi * T;
        for (int e = tid; e < total; e += kPipeThreads) {
            const int phi = e / T, i = e - phi * T;
            if constexpr (TD) { lpfb[(phi * TP + i) * 2] = g0[e]; lpfb[(phi * TP + i) * 2 + 1] = g1[e]; }
            else { lpfb[phi * TP + i] = g0[e]; ldpfb[phi * TP + i] = g1[e]; }
        }
    }
    const int RP = DMA ? ta.row_pitch : MS;                          // row pitch in samples (DMA: whole 16-byte chunks)
    const unsigned copy_bytes = DMA ? static_cast<unsigned>(ta.dma_slots) * 1024u : static_cast<unsigned>(CPL) * static_cast<unsigned>(MS) * SB;
    const unsigned copyb_off = copy_bytes + static_cast<unsigned>(ta.copyb_pad) * SB;       // copy B behind copy A, 128 B round the banks
    const unsigned xbuf_bytes = PAIR ? copyb_off + copy_bytes : copy_bytes;
    const unsigned lx0 = lds0 + static_cast<unsigned>(ta.x_offset_bytes);        // sample buffer b at lx0 + b*xbuf_bytes: [CPL][MS] (x2)

    // tile -> (stretch of 256 outputs tau, channel group cg), time-major: the workgroups that run together share the schedule
    long long ngroups_ll;
    tiles_take_dyn(a.n_out, ta, ngroups_ll, a.dyn);                 // (a device-planned call: the count from the call record)
    const int ngroups = static_cast<int>(ngroups_ll);
    long long tile = blockIdx.x;
#ifdef MRHIP_AP_TRACE
    if (tid == 0 && blockIdx.x < 2048) g_ap_trace[g_ap_slot & 63][blockIdx.x][0] = wall_clock64();
#endif
    // Tiles are HANDED OUT in runs of ta.run_tiles consecutive tiles (ta.counters): a workgroup's first run is its index, every
    // further one the next the whole grid has not taken -- time-major like the static order, so the workgroups that run together
    // still share a stretch of the schedule.  With tile += gridDim every workgroup owns 1/grid of the tiles whatever happens to
    // it: the four workgroups of a CU do not advance evenly (the kernel's last third ran on CUs with three, two, one of them
    // left), and a workgroup that is placed late -- the next call's schedule runs beside this kernel -- ends late by as much
    // (profiles/r04/experiments.md S).  Runs, not tiles: a device-scope atomic on one address takes ~13 ns, 650 000 of them are
    // the kernel.  A run's index is needed before the run in front of it ends (the tile after next is prefetched): lane 0 asks
    // at the top of a tile, publishes behind the tile's staging wait, the barrier at the tile's end makes it everyone's; two
    // runs are always in hand (q0, q1).
    unsigned *const ctr = ta.counters;
    __shared__ unsigned s_grab[2];
    const long long G = gridDim.x;
    const int RUN = ctr ? ta.run_tiles : 1;
    auto leave = [&]() {                                        // every workgroup, the ones without a tile too
        if (ctr && tid == 0) {
            __threadfence();                                      // (this workgroup's requests are in before it counts itself off)
            if (atomicAdd(ctr + 64, 1u) == static_cast<unsigned>(G) - 1u) {
                __threadfence();
                ctr[0] = 0u; ctr[64] = 0u;                        // re-armed for the next launch (stream order makes it visible)
            }
        }
    };
    if (ctr) tile *= RUN;
    if (tile >= ta.total_tiles) { leave(); dev::shiftin_by_last_workgroup<TX, NC>(a.fold, a.x, a.hist, a.x_stride, a.x_len, a.H, a.nch); return; }
    long long tau = tile / ngroups;
    int cg = static_cast<int>(tile - tau * ngroups);
    constexpr long long kNoRun = -1;
    long long q0 = kNoRun, q1 = kNoRun;                         // first tiles of the next two runs of this workgroup
    if (ctr) {
        if (tid == 0) { const unsigned b = atomicAdd(ctr, 2u); s_grab[0] = b; s_grab[1] = b + 1u; }
        __syncthreads();
        q0 = (G + s_grab[0]) * RUN; q1 = (G + s_grab[1]) * RUN;
    }
    auto after = [&](long long t_) -> long long {               // the tile this workgroup takes after t_
        if (!ctr) return t_ + G;
        if (((t_ + 1) & (RUN - 1)) != 0) return t_ + 1;        // (RUN is a power of two)
        const long long r_ = q0;
        q0 = q1; q1 = kNoRun;
        return r_;
    };
    long long t1 = after(tile), t2 = after(t1);                 // this workgroup's next tile and the one after it

    // n_idx[first output of a tile].  In the tile loop it is loaded TWO tiles ahead by an ordinary load and taken over into a scalar
    // behind the staging wait at the end of a tile, where nothing is in flight any more: left to a load at the top of the tile the
    // compiler waits s_waitcnt vmcnt(0) there -- the previous tile's output stores included, microseconds per tile.  (Round 3 first
    // issued it as an asynchronous s_load_dword from inline assembly and waited a tile later: the compiler, which takes an asm
    // output for valid at once, spilled and re-used that SGPR while the load was still in flight, and the landing data overwrote
    // whatever lived there -- whole tiles of zeros in workgroups that take more than one tile, for some instantiations only.)
    auto first_index_sync = [&](long long tau_) -> int {
        const int *p = a.n_idx + tau_ * kPipeThreads;
        const unsigned plo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>(p)));
        const unsigned phi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>(p) >> 32));
        const unsigned long long pu = (static_cast<unsigned long long>(phi) << 32) | plo;
        int v;
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(pu));   // (valid when the statement ends)
        return v;
    };
    struct Tile { long long k0, o; int nout, ch0, nchl, n_lo, interior; };
    auto make_tile = [&](long long tau_, int cg_, int n_lo_) {
        Tile t;
        t.ch0 = cg_ * CPL;
        t.nchl = a.nch - t.ch0 < CPL ? a.nch - t.ch0 : CPL;
        t.k0 = tau_ * kPipeThreads;
        const long long rem = a.n_out - t.k0;
        t.nout = rem < kPipeThreads ? static_cast<int>(rem) : kPipeThreads;
        t.n_lo = n_lo_;                                          // = n_idx[k0]
        t.o = static_cast<long long>(t.n_lo) - T;               // x[n_lo - T ...] (0-based); n = 1-based newest sample
        // every sample of the tile lies inside the signal for every channel of a full group: no per-lane checks
        t.interior = t.nchl == CPL && t.o >= 0 && t.o + MS <= a.x_len;
        return t;
    };
    // Staging: element j of a thread is sample r*256 + tid of channel cc (j = cc*ROWS + r).  The lane offsets are the same
    // for every tile (lanes past the span re-read its last sample: same cache line, no branch), the rest of the address is scalar.
    unsigned soff[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int sidx = r * kPipeThreads + tid;
        soff[r] = static_cast<unsigned>(sidx < MS ? sidx : MS - 1) * SB;
    }
    // DMA: a wave transfer moves 64 chunks of 16 bytes to 1 KiB of LDS; chunk c = slot*64 + lane is chunk c % row_chunks of row
    // c / row_chunks (chunks past the tile re-read chunk 0: they land in the padding behind the rows)
    constexpr int DSL = 4;                                            // slots per wave and copy (4 waves: 16 KiB per copy)
    unsigned dvoff[DSL];
    const int row_chunks = RP * static_cast<int>(SB) / 16, nslots = ta.dma_slots;
    if constexpr (DMA) {
#pragma unroll
        for (int q = 0; q < DSL; ++q) {
            const int c = ((tid >> 6) + 4 * q) * 64 + (tid & 63);
            const int cc = c / row_chunks, k = c - cc * row_chunks;
            dvoff[q] = cc < CPL ? static_cast<unsigned>(cc) * static_cast<unsigned>(a.x_stride) * SB + static_cast<unsigned>(k) * 16u : 0u;
        }
    }
    auto dma_ok = [&](const Tile &t) {                              // every chunk of every row (and of copy B) lies inside the signal
        return t.nchl == CPL && t.o >= 0 && t.o + RP + (PAIR ? 1 : 0) <= a.x_len;
    };
    auto dma_tile = [&](const Tile &t, int b) {
        const global_ptr<const unsigned char> base = opaque_uniform(static_cast<const unsigned char *>(a.x) + (static_cast<long long>(t.ch0) * a.x_stride + t.o) * static_cast<long long>(SB));
        unsigned char *const dst = smem + ta.x_offset_bytes + static_cast<size_t>(b) * xbuf_bytes;
#pragma unroll
        for (int q = 0; q < DSL; ++q) {
            const int slot = (tid >> 6) + 4 * q;                   // (wave-uniform)
            if (slot < nslots) {
                dev::dma16((const void *)(base + dvoff[q]), dst + static_cast<size_t>(slot) * 1024);
                if constexpr (PAIR) dev::dma16((const void *)(base + dvoff[q] + SB), dst + copyb_off + static_cast<size_t>(slot) * 1024);   // B[s] = sample s + 1
            }
        }
    };
    StageT pv[kPipeElems];                                        // the next tile's samples, as raw bits
    auto load_tile = [&](const Tile &t) {
        if (!DMA && t.interior) {
#pragma unroll
            for (int j = 0; j < kPipeElems; ++j) {
                // (unconditional, straight-line: a row past the span re-reads the span's last sample -- with a branch per load the
                // compiler waits for every earlier memory operation, the previous tile's stores included, in front of each)
                const int cc = j / ROWS, r = j - cc * ROWS;
                const global_ptr<const unsigned char> base = opaque_uniform(static_cast<const unsigned char *>(a.x) + (static_cast<long long>(t.ch0 + cc) * a.x_stride + t.o) * static_cast<long long>(SB));
                pv[j] = *reinterpret_cast<global_ptr<const StageT>>(base + soff[r]);
            }
        } else {                                                  // the first and last tiles of a channel group: history, zeros
#pragma unroll
            for (int j = 0; j < kPipeElems; ++j) {
                const int cc = j / ROWS, r = j - cc * ROWS;
                const int sidx = r * kPipeThreads + tid;
                const long long gi = t.o + sidx;
                const bool ok = cc < t.nchl && sidx < MS && gi < a.x_len && gi >= -static_cast<long long>(a.H);
                const StageT *px = static_cast<const StageT *>(a.x) + static_cast<long long>(t.ch0 + cc) * a.x_stride + gi;
                const StageT *ph = static_cast<const StageT *>(a.hist) + static_cast<long long>(t.ch0 + cc) * a.H + (a.H + gi);
                const StageT *p = gi >= 0 ? px : ph;
                const StageT v = *(ok ? p : static_cast<const StageT *>(a.taps));
                pv[j] = ok ? v : static_cast<StageT>(0);
            }
        }
    };
    auto store_tile = [&](int b) {
        StageT *const lx = reinterpret_cast<StageT *>(smem + ta.x_offset_bytes + static_cast<size_t>(b) * xbuf_bytes);
        StageT *const lxB = reinterpret_cast<StageT *>(smem + ta.x_offset_bytes + static_cast<size_t>(b) * xbuf_bytes + copyb_off);
#pragma unroll
        for (int j = 0; j < kPipeElems; ++j) {
            const int cc = j / ROWS, r = j - cc * ROWS;
            const int sidx = r * kPipeThreads + tid;
            if (r * kPipeThreads < MS && sidx < MS) {
                lx[cc * RP + sidx] = pv[j];
                if constexpr (PAIR) { if (sidx > 0) lxB[cc * RP + sidx - 1] = pv[j]; }        // B[s] = sample s + 1
            }
        }
    };

    const int n_lo0 = first_index_sync(tau);
    Tile cur = make_tile(tau, cg, n_lo0);
    load_tile(cur);
    int n_pre = 0;
    double acc_pre = 0.0;
    if (tid < cur.nout) { n_pre = a.n_idx[cur.k0 + tid]; acc_pre = a.acc[cur.k0 + tid]; }
    auto split = [&](long long t_, long long &tau_, int &c_) {   // tile index -> (stretch of outputs, channel group)
        tau_ = t_ / ngroups; c_ = static_cast<int>(t_ - tau_ * ngroups);
        return t_ < ta.total_tiles;
    };
    long long ntau = tau;
    int ncg = cg;
    bool have_next = t1 < ta.total_tiles;
    if (have_next) (void)split(t1, ntau, ncg);
    int n_lo_next = 0;
    if (have_next) n_lo_next = first_index_sync(ntau);
    store_tile(0);
    __syncthreads();       // the tap banks and the first tile are in LDS
    int buf = 0;
    unsigned it = 0;

    for (;;) {
        // the next tile's loads go out now and land while this tile is computed
        const int n_mine = n_pre;
        const double acc_mine = acc_pre;
        const Tile nxt = make_tile(have_next ? ntau : tau, have_next ? ncg : cg, have_next ? n_lo_next : cur.n_lo);
        bool by_dma = false;                                      // (uniform)
        if (have_next) {
            if constexpr (DMA) {
                by_dma = dma_ok(nxt);
                if (by_dma) dma_tile(nxt, buf ^ 1);
            } else {
                load_tile(nxt);
            }
            if (tid < nxt.nout) { n_pre = a.n_idx[nxt.k0 + tid]; acc_pre = a.acc[nxt.k0 + tid]; }
        }
        const bool refill = ctr && q1 == kNoRun;                 // (uniform) a run was taken into use: ask for another, publish it below
        unsigned grabbed = 0u;
        if (refill && tid == 0) grabbed = atomicAdd(ctr, 1u);
        long long n2tau = ntau;
        int n2cg = ncg;
        const bool have_next2 = have_next && t2 < ta.total_tiles;
        if (have_next2) (void)split(t2, n2tau, n2cg);
        int first2 = 0;                                         // n_idx[first output] of the tile after the next: taken over below
        if (have_next2) first2 = a.n_idx[n2tau * kPipeThreads];

        R res[CPL][NC];
        if (tid < cur.nout) {
            const double phif = __builtin_floor(acc_mine);
            const double alpha = acc_mine - phif;                 // src/Filters.jl:671-672
            const int phi = static_cast<int>(phif) - 1;           // 0-based column
            const int w = n_mine - cur.n_lo;                      // oldest sample of this output's window, within the tile
            unsigned tpa = lds0 + static_cast<unsigned>(phi * TP) * (TD ? 2u * RS : RS);  // taps of this phase
            unsigned dpa = tpa + static_cast<unsigned>(ta.bank_elems) * RS;              // ... of the difference bank (TD: unused)
            unsigned sa[CPL];
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) {
                if constexpr (PAIR) {     // an aligned pair read: even window starts from copy A, odd ones from copy B (= one sample later)
                    const unsigned odd = static_cast<unsigned>(w) & 1u;
                    sa[cc] = lx0 + static_cast<unsigned>(buf) * xbuf_bytes + (odd ? copyb_off : 0u) + (static_cast<unsigned>(cc * RP + w) - odd) * SB;
                } else {
                    sa[cc] = lx0 + static_cast<unsigned>(buf) * xbuf_bytes + static_cast<unsigned>(cc * RP + w) * SB;
                }
            }
            dev::pin(tpa); dev::pin(dpa);                         // (complete addresses in registers: the loop only adds to them)
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) dev::pin(sa[cc]);

            struct Blk { TapReg t0, t1, d0, d1; dev::v4u_t td0, td1; v2u_t s0[CPL], s1[PAIR ? 1 : CPL]; };
            auto issue = [&](Blk &b, auto off_c) {              // taps i, i + 1 at byte offset OFF from the running bases
                constexpr int OFF = decltype(off_c)::value;
                constexpr int TOFF = OFF / 8 * static_cast<int>(RS);           // the same tap pair in the tap banks
                constexpr int SOFF = OFF / 8 * static_cast<int>(SB);           // ... in the sample tile
                if constexpr (TD) {
                    b.td0 = dev::lds_read_b128<2 * TOFF>(tpa); b.td1 = dev::lds_read_b128<2 * TOFF + 16>(tpa);
                } else {
                    b.t0 = lds_read_tap<R, TOFF>(tpa); b.t1 = lds_read_tap<R, TOFF + static_cast<int>(RS)>(tpa);
                    b.d0 = lds_read_tap<R, TOFF>(dpa); b.d1 = lds_read_tap<R, TOFF + static_cast<int>(RS)>(dpa);
                }
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) {
                    b.s0[cc] = dev::lds_read_b64<SOFF>(sa[cc]);
                    if constexpr (!PAIR) b.s1[cc] = dev::lds_read_b64<SOFF + 8>(sa[cc]);
                }
            };
            auto landed = [&](Blk &b, auto n_c) {               // at most N later reads still in flight => b has landed
                constexpr int N = decltype(n_c)::value;
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
                if constexpr (TD) { dev::pin(b.td0); dev::pin(b.td1); }
                else { dev::pin(b.t0); dev::pin(b.t1); dev::pin(b.d0); dev::pin(b.d1); }
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) { dev::pin(b.s0[cc]); if constexpr (!PAIR) dev::pin(b.s1[cc]); }
            };
            // -0.0 + p == p for every p (signed zeros, NaN included): starting from -0.0 IS "the first product initialises"
            R lo[CPL][NC], up[CPL][NC];
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) {
#pragma unroll
                for (int c = 0; c < NC; ++c) { lo[cc][c] = static_cast<R>(-0.0); up[cc][c] = static_cast<R>(-0.0); }
            }
            auto compute = [&](const Blk &b) {
                R t0, t1, d0, d1;
                if constexpr (TD) {
                    t0 = __builtin_bit_cast(R, v2u_t{b.td0.x, b.td0.y}); d0 = __builtin_bit_cast(R, v2u_t{b.td0.z, b.td0.w});
                    t1 = __builtin_bit_cast(R, v2u_t{b.td1.x, b.td1.y}); d1 = __builtin_bit_cast(R, v2u_t{b.td1.z, b.td1.w});
                } else {
                    t0 = __builtin_bit_cast(R, b.t0); t1 = __builtin_bit_cast(R, b.t1);
                    d0 = __builtin_bit_cast(R, b.d0); d1 = __builtin_bit_cast(R, b.d1);
                }
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        R x0, x1;
                        if constexpr (PAIR) { x0 = sample_of_pair<R>(b.s0[cc], 0); x1 = sample_of_pair<R>(b.s0[cc], 1); }
                        else { x0 = sample_part<TX, R, NC>(b.s0[cc], c); x1 = sample_part<TX, R, NC>(b.s1[cc], c); }
                        lo[cc][c] = macd<FUSED>(t0, x0, lo[cc][c]);
                        up[cc][c] = macd<FUSED>(d0, x0, up[cc][c]);
                        lo[cc][c] = macd<FUSED>(t1, x1, lo[cc][c]);
                        up[cc][c] = macd<FUSED>(d1, x1, up[cc][c]);
                    }
                }
            };
            auto advance = [&](unsigned bytes) {                  // `bytes` of samples = bytes / 8 taps
                tpa += bytes / 8u * (TD ? 2u * RS : RS); dpa += bytes / 8u * RS;
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) sa[cc] += bytes / 8u * SB;
            };
            using I0 = std::integral_constant<int, 0>;
            using I16 = std::integral_constant<int, 16>;
            using I32 = std::integral_constant<int, 32>;
            using INR = std::integral_constant<int, NR>;
            const int nblk = T >> 1;                             // tap pairs (uniform)
            if constexpr (TEX > 0) {
                static_assert(TEX % 4 == 0 && TEX <= 64, "whole double pairs, offsets within the tile's pad");
                Blk A, B;
                issue(A, I0{});
                // (no read past the last pair here: the loop below reads one pair too many and retires it with a wait before its
                //  registers are touched again; unrolled, the compiler sees those results are never used, gives all of them ONE
                //  scratch register and reuses it at once -- the reads, still in flight, then land on top of an accumulator)
                dev::static_for<0, TEX / 4>([&](auto K) {
                    constexpr int o = decltype(K)::value * 32;
                    issue(B, std::integral_constant<int, o + 16>{});
                    landed(A, INR{});
                    compute(A);
                    if constexpr (decltype(K)::value + 1 < TEX / 4) {
                        issue(A, std::integral_constant<int, o + 32>{});
                        landed(B, INR{});
                    } else {
                        landed(B, I0{});
                    }
                    compute(B);
                });
            } else if (nblk > 0) {
                // Straight-line pipeline, no conditional issue (a conditional read makes the two register sets merge through
                // copies): the pair after the last one is read too -- it lies inside LDS (the next column, the pad behind the
                // sample buffers) and is never used.
                Blk A, B;
                issue(A, I0{});
                for (int p = 0; p < nblk; p += 2) {              // the bases point at pair p, A holds it
                    issue(B, I16{});
                    landed(A, INR{});
                    compute(A);
                    issue(A, I32{});
                    advance(32);
                    landed(B, INR{});
                    if (p + 1 < nblk) compute(B);
                }
                asm volatile("s_waitcnt lgkmcnt(0)");            // the unused last reads
                if constexpr (TD) dev::pin(A.td0); else dev::pin(A.t0);
            }
            if (TEX == 0 && (T & 1)) {                           // odd tapsPerPhi: the last tap alone
                const int i = T - 1;
                const unsigned tl = lds0 + static_cast<unsigned>(phi * TP + i) * (TD ? 2u * RS : RS);
                TapReg t = lds_read_tap<R, 0>(tl), d = lds_read_tap<R, 0>(tl + (TD ? RS : static_cast<unsigned>(ta.bank_elems) * RS));
                std::conditional_t<PAIR, unsigned, v2u_t> s[CPL];     // (copy A holds every sample at its own index)
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) {
                    const unsigned sad = lx0 + static_cast<unsigned>(buf) * xbuf_bytes + static_cast<unsigned>(cc * RP + w + i) * SB;
                    if constexpr (PAIR) s[cc] = dev::lds_read_b32<0>(sad);
                    else s[cc] = dev::lds_read_b64<0>(sad);
                }
                asm volatile("s_waitcnt lgkmcnt(0)");
                dev::pin(t); dev::pin(d);
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) dev::pin(s[cc]);
                const R tt = __builtin_bit_cast(R, t), dd = __builtin_bit_cast(R, d);
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        R x;
                        if constexpr (PAIR) x = static_cast<R>(__builtin_bit_cast(float, s[cc]));
                        else x = sample_part<TX, R, NC>(s[cc], c);
                        lo[cc][c] = macd<FUSED>(tt, x, lo[cc][c]);
                        up[cc][c] = macd<FUSED>(dd, x, up[cc][c]);
                    }
                }
            }
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const double prod = static_cast<double>(up[cc][c]) * alpha;   // Filters.jl:730, Float64 combine, rounded once
                    res[cc][c] = static_cast<R>(static_cast<double>(lo[cc][c]) + prod);
                }
            }
        }
        // The prefetched samples go to the other buffer BEFORE the outputs are stored: their wait (vmcnt) would otherwise
        // include the stores.  (The waves still computing this tile do not read that buffer.)
        asm volatile("" ::: "memory");   // (the LDS writes below stay below the hand-issued reads above: the compiler does not see those as memory operations)
        if (have_next) {
            buf ^= 1;
            if constexpr (DMA) {
                if (by_dma) {
                    // this wave's transfers and its schedule entries have landed (the barrier below: everyone's)
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(n_pre), "+v"(acc_pre)::"memory");
                } else {                                              // a first / last tile, a partial channel group: synchronously
                    load_tile(nxt);
                    store_tile(buf);
                }
            } else {
                store_tile(buf);
            }
        }
        if (refill && tid == 0) s_grab[it & 1] = grabbed;       // (its return has landed with the staging loads above)
        const int n_lo_next2 = __builtin_amdgcn_readfirstlane(first2);   // (its load has landed with the staging loads above)
        if (tid < cur.nout) {
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) {
                if (cc < cur.nchl) {
                    const global_ptr<unsigned char> yc = opaque_uniform(reinterpret_cast<unsigned char *>(static_cast<R *>(a.y) + (static_cast<long long>(cur.ch0 + cc) * a.y_stride + cur.k0) * NC));
                    if constexpr (NC == 2 && sizeof(R) == 4) {     // one ComplexF32 output: one 8-byte store
                        dev::v2f_t o2 = {res[cc][0], res[cc][1]};
                        *reinterpret_cast<global_ptr<dev::v2f_t>>(yc + static_cast<unsigned>(tid) * 8u) = o2;
                    } else {
#pragma unroll
                        for (int c = 0; c < NC; ++c) *reinterpret_cast<global_ptr<R>>(yc + static_cast<unsigned>(tid * NC + c) * RS) = res[cc][c];
                    }
                }
            }
        }
        if (!have_next) break;
        __syncthreads();       // one barrier per tile: the next tile is in LDS, and everyone is done with the buffer written after it
        tau = ntau; cg = ncg; cur = nxt;
        ntau = n2tau; ncg = n2cg; have_next = have_next2; n_lo_next = n_lo_next2;
        if (refill) q1 = (G + s_grab[it & 1]) * RUN;
        t1 = t2;
        t2 = after(t2);
        ++it;
    }
    leave();
    dev::shiftin_by_last_workgroup<TX, NC>(a.fold, a.x, a.hist, a.x_stride, a.x_len, a.H, a.nch);
#ifdef MRHIP_AP_TRACE
    if (tid == 0 && blockIdx.x < 2048) g_ap_trace[g_ap_slot & 63][blockIdx.x][1] = wall_clock64();
#endif
}

template <typename TX, typename R, int NC, bool DMA>
hipError_t launch_pipe_t(bool fused, const ArbArgs &a, const ArbTileArgs &ta, size_t lds, hipStream_t s, int num_cus)
{
    auto go = [&](auto kfn) -> hipError_t {
        int per_cu = 0;
        hipError_t eo = occupancy_cached(reinterpret_cast<const void *>(kfn), kPipeThreads, lds, &per_cu);
        if (eo != hipSuccess) return eo;
        if (per_cu < 1) per_cu = 1;
        const int bpc = MRHIP_ENV_INT("MRHIP_PIPE_BPC", 0);           // experiments: fewer workgroups per CU than fit
        if (bpc > 0 && bpc < per_cu) per_cu = bpc;
        long long g = static_cast<long long>(num_cus) * per_cu;
        if (g > ta.total_tiles) g = ta.total_tiles;
        if (g < 1) g = 1;
        if (MRHIP_ENV_INT("MRHIP_DEBUG", 0) == 1) {
            hipFuncAttributes fa;
            (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kfn));
            std::fprintf(stderr, "[mrhip] arb_pipe T=%d Nphi=%d cpl=%d grid=%lld lds=%zu occ/CU=%d regs=%d max_span=%d tiles=%lld\n",
                         a.T, a.Nphi, ta.cpl, g, lds, per_cu, fa.numRegs, ta.max_span, ta.total_tiles);
        }
#ifdef MRHIP_AP_TRACE
        {
            static unsigned slot = 0;
            static bool armed = false;
            static std::vector<long long> grids(64, 0);
            grids[slot & 63] = g;
            (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_ap_slot), &slot, sizeof(slot), 0, hipMemcpyHostToDevice, s);
            ++slot;
            if (!armed) {
                armed = true;
                std::atexit([] {
                    (void)hipDeviceSynchronize();
                    static unsigned long long h[64][2048][2];
                    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ap_trace), sizeof(h)) != hipSuccess) return;
                    for (unsigned l = 0; l < 64 && l < slot; ++l) {
                        const unsigned sl = (slot - 1 - l) & 63;
                        const long long gg = std::min<long long>(grids[sl], 2048);
                        if (gg < 2) continue;
                        unsigned long long s0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0;
                        int late = 0;
                        for (long long b = 0; b < gg; ++b) { s0 = std::min(s0, h[sl][b][0]); s1 = std::max(s1, h[sl][b][0]); e0 = std::min(e0, h[sl][b][1]); e1 = std::max(e1, h[sl][b][1]); }
                        for (long long b = 0; b < gg; ++b) late += h[sl][b][0] - s0 > 5000;     // 100 MHz clock: 50 us
                        std::fprintf(stderr, "[ap_trace] launch -%u grid=%lld: starts spread %.1f us (%d workgroups more than 50 us late), ends spread %.1f us, kernel %.1f us\n",
                                     l, gg, (s1 - s0) / 100.0, late, (e1 - e0) / 100.0, (e1 - s0) / 100.0);
                    }
                });
            }
        }
#endif
        // Hand-outs of tiles (ta.counters): for launches long enough to have a tail worth balancing and few enough requests for the
        // one address they all go to -- at least 64 tiles a workgroup, about 40 runs each (a power of two, 2 ... 32 tiles)
        ArbTileArgs tq = ta;
        const long long per_wg = ta.total_tiles / g;
        if (tq.counters && per_wg >= MRHIP_ENV_INT("MRHIP_PIPE_DYN_MIN", 64)) {
            int r = 2;
            while (r < 32 && per_wg / (2 * r) >= 40) r *= 2;
            const int env_r = MRHIP_ENV_INT("MRHIP_PIPE_RUN", 0);
            if (env_r >= 2 && (env_r & (env_r - 1)) == 0) r = env_r;
            tq.run_tiles = r;
        } else {
            tq.counters = nullptr;
        }
        launch_kernel(kfn, dim3(static_cast<unsigned>(g)), dim3(kPipeThreads), lds, s, a, tq);
        return hipGetLastError();
    };
    switch (ta.cpl) {
    case 4:
        // (four ComplexF32 channels per lane with Float64 arithmetic and register staging would spill: never planned, not built)
        if constexpr (!DMA && NC == 2 && sizeof(R) == 8) return hipErrorInvalidValue;
        else {
            if constexpr (DMA) {                                // BASELINE config 4's shape: 32 taps per phase, unrolled whole
                if (a.T == 32 && MRHIP_ENV_INT("MRHIP_ARB_EXACT", 1) != 0)
                    return fused ? go(arb_pipe_kernel<TX, R, NC, true, 4, DMA, 32>) : go(arb_pipe_kernel<TX, R, NC, false, 4, DMA, 32>);
            }
            return fused ? go(arb_pipe_kernel<TX, R, NC, true, 4, DMA>) : go(arb_pipe_kernel<TX, R, NC, false, 4, DMA>);
        }
    case 2: return fused ? go(arb_pipe_kernel<TX, R, NC, true, 2, DMA>) : go(arb_pipe_kernel<TX, R, NC, false, 2, DMA>);
    default: return fused ? go(arb_pipe_kernel<TX, R, NC, true, 1, DMA>) : go(arb_pipe_kernel<TX, R, NC, false, 1, DMA>);
    }
}

}  // namespace

// Eligible: Float64 arithmetic, 8-byte samples, tiles of 256 outputs whose sample span fits kPipeElems rows of 256 per
// thread (a decimating rate stretches the span: fewer channels per lane).  `span256` = the largest n[last] - n[first]
// over the aligned stretches of 256 outputs.
bool plan_arb_pipe(const TypeKey &tk, const ArbArgs &a, long long span256, ArbTileArgs *out, size_t *lds)
{
    if (MRHIP_ENV_INT("MRHIP_ARB_PIPE", 1) == 0) return false;
    const size_t sb = (tk.x_f64 ? 8 : 4) * (tk.complex_x ? 2 : 1);
    if (sb > 8 || a.n_out < 1) return false;               // (Float32 and 8-byte samples; ComplexF64 stays on arb_tiled_kernel)
    const size_t rs = tk.r_f64 ? 8 : 4;
    const int copies = sb == 4 ? 2 : 1;                    // Float32: the tile twice, one sample apart (aligned pair reads)
    const int TP = a.T | 1;                                // odd column pitch: lanes with different phases read different banks
    const size_t bank_elems = static_cast<size_t>(a.Nphi) * TP;
    const size_t banks_bytes = (2 * bank_elems * rs + 15) / 16 * 16;
    if (banks_bytes > 96 * 1024) return false;
    const long long max_span = (span256 + a.T + 1) / 2 * 2;
    int cpl = a.nch >= 32 ? 4 : (a.nch >= 8 ? 2 : 1);
    const int env_cpl = MRHIP_ENV_INT("MRHIP_ARB_CPL", 0);
    if (env_cpl == 1 || env_cpl == 2 || env_cpl == 4) cpl = env_cpl;
    while (cpl > 1 && max_span > static_cast<long long>(kPipeElems / cpl) * kPipeThreads) cpl /= 2;
    if (max_span > static_cast<long long>(kPipeElems / cpl) * kPipeThreads) return false;
    // copy B starts 128 B (mod 256) behind copy A: the lanes of one read that use it do not land on their neighbours' banks
    // LDS-DMA staging: rows of whole 16-byte chunks, a copy rounded up to whole 1 KiB wave transfers, at most 16 of them;
    // the lane offsets of the transfers are 32-bit: the channels of a group must lie within 2 GiB of each other
    const long long row_chunks = (max_span * static_cast<long long>(sb) + 15) / 16;
    const long long nslots = (row_chunks * cpl + 63) / 64;
    const bool dma = MRHIP_ENV_INT("MRHIP_PIPE_DMA", 1) != 0 && nslots <= 16 &&
                     static_cast<double>(cpl) * static_cast<double>(a.x_stride) * static_cast<double>(sb) < 2147483648.0;
    // (register staging of four ComplexF32 channels with Float64 arithmetic does not fit 128 VGPRs: that instantiation spills, and
    //  a spilled register of the hand-issued LDS pipeline would be saved before its data has landed -- two channels per lane then)
    if (!dma && tk.complex_x && tk.r_f64 && cpl > 2) cpl = 2;
    int copyb_pad = copies == 2 ? static_cast<int>((128 + 256 - (static_cast<size_t>(max_span) * sb * cpl) % 256) % 256 / sb) : 0;
    size_t buf_bytes = (static_cast<size_t>(max_span) * cpl * copies + copyb_pad) * sb;
    if (dma) {
        copyb_pad = copies == 2 ? static_cast<int>(128 / sb) : 0;            // (a copy is a multiple of 1 KiB)
        buf_bytes = static_cast<size_t>(nslots) * 1024 * copies + copyb_pad * sb;
    }
    const size_t total = banks_bytes + 2 * buf_bytes + 64;   // (+ pad: the pipeline reads one tap pair past a window)
    if (total > 150 * 1024) return false;
    ArbTileArgs ta{};
    ta.pipe = 1;
    ta.cpl = cpl;
    ta.tap_pitch = TP;
    ta.bank_elems = static_cast<int>(bank_elems);
    ta.x_offset_bytes = static_cast<int>(banks_bytes);
    ta.max_span = static_cast<int>(max_span);
    ta.copyb_pad = copyb_pad;
    ta.prefetch = dma ? 1 : 0;
    ta.row_pitch = static_cast<int>(row_chunks * 16 / static_cast<long long>(sb));
    ta.dma_slots = static_cast<int>(nslots);
    ta.tile_out = kPipeThreads;
    ta.tiles_per_channel = (a.n_out + kPipeThreads - 1) / kPipeThreads;
    ta.total_tiles = ta.tiles_per_channel * ((a.nch + cpl - 1) / cpl);
    *out = ta;
    *lds = total;
    return true;
}

hipError_t launch_arb_pipe(const TypeKey &tk, bool fused, const ArbArgs &a, const ArbTileArgs &ta, size_t lds, hipStream_t s,
                           const char **kname, int num_cus)
{
    *kname = "arb_pipe_kernel";
#define MRHIP_AP_GO(D)                                                                                                           \
    if (tk.complex_x) return tk.r_f64 ? launch_pipe_t<float, double, 2, D>(fused, a, ta, lds, s, num_cus) : launch_pipe_t<float, float, 2, D>(fused, a, ta, lds, s, num_cus);   \
    if (tk.x_f64) return launch_pipe_t<double, double, 1, D>(fused, a, ta, lds, s, num_cus);                                          \
    return tk.r_f64 ? launch_pipe_t<float, double, 1, D>(fused, a, ta, lds, s, num_cus) : launch_pipe_t<float, float, 1, D>(fused, a, ta, lds, s, num_cus);
    if (ta.prefetch) { MRHIP_AP_GO(true) }
    MRHIP_AP_GO(false)
#undef MRHIP_AP_GO
}

}  // namespace mrhip
