// kernels_farrow_pipe.hip -- FIRFarrow (src/Filters.jl:764-839) with Float64 arithmetic over 8-byte samples (Float64,
// ComplexF32) and at most 32 taps: the hand-scheduled form of farrow_tiled_kernel (kernels_arbitrary.hip), same
// arithmetic, bit-identical results.
//
// A workgroup takes 256 consecutive outputs (one per lane).  The lane evaluates ITS taps once (Float64 Horner, separately
// rounded multiply and add, rounded to the tap type: Polynomials.jl polyval + the store into currentTaps::Vector{Th}) into
// registers and then walks over all channels, CPL at a time: per channel group the run of samples the 256 outputs touch is
// staged in LDS and every lane forms CPL dot products.  Per tap and channel that is ONE 8-byte LDS read for one multiply
// and one add: LDS array and Float64 VALU are equally loaded, so the reads must cost exactly their 2 cycles --
//   * ds_read_b64 issued by hand (lane groups {0-31}, {32-63}, 64 banks; 32 consecutive outputs at a rate <= 1 touch 32
//     consecutive samples; the compiler's ds_read2_b64 / two-copy ds_read_b128 forms conflict or run at half rate, see
//     kernels_arb_pipe.hip), ONE copy of the tile;
//   * a ring of four register sets of one tap x CPL channels: the reads of the next three taps are in flight while this one is multiplied (counted
//     s_waitcnt lgkmcnt: LDS returns in order), fully unrolled with immediate offsets;
//   * two sample buffers: one barrier per channel group; the next group's samples (the next tile's first group behind the
//     last one) wait in registers while this one is computed; staging addresses are scalar base + lane offset.
// The accumulators start from -0.0 (x + -0.0 == x for every x: "the first product initialises"), from +0.0 on the seam
// (support.jl:46: the seam dot starts from zero).
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

constexpr int kFpThreads = 256;
constexpr int kFpElems = 8;          // 8-byte samples a thread stages per channel group: CPL channels x ROWS rows of 256 samples
constexpr int kFpDepth = 4;          // register sets: the reads of taps i + 1 and i + 2 are in flight while tap i is multiplied

template <bool FUSED, typename R>
__device__ __forceinline__ R fmacd(R t, R x, R acc)
{
    if constexpr (FUSED) {
        if constexpr (sizeof(R) == 4) return __builtin_fmaf(t, x, acc);
        else return __builtin_fma(t, x, acc);
    } else {
        const R p = t * x;
        return acc + p;
    }
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
__device__ __forceinline__ R fsample_part(v2u_t v, int c)
{
    if constexpr (NC == 1) {
        static_assert(sizeof(TX) == 8 && sizeof(R) == 8, "one 8-byte real sample");
        return __builtin_bit_cast(double, v);
    } else {
        static_assert(sizeof(TX) == 4 && NC == 2, "one ComplexF32 sample");
        return static_cast<R>(__builtin_bit_cast(float, c == 0 ? v.x : v.y));
    }
}

// EXACT: tapsPerPhi == TREG (no per-tap guards: the unrolled pipeline is one basic block)
// R = the arithmetic type: Float64, or Float32 (ComplexF32 samples x Float32 taps)
// DMA: the sample rows go HBM -> LDS by LDS-DMA (global_load_lds, 16 bytes per lane, no staging registers, no ds_write); the
// rows then have a pitch of whole 16-byte chunks.  First / last tiles and partial channel groups are staged synchronously
// through registers.
template <typename TX, typename R, int NC, bool FUSED, int CPL, int TREG, bool EXACT, bool DMA>
__global__ __launch_bounds__(kFpThreads, 3) void farrow_pipe_kernel(FarrowArgs a, ArbTileArgs ta)
{
    constexpr int ROWS = kFpElems / CPL;
    constexpr unsigned SB = sizeof(TX) * NC;                          // bytes per sample: 8, or 4 (Float32)
    constexpr bool PAIR = SB == 4;                                    // one 8-byte read = two Float32 samples: the tile is kept twice, one
                                                                      // sample apart (copy B serves the odd window starts)
    static_assert(SB == 8 || (SB == 4 && NC == 1), "8-byte samples, or Float32");
    using StageT = std::conditional_t<SB == 8, unsigned long long, unsigned>;
    constexpr int kFpSetTaps = PAIR ? 2 : 1;                          // taps per register set (= per 8-byte read and channel)
    constexpr int NSETS = (TREG + kFpSetTaps - 1) / kFpSetTaps;
    constexpr int NR = CPL;                                          // LDS reads per register set
    static_assert(NR * (kFpDepth - 1) <= 15, "lgkmcnt is a 4-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
    const int tid = threadIdx.x;
    const int T = a.T, P = a.polyorder, MS = ta.max_span;
    const int RP = DMA ? ta.row_pitch : MS;                         // row pitch in samples (DMA: whole 16-byte chunks)
    // one copy of the tile: [CPL][RP]; DMA rounds it up to whole 1 KiB wave transfers
    const unsigned copy_bytes = DMA ? static_cast<unsigned>(ta.dma_slots) * 1024u : static_cast<unsigned>(CPL) * static_cast<unsigned>(MS) * SB;
    const unsigned copyb_off = copy_bytes + static_cast<unsigned>(ta.copyb_pad) * SB;       // copy B behind copy A, 128 B round the banks
    const unsigned xbuf_bytes = PAIR ? copyb_off + copy_bytes : copy_bytes;                // sample buffer b at b*xbuf_bytes (x2)
    {
        long long ngroups;
        tiles_take_dyn(a.n_out, ta, ngroups, a.dyn);                // (a device-planned call: the count from the call record)
    }
    const long long ntiles = ta.total_tiles;
    const int ngroups = (a.nch + CPL - 1) / CPL;

    // Tiles (a stretch of 256 outputs, every channel group of it) are HANDED OUT when ta.counters is set, as in arb_pipe_kernel: the
    // first is the workgroup's index, every further one the next the grid has not taken; lane 0 asks for the tile after the
    // tile after next at the top of a tile and publishes it behind the first staging wait, the tile's barriers make it everyone's.
    // (One tile per request: a tile is all channel groups here -- tens of microseconds -- so the requests are few.)
    unsigned *const ctr = DMA ? ta.counters : nullptr;          // (register staging: at the register limit as it is, stays static)
    __shared__ unsigned s_grab[2];
    const long long G = gridDim.x;
    auto leave = [&]() {                                        // every workgroup, the ones without a tile too
        if (ctr && tid == 0) {
            __threadfence();
            if (atomicAdd(ctr + 64, 1u) == static_cast<unsigned>(G) - 1u) {
                __threadfence();
                ctr[0] = 0u; ctr[64] = 0u;                        // re-armed for the next launch
            }
        }
    };
    long long tau = blockIdx.x;
    if (tau >= ntiles) { leave(); dev::shiftin_by_last_workgroup<TX, NC>(a.fold, a.x, a.hist, a.x_stride, a.x_len, a.H, a.nch); return; }
    long long t1 = 0, t2 = 0;                                   // handed out: this workgroup's next tile and the one after it
    if (ctr) {
        if (tid == 0) { const unsigned b = atomicAdd(ctr, 2u); s_grab[0] = b; s_grab[1] = b + 1u; }
        __syncthreads();
        t1 = G + s_grab[0]; t2 = G + s_grab[1];
    }
    unsigned it = 0;
    // the polynomial coefficients -> LDS once (read from global memory in the Horner loops they are ~160 dependent
    // vector loads per tile, each waited for: as long as all the dot products of the tile)
    double *const lcoef = reinterpret_cast<double *>(smem + ta.x_offset_bytes);
    for (int e = tid; e < T * (P + 1); e += kFpThreads) lcoef[e] = a.pnfb[e];

    // n_idx[first output of a tile]: loaded two tiles ahead by an ordinary load, taken over into a scalar behind the first staging
    // wait of a tile (see kernels_arb_pipe.hip: an asynchronous s_load from inline assembly is not safe, the compiler re-uses its SGPR)
    auto first_index_sync = [&](long long tau_) -> int {
        const int *p = a.n_idx + tau_ * kFpThreads;
        const unsigned plo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>(p)));
        const unsigned phi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>(p) >> 32));
        const unsigned long long pu = (static_cast<unsigned long long>(phi) << 32) | plo;
        int v;
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(pu));   // (valid when the statement ends)
        return v;
    };

    // Staging: element j of a thread is sample r*256 + tid of channel cc (j = cc*ROWS + r); lanes past the span re-read its
    // last sample (same cache line, no branch)
    unsigned soff[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int sidx = r * kFpThreads + tid;
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
    auto dma_ok = [&](long long o, int ch0) {                      // every chunk of every row (and of copy B) lies inside the signal
        return a.nch - ch0 >= CPL && o >= 0 && o + RP + (PAIR ? 1 : 0) <= a.x_len;
    };
    auto dma_group = [&](long long o, int ch0, int b) {
        const global_ptr<const unsigned char> base = opaque_uniform(static_cast<const unsigned char *>(a.x) + (static_cast<long long>(ch0) * a.x_stride + o) * static_cast<long long>(SB));
        unsigned char *const dst = smem + static_cast<size_t>(b) * xbuf_bytes;
#pragma unroll
        for (int q = 0; q < DSL; ++q) {
            const int slot = (tid >> 6) + 4 * q;                   // (wave-uniform)
            if (slot < nslots) {
                dev::dma16((const void *)(base + dvoff[q]), dst + static_cast<size_t>(slot) * 1024);
                if constexpr (PAIR) dev::dma16((const void *)(base + dvoff[q] + SB), dst + copyb_off + static_cast<size_t>(slot) * 1024);   // B[s] = sample s + 1
            }
        }
    };
    StageT pv[kFpElems];
    auto load_group = [&](long long o, int ch0) {                  // samples x[o ..] of channels ch0 .. ch0 + CPL - 1
        const int nchl = a.nch - ch0 < CPL ? a.nch - ch0 : CPL;
        const bool interior = !DMA && nchl == CPL && o >= 0 && o + MS <= a.x_len;
        if (interior) {
#pragma unroll
            for (int j = 0; j < kFpElems; ++j) {
                const int cc = j / ROWS, r = j - cc * ROWS;
                const global_ptr<const unsigned char> base = opaque_uniform(static_cast<const unsigned char *>(a.x) + (static_cast<long long>(ch0 + cc) * a.x_stride + o) * static_cast<long long>(SB));
                pv[j] = *reinterpret_cast<global_ptr<const StageT>>(base + soff[r]);
            }
        } else {                                                  // the first and last tiles, the last channel group: history, zeros
#pragma unroll
            for (int j = 0; j < kFpElems; ++j) {
                const int cc = j / ROWS, r = j - cc * ROWS;
                const int sidx = r * kFpThreads + tid;
                const long long gi = o + sidx;
                const bool ok = cc < nchl && sidx < MS && gi < a.x_len && gi >= -static_cast<long long>(a.H);
                const StageT *px = static_cast<const StageT *>(a.x) + static_cast<long long>(ch0 + cc) * a.x_stride + gi;
                const StageT *ph = static_cast<const StageT *>(a.hist) + static_cast<long long>(ch0 + cc) * a.H + (a.H + gi);
                const StageT *p = gi >= 0 ? px : ph;
                const StageT v = *(ok ? p : reinterpret_cast<const StageT *>(a.pnfb));
                pv[j] = ok ? v : static_cast<StageT>(0);
            }
        }
    };
    auto store_group = [&](int b) {
        StageT *const lx = reinterpret_cast<StageT *>(smem + static_cast<size_t>(b) * xbuf_bytes);
        StageT *const lxB = reinterpret_cast<StageT *>(smem + static_cast<size_t>(b) * xbuf_bytes + copyb_off);
#pragma unroll
        for (int j = 0; j < kFpElems; ++j) {
            const int cc = j / ROWS, r = j - cc * ROWS;
            const int sidx = r * kFpThreads + tid;
            if (r * kFpThreads < MS && sidx < MS) {
                lx[cc * RP + sidx] = pv[j];
                if constexpr (PAIR) { if (sidx > 0) lxB[cc * RP + sidx - 1] = pv[j]; }        // B[s] = sample s + 1
            }
        }
    };

    int n_lo = first_index_sync(tau);
    int n_lo_next = (ctr ? t1 : tau + G) < ntiles ? first_index_sync(ctr ? t1 : tau + G) : 0;
    int n_pre = 0;
    double ph_pre = 0.0;
    {
        const long long rem = a.n_out - tau * kFpThreads;
        if (tid < rem) { n_pre = a.n_idx[tau * kFpThreads + tid]; ph_pre = a.acc[tau * kFpThreads + tid]; }
    }
    load_group(static_cast<long long>(n_lo) - T, 0);                // (the first group: through registers in either form)
    store_group(0);
    __syncthreads();
    int buf = 0;

    for (;;) {                                                    // tiles of 256 outputs
        const long long k0 = tau * kFpThreads;
        const long long rem = a.n_out - k0;
        const int nout = rem < kFpThreads ? static_cast<int>(rem) : kFpThreads;
        const long long o = static_cast<long long>(n_lo) - T;    // x[n_lo - T ...] (0-based); n = 1-based newest sample
        const long long ntau = ctr ? t1 : tau + G;
        const bool have_next_tile = ntau < ntiles;
        const long long n2tau = ctr ? t2 : ntau + G;
        int first2 = 0;                                           // n_idx[first output] of the tile after the next: taken over below
        if (n2tau < ntiles) first2 = a.n_idx[n2tau * kFpThreads];
        int n_lo_next2 = 0;
        const bool have = tid < nout;
        const int n = n_pre;
        const double phase = ph_pre;
        if (have_next_tile) {
            const long long remn = a.n_out - ntau * kFpThreads;
            if (tid < remn) { n_pre = a.n_idx[ntau * kFpThreads + tid]; ph_pre = a.acc[ntau * kFpThreads + tid]; }
        }
        R treg[TREG];
        if (have) {
            // Horner in Float64, separately rounded multiply and add (Polynomials.jl polyval: y = p[end]; y = p[i] + x*y) -- the
            // degree in the OUTER loop, the taps unrolled inside it: every step then issues the coefficient reads of all taps
            // together and runs TREG independent multiply-add chains.  (Tap by tap, as rounds 2-3 had it, each of the
            // tapsPerPhi x polyorder steps waited for its own LDS read: ~5 000 cycles of latency per tile, which a 64-channel
            // launch hides behind its dot products and a one-channel launch -- the reference's own FIRFarrow benchmark shape,
            // examples/Arb-Farrow Speed Comparison.jl -- does not: 0.21 ms per 1e7 samples against 0.07 for FIRArbitrary.)
            double yv[TREG];
#pragma unroll
            for (int i = 0; i < TREG; ++i) yv[i] = (EXACT || i < T) ? lcoef[i * (P + 1) + P] : 0.0;
            for (int j = P - 1; j >= 0; --j) {
#pragma unroll
                for (int i = 0; i < TREG; ++i) {
                    if (EXACT || i < T) {
                        const double t = phase * yv[i];
                        yv[i] = lcoef[i * (P + 1) + j] + t;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < TREG; ++i)
                treg[i] = (EXACT || i < T) ? (a.tap_f32 ? static_cast<R>(static_cast<float>(yv[i])) : static_cast<R>(yv[i])) : static_cast<R>(0);
        }
        const bool seam = n < a.seam_below;                       // kernel.xIdx < kernel.tapsPer𝜙, Filters.jl:818 (never in a piece that continues a call)
        const R acc0 = seam ? static_cast<R>(0.0) : static_cast<R>(-0.0);
        const int w = have ? n - n_lo : 0;                        // oldest sample of this output's window, within the tile

        for (int cg = 0; cg < ngroups; ++cg) {
            const int ch0 = cg * CPL;
            const int nchl = a.nch - ch0 < CPL ? a.nch - ch0 : CPL;
            const bool last = cg + 1 == ngroups;
            const bool have_next = !last || have_next_tile;
            // the next group's loads go out now and land while this group is computed
            long long o_nx = o;
            int ch_nx = ch0 + CPL;
            if (last && have_next_tile) {
                o_nx = static_cast<long long>(n_lo_next) - T;
                ch_nx = 0;
            }
            bool by_dma = false;                                  // (uniform)
            if (have_next) {
                if constexpr (DMA) {
                    by_dma = dma_ok(o_nx, ch_nx);
                    if (by_dma) dma_group(o_nx, ch_nx, buf ^ 1);
                } else {
                    load_group(o_nx, ch_nx);
                }
            }

            R res[CPL][NC];
            if (have) {
                unsigned sa[CPL];
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) {
                    if constexpr (PAIR) {     // an aligned pair read: even window starts from copy A, odd ones from copy B (= one sample later)
                        const unsigned odd = static_cast<unsigned>(w) & 1u;
                        sa[cc] = lds0 + static_cast<unsigned>(buf) * xbuf_bytes + (odd ? copyb_off : 0u) + (static_cast<unsigned>(cc * RP + w) - odd) * SB;
                    } else {
                        sa[cc] = lds0 + static_cast<unsigned>(buf) * xbuf_bytes + static_cast<unsigned>(cc * RP + w) * SB;
                    }
                    dev::pin(sa[cc]);
                }
                struct Set { v2u_t s[CPL]; };                    // one 8-byte read per channel: one sample, or two Float32 samples
                auto issue = [&](Set &q, auto set_c) {           // the samples of the taps of register set SET, every channel
                    constexpr int SET = decltype(set_c)::value;
#pragma unroll
                    for (int cc = 0; cc < CPL; ++cc) q.s[cc] = dev::lds_read_b64<SET * 8>(sa[cc]);
                };
                auto landed = [&](Set &q, auto n_c) {            // at most N later reads still in flight => q has landed
                    constexpr int N = decltype(n_c)::value;
                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
#pragma unroll
                    for (int cc = 0; cc < CPL; ++cc) dev::pin(q.s[cc]);
                };
                R acc[CPL][NC];
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) acc[cc][c] = acc0;
                }
                auto compute = [&](const Set &q, auto set_c) {
                    constexpr int SET = decltype(set_c)::value;
                    dev::static_for<0, kFpSetTaps>([&](auto t_c) {
                        constexpr int TT = decltype(t_c)::value;
                        constexpr int I = SET * kFpSetTaps + TT;
                        if constexpr (I < TREG) {
                            if (EXACT || I < T) {                 // (uniform)
#pragma unroll
                                for (int cc = 0; cc < CPL; ++cc) {
#pragma unroll
                                    for (int c = 0; c < NC; ++c) {
                                        R xs;
                                        if constexpr (PAIR) xs = static_cast<R>(__builtin_bit_cast(float, TT == 0 ? q.s[cc].x : q.s[cc].y));
                                        else xs = fsample_part<TX, R, NC>(q.s[cc], c);
                                        acc[cc][c] = fmacd<FUSED, R>(treg[I], xs, acc[cc][c]);
                                    }
                                }
                            }
                        }
                    });
                };
                // straight-line pipeline over a ring of kFpDepth register sets (reads past the window stay inside LDS: the pad
                // behind the buffers; they are never used)
                Set ring[kFpDepth];
                dev::static_for<0, (kFpDepth - 1 < NSETS ? kFpDepth - 1 : NSETS)>([&](auto s_c) { issue(ring[decltype(s_c)::value], s_c); });
                dev::static_for<0, NSETS>([&](auto s_c) {
                    constexpr int S = decltype(s_c)::value;
                    // wait for set S (the sets issued behind it stay in flight), reuse the register set freed one step ago
                    constexpr int behind = (NSETS - 1 - S) < (kFpDepth - 2) ? (NSETS - 1 - S) : (kFpDepth - 2);
                    landed(ring[S % kFpDepth], std::integral_constant<int, behind * NR>{});
                    if constexpr (S + kFpDepth - 1 < NSETS) issue(ring[(S + kFpDepth - 1) % kFpDepth], std::integral_constant<int, S + kFpDepth - 1>{});
                    compute(ring[S % kFpDepth], s_c);
                });
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) res[cc][c] = acc[cc][c];
                }
            }
            // The prefetched samples go to the other buffer BEFORE the outputs are stored: their wait (vmcnt) would otherwise
            // include the stores.  (The waves still computing this group do not read that buffer.)
            asm volatile("" ::: "memory");   // (the LDS writes below stay below the hand-issued reads above)
            // the tile after t2: asked for behind the first channel group's arithmetic (a register that lived through it would push
            // the widest instantiations into scratch), answered with that group's staging wait
            unsigned grabbed = 0u;
            if (cg == 0 && ctr && tid == 0) grabbed = atomicAdd(ctr, 1u);
            if (have_next) {
                buf ^= 1;
                if constexpr (DMA) {
                    if (by_dma) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's transfers have landed (the barrier below: everyone's)
                    } else {                                                  // a first / last tile, a partial channel group: synchronously
                        load_group(o_nx, ch_nx);
                        store_group(buf);
                    }
                } else {
                    store_group(buf);
                }
            }
            if (cg == 0) {
                n_lo_next2 = __builtin_amdgcn_readfirstlane(first2);   // (its load has landed with the staging above)
                if (ctr && tid == 0) s_grab[it & 1] = grabbed;        // (and so has the request's answer)
            }
            if (have) {
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) {
                    if (cc < nchl) {
                        const global_ptr<unsigned char> yc = opaque_uniform(reinterpret_cast<unsigned char *>(static_cast<R *>(a.y) + (static_cast<long long>(ch0 + cc) * a.y_stride + k0) * NC));
                        if constexpr (NC == 2 && sizeof(R) == 4) {     // one ComplexF32 output: one 8-byte store
                            dev::v2f_t o2 = {res[cc][0], res[cc][1]};
                            *reinterpret_cast<global_ptr<dev::v2f_t>>(yc + static_cast<unsigned>(tid) * 8u) = o2;
                        } else {
#pragma unroll
                            for (int c = 0; c < NC; ++c) *reinterpret_cast<global_ptr<R>>(yc + static_cast<unsigned>(tid * NC + c) * static_cast<unsigned>(sizeof(R))) = res[cc][c];
                        }
                    }
                }
            }
            if (have_next) __syncthreads();   // one barrier per channel group
        }
        if (!have_next_tile) break;
        tau = ntau;
        n_lo = n_lo_next;
        n_lo_next = n_lo_next2;
        if (ctr) {
            t1 = t2;
            t2 = G + s_grab[it & 1];
            ++it;
        }
    }
    leave();
    dev::shiftin_by_last_workgroup<TX, NC>(a.fold, a.x, a.hist, a.x_stride, a.x_len, a.H, a.nch);
}

template <typename TX, typename R, int NC>
hipError_t launch_fpipe_t(bool fused, const FarrowArgs &a, const ArbTileArgs &ta, size_t lds, hipStream_t s, int num_cus)
{
    auto go = [&](auto kfn) -> hipError_t {
        int per_cu = 0;
        hipError_t eo = occupancy_cached(reinterpret_cast<const void *>(kfn), kFpThreads, lds, &per_cu);
        if (eo != hipSuccess) return eo;
        if (per_cu < 1) per_cu = 1;
        long long g = static_cast<long long>(num_cus) * per_cu;
        if (g > ta.total_tiles) g = ta.total_tiles;
        if (g < 1) g = 1;
        if (MRHIP_ENV_INT("MRHIP_DEBUG", 0) == 1) {
            hipFuncAttributes fa;
            (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kfn));
            std::fprintf(stderr, "[mrhip] farrow_pipe T=%d P=%d cpl=%d grid=%lld lds=%zu occ/CU=%d regs=%d max_span=%d tiles=%lld\n",
                         a.T, a.polyorder, ta.cpl, g, lds, per_cu, fa.numRegs, ta.max_span, ta.total_tiles);
        }
        ArbTileArgs tq = ta;                                       // hand-outs only where a workgroup has a few tiles to balance
        if (tq.counters && ta.total_tiles / g < MRHIP_ENV_INT("MRHIP_PIPE_DYN_MIN_F", 8)) tq.counters = nullptr;
        launch_kernel(kfn, dim3(static_cast<unsigned>(g)), dim3(kFpThreads), lds, s, a, tq);
        return hipGetLastError();
    };
#define MRHIP_FP_GO(C, D)                                                                                         \
    if (a.T == 32) return fused ? go(farrow_pipe_kernel<TX, R, NC, true, C, 32, true, D>) : go(farrow_pipe_kernel<TX, R, NC, false, C, 32, true, D>);   \
    return a.T <= 16 ? (fused ? go(farrow_pipe_kernel<TX, R, NC, true, C, 16, false, D>) : go(farrow_pipe_kernel<TX, R, NC, false, C, 16, false, D>))  \
                     : (fused ? go(farrow_pipe_kernel<TX, R, NC, true, C, 32, false, D>) : go(farrow_pipe_kernel<TX, R, NC, false, C, 32, false, D>));
    if (ta.prefetch) {      // LDS-DMA staging
        switch (ta.cpl) {
        case 4: MRHIP_FP_GO(4, true)
        case 2: MRHIP_FP_GO(2, true)
        default: MRHIP_FP_GO(1, true)
        }
    }
    switch (ta.cpl) {
    case 4: MRHIP_FP_GO(4, false)
    case 2: MRHIP_FP_GO(2, false)
    default: MRHIP_FP_GO(1, false)
    }
#undef MRHIP_FP_GO
}

}  // namespace

// Eligible: Float64 arithmetic, 8-byte samples, at most 32 taps, tiles of 256 outputs whose sample span fits kFpElems rows
// of 256 per thread (a decimating rate stretches the span: fewer channels per lane).
bool plan_farrow_pipe(const TypeKey &tk, const FarrowArgs &a, long long span256, ArbTileArgs *out, size_t *lds)
{
    if (MRHIP_ENV_INT("MRHIP_FARROW_PIPE", 1) == 0) return false;
    const size_t sb = (tk.x_f64 ? 8 : 4) * (tk.complex_x ? 2 : 1);
    if (sb > 8 || a.n_out < 1 || a.T > 32) return false;     // (Float32 and 8-byte samples; ComplexF64 stays on farrow_tiled_kernel)
    const int copies = sb == 4 ? 2 : 1;                      // Float32: the tile twice, one sample apart (aligned pair reads)
    const long long max_span = (span256 + a.T + 1) / 2 * 2;
    int cpl = a.nch >= 4 ? 4 : (a.nch >= 2 ? 2 : 1);
    while (cpl > 1 && max_span > static_cast<long long>(kFpElems / cpl) * kFpThreads) cpl /= 2;
    if (max_span > static_cast<long long>(kFpElems / cpl) * kFpThreads) return false;
    // LDS-DMA staging: rows of whole 16-byte chunks, a copy rounded up to whole 1 KiB wave transfers, at most 16 of them;
    // the lane offsets of the transfers are 32-bit: the channels of a group must lie within 2 GiB of each other
    const long long row_chunks = (max_span * static_cast<long long>(sb) + 15) / 16;
    const long long nslots = (row_chunks * cpl + 63) / 64;
    const bool dma = MRHIP_ENV_INT("MRHIP_PIPE_DMA", 1) != 0 && nslots <= 16 &&
                     static_cast<double>(cpl) * static_cast<double>(a.x_stride) * static_cast<double>(sb) < 2147483648.0;
    int copyb_pad = copies == 2 ? static_cast<int>((128 + 256 - (static_cast<size_t>(max_span) * sb * cpl) % 256) % 256 / sb) : 0;
    size_t buf_bytes = (static_cast<size_t>(max_span) * cpl * copies + copyb_pad) * sb;
    if (dma) {
        copyb_pad = copies == 2 ? static_cast<int>(128 / sb) : 0;            // (a copy is a multiple of 1 KiB)
        buf_bytes = static_cast<size_t>(nslots) * 1024 * copies + copyb_pad * sb;
    }
    // (+ pad: the pipeline reads whole register sets, up to 33 samples from a window's start)
    const size_t coef_off = (2 * buf_bytes + 320 + 15) / 16 * 16;
    const size_t total = coef_off + static_cast<size_t>(a.T) * (a.polyorder + 1) * 8;
    if (total > 150 * 1024) return false;
    ArbTileArgs ta{};
    ta.pipe = 1;
    ta.cpl = cpl;
    ta.max_span = static_cast<int>(max_span);
    ta.copyb_pad = copyb_pad;
    ta.prefetch = dma ? 1 : 0;
    ta.row_pitch = static_cast<int>(row_chunks * 16 / static_cast<long long>(sb));
    ta.dma_slots = static_cast<int>(nslots);
    ta.x_offset_bytes = static_cast<int>(coef_off);   // (here: where the polynomial coefficients live)
    ta.tile_out = kFpThreads;
    ta.tiles_per_channel = (a.n_out + kFpThreads - 1) / kFpThreads;
    ta.total_tiles = ta.tiles_per_channel;            // a tile covers all channels
    *out = ta;
    *lds = total;
    return true;
}

hipError_t launch_farrow_pipe(const TypeKey &tk, bool fused, const FarrowArgs &a, const ArbTileArgs &ta, size_t lds, hipStream_t s,
                              const char **kname, int num_cus)
{
    *kname = "farrow_pipe_kernel";
    if (tk.complex_x) return tk.r_f64 ? launch_fpipe_t<float, double, 2>(fused, a, ta, lds, s, num_cus) : launch_fpipe_t<float, float, 2>(fused, a, ta, lds, s, num_cus);
    if (tk.x_f64) return launch_fpipe_t<double, double, 1>(fused, a, ta, lds, s, num_cus);
    return tk.r_f64 ? launch_fpipe_t<float, double, 1>(fused, a, ta, lds, s, num_cus) : launch_fpipe_t<float, float, 1>(fused, a, ta, lds, s, num_cus);
}

}  // namespace mrhip
