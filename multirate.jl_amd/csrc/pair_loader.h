// pair_loader.h -- the loader wave shared by the ring-pipelined FIRRational kernels (kernels_rational_opair.hip: two outputs per lane):
// grouped dynamic scheduling of the steps, tiles staged HBM -> LDS by LDS-DMA ns-1 tiles ahead of the compute waves,
// tile descriptors published through LDS, counters re-armed by the last workgroup, shiftin! fused at the end.
// gfx950 only.
#ifndef MRHIP_PAIR_LOADER_H
#define MRHIP_PAIR_LOADER_H

#include "mrhip_internal.h"
#include "pair_device.h"

namespace mrhip {
namespace dev {

struct TileAt { int ch, st, jt; };                      // channel, first step within the channel, steps

// step number (channel-major) -> tile: multiply-high by floor(2^32/spc) + fix-ups
__device__ __forceinline__ TileAt pair_tile_at(const PairArgs &pa, unsigned g, unsigned jt)
{
    const unsigned spc = pa.steps_per_channel;
    unsigned q = __umulhi(g, pa.spc_magic);
    unsigned r = g - q * spc;
    if (r >= spc) { ++q; r -= spc; }
    if (r >= spc) { ++q; r -= spc; }
    return TileAt{static_cast<int>(q), static_cast<int>(r), static_cast<int>(umin(jt, spc - r))};
}

// A device-planned call (PolyArgs::dyn, mrhip_internal.h: DevCall): the output count, the call-start (u0, d0) and the step
// walk come from the record the call's plan kernel filled; the kernel arguments hold upper bounds the launch was sized with.
__device__ __forceinline__ void pair_take_dyn(PolyArgs &a, PairArgs &pa)
{
#ifdef MRHIP_NO_TAKE_DYN        // (A/B builds only: profiles/r04/experiments.md A)
    return;
#endif
    if (a.multi) {
        // several independent streams in one launch: this workgroup belongs to stream blockIdx.x % multi_n (= its scheduling
        // group: pa.ngroups == multi_n, grabs dealt round-robin among the stream's workgroups) and to nobody else
        const MultiDesc *__restrict__ d = a.multi + (blockIdx.x % static_cast<unsigned>(a.multi_n));
        a.x = d->x; a.y = d->y; a.hist = d->hist; a.hist_new = d->hist_new; a.taps = d->taps; a.rec = d->rec;
        a.x_stride = d->x_stride; a.y_stride = d->y_stride; a.x_len = d->x_len; a.n_out = d->n_out;
        a.u0 = d->u0; a.d0 = d->d0; a.phi_end = d->phi_end; a.d_end = d->d_end; a.nch = d->nch;
        pa.o0 = d->d0 - a.T;
        if (d->P_blk > 0) { pa.P = d->P_blk; pa.q0 = d->q0; pa.o0 += d->q0; }   // a block of the period (L > 512): its outputs, its sub-range of the input
        pa.steps_per_channel = d->steps_per_channel;
        pa.total_steps = d->total_steps;
        pa.spc_magic = d->spc_magic;
        pa.steps_per_group = d->total_steps;                  // the group IS the stream: steps [0, total_steps)
    }
    if (a.dyn) {
        const DevCall *__restrict__ d = a.dyn;
        a.n_out = d->n_out; a.u0 = d->u0; a.d0 = d->d0;
        a.x_len = d->x_len;                                   // (a chained call: the previous stage's count; else the launch's own value)
        pa.o0 = d->d0 - a.T;
        pa.steps_per_channel = d->steps_per_channel;
        pa.total_steps = d->total_steps;
        pa.spc_magic = d->spc_magic;
        pa.steps_per_group = (pa.total_steps + static_cast<unsigned>(pa.ngroups) - 1u) / static_cast<unsigned>(pa.ngroups);
    }
}

// The whole life of the loader wave (the last wave of the workgroup).  A tile of jt steps needs
// jt * pa.cM + pa.tail samples of one channel; NC = components per sample (1: Float32, 2: ComplexF32 / one Float64).
// FAST_SEAM = false (the PLAIN instantiations: big one-filter launches, where a channel's two seam tiles are nothing): the first / last
// tile of a channel is staged element by element as in rounds 1-4 -- the DMA variant below costs the headline 1.3 % (its loader code, A/B
// against the round-4 library) and buys it nothing.
template <int NC, bool FAST_SEAM = true>
__device__ __forceinline__ void pair_loader_wave(const PolyArgs &a, const PairArgs &pa, unsigned char *smem, int lane)
{
    volatile unsigned *const tile_flag = reinterpret_cast<volatile unsigned *>(smem + pa.flags_off);   // [ns][2]: first step, steps (0 = end)
    auto tile_at = [&](unsigned g, unsigned jt) -> TileAt { return pair_tile_at(pa, g, jt); };
    // ================= loader wave: HBM -> LDS, one tile ahead of the compute waves =================
    // It is the only wave that waits on vmcnt, so the compute waves' output stores stay in flight
    // across tiles (their barrier carries no memory wait).
    // Stages one tile; returns the number of LDS-DMA operations it left in flight (0 for the
    // checked register path, which drains everything before returning).
    const unsigned pad_magic = pa.pad_every > 0 ? 0xffffffffu / static_cast<unsigned>(pa.pad_every + 1) + 1u : 0u;   // ceil(2^32 / (cd + 1)) (cd + 1 is no power of two's divisor issue: cd + 1 >= 3 odd)
    // (pa.run_chunks = rc > 0: the workgroup owns a BLOCK of the period (L > 512) and stages, per step, only the run of samples its
    //  lanes' windows touch: ta.jt runs of rc chunks, pa.cM samples apart in the signal, packed in LDS)
    auto stage_tile = [&](const TileAt &ta, int stage) -> int {
        const int sch = ta.ch;
        constexpr int EPC = 4 / NC;                                 // samples per 16-byte DMA chunk
        const int rc = pa.run_chunks;
        const int tlen = rc > 0 ? ta.jt * rc * EPC : (ta.jt * pa.cM + pa.tail + EPC - 1) / EPC * EPC;   // samples this tile stages, whole chunks
        const int nchunks = tlen / EPC;
        // pa.pad_every = cd > 0: one 16-byte pad chunk after every cd data chunks (data chunk d lives at LDS chunk
        // d + d / cd), so that lanes whose runs start cd chunks apart hit different banks (kernels_fir_stream.hip)
        const int cd = pa.pad_every;
        const int nlds = cd > 0 ? (nchunks + cd - 1) / cd * (cd + 1) : nchunks;
        const int nslots = (nlds + 63) >> 6;                        // 1 KiB LDS slots
        const float *__restrict__ xc = static_cast<const float *>(a.x) + static_cast<long long>(sch) * a.x_stride * NC;
        const long long o = pa.o0 + static_cast<long long>(ta.st) * pa.cM;   // x index of LDS sample 0 (may be < 0)
        unsigned char *st = smem + static_cast<size_t>(stage) * pa.stage_bytes;
        const long long o_end = rc > 0 ? o + static_cast<long long>(ta.jt - 1) * pa.cM + rc * EPC : o + tlen;
        const bool interior = o >= 0 && o_end <= a.x_len;                // wave-uniform
        // chunk number within the staged tile -> sample offset from o (runs: chunk / rc by multiply-high, exact below 2^16 chunks)
        auto chunk_sample = [&](int c) -> long long {
            if (rc <= 0) return static_cast<long long>(EPC) * c;
            unsigned j = __umulhi(static_cast<unsigned>(c), pa.run_magic);
            int w = c - static_cast<int>(j) * rc;
            if (w < 0) { --j; w += rc; }
            return static_cast<long long>(j) * pa.cM + static_cast<long long>(EPC) * w;
        };
        if (interior) {
            const unsigned char *src = reinterpret_cast<const unsigned char *>(xc + o * NC);
            if (rc > 0) {                                                // runs (period blocks): chunk -> (run, chunk within the run)
                for (int slot = 0; slot < nslots; ++slot) {
                    const int ci = slot * 64 + lane;
                    const int cis = ci < nchunks ? ci : 0;               // (padding lanes re-read chunk 0)
                    dma16(src + static_cast<size_t>(chunk_sample(cis)) * (NC * 4u), st + static_cast<size_t>(slot) * 1024);
                }
                return nslots;
            }
            for (int slot = 0; slot < nslots; ++slot) {
                const int ci = slot * 64 + lane;
                int d = ci;
                // (the pad period is a run-time value: a multiply-high by ceil(2^32 / (cd + 1)), exact for ci < 2^16, instead of
                //  an integer division per lane and slot -- ~40 VALU instructions each, on the SIMD the loader shares with a
                //  compute wave: a third of C3b's VALU instructions were this, profiles/r04/item7/)
                if (cd > 0) { const int g = static_cast<int>(__umulhi(static_cast<unsigned>(ci), pad_magic)), r = ci - g * (cd + 1); d = r == cd ? 0 : g * cd + r; }
                const int cis = d < nchunks ? d : 0;                     // pad chunks and padding lanes re-read chunk 0
                dma16(src + static_cast<size_t>(cis) * 16, st + static_cast<size_t>(slot) * 1024);
            }
            return nslots;
        }
        // First / last tile of a channel: the history seam and the end of the input.  Every 16-byte chunk that lies inside x goes by
        // LDS-DMA like an interior tile's (the others from a dummy source); the few chunks at the seam and past the end are then put
        // right element by element.  (Round 4 staged the WHOLE tile element-wise: 2 000 chunks of dependent loads per seam tile -- most
        // of a one-channel 1e6-sample call.)
        if constexpr (FAST_SEAM) {
            const unsigned char *src = reinterpret_cast<const unsigned char *>(xc + o * NC);
            const unsigned char *dummy = static_cast<const unsigned char *>(a.taps);       // (library-owned, padded: upload_taps)
            for (int slot = 0; slot < nslots; ++slot) {
                const int ci = slot * 64 + lane;
                int d = ci;
                bool data = ci < nlds;
                if (cd > 0) { const int g = static_cast<int>(__umulhi(static_cast<unsigned>(ci), pad_magic)), r = ci - g * (cd + 1); data = data && r != cd; d = g * cd + r; }
                data = data && d < nchunks;
                const long long smp = data ? chunk_sample(d) : 0;
                const bool inside = data && o + smp >= 0 && o + smp + EPC <= a.x_len;
                dma16(inside ? src + smp * (NC * 4) : dummy, st + static_cast<size_t>(slot) * 1024);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const float *__restrict__ hc = static_cast<const float *>(a.hist) + static_cast<long long>(sch) * a.H * NC;
        float *l = reinterpret_cast<float *>(st);
        for (int ci = lane; ci < nchunks; ci += 64) {
            const long long g0 = o + chunk_sample(ci);
            if constexpr (FAST_SEAM) { if (g0 >= 0 && g0 + EPC <= a.x_len) continue; }   // staged above
            float4 v;
            float *pv = reinterpret_cast<float *>(&v);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const long long gi = g0 + e;
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) {
                    float val = 0.f;
                    if (gi >= 0) { if (gi < a.x_len) val = xc[gi * NC + cc]; }
                    else if (gi >= -static_cast<long long>(a.H)) val = hc[(a.H + gi) * NC + cc];
                    pv[e * NC + cc] = val;
                }
            }
            *reinterpret_cast<float4 *>(l + (cd > 0 ? ci + ci / cd : ci) * 4) = v;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        return 0;
    };
    // pa.ns LDS stages, the DMA runs ns-1 tiles ahead: while the compute waves work on tile i the loader
    // has tiles i+1 .. i+ns-2 landing and tile i+ns-1 being issued, so the HBM stream never pauses.
    // `hist` is a shift register of the LDS-DMA operation counts of the tiles issued so far (6 bits each,
    // newest in the low bits): tile i+1 has landed once no more operations are outstanding than the ns-2
    // newest tiles own.
    const unsigned grp = blockIdx.x % static_cast<unsigned>(pa.ngroups);
    const unsigned grp_lo = a.multi ? 0u : umin(grp * pa.steps_per_group, pa.total_steps);     // (multi: the stream's own step numbers)
    const unsigned grp_hi = umin(grp_lo + pa.steps_per_group, pa.total_steps);
    unsigned *const ctr = pa.counters + grp * 64u;            // one counter per 256 bytes
    unsigned pend = 0;                                        // lane 0: the grab number drawn ahead of need
    // small launches (at most a couple of grabs per workgroup) skip the atomics: grab numbers are dealt
    // round-robin from the workgroup number, which costs nothing and balances just as well
    unsigned static_next = blockIdx.x / static_cast<unsigned>(pa.ngroups);
    const unsigned static_stride = (gridDim.x + static_cast<unsigned>(pa.ngroups) - 1u - grp) / static_cast<unsigned>(pa.ngroups);
    auto grab_issue = [&]() {
        if (pa.static_grabs) { pend = static_next; static_next += static_stride; }
        else if (lane == 0) pend = atomicAdd(ctr, 1u);
    };
    unsigned ra = 0, rb = 0;                                  // the current grab's steps [ra, rb)
    bool more = true;                                         // false after the first empty grab
    auto grab_take = [&]() {                                  // the compiler waits for `pend` here (vmcnt(0))
        const unsigned t = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(pend)));
        const unsigned long long lo = static_cast<unsigned long long>(grp_lo) + static_cast<unsigned long long>(t) * pa.J;
        if (lo < grp_hi) { ra = static_cast<unsigned>(lo); rb = umin(ra + pa.J, grp_hi); grab_issue(); }
        else { more = false; ra = rb = 0; }
    };
    unsigned long long hist = 0;
    auto newest_ops = [&](int ntiles) -> int {
        int n = 0;
        for (int k = 0; k < ntiles; ++k) n += static_cast<int>((hist >> (6 * k)) & 63u);
        return n < 60 ? n : 60;               // the counter itself holds at most 63
    };
    // next tile of the stream -> stage `stage`; returns false at the end of the stream (end marker published)
    auto produce = [&](int stage) -> bool {
        if (ra >= rb && more) grab_take();
        if (ra >= rb) {
            if (lane == 0) { tile_flag[2 * stage] = 0u; tile_flag[2 * stage + 1] = 0u; }
            hist <<= 6;
            return false;
        }
        const TileAt ta = tile_at(ra, rb - ra);
        if (lane == 0) { tile_flag[2 * stage] = ra; tile_flag[2 * stage + 1] = static_cast<unsigned>(ta.jt); }
        hist = (hist << 6) | static_cast<unsigned>((pa.ablate & 1) ? 0 : stage_tile(ta, stage));
        ra += static_cast<unsigned>(ta.jt);
        return true;
    };
    grab_issue();
    unsigned pipeline = 0;                    // bit k: the tile opened k barriers from now exists
    for (int k = 0; k < pa.ns - 1; ++k)
        if (produce(k)) pipeline |= 1u << k;  // after the end of the stream produce() keeps publishing end markers
    if (pa.bank_off >= 0) __builtin_amdgcn_s_barrier();   // the compute waves' tap-bank barrier (the bank lives in stage ns-1, first written below after the next barrier)
    wait_vmcnt_le(newest_ops(pa.ns - 2));     // tile 0 has landed (only the later tiles' operations may remain)
    int pstage = pa.ns - 1;
#ifdef MRHIP_OPAIR_PROBE
    unsigned long long pr_bar = 0, pr_vm = 0;
    const unsigned long long pr_t0 = clock64();
#endif
    for (;;) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the tile descriptors published so far are in LDS
#ifdef MRHIP_OPAIR_PROBE
        const unsigned long long pr_a = clock64();
#endif
        __builtin_amdgcn_s_barrier();         // the next tile is published; the stage of the tile before it is free again
#ifdef MRHIP_OPAIR_PROBE
        pr_bar += clock64() - pr_a;
#endif
        if (!(pipeline & 1u)) break;          // that was the end marker: every wave of the workgroup leaves
        pipeline >>= 1;
        if (produce(pstage)) pipeline |= 1u << (pa.ns - 2);
        pstage = pstage + 1 == pa.ns ? 0 : pstage + 1;
#ifdef MRHIP_OPAIR_PROBE
        const unsigned long long pr_b = clock64();
#endif
        wait_vmcnt_le(newest_ops(pa.ns - 2)); // everything older than the ns-2 newest tiles has landed
#ifdef MRHIP_OPAIR_PROBE
        pr_vm += clock64() - pr_b;
#endif
    }
#ifdef MRHIP_OPAIR_PROBE
    if (pa.probe && lane == 0) {   // loader record (slot 7 of the workgroup): hardware id, total, barrier wait, vmcnt wait
        unsigned long long *r = pa.probe + (static_cast<unsigned long long>(blockIdx.x) * 8u + 7u) * 4u;
        const unsigned hw = __builtin_amdgcn_s_getreg(0xF804), xcc = __builtin_amdgcn_s_getreg(0xF814);
        r[0] = hw | (static_cast<unsigned long long>(xcc) << 32) | (1ULL << 63);
        r[1] = clock64() - pr_t0; r[2] = pr_bar; r[3] = pr_vm;
    }
#endif
    // the last workgroup to finish re-arms the counters for the next launch (stream order makes it visible)
    if (lane == 0 && !pa.static_grabs) {
        unsigned *const done = pa.counters + static_cast<unsigned>(pa.ngroups) * 64u;
        if (atomicAdd(done, 1u) == gridDim.x - 1) {
            for (int k = 0; k < pa.ngroups; ++k) pa.counters[k * 64] = 0u;
            *done = 0u;
        }
    }
    // the stream state on the device (mrhip_internal.h: DevStream): a call the host planned carries its end state
    // (Filters.jl:571-572, 627-628) in the arguments and one lane files it; a device-planned call's plan kernel already has
    const unsigned wg_in_grp = blockIdx.x / static_cast<unsigned>(pa.ngroups);                // (multi: this workgroup's number within its stream)
    if ((a.multi ? wg_in_grp == 0u : blockIdx.x == 0) && lane == 0 && a.rec && !a.dyn) {
        a.rec->phiIdx = a.phi_end;
        a.rec->inputDeficit = a.d_end;
        a.rec->n_written = a.n_out;
        a.rec->calls += 1;
    }
    // shiftin! (support.jl:61-80), fused: hist_new <- last H samples of [hist ; x] for the channels this
    // workgroup is responsible for (round-robin); hist_new is the other ping-pong buffer, nobody reads it
    // during this launch.  Saves one kernel launch per filt! call.
    if (a.H > 0) {
        const float *__restrict__ xin = static_cast<const float *>(a.x);
        const float *__restrict__ hold = static_cast<const float *>(a.hist);
        float *__restrict__ hnew = static_cast<float *>(a.hist_new);
        const int c_first = a.multi ? static_cast<int>(wg_in_grp) : static_cast<int>(blockIdx.x);
        const int c_step = a.multi ? static_cast<int>(static_stride) : static_cast<int>(gridDim.x);   // (multi: the workgroups of this stream)
        for (int c2 = c_first; c2 < a.nch; c2 += c_step)
            for (int i = lane; i < a.H; i += 64) {
                const long long e = static_cast<long long>(i) + a.x_len;          // index into [hist ; x]
#pragma unroll
                for (int cc = 0; cc < NC; ++cc)
                    hnew[(static_cast<long long>(c2) * a.H + i) * NC + cc] =
                        e < a.H ? hold[(static_cast<long long>(c2) * a.H + e) * NC + cc]
                                : xin[(static_cast<long long>(c2) * a.x_stride + (e - a.H)) * NC + cc];
            }
    }
}


// =====================================================================================================================
// THE RESIDENT RING CONSUMER (mrhip_internal.h: RingDesc / RingHost / RingDev; host side: ring.hip)
// =====================================================================================================================
// LDS words of a tile descriptor in ring mode: [0] steps (0: the kernel ends), [1] flags (bit 0: the tile completes a grab),
// [2,3] address of the tile's first output, [4] outputs of the channel from this tile on, [5] u0 of the chunk,
// [6] ring slot, [7] the grab's shard of the chunk's completion counters, [8,9] chunk number, [10] grabs of the chunk on that shard,
// [11] non-empty shards of the chunk; bit 31: its outputs are stored write-through (RingDesc::flags)
constexpr unsigned kRingTileWords = 16;

// One grab of chunk `seq` is complete (every wave that stored for it has drained its write-through stores).  The grab counts on ITS
// shard of the chunk's counters (ticket mod kRingShards; `cnt` grabs of the chunk land there); the grab that completes a shard counts
// the shard, the one that completes the last of the chunk's `nshards` non-empty shards tells the host.
// (nshards bit 31: the chunk's outputs were stored write-through)
__device__ __forceinline__ void ring_grab_done(RingDev *rd, RingHost *rh, unsigned slot, unsigned long long seq, unsigned shard, unsigned cnt, unsigned nshards_wt)
{
    const bool wt = (nshards_wt >> 31) != 0u;
    const unsigned nshards = nshards_wt & 0x7fffffffu;
    const unsigned prev = __hip_atomic_fetch_add(&rd->shard_done[slot][shard][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev + 1u == cnt) {
        st_sc1(&rd->shard_done[slot][shard][0], 0u);  // (the slot is reused only after the host has seen the flag below)
        vm_drain();
        const unsigned top = __hip_atomic_fetch_add(&rd->chunk_done[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (top + 1u == nshards) {
            st_sc1(&rd->chunk_done[slot], 0u);
            vm_drain();
            if (wt) st_sys(&rh->done[slot], seq + 1ull);                               // write-through stores: the outputs are in memory
            else {                                                                     // plain stores: in the L2s -- a request for their write-back
                const unsigned long long r = __hip_atomic_fetch_add(&rd->flush_req, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
                const unsigned long long ent[2] = {r, seq};
                store_wt<16>(&rd->flush_ent[r % kRingDepth][0], ent);
                vm_drain();
            }
        }
    }
}
// tickets [tb, tb + n) of a chunk: how many fall on shard s, and how many shards get any
__device__ __forceinline__ unsigned ring_shard_count(unsigned long long tb, unsigned n, unsigned s)
{
    const unsigned long long S = kRingShards;
    // integers congruent to s modulo S in [tb, tb + n): those below tb + n minus those below tb
    auto below = [&](unsigned long long e) -> unsigned long long { return e > s ? (e - s + S - 1ull) / S : 0ull; };
    return static_cast<unsigned>(below(tb + n) - below(tb));
}

// Workgroup 0's last wave: descriptors from the host's ring (pinned memory) into device memory, `head` behind them; the end of the
// kernel (the host closed the ring, or it stayed empty for idle_ticks).  A descriptor crosses PCIe as six 16-byte reads (its first 96
// bytes; lane 8k + j reads unit j of descriptor k: sixteen descriptors per round trip: 3.5 us per batch, measured;
// the first version read sixteen 8-byte words per descriptor -- the same ring rate either way, profiles/r05/experiments.md N).
__device__ __forceinline__ void ring_feeder(RingDev *rd, RingHost *rh, int lane)
{
    unsigned long long last = 0, t0 = wall_clock64(), code = 0;
    unsigned long long fb = 0, ft = 0, fd = 0;
    const unsigned long long idle = ld_sc1(&rd->idle_ticks);
    const int du = lane & 7, dk = lane >> 3;                // 16-byte unit of the descriptor, descriptor of the round (two rounds: dk, dk + 8)
    constexpr int KU = kRingKeyQword / 2;                   // the key's unit
    bool whole = false;                                     // every workgroup of the grid has started (told to the host: ring_launch)
    for (;;) {
        if (!whole) {
            const unsigned long long arr = ld_sc1(&rd->arrived);
            if (lane == 0) { st_sys(&rh->grid, static_cast<unsigned long long>(gridDim.x)); st_sys(&rh->arrived, arr); }
            whole = arr >= gridDim.x;
        }
        const unsigned long long h = ld_sys(&rh->head);
        const unsigned long long tseen = wall_clock64();
        if (h > last) {
            const unsigned nb = h - last < 16ull ? static_cast<unsigned>(h - last) : 16u;
            const unsigned k0 = static_cast<unsigned>(dk), k1 = static_cast<unsigned>(dk) + 8u;
            const bool on0 = k0 < nb && du < 6, on1 = k1 < nb && du < 6;
            const unsigned char *h0 = reinterpret_cast<const unsigned char *>(&rh->desc[(last + (on0 ? k0 : 0u)) % kRingDepth]) + 16 * (du < 6 ? du : 0);
            const unsigned char *h1 = reinterpret_cast<const unsigned char *>(&rh->desc[(last + (on1 ? k1 : 0u)) % kRingDepth]) + 16 * (du < 6 ? du : 0);
            v4u_t u0, u1;
            ld32_sys(h0, h1, u0, u1);                       // (every lane loads: the inactive ones re-read descriptor `last`)
            // fields first ...
            unsigned char *d0 = reinterpret_cast<unsigned char *>(&rd->desc[(last + k0) % kRingDepth]) + 16 * du;
            unsigned char *d1 = reinterpret_cast<unsigned char *>(&rd->desc[(last + k1) % kRingDepth]) + 16 * du;
            if (on0 && du != KU) store_wt<16>(d0, &u0);
            if (on1 && du != KU) store_wt<16>(d1, &u1);
            vm_drain();
            // ... then the key, by ONE 16-byte store: (tile_base, ngrabs, low half of the chunk number).  A reader that meets the slot's OLD
            // key next to new fields finds no pending ticket in it (that chunk is complete): no need to kill the key first.
            if (on0 && du == KU) { u0.w = static_cast<unsigned>((last + k0) & 0xffffffffull); store_wt<16>(d0, &u0); }
            if (on1 && du == KU) { u1.w = static_cast<unsigned>((last + k1) & 0xffffffffull); store_wt<16>(d1, &u1); }
            vm_drain();
            last += nb;
            if (lane == 0) st_sc1(&rd->head, last);
            {   // diagnostics (RingDev::stats [13] batches, [14] ticks from seeing `head` move to having published it, [15] descriptors)
                const unsigned long long t1 = wall_clock64();
                if (lane == 0) {
                    if (fb < static_cast<unsigned long long>(kRingTraceTiles)) { rd->trace[0][3 * fb] = tseen; rd->trace[0][3 * fb + 1] = t1; rd->trace[0][3 * fb + 2] = last; }
                    fb += 1; ft += t1 - tseen; fd += nb;
                }
                t0 = t1;
            }
            continue;
        }
        if (ld_sc1(&rd->closed) != 0ull) { code = 3; break; }               // a worker ran into its deadline
        if (ld_sys(&rh->close) != 0ull) {
            if (ld_sys(&rh->head) == last) { code = 1; break; }
            continue;
        }
        if (wall_clock64() - t0 > idle) { code = 2; break; }
        __builtin_amdgcn_s_sleep(8);
    }
    vm_drain();
    if (lane == 0) {
        atomicAdd(&rd->stats[13], fb); atomicAdd(&rd->stats[14], ft); atomicAdd(&rd->stats[15], fd);
        if (code != 3) st_sc1(&rd->closed, code);
        vm_drain();
        st_sys(&rh->stopped, code);
    }
    vm_drain();
}

// The write-back duty of a loader wave (RingDev::flush_req): `req` = flush_req and `claimed` = flush_claim[xcc] as loaded a moment ago.  Rare (once per
// chunk and XCD), hence a function of its own.
__device__ __attribute__((noinline)) void ring_flush_duty(RingDev *rd, RingHost *rh, unsigned xcc, unsigned long long req, int lane)
{
    unsigned long long won = 0;
    if (lane == 0) won = __hip_atomic_fetch_max(&rd->flush_claim[xcc][0], req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < req ? 1ull : 0ull;
    if (!__builtin_amdgcn_readfirstlane(static_cast<int>(won))) return;                 // another workgroup of this XCD has taken these requests on
    // every store of the chunks behind requests 1..req reached this XCD's L2 before its grab was counted: write the L2 back
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane != 0) return;
    (void)__hip_atomic_fetch_max(&rd->flush_done[xcc][0], req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the requests every XCD is past (read by atomics: at the point of coherence, in program order behind this XCD's own)
    unsigned long long F = ~0ull;
    for (int x = 0; x < 8; ++x) {
        const unsigned long long d = __hip_atomic_fetch_max(&rd->flush_done[x][0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        F = d < F ? d : F;
    }
    const unsigned long long P = __hip_atomic_fetch_max(&rd->flush_pub, F, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (unsigned long long r = P + 1ull; r <= F; ++r) {                                  // (empty unless this lane raised flush_pub: published once)
        unsigned long long seq;
        for (;;) {                                                                         // (the entry is written right behind the request's number)
            v4u_t e;
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(e) : "v"(&rd->flush_ent[r % kRingDepth][0]) : "memory");
            if (((static_cast<unsigned long long>(e.y) << 32) | e.x) == r) { seq = (static_cast<unsigned long long>(e.w) << 32) | e.z; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        st_sys(&rh->done[seq % kRingDepth], seq + 1ull);
    }
}

// The ring loader's COLD paths as functions of their own (never inlined): a seam tile happens twice per channel and chunk, the tail copy
// once per chunk, and inlined they raised the loader's register demand to 168 VGPRs -- the kernel is built for 128, and the allocator then
// parked values in scratch memory INSIDE the LDS-DMA loop of every tile, where the reload's s_waitcnt vmcnt(0) also waits for the transfer
// issued just before it: one memory round trip per kilobyte (profiles/r05/experiments.md P).
template <int NC>
__device__ __attribute__((noinline)) void ring_stage_seam(const float *xc, long long o, long long xlen, const float *hc, int H, int nchunks, int cd,
                                                          unsigned char *st, const unsigned char *dummy, int nslots, int lane)
{
    constexpr int EPC = 4 / NC;
    {
        const unsigned char *src = reinterpret_cast<const unsigned char *>(xc + o * NC);
        for (int slot = 0; slot < nslots; ++slot) {
            const int ci = slot * 64 + lane;
            const bool data = ci < nchunks;
            const long long smp = static_cast<long long>(EPC) * (data ? ci : 0);
            const bool inside = data && o + smp >= 0 && o + smp + EPC <= xlen;
            dma16_sc1(inside ? src + smp * (NC * 4) : dummy, st + static_cast<size_t>(slot) * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float *l = reinterpret_cast<float *>(st);
    for (int ci = lane; ci < nchunks; ci += 64) {
        const long long g0 = o + static_cast<long long>(EPC) * ci;
        if (g0 >= 0 && g0 + EPC <= xlen) continue;              // staged above
        float4 v;
        float *pv = reinterpret_cast<float *>(&v);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const long long gi = g0 + e;
#pragma unroll
            for (int cc = 0; cc < NC; ++cc) {
                float val = 0.f;
                if (gi >= 0) { if (gi < xlen) val = ld_sc1(xc + gi * NC + cc); }
                else if (gi >= -static_cast<long long>(H)) val = ld_sc1(hc + (H + gi) * NC + cc);
                pv[e * NC + cc] = val;
            }
        }
        *reinterpret_cast<float4 *>(l + (cd > 0 ? ci + ci / cd : ci) * 4) = v;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// shiftin! (support.jl:61-80) for the chunk AFTER this one: hnew <- the last H samples of [hold ; x], every channel
template <int NC>
__device__ __attribute__((noinline)) void ring_tail_copy(const float *xin, const float *hold, float *hnew, long long xs, long long xlen, int H, int nch, int lane)
{
    const long long words = static_cast<long long>(nch) * H * NC;
    for (long long base = 0; base < words; base += 64 * 4) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long w = base + k * 64 + lane;
            v[k] = 0.f;
            if (w < words) {
                const long long smp = w / NC, cc = w - smp * NC;
                const long long c2 = smp / H, i = smp - c2 * H;
                const long long e = i + xlen;                       // index into [history ; x]
                v[k] = e < H ? ld_sc1(hold + (c2 * H + e) * NC + cc) : ld_sc1(xin + (c2 * xs + (e - H)) * NC + cc);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long w = base + k * 64 + lane;
            if (w < words) st_sc1(hnew + w, v[k]);
        }
    }
    vm_drain();
}

// The whole life of a worker's loader wave in ring mode.  NW = 32-bit words per input sample, OS = bytes per output sample.
template <int NC, int OS>
__device__ __forceinline__ void pair_ring_loader_wave(const PolyArgs &a, const PairArgs &pa, unsigned char *smem, int lane)
{
    RingDev *const rd = a.ring_dev;
    RingHost *const rh = a.ring_host;
    if (lane == 0) (void)__hip_atomic_fetch_add(&rd->arrived, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (ring_launch counts the workgroups that started)
    if (blockIdx.x == 0) { if (lane == 0) st_sc1(&rd->grid, static_cast<unsigned long long>(gridDim.x)); ring_feeder(rd, rh, lane); return; }
    volatile unsigned *const td = reinterpret_cast<volatile unsigned *>(smem + pa.flags_off);   // [ns][kRingTileWords]
    // (what every lane loads alike goes to scalar registers at once: the compiler takes a vector load's result for divergent, and with it every
    //  branch on it and every value assigned under such a branch -- the chunk context, the scan position -- which then live in VGPRs)
    auto uni = [](unsigned long long v) -> unsigned long long {
        const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(v & 0xffffffffull)));
        const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(v >> 32)));
        return (static_cast<unsigned long long>(hi) << 32) | lo;
    };
    const unsigned long long idle = uni(ld_sc1(&rd->idle_ticks));
    const unsigned opts = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(ld_sc1(&rd->opts))));
    // the loader's instruction stream is long in ring mode (descriptor look-up, bookkeeping) and shares its SIMD with compute waves that
    // never stall: let it issue first (experiment switch: bit 9 = 512 keeps the default priority)
    if (!(opts & 512u)) __builtin_amdgcn_s_setprio(3);
    // tickets are dealt statically (ticket t -> workgroup t mod G).  MRHIP_RING_OPTS bit 10 = 1024: HANDED OUT instead, from the queue of
    // this workgroup's XCD group (RingDev::next_ticket) -- measured slower (1 ch 2.65 against 2.38 us per chunk, 64 ch 42.8 against 45.7 %:
    // the loader is the long leg and the draw is one more thing in it; profiles/r05/experiments.md L)
    const unsigned long long G = gridDim.x - 1u;
    const unsigned tq = (blockIdx.x - 1u) & 7u;
    const bool static_deal = (opts & 1024u) == 0u;
    // (the position drawn for the grab AFTER the next is always in flight: `pend` is issued when a ticket is consumed and read when
    //  the next one is -- its round trip never stalls the wave)
    unsigned long long pend = 0;
    auto ticket_issue = [&]() {
        if (!static_deal && lane == 0) pend = __hip_atomic_fetch_add(&rd->next_ticket[tq][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto ticket_take = [&](unsigned long long prev) -> unsigned long long {
        if (static_deal) return prev + G;
        const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(pend & 0xffffffffull)));
        const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(pend >> 32)));
        ticket_issue();
        return ((static_cast<unsigned long long>(hi) << 32) | lo) * 8ull + tq;
    };
    ticket_issue();
    unsigned long long ticket = static_deal ? blockIdx.x - 1u : ticket_take(0);          // this workgroup's next grab, counted over the whole life of the ring
    unsigned long long cur = 0, head_seen = 0;            // chunks below cur cannot hold `ticket`
    bool aborted = false;
    unsigned long long st_poll = 0, st_flush = 0, st_hist = 0;      // diagnostics (opts bit 8)
    unsigned long long st_flush_run = 0, st_flush_mid = 0, st_first = 0, st_last = 0, st_real = 0;
    // the chunk of the current grab (wave-uniform)
    unsigned long long c_x = 0, c_y = 0, c_tile_base = 0, c_seq = 0;
    long long c_xs = 0, c_ys = 0, c_xlen = 0, c_nout = 0, c_u0 = 0, c_o0 = 0;
    unsigned c_ngrabs = 1, c_spc = 1, c_total = 0, c_magic = 0xffffffffu;
    bool c_wt = true;                                         // the chunk's outputs go write-through (RingDesc::flags bit 0)
    const size_t slot_bytes = static_cast<size_t>(a.nch) * a.H * NC * 4u;
    auto hist_slot = [&](unsigned long long s) -> float * {
        return reinterpret_cast<float *>(static_cast<unsigned char *>(const_cast<void *>(a.hist)) + (s % kRingDepth) * slot_bytes);
    };
    // chunk s's call-start history is in its slot (written by the workgroup that took the first grab of chunk s - 1: an
    // earlier ticket, held by a workgroup that is running)
    auto wait_hist = [&](unsigned long long s) -> bool {
        const unsigned long long t0 = wall_clock64();
        while (uni(ld_sc1(&rd->hist_seq[s % kRingDepth])) != s + 1ull) {
            if (wall_clock64() - t0 > idle) {
                if (lane == 0) { st_sc1(&rd->stats[12], (1ull << 60) | s); st_sc1(&rd->stats[22], (static_cast<unsigned long long>(blockIdx.x) << 40) | ticket); st_sc1(&rd->closed, 3ull); }   // (what gave up: RingDev::stats[12], [22])
                vm_drain(); aborted = true; return false;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        st_hist += wall_clock64() - t0;
        return true;
    };
    // ---- which chunk holds `ticket`: the descriptors of chunks [cur, cur + 16) come into LDS by two LDS-DMA operations (lane 8k + j moves
    // unit j of descriptor k: the 2 KiB are the sixteen descriptors as the feeder wrote them), AHEAD of need: issued right behind the look-up
    // of the grab before, IN FRONT of that grab's tile -- LDS-DMA operations land in order, so "everything but the newest tile has landed"
    // (the wait this wave does anyway before it opens a tile to the compute waves) says the descriptors are there.  No registers are
    // held across the barrier for them and nothing drains the newest tile's transfers (round 5's first form loaded them into 24 VGPRs
    // with a vmcnt(0) of their own: with three pipeline stages that wait emptied the pipeline, profiles/r05/experiments.md N, P) ----
    unsigned char *const dbuf = smem + ((static_cast<unsigned>(pa.flags_off) + (kRingTileWords * 4u + 4u) * static_cast<unsigned>(pa.ns) + 15u) & ~15u);
    auto lds_offset_of = [](const void *q) -> unsigned { return static_cast<unsigned>(reinterpret_cast<size_t>((const __attribute__((address_space(3))) void *)q)); };
    unsigned long long pf_cur = ~0ull;                     // dbuf holds (or will hold) chunks [pf_cur, pf_cur + pf_nb)
    unsigned pf_nb = 0;
    auto prefetch_issue = [&]() {
        pf_cur = ~0ull;
        if (cur >= head_seen) return;
        if (head_seen >= static_cast<unsigned long long>(kRingDepth) && cur + kRingDepth <= head_seen) cur = head_seen - kRingDepth + 1ull;   // (complete long ago, slots recycled)
        pf_nb = head_seen - cur < 16ull ? static_cast<unsigned>(head_seen - cur) : 16u;
        pf_cur = cur;
#pragma unroll
        for (int op = 0; op < 2; ++op) {
            const unsigned k = static_cast<unsigned>(op) * 8u + (static_cast<unsigned>(lane) >> 3);
            const unsigned char *src = reinterpret_cast<const unsigned char *>(&rd->desc[(cur + (k < pf_nb ? k : 0u)) % kRingDepth]) + 16u * (static_cast<unsigned>(lane) & 7u);
            dma16_sc1(src, dbuf + op * 1024);
        }
    };
    enum { kFound = 0, kClosed = 1, kWouldBlock = 2 };
    // positions the chunk context on the chunk that holds `ticket`; may_block = false: returns kWouldBlock instead of polling
    auto find_chunk = [&](bool may_block) -> int {
        const unsigned long long t0 = wall_clock64();
        for (;;) {
            if (cur < head_seen) {
                if (pf_cur != cur) { prefetch_issue(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }   // (nothing requested ahead: first grab, or the window moved)
                const unsigned nb = pf_nb;
                pf_cur = ~0ull;
                // key of descriptor `lane` (lanes 0..15): (tile_base lo, hi, ngrabs, seq_lo)
                const v4u_t key = lds_read16_now(lds_offset_of(dbuf) + (static_cast<unsigned>(lane) & 15u) * 128u + kRingKeyQword * 8u);
                const unsigned long long tb = (static_cast<unsigned long long>(key.y) << 32) | key.x;
                const bool mine = lane < static_cast<int>(nb) && key.w == static_cast<unsigned>((cur + static_cast<unsigned>(lane)) & 0xffffffffull) &&
                                  ticket >= tb && ticket - tb < key.z;
                const unsigned long long hit = __ballot(mine);
                if (hit == 0ull) { cur += nb; continue; }
                const int fl = __builtin_ctzll(hit);
                // the six units of that descriptor: lane j reads unit j, the fields come by v_readlane
                const v4u_t un = lds_read16_now(lds_offset_of(dbuf) + static_cast<unsigned>(fl) * 128u + (static_cast<unsigned>(lane) < 6u ? static_cast<unsigned>(lane) : 0u) * 16u);
                auto rl = [&](unsigned v, int u) -> unsigned { return static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), u)); };
                auto rq = [&](unsigned lo, unsigned hi, int u) -> unsigned long long { return (static_cast<unsigned long long>(rl(hi, u)) << 32) | rl(lo, u); };
                cur += static_cast<unsigned>(fl);
                c_x = rq(un.x, un.y, 0); c_y = rq(un.z, un.w, 0);
                c_xs = static_cast<long long>(rq(un.x, un.y, 1)); c_ys = static_cast<long long>(rq(un.z, un.w, 1));
                c_xlen = static_cast<long long>(rq(un.x, un.y, 2)); c_nout = static_cast<long long>(rq(un.z, un.w, 2));
                c_u0 = static_cast<long long>(rq(un.x, un.y, 3)); c_o0 = static_cast<long long>(rq(un.z, un.w, 3)) - a.T;
                c_tile_base = rq(un.x, un.y, 4); c_ngrabs = rl(un.z, 4);
                c_spc = rl(un.x, 5); c_total = rl(un.y, 5); c_magic = rl(un.z, 5); c_wt = (rl(un.w, 5) & 1u) != 0u;
                c_seq = cur;
                if (c_total > 0u && (c_x == 0ull || c_y == 0ull || c_spc == 0u)) {   // (never: a descriptor that cannot be one -- leave rather than fault)
                    if (lane == 0) { st_sc1(&rd->stats[12], (2ull << 60) | cur); st_sc1(&rd->stats[22], (static_cast<unsigned long long>(blockIdx.x) << 40) | ticket); st_sc1(&rd->closed, 3ull); }
                    vm_drain();
                    aborted = true;
                    return kClosed;
                }
                return kFound;
            }
            head_seen = uni(ld_sc1(&rd->head));
            if (cur < head_seen) { st_poll += wall_clock64() - t0; continue; }
            if (uni(ld_sc1(&rd->closed)) != 0ull) {
                head_seen = uni(ld_sc1(&rd->head));       // `closed` is set behind the last `head`: look once more
                if (cur < head_seen) continue;
                return kClosed;
            }
            if (!may_block) return kWouldBlock;
            if (wall_clock64() - t0 > 2ull * idle + 100000000ull) return kClosed;     // (backstop: the feeder ends an idle ring itself)
            __builtin_amdgcn_s_sleep(16);
        }
    };
    auto tile_at = [&](unsigned g, unsigned jt) -> TileAt {
        unsigned q = __umulhi(g, c_magic);
        unsigned r = g - q * c_spc;
        if (r >= c_spc) { ++q; r -= c_spc; }
        if (r >= c_spc) { ++q; r -= c_spc; }
        return TileAt{static_cast<int>(q), static_cast<int>(r), static_cast<int>(umin(jt, c_spc - r))};
    };
    // shiftin! (support.jl:61-80) for the chunk AFTER this one: its call-start history = the last H samples of [history ; x] of this
    // chunk, every channel, into the next slot -- by the workgroup that takes the chunk's first grab, before anything else
    auto tail_copy = [&]() -> bool {
        if (a.H > 0) {
            if (c_xlen < a.H && !wait_hist(c_seq)) return false;
            ring_tail_copy<NC>(reinterpret_cast<const float *>(c_x), hist_slot(c_seq), hist_slot(c_seq + 1ull), c_xs, c_xlen, a.H, a.nch, lane);
        }
        if (lane == 0) st_sc1(&rd->hist_seq[(c_seq + 1ull) % kRingDepth], c_seq + 2ull);
        vm_drain();
        return true;
    };
    unsigned ra = 0, rb = 0;                                  // the current grab's steps [ra, rb) still to be staged
    unsigned g_shard = 0, g_cnt = 1;                          // ... its shard of the chunk's completion counters, the grabs that count there
    // the next grab of this workgroup that has steps: kFound; kClosed: the ring is closed (or a wait ran into its deadline);
    // kWouldBlock (only with may_block = false): nothing published yet
    auto next_grab = [&](bool may_block) -> int {
        for (;;) {
            const int fc = find_chunk(may_block);
            if (fc != kFound) return fc;
            const unsigned g = static_cast<unsigned>(ticket - c_tile_base);
            g_shard = static_cast<unsigned>(ticket % kRingShards);
            g_cnt = ring_shard_count(c_tile_base, c_ngrabs, g_shard);
            ticket = ticket_take(ticket);
            if (g == 0u && !tail_copy()) return kClosed;
            // a grab = pa.steps_per_group steps = several tiles: one descriptor look-up per grab, not per tile
            const unsigned long long lo = static_cast<unsigned long long>(g) * pa.steps_per_group;
            if (lo < c_total) { ra = static_cast<unsigned>(lo); rb = umin(ra + pa.steps_per_group, c_total); return kFound; }
            // a chunk without outputs (a short input, Filters.jl:543-547) has one grab without steps: nothing is stored for it
            if (lane == 0) ring_grab_done(rd, rh, static_cast<unsigned>(c_seq % kRingDepth), c_seq, g_shard, g_cnt, (c_ngrabs < kRingShards ? c_ngrabs : kRingShards) | (c_wt ? 0x80000000u : 0u));
            vm_drain();
        }
    };
    const unsigned pad_magic = pa.pad_every > 0 ? 0xffffffffu / static_cast<unsigned>(pa.pad_every + 1) + 1u : 0u;
    auto stage_tile = [&](const TileAt &ta, int stage) -> int {
        constexpr int EPC = 4 / NC;
        const int tlen = (ta.jt * pa.cM + pa.tail + EPC - 1) / EPC * EPC;
        const int nchunks = tlen / EPC;
        const int cd = pa.pad_every;
        const int nlds = cd > 0 ? (nchunks + cd - 1) / cd * (cd + 1) : nchunks;
        const int nslots = (nlds + 63) >> 6;
        const float *xc = reinterpret_cast<const float *>(c_x) + static_cast<long long>(ta.ch) * c_xs * NC;
        const long long o = c_o0 + static_cast<long long>(ta.st) * pa.cM;
        unsigned char *st = smem + static_cast<size_t>(stage) * pa.stage_bytes;
        const bool interior = o >= 0 && o + tlen <= c_xlen;
        if (interior) {
            const unsigned char *src = reinterpret_cast<const unsigned char *>(xc + o * NC);
            for (int slot = 0; slot < nslots; ++slot) {
                const int ci = slot * 64 + lane;
                int d = ci;
                if (cd > 0) { const int g = static_cast<int>(__umulhi(static_cast<unsigned>(ci), pad_magic)), r = ci - g * (cd + 1); d = r == cd ? 0 : g * cd + r; }
                const int cis = d < nchunks ? d : 0;
                dma16_sc1(src + static_cast<size_t>(cis) * 16, st + static_cast<size_t>(slot) * 1024);
            }
            return nslots;
        }
        // first / last tile of a channel: history seam and end of input.  As in pair_loader_wave: what lies inside x by LDS-DMA, the few
        // chunks at the seam and past the end element by element afterwards (L2-served loads: the history slot was written by another
        // workgroup of this launch, the signal by whoever filled the caller's buffer)
        if (o < 0 && !wait_hist(c_seq)) return 0;
        ring_stage_seam<NC>(xc, o, c_xlen, hist_slot(c_seq) + static_cast<long long>(ta.ch) * a.H * NC, a.H, nchunks, cd, st,
                            static_cast<const unsigned char *>(a.taps), nslots, lane);
        return 0;
    };
    unsigned long long ops = 0;
    auto newest_ops = [&](int ntiles) -> int {
        int n = 0;
        for (int k = 0; k < ntiles; ++k) n += static_cast<int>((ops >> (6 * k)) & 63u);
        return n < 60 ? n : 60;
    };
    auto end_marker = [&](int stage) {
        if (lane == 0) { td[kRingTileWords * stage] = 0u; td[kRingTileWords * stage + 1] = 0u; }
        ops <<= 6;
    };
    // The compute waves report a grab at the START of the tile after it (their write-through stores have had that tile's landing time
    // to drain).  When nothing is published and grabs are still unreported, a FLUSH tile (no steps) takes them through the barrier to
    // report; only then does this wave sit down and poll.
    bool unreported = false;
    unsigned long long idle_since = 0;                        // != 0: nothing to stage since then
    auto produce = [&](int stage) -> bool {
        const unsigned long long tr_begin = (opts & 256u) ? wall_clock64() : 0ull;
        if (ra >= rb && !aborted) {
            // never a wait in here: tiles published earlier may still be on their way through the barriers (with more than two stages
            // the compute waves would never reach them).  Nothing published yet: an idle tile (no steps) goes round instead.
            int ng = next_grab(false);
            if (ng == kWouldBlock) {
                if (!unreported) {
                    const unsigned long long now = wall_clock64();
                    if (idle_since == 0ull) idle_since = now;
                    if (now - idle_since > 2ull * idle + 100000000ull) ng = kClosed;     // (backstop: the feeder ends an idle ring itself)
                    else __builtin_amdgcn_s_sleep(32);
                }
                if (ng == kWouldBlock) {
                    if (lane == 0) { td[kRingTileWords * stage] = 1u; td[kRingTileWords * stage + 1] = 2u; }   // FLUSH / idle
                    ++st_flush; ++st_flush_run;
                    ops <<= 6;
                    unreported = false;
                    return true;
                }
            }
            idle_since = 0ull;
            if (ng != kFound) ra = rb = 0;
        }
        if (ra >= rb || aborted) { end_marker(stage); return false; }
        const TileAt ta = tile_at(ra, umin(rb - ra, static_cast<unsigned>(pa.J)));
        const unsigned long long tr_ticket = ticket;
        const unsigned long long tr_found = (opts & 256u) ? wall_clock64() : 0ull;
        // the descriptors the NEXT grab will be looked up in, requested IN FRONT of this tile's transfers (see prefetch_issue)
        if (ra + static_cast<unsigned>(ta.jt) >= rb && pf_cur == ~0ull && !(opts & 1u)) prefetch_issue();
        const int n_ops = stage_tile(ta, stage);
        const unsigned long long tr_issued = (opts & 256u) ? wall_clock64() : 0ull;
        if (aborted) { end_marker(stage); return false; }
        {   // the tile's descriptor: twelve wave-uniform words, lane k stores word k (one LDS store; computed on the scalar side)
            const unsigned long long yaddr = c_y + (static_cast<unsigned long long>(ta.ch) * static_cast<unsigned long long>(c_ys) +
                                                    static_cast<unsigned long long>(ta.st) * static_cast<unsigned>(pa.Sout)) * OS;
            const long long rem = c_nout - static_cast<long long>(ta.st) * pa.Sout;
            const unsigned w[12] = {static_cast<unsigned>(ta.jt), ra + static_cast<unsigned>(ta.jt) >= rb ? 1u : 0u,
                                    static_cast<unsigned>(yaddr & 0xffffffffull), static_cast<unsigned>(yaddr >> 32),
                                    static_cast<unsigned>(rem < 0x7fffffffLL ? rem : 0x7fffffffLL), static_cast<unsigned>(c_u0),
                                    static_cast<unsigned>(c_seq % kRingDepth), g_shard,
                                    static_cast<unsigned>(c_seq & 0xffffffffull), static_cast<unsigned>(c_seq >> 32), g_cnt,
                                    (c_ngrabs < static_cast<unsigned>(kRingShards) ? c_ngrabs : static_cast<unsigned>(kRingShards)) | (c_wt ? 0x80000000u : 0u)};
            unsigned mine = w[0];
#pragma unroll
            for (int k = 1; k < 12; ++k) mine = lane == k ? w[k] : mine;
            if (lane < 12) td[kRingTileWords * stage + lane] = mine;
        }
        ops = (ops << 6) | static_cast<unsigned>(n_ops);
        ra += static_cast<unsigned>(ta.jt);
        unreported = !(opts & 2u);
        const unsigned long long tr_pub = (opts & 256u) ? wall_clock64() : 0ull;
        if (opts & 256u) {
            st_last = wall_clock64();
            if (st_real == 0ull) st_first = st_last;
            if (lane == 0 && st_real < static_cast<unsigned long long>(kRingTraceTiles) && blockIdx.x < static_cast<unsigned>(kRingTraceRows)) {
                unsigned long long *tr = &rd->trace[blockIdx.x][3 * st_real];
                auto q = [](unsigned long long d) { return d < 0xffffull ? d : 0xffffull; };
                // [0] produce() began, [1] ticks to: chunk found | DMA issued | descriptor published | prefetch landed, [2] the NEXT ticket, idle tiles before
                tr[0] = tr_begin;
                tr[1] = q(tr_found - tr_begin) | (q(tr_issued - tr_found) << 16) | (q(tr_pub - tr_issued) << 32) | (q(st_last - tr_pub) << 48);
                tr[2] = (tr_ticket << 16) | (st_flush_run & 0xffffull);
            }
            ++st_real; st_flush_mid += st_flush_run; st_flush_run = 0;
        }
        return true;
    };
    if (lane < pa.ns) td[kRingTileWords * pa.ns + lane] = 0u;   // per stage: compute waves through with a grab's last tile (opair_kernel.inc)
    const bool stats = (opts & 256u) != 0u;
    const bool flush_mode = true;                             // (every loader does the write-back duty; chunks choose their protocol: RingDesc::flags)
    unsigned long long fl_since = 0, fl_last = 0, fl_changed = 0;   // when this wave first saw a request its XCD has not taken on; the newest request it has seen, since when
    const unsigned xcc = static_cast<unsigned>(__builtin_amdgcn_s_getreg(0xF814)) & 7u;     // XCC_ID: the XCD this workgroup runs on
    const unsigned long long st_t0 = wall_clock64();
    unsigned long long st_bar = 0, st_prod = 0, st_tiles = 0;
    // ONE site of produce(): the pa.ns - 1 tiles of the prologue and the steady state share the loop (inlined twice, the loader's code --
    // 34 KB, most of it produce() -- also held 26 more VGPRs: profiles/r05/experiments.md P)
    unsigned pipeline = 0;
    int pstage = 0, primed = 0;
    for (;;) {
        const bool steady = primed >= pa.ns - 1;
        unsigned long long tb1 = 0;
        if (steady) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long tb0 = stats ? wall_clock64() : 0ull;
            __builtin_amdgcn_s_barrier();
            tb1 = stats ? wall_clock64() : 0ull;
            st_bar += tb1 - tb0;
            if (!(pipeline & 1u)) break;
            pipeline >>= 1;
        }
        const unsigned bit = steady ? 1u << (pa.ns - 2) : 1u << primed;
        if (produce(pstage)) pipeline |= bit;
        if (!steady) ++primed;
        pstage = pstage + 1 == pa.ns ? 0 : pstage + 1;
        // (the write-back duty: two loads in front of the wait this wave does anyway)
        unsigned long long fl_req = 0, fl_claim = 0;
        if (flush_mode) { fl_req = ld_sc1(&rd->flush_req); fl_claim = ld_sc1(&rd->flush_claim[xcc][0]); }
        if (primed >= pa.ns - 1) { if (pa.ns > 2) wait_vmcnt_le(newest_ops(pa.ns - 2)); else vm_drain(); }
        // (one write-back serves every request filed so far and costs the same whatever the L2 holds: requests are batched -- 32 of them, or the
        //  oldest 40 us old, or none new for 5 us: a stream that went quiet, a lone chunk)
        if (flush_mode) {
            const unsigned long long fr = uni(fl_req), fc = uni(fl_claim);
            if (fr > fc) {
                const unsigned long long now = wall_clock64();
                if (fl_since == 0ull) fl_since = now;
                if (fr != fl_last) { fl_last = fr; fl_changed = now; }
                if (fr - fc >= 32ull || now - fl_since > 4000ull || now - fl_changed > 500ull) { ring_flush_duty(rd, rh, xcc, fr, lane); fl_since = 0ull; }
            } else fl_since = 0ull;
        }
        if (stats && steady) { st_prod += wall_clock64() - tb1; ++st_tiles; }
    }
    if (stats && lane == 0) {
        atomicAdd(&rd->stats[4], wall_clock64() - st_t0);
        atomicAdd(&rd->stats[5], st_poll);
        atomicAdd(&rd->stats[6], st_prod - st_poll);
        atomicAdd(&rd->stats[7], st_bar);
        atomicAdd(&rd->stats[8], 1ull);
        atomicAdd(&rd->stats[9], st_tiles);
        atomicAdd(&rd->stats[10], st_flush);
        atomicAdd(&rd->stats[11], st_hist);
        atomicAdd(&rd->stats[16], st_flush_mid);
        if (st_real) {
            atomicMax(&rd->stats[17], ~(st_last - st_t0)); atomicMax(&rd->stats[18], st_last - st_t0); atomicMax(&rd->stats[19], st_first - st_t0);
        }
        atomicMax(&rd->stats[20], ~st_real); atomicMax(&rd->stats[21], st_real);
    }
}

}  // namespace dev
}  // namespace mrhip
#endif
