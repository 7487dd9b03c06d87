// kernels_rational_pair.hip -- the headline kernel: FIRRational with M > L (sample-rate reduction by a
// ratio close to one, e.g. 147//160), Float32 taps and samples, tapsPerPhi <= 32.
//
// Mapping ("pair of positions per lane, phase-stationary").  The phase of output k is (u0 + k*M) mod L,
// so outputs k and k + c*L share a tap column.  A workgroup covers one period of c*M input positions
// (c*L outputs) per step; lane l owns the two adjacent positions 2l and 2l+1 of that period, at every
// step j of every tile it is handed.  Each position produces zero or one output (M > L), always with
// the same phase, so the lane keeps TWO tap columns in VGPRs for its whole life and runs two
// independent accumulation chains (ILP hides the dependent-add latency of a single dot product).
// The two windows overlap in T-1 samples: one aligned run of T+1 (rounded to T+2) samples, fetched
// with (T+2)/2 ds_read_b64 through a small register ring (see run_steps), feeds both outputs -- half the
// LDS read traffic of one-output-per-lane, and because every lane's run starts on an even sample index
// the reads are 8-byte aligned and conflict-free with a single copy of the data in LDS.
//
// Staging.  A tile is up to J steps: J*c*M + T + 2 input samples of one channel, brought HBM -> LDS by
// LDS-DMA (global_load_lds_dwordx4, 16 B per lane, no VGPRs; the source only needs 4-byte alignment,
// scripts/ubench/dma_test.hip) by a dedicated loader wave, pa.ns stages deep (three stages of J = 5 steps for short
// launches: the DMA for tile i+2 is issued right after the one barrier that opens tile i; two stages of J = 8 steps for
// long launches: one tile ahead, fewer tile boundaries -- see plan_rational_pair).  The first/last tile of a channel (history seam,
// end of input) is staged through a checked register path into the same buffer.  The loader also draws
// the work (grouped dynamic scheduling, see the kernel) and performs shiftin! at the end of the launch.
//
// Arithmetic: exactly the generic kernel's (STRICT: separately rounded multiply and add, oldest sample
// first, first product initialises the accumulator; FUSED: explicit fma) => bit-identical results.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "mrhip_internal.h"
#include "pair_device.h"
#include "pair_loader.h"

#pragma clang fp contract(off)

namespace mrhip {
namespace {

constexpr int kPairMaxThreads = 512;
constexpr int kPairGroups = 32;         // scheduling groups (one step counter each); a multiple of the 8 XCDs

inline bool pair_debug_once()
{
    static int state = -1;
    if (state < 0) { const char *v = std::getenv("MRHIP_DEBUG"); state = (v && v[0] == '1') ? 1 : 0; }
    if (state == 1) { state = 0; return true; }
    return false;
}
inline int pair_env_int(const char *name, int dflt)
{
    const char *v = std::getenv(name);
    return v && *v ? std::atoi(v) : dflt;
}

using namespace dev;

// NC = 1: Float32 samples; NC = 2: ComplexF32 samples (interleaved re, im) with real taps = two independent real
// dots per output (SURVEY.md Appendix A "Types").  A sample is ES = 4*NC bytes; a lane's pair of samples is one
// ds_read_b64 (NC = 1) or one 16-byte aligned ds_read_b128 (NC = 2).
template <int T, bool FUSED, int NC>
__global__ __launch_bounds__(kPairMaxThreads + 64)
void rational_pair_kernel(PolyArgs a, PairArgs pa)
{
    constexpr unsigned ES = 4u * NC;            // bytes per sample
    using pair_t = std::conditional_t<NC == 1, v2u_t, v4u_t>;   // two consecutive samples
    constexpr int NPR = (T + 2) / 2;           // aligned 8-byte reads per lane per step: T+1 samples, rounded up

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // wave index as a scalar
    const int ncw = (blockDim.x >> 6) - 1;      // compute waves; the last wave of the workgroup is the loader

    // ---- tile walk shared by both roles.  The launch is a sequence of STEPS (one period of c*M input positions
    // each), numbered channel-major: g = channel * steps_per_channel + step.  Steps are handed out DYNAMICALLY in
    // grabs of J: workgroups that share the CUs and the HBM run the same loop at speeds that differ by +-25 %
    // (MRHIP_PAIR_PROBE=1: with equal static shares the first workgroup finished at 70 us, the last at 120 us),
    // so a static split leaves a quarter of the machine idle at the end.  A single device-wide counter would
    // serialise (~11 ns per same-address atomic x 22 000 grabs); instead the step range is cut into pa.ngroups
    // contiguous group ranges, workgroup b draws from group b % ngroups (32 groups: an XCD-local set of ~32
    // workgroups shares one counter, ~700 atomics per launch each).  The loader wave draws the grabs, cuts them
    // into tiles (never across a channel boundary) and publishes each tile (first step, steps) through two
    // LDS words per pipeline stage; the compute waves read them after the barrier that opens the tile.
    auto tile_at = [&](unsigned g, unsigned jt) -> TileAt { return pair_tile_at(pa, g, jt); };
    volatile unsigned *const tile_flag = reinterpret_cast<volatile unsigned *>(smem + pa.flags_off);   // [ns][2]: first step, steps (0 = end)

    if (wave == ncw) {
        // ================= loader wave: HBM -> LDS, one tile ahead of the compute waves (pair_loader.h) =================
        pair_loader_wave<NC>(a, pa, smem, lane);
        return;
    }

    // ================= compute waves =================
    // the lane's two positions -> (output index within a step, phase, active)
    int t_out[2];
    bool act[2];
    float taps[2][T];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        // 32-bit: plan_rational_pair guarantees c*M <= 1024 and L < M, so every product below is < 2^21
        const int p = 2 * tid + s;                                       // position inside the c*M period
        const int L = static_cast<int>(a.L), M = static_cast<int>(a.M), u0 = static_cast<int>(a.u0);
        const int num = p * L - u0;
        const int tt = num <= 0 ? 0 : (num + M - 1) / M;                 // first output at or after p
        const int u = u0 + tt * M;
        const int qq = u / L;
        act[s] = p < pa.cM && tt < pa.P && qq == p;
        t_out[s] = tt;
        const int phi = u - qq * L;
        const float *__restrict__ tp = static_cast<const float *>(a.taps) + static_cast<long long>(act[s] ? phi : 0) * T;
#pragma unroll
        for (int i = 0; i < T; ++i) taps[s][i] = tp[i];
    }
    // Output staging.  A wave's 128 positions produce one contiguous run of outputs [t_lo, t_hi) of the
    // step (about 128*L/M of them), scattered over its lanes' two accumulators with gaps where a position
    // has no output.  The lanes drop their results into a 512-byte LDS strip at (t - t_lo) and read the
    // strip back two per lane, so every step ends in ONE dense 8-byte-per-lane store instead of an
    // 8-byte store with holes plus a 4-byte store for the odd ones (half the write requests to L2).
    auto first_output_at = [&](int p) -> int {   // first output whose position is >= p
        const int num = p * static_cast<int>(a.L) - static_cast<int>(a.u0);
        return num <= 0 ? 0 : (num + static_cast<int>(a.M) - 1) / static_cast<int>(a.M);
    };
    const int t_lo_ll = first_output_at(128 * wave);
    const int t_hi_ll = first_output_at(128 * (wave + 1));
    const unsigned t_lo = static_cast<unsigned>(t_lo_ll < pa.P ? t_lo_ll : pa.P);
    const unsigned n_w = static_cast<unsigned>((t_hi_ll < pa.P ? t_hi_ll : pa.P)) - t_lo;   // outputs of this wave per step
    // strip: 1 KiB per compute wave = 128 output slots + 2 x 64 dump slots for the accumulators of positions
    // that produce no output (the writes are unconditional: no exec masking in the step loop)
    const unsigned strip_base = lds_base + static_cast<unsigned>(pa.ns) * static_cast<unsigned>(pa.stage_bytes) + static_cast<unsigned>(wave) * 256u * ES;
    const unsigned strip_w0 = strip_base + ES * (act[0] ? static_cast<unsigned>(t_out[0]) - t_lo : 128u + static_cast<unsigned>(lane));
    const unsigned strip_w1 = strip_base + ES * (act[1] ? static_cast<unsigned>(t_out[1]) - t_lo : 192u + static_cast<unsigned>(lane));
    const unsigned strip_r = strip_base + 2u * ES * static_cast<unsigned>(lane);
    const unsigned my_pair_pre = 2u * static_cast<unsigned>(lane);           // outputs my_pair, my_pair+1 of the strip
    const bool st8_full = my_pair_pre + 1 < n_w, st4_full = my_pair_pre + 1 == n_w;  // store predicates of a full tile

    // The compute waves' tile walk is 32-bit and scalar (plan_rational_pair guarantees n_out and total_tiles
    // < 2^31): 64-bit compares would park wave-uniform values in VGPRs for the life of the kernel.
    const int n_out = static_cast<int>(a.n_out);
    unsigned long long probe_c0 = 0, probe_r0 = 0;
    if (pa.probe) { probe_c0 = __builtin_amdgcn_s_memtime(); probe_r0 = __builtin_amdgcn_s_memrealtime(); }
    unsigned long long probe_bar = 0;             // cycles this wave spends at the tile barrier (diagnostics)
    unsigned long long probe_pre = 0;             // cycles between leaving the barrier and entering the step loop (descriptor read + address arithmetic)
    for (int s = 0;; s = (s + 1 == pa.ns ? 0 : s + 1)) {
        // One barrier per tile and no memory wait: the loader wave arrives only after this tile's data has
        // landed and its descriptor is in LDS; all compute waves arriving proves the oldest stage is no longer read.
        unsigned long long probe_b0 = 0;
        if (pa.probe) probe_b0 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        unsigned long long probe_b1 = 0;
        if (pa.probe) { probe_b1 = __builtin_amdgcn_s_memtime(); probe_bar += probe_b1 - probe_b0; }
        asm volatile("" ::: "memory");
        const unsigned tg = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(tile_flag[2 * s])));
        const unsigned tj = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(tile_flag[2 * s + 1])));
        if (tj == 0u) break;                      // end marker
        const TileAt ta = tile_at(tg, tj);
        const int ch = ta.ch, J = ta.jt;

        float *__restrict__ yc = static_cast<float *>(a.y) + (static_cast<long long>(ch) * a.y_stride + static_cast<long long>(ta.st) * pa.P) * NC;
        const int remaining = n_out - ta.st * pa.P;                       // outputs of this channel from this tile on
        const bool full = remaining >= J * pa.P;                          // wave-uniform
        // Lane-constant addresses are re-derived per tile from a fresh lane id (opaque to the compiler): kept live
        // across the tile loop they cost registers the step loop never uses (and, spilled, a vmcnt(0) per tile).
        unsigned lane_t;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_t));
        const unsigned my_pair = 2u * lane_t;
        const unsigned lane_win = (static_cast<unsigned>(wave) * 64u + lane_t) * 2u * ES;   // byte offset of sample 2*tid inside a stage
        const unsigned wbase = lds_base + static_cast<unsigned>(s) * pa.stage_bytes + lane_win;

        // Ring-buffered software pipeline.  All LDS traffic of the compute waves is hand-issued asm, so the order
        // of a wave's LDS operations is exactly the program order below and the counted waits hold.
        //
        // A wave issues at most one instruction every ~6 cycles and a SIMD needs 3-4 resident waves to keep its
        // VALU busy (scripts/ubench/valu_bank.hip), so registers are the budget: the T+2 window samples never
        // sit in registers all at once.  A "unit" (one pair of samples: ds_read_b64, or ds_read_b128 for complex;
        // ds_read2_b64 quads were 1.5-2 % slower) is fetched into slot u % K of a K-unit register ring K units before it is
        // consumed, and the slot is re-targeted as soon as it has fed its multiply-adds: LDS reads are spread
        // evenly through the arithmetic instead of arriving as a burst the in-order wave must push through the
        // shared LDS queue.  A step is padded to a multiple of K virtual units so that slot numbers repeat:
        //   per step:  W W S  [use 0, read 0+K] [use 1, read 1+K] ... ; virtual units past the window only
        //   issue reads; a read index past the padded step is a unit of the NEXT step's window.
        //   W W = the previous step's two accumulators -> output strip, S = strip read-back (two dense outputs
        //   per lane), stored to HBM mid-step.  ring_younger(u) counts the LDS operations issued between the
        //   read of unit u and its use (compile time; verified for every T by simulation).
        auto run_steps = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            constexpr int KP = NPR < (NC == 1 ? 5 : 3) ? NPR : (NC == 1 ? 5 : 3);   // ring: 5 pairs of 8 B or 3 pairs of 16 B
            constexpr int NPRV = (NPR + KP - 1) / KP * KP;
            constexpr int RS = NPR / 2;                       // the strip read-back is consumed after pair RS
            constexpr bool PAIRED_WAITS = KP >= 5;   // the three-unit ring of the complex kernel keeps its full look-ahead
            pair_t pring[KP];
            pair_t sv{};
            char *const ybytes = reinterpret_cast<char *>(yc);
            auto store_step = [&](int j, pair_t v) {
                // byte offsets from the (wave-uniform) tile base stay 32-bit: scalar base + VGPR offset stores
                const unsigned kj = static_cast<unsigned>(j) * static_cast<unsigned>(pa.P) + t_lo;
                char *const dst = ybytes + (kj + my_pair) * ES;                       // ES-byte aligned
                if (pa.ablate & 2) {   // timing experiments only: keep the arithmetic live, drop the stores
                    if (v.x == 0x7f123456u) *reinterpret_cast<unsigned *>(dst) = v.y;
                    return;
                }
                if constexpr (FULL) {
                    if (st8_full) __builtin_memcpy(dst, &v, 2 * ES);                  // two dense outputs
                    if (st4_full) __builtin_memcpy(dst, &v, ES);                      // the odd last one
                } else {
                    const unsigned rem_j = static_cast<unsigned>(remaining) > kj ? static_cast<unsigned>(remaining) - kj : 0u;
                    const unsigned lim = rem_j > n_w ? n_w : rem_j;
                    if (my_pair + 1 < lim) __builtin_memcpy(dst, &v, 2 * ES);
                    else if (my_pair < lim) __builtin_memcpy(dst, &v, ES);
                }
            };
            auto read_pair = [&](auto off_tag, unsigned addr) -> pair_t {
                constexpr int OFFP = decltype(off_tag)::value;                        // pair index inside the window
                if constexpr (NC == 1) return lds_read_b64<OFFP * 8>(addr);
                else return lds_read_b128<OFFP * 16>(addr);
            };
            auto strip_write = [&](unsigned addr, const float (&acc)[NC]) {
                if constexpr (NC == 1) lds_write_b32(addr, acc[0]);
                else lds_write_b64(addr, v2f_t{acc[0], acc[1]});
            };
            static_for<0, KP>([&](auto I) { pring[decltype(I)::value] = read_pair(I, wbase); });
            float pacc0[NC] = {}, pacc1[NC] = {};             // step -1 "results": written to the strip, never stored
#pragma unroll 1
            for (int j = 0; j < J; ++j) {
                const int jn = j + 1 < J ? j + 1 : j;         // the last step re-reads its own window (never used)
                const unsigned wcur = wbase + static_cast<unsigned>(j) * pa.cM * ES;
                const unsigned wnext = wbase + static_cast<unsigned>(jn) * pa.cM * ES;
                strip_write(strip_w0, pacc0);
                strip_write(strip_w1, pacc1);
                sv = read_pair(std::integral_constant<int, 0>{}, strip_r);   // same wave: LDS operations complete in order
                float acc0[NC] = {}, acc1[NC] = {};
                // pair r = (w[2r], w[2r+1]); output 0 uses w[i], output 1 uses w[i+1], i = 0..T-1: w[2r] feeds tap 2r of
                // output 0 and tap 2r-1 of output 1, w[2r+1] feeds tap 2r+1 of output 0 and tap 2r of output 1.
                // This file is compiled with -fno-slp-vectorize: hipcc would otherwise SLP-pack the chains into
                // v_pk_* (no faster per flop on gfx950: scripts/ubench/valu_rate.hip) and serialise them.
                static_for<0, NPRV>([&](auto I) {
                    constexpr int r = decltype(I)::value;
                    constexpr int slot = r % KP;
                    if constexpr (r < NPR) {
                        // one s_waitcnt per two units (the loop is bound by instruction issue, every kind counts)
                        if constexpr (!PAIRED_WAITS) lgkm_wait<ring_younger(r, NPR, KP)>(pring[slot]);
                        else if constexpr (r % 2 == 0) {
                            if constexpr (r + 1 < NPR) lgkm_wait2<ring_younger_pair(r, NPR, KP)>(pring[slot], pring[(r + 1) % KP]);
                            else lgkm_wait<ring_younger(r, NPR, KP)>(pring[slot]);
                        }
                        float wlo[NC], whi[NC];
                        if constexpr (NC == 1) { wlo[0] = __uint_as_float(pring[slot].x); whi[0] = __uint_as_float(pring[slot].y); }
                        else {
                            wlo[0] = __uint_as_float(pring[slot].x); wlo[1] = __uint_as_float(pring[slot].y);
                            whi[0] = __uint_as_float(pring[slot].z); whi[1] = __uint_as_float(pring[slot].w);
                        }
#pragma unroll
                        for (int cc = 0; cc < NC; ++cc) {
                            if constexpr (2 * r < T) { if constexpr (r == 0) acc0[cc] = taps[0][0] * wlo[cc]; else acc0[cc] = macf<FUSED>(taps[0][2 * r], wlo[cc], acc0[cc]); }
                            if constexpr (2 * r - 1 >= 0 && 2 * r - 1 < T) acc1[cc] = macf<FUSED>(taps[1][2 * r - 1], wlo[cc], acc1[cc]);
                            if constexpr (2 * r + 1 < T) acc0[cc] = macf<FUSED>(taps[0][2 * r + 1], whi[cc], acc0[cc]);
                            if constexpr (2 * r < T) { if constexpr (r == 0) acc1[cc] = taps[1][0] * whi[cc]; else acc1[cc] = macf<FUSED>(taps[1][2 * r], whi[cc], acc1[cc]); }
                        }
#pragma unroll
                        for (int cc = 0; cc < NC; ++cc) { pin(acc0[cc]); pin(acc1[cc]); }   // pair r is issued before its slot is re-targeted
                    }
                    if constexpr (r + KP < NPR) pring[slot] = read_pair(std::integral_constant<int, r + KP>{}, wcur);
                    else if constexpr (r + KP >= NPRV) pring[slot] = read_pair(std::integral_constant<int, r + KP - NPRV>{}, wnext);
                    if constexpr (r == RS) {
                        lgkm_wait<ring_reads_upto(RS, NPR, KP)>(sv);   // only this step's reads are younger than S
                        if (j > 0) store_step(j - 1, sv);     // wave-uniform
                    }
                });
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) { pacc0[cc] = acc0[cc]; pacc1[cc] = acc1[cc]; }
            }
            strip_write(strip_w0, pacc0);
            strip_write(strip_w1, pacc1);
            sv = read_pair(std::integral_constant<int, 0>{}, strip_r);
            lgkm_wait<0>(sv);                                 // also retires the last step's unused reads
            static_for<0, KP>([&](auto I) { pin(pring[decltype(I)::value]); });
            store_step(J - 1, sv);
        };
        if (pa.probe) { probe_pre += __builtin_amdgcn_s_memtime() - probe_b1; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        if (full) run_steps(std::true_type{});
        else run_steps(std::false_type{});

    }
    if (pa.probe && tid == 0) {   // in-kernel clock = shader cycles / (100 MHz ticks) * 100 MHz (MI355X_MICROARCH.md, DVFS)
        pa.probe[3 * blockIdx.x] = __builtin_amdgcn_s_memtime() - probe_c0;
        pa.probe[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - probe_r0;
        pa.probe[3 * blockIdx.x + 2] = probe_r0;
    }
    if (pa.probe && lane == 0 && wave < 8) {   // per-wave barrier cycles and SIMD id (HW_ID bits 5:4)
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        pa.probe[4 * gridDim.x + 8 * blockIdx.x + wave] = (probe_bar << 8) | ((hwid >> 4) & 3u);
        if (wave == 0) pa.probe[4 * gridDim.x + 8 * blockIdx.x + 7] = probe_pre;   // slot 7 (no eighth compute wave): wave 0's pre-loop cycles
    }
}

template <bool FUSED, int NC>
hipError_t launch_pair_T(int T, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, PairArgs pa, int num_cus,
                         int blocks_per_cu_override)
{
#define MRHIP_CASE(TT)                                                                              \
    case TT: {                                                                                      \
        auto kfn = rational_pair_kernel<TT, FUSED, NC>;                                                \
        if (lds > 48 * 1024) {                                                                      \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),                 \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)); \
            if (e != hipSuccess) return e;                                                          \
        }                                                                                           \
        int per_cu = 0;                                                                             \
        hipError_t eo = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, static_cast<int>(block.x), lds); \
        if (eo != hipSuccess) return eo;                                                            \
        if (per_cu < 1) per_cu = 1;                                                                 \
        {   /* see plan_rational_pair: the fourth six-wave workgroup never fits next to three running ones; short */ \
            /* launches (a tile or two per workgroup, static dealing) keep the full grid: slots free up at once */ \
            const long long tiles = (static_cast<long long>(pa.total_steps) + pa.J - 1) / pa.J;     \
            if (block.x == 6 * 64 && per_cu > 3 && tiles > 3LL * num_cus * per_cu) per_cu = 3;      \
        }                                                                                           \
        if (blocks_per_cu_override > 0) per_cu = blocks_per_cu_override;                            \
        long long g = static_cast<long long>(num_cus) * per_cu;                                     \
        if (g > static_cast<long long>(pa.total_steps)) g = pa.total_steps;                           \
        if (g < 1) g = 1;                                                                           \
        pa.ngroups = static_cast<int>(g < kPairGroups ? g : kPairGroups);   /* every group needs a workgroup */ \
        pa.steps_per_group = static_cast<unsigned>((pa.total_steps + pa.ngroups - 1) / pa.ngroups); \
        pa.static_grabs = (static_cast<long long>(pa.total_steps) + pa.J - 1) / pa.J <= 3 * g;      \
        if (g < 1) g = 1;                                                                           \
        if (pair_debug_once()) {                                                                    \
            hipFuncAttributes fa;                                                                   \
            (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kfn));                   \
            std::fprintf(stderr, "[mrhip] rational_pair T=%d grid=%lld block=%u lds=%zu occ/CU=%d regs=%d c=%d P=%d cM=%d J=%d " \
                         "tile_len=%d rounds=%d tiles=%lld\n", TT, g, block.x, lds, per_cu, fa.numRegs, pa.c, pa.P, pa.cM, \
                         pa.J, pa.tile_len, pa.dma_rounds, pa.total_tiles);                         \
        }                                                                                           \
        static const int probe_on = pair_env_int("MRHIP_PAIR_PROBE", 0);                            \
        static unsigned long long *probe_buf = nullptr;                                             \
        static int probe_left = 6;                                                                  \
        pa.probe = nullptr;                                                                         \
        if (probe_on && probe_left > 0) {                                                           \
            if (!probe_buf && hipMalloc(&probe_buf, sizeof(unsigned long long) * 12 * 65536) != hipSuccess) probe_buf = nullptr; \
            if (g <= 65536) pa.probe = probe_buf;                                                   \
        }                                                                                           \
        launch_kernel(kfn, dim3(static_cast<unsigned>(g)), block, lds, s, a, pa);              \
        if (pa.probe) {                                                                             \
            --probe_left;                                                                           \
            std::vector<unsigned long long> hb(12 * static_cast<size_t>(g));                         \
            (void)hipStreamSynchronize(s);                                                          \
            (void)hipMemcpy(hb.data(), probe_buf, hb.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost); \
            std::vector<double> ghz, us;                                                            \
            std::vector<double> st, en;                                                             \
            unsigned long long t0 = ~0ull;                                                          \
            for (long long i = 0; i < g; ++i) if (hb[3 * i + 1] && hb[3 * i + 2] < t0) t0 = hb[3 * i + 2]; \
            for (long long i = 0; i < g; ++i) if (hb[3 * i + 1]) { ghz.push_back(0.1 * hb[3 * i] / hb[3 * i + 1]); us.push_back(hb[3 * i + 1] * 0.01); \
                st.push_back((hb[3 * i + 2] - t0) * 0.01); en.push_back((hb[3 * i + 2] - t0 + hb[3 * i + 1]) * 0.01); } \
            std::sort(ghz.begin(), ghz.end()); std::sort(us.begin(), us.end()); std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end()); \
            if (!st.empty()) std::fprintf(stderr, "[mrhip] probe: tile-loop START after first wg: median %.1f p90 %.1f max %.1f us; END: p10 %.1f median %.1f max %.1f us\n", \
                                          st[st.size() / 2], st[st.size() * 9 / 10], st.back(), en[en.size() / 10], en[en.size() / 2], en.back()); \
            {                                                                                       \
                const int ncwv = static_cast<int>(block.x / 64) - 1;                                \
                for (int w = 0; w < ncwv && w < 8; ++w) {                                           \
                    std::vector<double> wf; int simd_hist[4] = {0, 0, 0, 0};                         \
                    for (long long i = 0; i < g; ++i) if (hb[3 * i] > 1000) {                       \
                        const unsigned long long v = hb[4 * g + 8 * i + w];                         \
                        wf.push_back(static_cast<double>(v >> 8) / static_cast<double>(hb[3 * i])); ++simd_hist[v & 3]; } \
                    std::sort(wf.begin(), wf.end());                                                \
                    if (!wf.empty()) std::fprintf(stderr, "[mrhip] probe: wave %d spends %.3f of the tile loop at the barrier (median); SIMD histogram %d %d %d %d\n", \
                                                  w, wf[wf.size() / 2], simd_hist[0], simd_hist[1], simd_hist[2], simd_hist[3]); \
                }                                                                                   \
            }                                                                                       \
            {                                                                                       \
                std::vector<double> pf;                                                             \
                for (long long i = 0; i < g; ++i) if (hb[3 * i] > 1000) pf.push_back(static_cast<double>(hb[4 * g + 8 * i + 7]) / static_cast<double>(hb[3 * i])); \
                std::sort(pf.begin(), pf.end());                                                    \
                if (!pf.empty() && static_cast<int>(block.x / 64) - 1 < 8) std::fprintf(stderr, "[mrhip] probe: wave 0 spends %.3f of the tile loop between the barrier and the step loop (descriptor read, address arithmetic; median)\n", pf[pf.size() / 2]); \
            }                                                                                       \
            if (!ghz.empty()) std::fprintf(stderr, "[mrhip] probe: in-kernel clock median %.3f GHz (min %.3f max %.3f); tile loop p10 %.1f median %.1f p90 %.1f max %.1f us\n", \
                                           ghz[ghz.size() / 2], ghz.front(), ghz.back(), us[us.size() / 10], us[us.size() / 2], us[us.size() * 9 / 10], us.back()); \
        }                                                                                           \
        return hipGetLastError();                                                                   \
    }
    switch (T) {
#ifdef MRHIP_PS_FAST_BUILD
        MRHIP_CASE(24)
#else
        MRHIP_CASE(1) MRHIP_CASE(2) MRHIP_CASE(3) MRHIP_CASE(4) MRHIP_CASE(5) MRHIP_CASE(6) MRHIP_CASE(7) MRHIP_CASE(8)
        MRHIP_CASE(9) MRHIP_CASE(10) MRHIP_CASE(11) MRHIP_CASE(12) MRHIP_CASE(13) MRHIP_CASE(14) MRHIP_CASE(15) MRHIP_CASE(16)
        MRHIP_CASE(17) MRHIP_CASE(18) MRHIP_CASE(19) MRHIP_CASE(20) MRHIP_CASE(21) MRHIP_CASE(22) MRHIP_CASE(23) MRHIP_CASE(24)
        MRHIP_CASE(25) MRHIP_CASE(26) MRHIP_CASE(27) MRHIP_CASE(28) MRHIP_CASE(29) MRHIP_CASE(30) MRHIP_CASE(31) MRHIP_CASE(32)
#endif
    default: return hipErrorInvalidValue;
    }
#undef MRHIP_CASE
}

}  // namespace

// Covers: Float32 or ComplexF32 samples with Float32 taps (R = Float32), M > L with L/M >= 0.7, tapsPerPhi <= 32, no zero-start
// quirk (i.e. a pfb kernel: FIRRational).  Returns false otherwise (caller tries the next kernel).
bool plan_rational_pair(const TypeKey &tk, const PolyArgs &a, int num_cus, PairArgs *out, dim3 *block, size_t *lds)
{
    if (!pair_env_int("MRHIP_PAIR", 1)) return false;   // read per call: tests switch kernels at run time
    if (tk.x_f64 || tk.r_f64) return false;
    const int nc = tk.complex_x ? 2 : 1;
    const long long es = 4 * nc;                      // bytes per sample
#ifdef MRHIP_PS_FAST_BUILD
    if (a.T != 24) return false;
#endif
    if (a.T < 1 || a.T > 32) return false;
    if (!(a.M > a.L) || static_cast<double>(a.L) / a.M < 0.70) return false;
    if (a.zero_start_below > 0) return false;
    static const int env_c = pair_env_int("MRHIP_PAIR_C", 0), env_r = pair_env_int("MRHIP_PAIR_ROUNDS", 0);
    static const int env_ns = pair_env_int("MRHIP_PAIR_NS", 0), env_j = pair_env_int("MRHIP_PAIR_J", 0);
    int ns = env_ns >= 2 && env_ns <= 10 ? env_ns : 3;
    // c: lanes = c*M/2 (c*M must be even), <= 512.  The loop is VALU-issue bound, so idle lanes in the last
    // wave cost in proportion: among the sizes with 3..5 compute waves take the fullest (147//160: c = 4, 320
    // lanes = 5 full waves; measured 4.02 TB/s vs 3.88 for c = 3 and 3.68 for c = 2 on the same box); if there
    // is none (large M), the best lane utilisation overall.
    int best_c = 0;
    double best = -1.0;
    for (int pass = 0; pass < 2 && !best_c; ++pass)
        for (int c = 1; static_cast<long long>(c) * a.M / 2 <= kPairMaxThreads; ++c) {
            if ((static_cast<long long>(c) * a.M) % 2) continue;
            const int lanes = static_cast<int>(static_cast<long long>(c) * a.M / 2);
            const int padded = (lanes + 63) / 64 * 64;
            if (pass == 0 && (padded < 192 || padded > (nc == 1 ? 320 : 256))) continue;   // complex: 4 compute waves (measured: c=3, J=3)
            const double score = static_cast<double>(lanes) / padded * (padded < 192 ? 0.5 + 0.5 * padded / 192.0 : 1.0);
            if (score > best + 1e-9) { best = score; best_c = c; }
        }
    if (env_c > 0 && (static_cast<long long>(env_c) * a.M) % 2 == 0 && static_cast<long long>(env_c) * a.M / 2 <= kPairMaxThreads)
        best_c = env_c;
    if (!best_c) return false;
    const int c = best_c;
    const long long cM = static_cast<long long>(c) * a.M;
    const int lanes = static_cast<int>(cM / 2);
    const int padded = (lanes + 63) / 64 * 64;
    const int nwaves = padded / 64;
    // tile: J steps; the stage is a whole number of 1 KiB DMA slots, sized so that FOUR workgroups fit the CU's
    // 160 KiB of LDS (ns stages + 1 KiB strip per compute wave + the tile descriptors).  MRHIP_PAIR_ROUNDS
    // (experiments) sets the stage size directly.
    // Complex samples double every buffer: three workgroups per CU there.
    const long long strip_bytes = static_cast<long long>(nwaves) * 256 * es;
    // Six-wave workgroups (147//160: five compute waves + the loader): the occupancy query promises four per CU
    // (24 waves = 6 per SIMD at 78 VGPRs), but a workgroup's waves land 2,2,1,1 on the four SIMDs and the fourth
    // workgroup only fits when the first three happen to be rotated evenly -- MRHIP_PAIR_PROBE=1 shows a quarter of a
    // 4-per-CU grid starting after the others have finished.  Plan for the three that are really resident.
    const int wg_per_cu = (nc == 1 && nwaves + 1 != 6) ? 4 : 3;
    // J (steps per tile) from the LDS budget of `stages` pipeline stages
    auto j_for = [&](int stages) -> long long {
        // three per CU: LDS is granted in coarse granules, leave slack (3 x 54 296 B did not fit)
        const long long budget_kib = ((wg_per_cu == 3 ? 150 : 160) * 1024 / wg_per_cu - 64 - strip_bytes) / stages / 1024;
        const int stage_kib = env_r > 0 ? env_r * nwaves : static_cast<int>(budget_kib > 1 ? budget_kib : 1);
        const long long j = (static_cast<long long>(stage_kib) * 1024 / es - a.T - 2) / cM;
        return j < 1 ? 1 : j;
    };
    long long J = j_for(ns);
    // Long launches: every tile costs ~1000 cycles of barrier skew, ring priming and drain (compute-only time follows
    // 72 us + 68 us / J per 491 MB), so the LDS is better spent on TWO stages of larger tiles: the DMA then runs one
    // tile (J steps, ~5 us) ahead, still far more than the HBM latency.  Measured, 64 ch x 1e8 in one call: J = 8 /
    // two stages 4.87-4.89 TB/s vs 4.76 with J = 5 / three stages; a 491 MB launch (16 tiles per workgroup) loses 2 %
    // to the coarser tail instead, so short launches keep three stages.
    if (env_ns <= 0 && env_j <= 0 && env_r <= 0) {
        const long long j2 = std::min<long long>(j_for(2), 64);
        const long long nslots2 = (((j2 * cM + a.T + 2 + 3) / 4 * 4) * es / 16 + 63) / 64;
        const long long tiles2 = ((a.n_out + j2 * c * a.L - 1) / (j2 * c * a.L)) * a.nch;
        if (j2 > J && nslots2 <= 60 && tiles2 >= 48LL * num_cus * wg_per_cu) { ns = 2; J = j2; }
    }
    if (env_j > 0) J = env_j;
    if (J > 64) J = 64;
    // small problems (few channels, short calls): shrink the tile until there are enough tiles to give
    // every CU a few workgroups -- a launch that occupies a third of the chip is latency-bound
    if (env_r <= 0 && env_j <= 0) {
        const long long want_tiles = 4LL * num_cus;
        while (J > 2 && ((a.n_out + J * c * a.L - 1) / (J * c * a.L)) * a.nch < want_tiles) J = (J + 1) / 2;
    }
    long long tile_len = J * cM + a.T + 2;
    tile_len = (tile_len + 3) / 4 * 4;
    const long long nslots = (tile_len * es / 16 + 63) / 64;
    const size_t stage_bytes = static_cast<size_t>(nslots) * 1024;
    if (nslots > 60 / (ns > 2 ? ns - 2 : 1) || ns * stage_bytes + static_cast<size_t>(strip_bytes) > 156 * 1024) return false;
    const long long need_rounds = nslots;
    PairArgs pa{};
    pa.c = c; pa.P = static_cast<int>(static_cast<long long>(c) * a.L); pa.cM = static_cast<int>(cM);
    pa.J = static_cast<int>(J);
    pa.tile_len = static_cast<int>(tile_len);
    pa.tail = a.T + 2;
    pa.dma_rounds = static_cast<int>(need_rounds);
    pa.stage_bytes = static_cast<int>(stage_bytes);
    pa.ns = ns;
    pa.o0 = a.d0 - a.T;
    static const int env_ablate = pair_env_int("MRHIP_PS_ABLATE", 0);   // timing experiments: 1 = no staging, 2 = no stores
    pa.ablate = env_ablate;
    pa.tile_in = J * cM;
    pa.tile_out = J * pa.P;
    pa.tiles_per_channel = (a.n_out + pa.tile_out - 1) / pa.tile_out;
    pa.total_tiles = pa.tiles_per_channel * a.nch;
    if (a.n_out >= (1LL << 31) - pa.tile_out || pa.total_tiles >= (1LL << 31) - 65536) return false;   // 32-bit tile walk
    *out = pa;
    *block = dim3(static_cast<unsigned>(padded + 64));   // + the loader wave
    {
        const long long spc = (a.n_out + pa.P - 1) / pa.P;
        if (spc * a.nch >= (1LL << 31)) return false;
        pa.steps_per_channel = static_cast<unsigned>(spc);
        pa.total_steps = static_cast<unsigned>(spc * a.nch);
        pa.spc_magic = spc == 1 ? 0xffffffffu : static_cast<unsigned>((1ULL << 32) / static_cast<unsigned long long>(spc));
    }
    *out = pa;
    pa.flags_off = static_cast<int>(ns * stage_bytes + static_cast<size_t>(strip_bytes));
    pa.nc = nc;
    *out = pa;
    *lds = ns * stage_bytes + static_cast<size_t>(strip_bytes) + 8 * ns;   // the pipeline stages + one output strip per compute wave + tile descriptors
    return true;
}

hipError_t launch_rational_pair(bool fused, const PolyArgs &a, const PairArgs &pa_in, dim3 block, size_t lds, hipStream_t s,
                                const char **kname, int num_cus, unsigned *counters)
{
    if (!counters) return hipErrorInvalidValue;
    PairArgs pa = pa_in;
    pa.counters = counters;
    *kname = "rational_pair_kernel";
    static const int bpc = pair_env_int("MRHIP_PAIR_BPC", 0);
    if (pa.nc == 2)
        return fused ? launch_pair_T<true, 2>(a.T, block, lds, s, a, pa, num_cus, bpc)
                     : launch_pair_T<false, 2>(a.T, block, lds, s, a, pa, num_cus, bpc);
    return fused ? launch_pair_T<true, 1>(a.T, block, lds, s, a, pa, num_cus, bpc)
                 : launch_pair_T<false, 1>(a.T, block, lds, s, a, pa, num_cus, bpc);
}

}  // namespace mrhip
