// kernels_rational_opair.hip -- the polyphase kernel: FIRRational with L > M (any ratio: 160//147, 441//160, 7//2, ...) and
// with M > L up to M/L < 6 (147//160: the headline; 160//441, 3//17, ...), and FIRInterpolator (M = 1); tapsPerPhi <= 48
// (Float32 samples: 64; Float64 arithmetic: 32; M >= 2L: 32, Float32 arithmetic): Float32, ComplexF32 and Float64 samples, Float32 or Float64
// arithmetic.  This file holds the planning and the dispatch; the kernel itself is opair_kernel.inc, instantiated by
// kernels_rational_opair_*.hip (one unit per arithmetic width and per window distance SMIN = floor(M/L) = 0..5).
//
// Mapping.  (Round 1's kernel gave a lane two adjacent INPUT positions: with M > L a position produces at most one
// output, 8 % of the positions of 147//160 none, and with L > M a position produces one or two, which would need a
// third, mostly idle chain.  It is gone: this mapping is faster in both directions.)  A lane owns two adjacent OUTPUTS A = 2l, B = 2l+1 of the
// period of c*L outputs (c*M input positions) a workgroup covers per step.  The phase of output t is
// (u0 + t*M) mod L in every step ("phase-stationary"), so both tap columns live in VGPRs for the life of the
// workgroup, and the two windows start at LDS samples qA = (u0 + A*M) div L and qB = qA + {SMIN, SMIN+1},
// SMIN = floor(M/L): they overlap in all but one or two samples, so ONE run of samples feeds both dots (half the
// LDS reads of one output per lane).  The outputs of a lane are adjacent: the step ends in one dense 8-byte store
// per lane straight from the accumulators (no LDS transposition).
//
// The catch is alignment.  The run is fetched with 8-byte (16-byte for complex) LDS reads, which need an even
// sample index, and the distance between the two windows differs from lane to lane, while all 64 lanes execute one
// instruction stream.  So the run starts at R = qA rounded down to even, window A starts offA = qA - R in {0, 1}
// samples into it and window B offB = qB - R in {SMIN .. SMIN+2}.  Sample j of the run meets "slot" j of each
// output; a lane's tap registers hold its column SHIFTED by its offset (slot j of A holds tap j - offA), and the
// slots at the two ends that are inside the window for some lanes and outside for others (2 for A, 4 for B) are
// made exact no-ops for the lanes that must skip them: the product is replaced by -0.0 (one v_cndmask with a
// wave-level lane mask held in SGPRs), and x + (-0.0) == x bit for bit for every x (also -0.0, NaN, Inf), even as
// the very first term ("first product initialises the accumulator": -0.0 + p == p).  Multiplying by a zero tap
// instead would not be exact (0 * Inf = NaN, and the sign of an all-zero sum).  Every lane therefore performs
// exactly the reference's operations in the reference's order (support.jl:5-31) => bit-identical to the generic
// kernel; the price is 12 extra VALU instructions per 2 outputs (94 -> 106 for 24 taps).
//
// Everything else is the pair kernel's machinery (pair_loader.h): a loader wave stages tiles HBM -> LDS by LDS-DMA
// ahead of the compute waves, steps are handed out dynamically in XCD-local groups, window reads go through a
// register ring with compile-time counted waits, shiftin! is fused into the loader.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "mrhip_internal.h"

namespace mrhip {

// instantiation units (kernels_rational_opair_*.hip): Float32 arithmetic (nc = 1 | 2) and Float64 arithmetic
// (Float64 samples, or Float32 samples widened: the README's mixed case), one unit per SMIN = floor(M/L)
hipError_t launch_opair_f32_s0r(bool fused, int T, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus);
hipError_t launch_opair_f32_s0c(bool fused, int T, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus);
hipError_t launch_opair_f32_s1r(bool fused, int T, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus);
hipError_t launch_opair_f32_s1c(bool fused, int T, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus);
// 49..64 taps per phase (Float32 arithmetic, SMIN <= 1: 140-168 VGPRs, one 8-wave workgroup per CU)
hipError_t launch_opair_f32_s0_long(int nc, bool fused, int T, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus);
hipError_t launch_opair_f32_s1_long(int nc, bool fused, int T, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus);
// SMIN = 2..5 (decimating ratios up to M/L < 6): Float32 arithmetic, tapsPerPhi <= 32
hipError_t launch_opair_f32_s2(int nc, bool fused, int T, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus);
hipError_t launch_opair_f32_s3(int nc, bool fused, int T, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus);
hipError_t launch_opair_f32_s4(int nc, bool fused, int T, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus);
hipError_t launch_opair_f32_s5(int nc, bool fused, int T, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus);
hipError_t launch_opair_wide_s0(bool x_f64, bool fused, int T, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus);
hipError_t launch_opair_wide_s1(bool x_f64, bool fused, int T, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus);
// complex samples with Float64 arithmetic: ComplexF32 samples with Float64 taps (what firdes' Float64 taps and an SDR's
// ComplexF32 samples promote to) and ComplexF64 samples
hipError_t launch_opair_cmix_s0(bool x_f64, bool fused, int T, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus);
hipError_t launch_opair_cmix_s1(bool x_f64, bool fused, int T, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus);

namespace {
constexpr int kOMaxThreads = 512;
#define opair_env_int(name, dflt) MRHIP_ENV_INT(name, dflt)   // cached per call site (mrhip_internal.h)
}  // namespace

// Covers FIRRational and FIRInterpolator with tapsPerPhi <= 48 (Float32 samples with M < 2L: 64; Float64 arithmetic: 32) and M/L < 6 (L >= 2, SMIN = floor(M/L);
// SMIN >= 2: Float32 arithmetic and tapsPerPhi <= 32) for Float32 arithmetic (Float32 or ComplexF32 samples, Float32 taps)
// and Float64 arithmetic (Float64 or ComplexF64 samples; Float64 taps x Float32 or ComplexF32 samples).  Returns false otherwise
// (the caller tries the next kernel).
bool plan_rational_opair(const TypeKey &tk, bool fused, const PolyArgs &a, int num_cus, PairArgs *out, dim3 *block, size_t *lds, int force_wgpc)
{
    if (!opair_env_int("MRHIP_OPAIR", 1)) return false;   // read per call: tests switch kernels at run time
    if (tk.x_f64 && !tk.r_f64) return false;
    const int nc = tk.complex_x ? 2 : 1;
    const long long es = (tk.x_f64 ? 8 : 4) * nc;        // bytes per input sample
#ifdef MRHIP_PS_FAST_BUILD
    if (a.T != 24 && !((a.T == 36 || a.T == 48) && !tk.complex_x) && !(!tk.r_f64 && (a.T == 36 || a.T == 48 || a.T == 56 || a.T == 64))) return false;
#endif
    if (a.T < 1 || a.T > (tk.r_f64 ? (tk.complex_x ? 32 : 48) : 64)) return false;
    if (a.L < 2 || a.M < 1 || a.zero_start_below > 0) return false;   // (L == 1: the single-column kernels; M == 1: FIRInterpolator)
    const int smin = static_cast<int>(a.M / a.L);        // the two windows of a lane start SMIN or SMIN + 1 samples apart
    if (smin > 5) return false;
    if (!opair_instantiated(fused, smin, a.T)) return false;   // (FUSED: M/L < 2 and tapsPerPhi a multiple of 4)
    if (smin >= 2 && (tk.r_f64 || a.T > 32)) return false;   // instantiated for Float32 arithmetic, tapsPerPhi <= 32
    const int env_c = opair_env_int("MRHIP_OPAIR_C", 0), env_j = opair_env_int("MRHIP_OPAIR_J", 0), env_ns = opair_env_int("MRHIP_OPAIR_NS", 0);
    // c: lanes = c*L/2 <= 512; c*L and c*M even (a lane owns two outputs; the run base keeps its parity from step to
    // step) => c even, L and M being coprime.  Among the sizes with 3..7 full-ish compute waves take the fullest.
    // Among the sizes with 3..7 compute waves take the fullest: 147//160 -> c = 6, 441 lanes = 98 % of 7 waves (measured
    // against c = 4, 92 % of 5 waves: 65.7 vs 59.7 % of the HBM roofline on the one-call batch); 160//147 -> c = 4, 5 full waves.
    int best_c = 0;
    double best = -1.0;
    for (int pass = 0; pass < 2 && !best_c; ++pass)
        for (int c = 2; static_cast<long long>(c) * a.L / 2 <= kOMaxThreads && static_cast<long long>(c) * a.L <= 1024; c += 2) {
            const int lanes = static_cast<int>(static_cast<long long>(c) * a.L / 2);
            const int padded = (lanes + 63) / 64 * 64;
            if (pass == 0 && (padded < 192 || padded > 448)) continue;
            const double score = static_cast<double>(lanes) / padded * (padded < 192 ? 0.5 + 0.5 * padded / 192.0 : 1.0);
            if (score > best + 1e-9) { best = score; best_c = c; }
        }
    // Small launches (a 1e6-sample chunk of one channel: ~1000 steps) are all ramp: the smallest workgroup with at least
    // two compute waves starts sooner and gives every CU something to do (C1: 11.9 vs 13.9 us, per-chunk streaming 7.4 vs 6.0 %)
    if (a.n_out * a.nch < (1LL << 22))
        for (int c = 2; static_cast<long long>(c) * a.L / 2 <= kOMaxThreads && static_cast<long long>(c) * a.L <= 1024; c += 2)
            if (static_cast<long long>(c) * a.L / 2 >= 128) { best_c = c; break; }
    if (env_c > 0 && env_c % 2 == 0 && static_cast<long long>(env_c) * a.L / 2 <= kOMaxThreads && static_cast<long long>(env_c) * a.L <= 1024) best_c = env_c;
    if (!best_c) return false;
    if (tk.r_f64 && a.T > 32 && (static_cast<long long>(best_c) * a.L / 2 + 63) / 64 > 7) return false;   // eight waves at most there (opair_kernel.inc)
    const int c = best_c;
    const long long cM = static_cast<long long>(c) * a.M, cL = static_cast<long long>(c) * a.L;
    const int lanes = static_cast<int>(cL / 2);
    const int padded = (lanes + 63) / 64 * 64;
    const int nwaves = padded / 64;
    const int tail = a.T + smin + 4;                    // run overhang beyond the period: offsets <= smin + 2, + even rounding
    // resident workgroups per CU: 76-81 VGPRs allow 6 waves per SIMD = 24 per CU; a six-wave workgroup gets three, not
    // four (its waves land 2,2,1,1 on the SIMDs and the fourth rarely fits: kernels measured with MRHIP_PAIR_PROBE in round 1)
    // waves per CU the registers allow: two tap columns + ring + accumulators = 2T + 30 VGPRs (ComplexF32: + 36), in
    // granules of 8: T = 24 -> 80 -> 6 per SIMD = 24 per CU (ComplexF32: 88 -> 20); Float64 arithmetic: ~135 -> 3 per SIMD
    const int vgpr_est = (2 * a.T + (nc == 2 ? 36 : 30) + 7) / 8 * 8;
    const int waves_per_cu = tk.r_f64 ? 12 : 4 * std::min(8, 512 / vgpr_est);
    int wg_per_cu = std::max(1, std::min(4, waves_per_cu / (nwaves + 1)));
    if (nwaves + 1 == 6 && !tk.r_f64) wg_per_cu = 3;
    // (Float64 arithmetic: 12 waves per CU are TWO six-wave workgroups on paper only -- 2,2,1,1 waves per SIMD each, and three per SIMD is the
    //  limit: the ring's launch check found 257 of 512 on the chip.  Planned as what runs: one per CU, three stages of larger tiles:
    //  147//160 Float64 51.8 -> 54.3 %, the README's mixed case 40.0 -> 41.0 %; profiles/r05/experiments.md T)
    if (nwaves + 1 == 6 && tk.r_f64) wg_per_cu = 1;
    if (force_wgpc > 0) wg_per_cu = force_wgpc;   // (the ring's RING instantiation holds 128 VGPRs: two workgroups per CU, larger tiles)
    if (const int env_w = opair_env_int("MRHIP_OPAIR_WGPC", 0); env_w > 0) wg_per_cu = env_w;   // experiments
    // TWO pipeline stages of tiles as large as the LDS allows (the DMA runs one tile ahead, far more than the HBM
    // latency; every tile costs ~1000 cycles of barrier skew, ring priming and drain).  Measured on 147//160 Float32,
    // c = 6: two stages of J = 6 steps 65.7 % / 56.1 % (one call / 1e6-sample launches) vs three stages of J = 4 63.5 / 52.3.
    // A workgroup alone on its CU (Float64 arithmetic, eight waves) has nobody to hide its loader's latency behind:
    // three stages there (147//160 Float64: 57.5 % with three stages of J = 6 vs 52.4 % with two of J = 5).
    int ns = env_ns >= 2 && env_ns <= 8 ? env_ns : (wg_per_cu == 1 ? 3 : 2);
    auto j_for = [&](int stages) -> long long {
        const long long budget_kib = ((wg_per_cu <= 3 ? 150 : 160) * 1024 / wg_per_cu - 64) / stages / 1024;
        const long long j = ((budget_kib > 1 ? budget_kib : 1) * 1024 / es - tail) / cM;
        return j < 1 ? 1 : j;
    };
    long long J = j_for(ns);
    if (env_j > 0) J = env_j;
    if (J > 64) J = 64;
    if (env_j <= 0) {                       // small problems: enough tiles to give every CU a few workgroups
        const long long want_tiles = 4LL * num_cus;
        while (J > 2 && ((a.n_out + J * cL - 1) / (J * cL)) * a.nch < want_tiles) J = (J + 1) / 2;
    }
    // a stage is at most 60 / (ns - 1) LDS-DMA operations of 1 KiB (the loader counts them in vmcnt, 6 bits)
    const long long max_slots = 60 / (ns > 2 ? ns - 2 : 1);
    auto slots_for = [&](long long j) { return (((j * cM + tail + 3) / 4 * 4) * es / 16 + 63) / 64; };
    while (J > 1 && slots_for(J) > max_slots) --J;
    long long tile_len = (J * cM + tail + 3) / 4 * 4;
    const long long nslots = slots_for(J);
    const size_t stage_bytes = static_cast<size_t>(nslots) * 1024;
    if (nslots > max_slots || ns * stage_bytes > 156 * 1024) return false;
    PairArgs pa{};
    pa.c = c; pa.P = static_cast<int>(cL); pa.cM = static_cast<int>(cM);
    pa.Sout = pa.P; pa.lds_step = pa.cM; pa.run_chunks = 0; pa.q0 = 0; pa.run_magic = 0;
    pa.J = static_cast<int>(J);
    pa.tile_len = static_cast<int>(tile_len);
    pa.tail = tail;
    pa.dma_rounds = static_cast<int>(nslots);
    pa.stage_bytes = static_cast<int>(stage_bytes);
    pa.ns = ns;
    pa.nc = nc;
    pa.x_f64 = tk.x_f64 ? 1 : 0; pa.r_f64 = tk.r_f64 ? 1 : 0;
    pa.o0 = a.d0 - a.T;                      // x index of LDS sample 0 of a channel's first tile (negative => history)
    pa.tile_in = J * cM;
    pa.tile_out = J * cL;
    pa.tiles_per_channel = (a.n_out + pa.tile_out - 1) / pa.tile_out;
    pa.total_tiles = pa.tiles_per_channel * a.nch;
    if (a.n_out >= (1LL << 31) - pa.tile_out || pa.total_tiles >= (1LL << 31) - 65536) return false;   // 32-bit tile walk
    const long long spc = (a.n_out + cL - 1) / cL;
    if (spc * a.nch >= (1LL << 31)) return false;
    // the last tile of a channel may read up to `tail` samples past the last window: x index o + tlen <= x_len is
    // checked per tile by the loader (checked register path otherwise), so nothing is assumed here
    pa.steps_per_channel = static_cast<unsigned>(spc);
    pa.total_steps = static_cast<unsigned>(spc * a.nch);
    pa.spc_magic = spc == 1 ? 0xffffffffu : static_cast<unsigned>((1ULL << 32) / static_cast<unsigned long long>(spc));
    pa.flags_off = static_cast<int>(ns * stage_bytes);
    {   // the tap bank goes through the last stage's LDS when it fits (see the kernel); MRHIP_OPAIR_BANK=0: gather from global
        const long long pt = (a.T + 4) | 1;
        const bool fits = static_cast<size_t>(a.L * pt * (tk.r_f64 ? 8 : 4)) <= stage_bytes;
        pa.bank_off = (fits && opair_env_int("MRHIP_OPAIR_BANK", 1)) ? static_cast<int>((ns - 1) * stage_bytes) : -1;
    }
    *out = pa;
    *block = dim3(static_cast<unsigned>(padded + 64));   // + the loader wave
    *lds = ns * stage_bytes + 8 * ns;
    return true;
}

// L > 512 (625//512, 1000//999, 640//441: ordinary clock-trim ratios): a lane pair per output pair of the whole period no longer fits a
// workgroup.  The period of 2L outputs (2M input positions: both even, so the lanes' runs keep their parity from step to step) is cut into
// `nblocks` BLOCKS of P consecutive outputs; a workgroup owns ONE block for its whole life -- phase-stationary as before: output
// b*P + t of every period has phase (u0 + (b*P + t)*M) mod L -- and walks the periods: steps are 2L outputs apart in y and 2M samples
// apart in x, of which the block's windows touch one run of about P*M/L + T samples: only that run is staged (pair_loader.h:
// stage_runs), packed in LDS.  Blocks ride on the multi-stream machinery (group = block: MultiDesc::P_blk, q0; api.hip fills one
// descriptor per block and call).  Same slots, same order, same roundings as every other shape of the kernel.
bool plan_rational_opair_blocks(const TypeKey &tk, bool fused, const PolyArgs &a, int num_cus, PairArgs *out, dim3 *block, size_t *lds, int *nblocks_out)
{
    if (!opair_env_int("MRHIP_OPAIR", 1) || !opair_env_int("MRHIP_OPAIR_BLOCKS", 1)) return false;
    if (tk.x_f64 && !tk.r_f64) return false;
    if (a.L <= 512 || a.L > 4096 || a.M < 1 || a.zero_start_below > 0 || a.dyn || a.multi) return false;
    const int smin = static_cast<int>(a.M / a.L);
    if (smin > 1 || !opair_instantiated(fused, smin, a.T)) return false;
    if (a.T < 1 || a.T > (tk.r_f64 ? (tk.complex_x ? 32 : 48) : 64)) return false;
#ifdef MRHIP_PS_FAST_BUILD
    if (a.T != 24) return false;
#endif
    const int nc = tk.complex_x ? 2 : 1;
    const long long es = (tk.x_f64 ? 8 : 4) * nc;
    const int epc = static_cast<int>(16 / es) > 0 ? static_cast<int>(16 / es) : 1;     // samples per 16-byte chunk
    const long long Sout = 2LL * a.L, Sin = 2LL * a.M;
    // lanes per workgroup: up to seven compute waves (Float64 arithmetic with many taps: fewer registers, see plan_rational_opair)
    int max_lanes = opair_env_int("MRHIP_OPAIR_BLK_LANES", 0);
    if (max_lanes < 64 || max_lanes > 448) max_lanes = tk.r_f64 ? 320 : 448;
    int nblocks = static_cast<int>((Sout / 2 + max_lanes - 1) / max_lanes);
    if (nblocks < 2) nblocks = 2;
    long long P = (Sout + nblocks - 1) / nblocks;
    P = (P + 1) / 2 * 2;
    nblocks = static_cast<int>((Sout + P - 1) / P);
    const int lanes = static_cast<int>(P / 2), padded = (lanes + 63) / 64 * 64, nwaves = padded / 64;
    if (tk.r_f64 && a.T > 32 && nwaves > 7) return false;
    // the run a block's lanes touch within one step: window starts span floor(((P - 2) * M + L - 1) / L) + 1 samples, + the run of a lane
    // (T + smin + 2), + 2 for the even rounding of the block's base and of the lanes' run starts
    long long span = ((P - 2) * a.M + a.L - 1) / a.L + 1 + a.T + smin + 2 + 2;
    const long long rc = (span + epc - 1) / epc;                           // chunks per run
    long long lds_step = rc * epc;
    // (lds_step is even whenever a ring unit holds two samples (epc >= 2): the lanes' aligned reads keep their parity from step to step;
    //  ComplexF64 -- one sample per 16-byte read -- has no parity to keep)
    const int vgpr_est = (2 * a.T + (nc == 2 ? 36 : 30) + 7) / 8 * 8;
    const int waves_per_cu = tk.r_f64 ? 12 : 4 * std::min(8, 512 / vgpr_est);
    int wg_per_cu = std::max(1, std::min(4, waves_per_cu / (nwaves + 1)));
    if (nwaves + 1 == 6 && !tk.r_f64) wg_per_cu = 3;
    const int ns = wg_per_cu == 1 ? 3 : 2;
    const long long budget = ((wg_per_cu <= 3 ? 150 : 160) * 1024 / wg_per_cu - 64) / ns;
    long long J = budget / (lds_step * es);
    if (const int env_j = opair_env_int("MRHIP_OPAIR_J", 0); env_j > 0) J = env_j;
    if (J > 64) J = 64;
    if (J < 1) return false;
    const long long max_slots = 60 / (ns > 2 ? ns - 2 : 1);
    auto slots_for = [&](long long j) { return (j * rc + 63) / 64; };
    while (J > 1 && slots_for(J) > max_slots) --J;
    {   // small problems: enough tiles to give every CU a few workgroups
        const long long want_tiles = 4LL * num_cus;
        const long long spc0 = (a.n_out + Sout - 1) / Sout;
        while (J > 2 && ((spc0 + J - 1) / J) * a.nch * nblocks < want_tiles) J = (J + 1) / 2;
    }
    const long long nslots = slots_for(J);
    const size_t stage_bytes = static_cast<size_t>(nslots) * 1024;
    if (nslots > max_slots || ns * stage_bytes > 156 * 1024) return false;
    const long long spc = (a.n_out + Sout - 1) / Sout;
    if (spc * a.nch >= (1LL << 31) || a.n_out >= (1LL << 31) - Sout * J) return false;
    if ((a.u0 + Sout * a.M) >= (1LL << 31)) return false;                   // 32-bit phase arithmetic in the kernel
    PairArgs pa{};
    pa.c = 2; pa.P = static_cast<int>(P); pa.cM = static_cast<int>(Sin);
    pa.Sout = static_cast<int>(Sout); pa.lds_step = static_cast<int>(lds_step); pa.run_chunks = static_cast<int>(rc); pa.q0 = 0;
    pa.run_magic = static_cast<unsigned>(0xffffffffu / static_cast<unsigned>(rc) + 1u);
    pa.J = static_cast<int>(J);
    pa.tile_len = static_cast<int>(J * lds_step);
    pa.tail = a.T + smin + 4;
    pa.dma_rounds = static_cast<int>(nslots);
    pa.stage_bytes = static_cast<int>(stage_bytes);
    pa.ns = ns; pa.nc = nc;
    pa.x_f64 = tk.x_f64 ? 1 : 0; pa.r_f64 = tk.r_f64 ? 1 : 0;
    pa.o0 = a.d0 - a.T;
    pa.tile_in = J * Sin; pa.tile_out = J * Sout;
    pa.tiles_per_channel = (a.n_out + pa.tile_out - 1) / pa.tile_out;
    pa.total_tiles = pa.tiles_per_channel * a.nch * nblocks;
    pa.steps_per_channel = static_cast<unsigned>(spc);
    pa.total_steps = static_cast<unsigned>(spc * a.nch);
    pa.spc_magic = spc <= 1 ? 0xffffffffu : static_cast<unsigned>((1ULL << 32) / static_cast<unsigned long long>(spc));
    pa.flags_off = static_cast<int>(ns * stage_bytes);
    pa.bank_off = -1;                                                       // (a bank of L > 512 columns does not fit a stage: the two columns of a lane come straight from global memory, once per workgroup)
    *out = pa;
    *block = dim3(static_cast<unsigned>(padded + 64));
    *lds = ns * stage_bytes + 8 * ns;
    *nblocks_out = nblocks;
    return true;
}

// Shapes the resident ring consumer is instantiated for (opair_kernel.inc: the RING mode exists where the PLAIN one does -- the BASELINE
// configs' 24 and 32 taps per phase, STRICT, M < 2L, every sample / tap type); everything else runs a ring as one launch per chunk.
bool opair_ring_available(const TypeKey &tk, bool fused, const PolyArgs &a)
{
    if (fused || a.L < 2 || a.M < 1 || a.M / a.L > 1) return false;
    if (tk.x_f64 && !tk.r_f64) return false;
#ifdef MRHIP_PS_FAST_BUILD
    return a.T == 24;
#else
    return a.T == 24 || a.T == 32;
#endif
}

hipError_t launch_rational_opair(bool fused, const PolyArgs &a, const PairArgs &pa_in, dim3 block, size_t lds, hipStream_t s,
                                 const char **kname, int num_cus, unsigned *counters)
{
    if (!counters) return hipErrorInvalidValue;
    PairArgs pa = pa_in;
    pa.counters = counters;
    *kname = "rational_opair_kernel";
    const int smin = static_cast<int>(a.M / a.L);
    if (pa.r_f64 && pa.nc == 2)
        return smin == 0 ? launch_opair_cmix_s0(pa.x_f64 != 0, fused, a.T, block, lds, s, a, pa, num_cus)
                         : launch_opair_cmix_s1(pa.x_f64 != 0, fused, a.T, block, lds, s, a, pa, num_cus);
    if (pa.r_f64)
        return smin == 0 ? launch_opair_wide_s0(pa.x_f64 != 0, fused, a.T, block, lds, s, a, pa, num_cus)
                         : launch_opair_wide_s1(pa.x_f64 != 0, fused, a.T, block, lds, s, a, pa, num_cus);
    if (a.T > 48)
        return smin == 0 ? launch_opair_f32_s0_long(pa.nc, fused, a.T, block, lds, s, a, pa, num_cus) : launch_opair_f32_s1_long(pa.nc, fused, a.T, block, lds, s, a, pa, num_cus);
    switch (smin) {
    case 0: return pa.nc == 2 ? launch_opair_f32_s0c(fused, a.T, block, lds, s, a, pa, num_cus) : launch_opair_f32_s0r(fused, a.T, block, lds, s, a, pa, num_cus);
    case 1: return pa.nc == 2 ? launch_opair_f32_s1c(fused, a.T, block, lds, s, a, pa, num_cus) : launch_opair_f32_s1r(fused, a.T, block, lds, s, a, pa, num_cus);
    case 2: return launch_opair_f32_s2(pa.nc, fused, a.T, block, lds, s, a, pa, num_cus);
    case 3: return launch_opair_f32_s3(pa.nc, fused, a.T, block, lds, s, a, pa, num_cus);
    case 4: return launch_opair_f32_s4(pa.nc, fused, a.T, block, lds, s, a, pa, num_cus);
    case 5: return launch_opair_f32_s5(pa.nc, fused, a.T, block, lds, s, a, pa, num_cus);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace mrhip
