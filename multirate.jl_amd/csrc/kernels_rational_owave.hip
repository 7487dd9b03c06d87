// kernels_rational_owave.hip -- FIRRational with 1/2 < M/L < 2, the WAVE-AUTONOMOUS form of the output-pair kernel
// (kernels_rational_opair.hip has the mapping: a lane owns two adjacent outputs of a period of c*L outputs, both tap columns
// live in VGPRs, one run of samples feeds both dots, alignment by exact -0.0 no-op slots).  This file holds the planning
// and the dispatch; the kernel is owave_kernel.inc.
//
// Why.  In the workgroup form (one loader wave + 7 compute waves, one s_barrier per tile) the per-wave records of
// scripts/exp_probe.py show what the barrier costs on this chip: the 21 compute waves of a CU land 5,5,5,6 on the SIMDs,
// instruction issue favours the OLDER waves of a SIMD, so in every workgroup one wave -- the youngest wave on the
// fullest SIMD -- never waits and its six siblings spend 30-59 % of their time at the barrier waiting for it (23 % of
// all compute-wave time; 19 % even where the SIMDs hold 5,5,5,5).  Nothing in the arithmetic couples the waves: a
// wave's two tap columns, its window and its outputs are its own.  So here a wave is the unit of everything:
//   * TYPE.  The period of P = c*L outputs is cut into slices of 128 outputs; a wave of type k owns outputs
//     [128k, 128k+128) of every step it works on (lane l: outputs 128k+2l, +1), so its tap columns stay in VGPRs for
//     its life.  type = global wave number mod nt.
//   * PRIVATE PIPELINE.  The samples a type touches in one step are one run of ~128*M/L + T + 3 samples; the wave
//     stages tiles of J such runs (J steps of ITS slice, cM samples apart in x) into two private LDS stages with
//     LDS-DMA (per-lane source addresses, so the J runs of a tile cost ceil(J*cps/64) instructions), one tile ahead,
//     and waits for them with a COUNTED vmcnt: vector-memory operations retire in order, so "at most as many
//     outstanding as I issued after the tile's DMA" means the tile has landed while the output stores issued since stay
//     in flight.  No barrier, no flags, no loader wave: all waves of a CU compute (24 instead of 21 for Float32).
//   * POOLS.  Steps are numbered channel-major and cut into 32 XCD-local groups as before; every (group, type) has its
//     own counter, drawn one grab (= one tile) ahead with a hand-issued returning atomic; a wave whose home pool is dry
//     moves on to the next group's pool of its type, so the launch ends when the last pool does.
// Neighbouring types overlap by T + 3 samples per step (read twice from L2, once from HBM).
// Workgroups are four waves -- one per SIMD by the hardware's placement rule -- and share nothing but the LDS
// allocation and the tap bank staged through it at the start.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "mrhip_internal.h"

namespace mrhip {

hipError_t launch_owave_f32_s0(int nc, bool fused, int T, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus);
hipError_t launch_owave_f32_s1(int nc, bool fused, int T, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus);
hipError_t launch_owave_wide_s0(bool x_f64, bool fused, int T, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus);
hipError_t launch_owave_wide_s1(bool x_f64, bool fused, int T, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus);

namespace {
inline int owave_env(const char *name, int dflt)
{
    const char *v = std::getenv(name);
    return v && *v ? std::atoi(v) : dflt;
}
}  // namespace

// Same coverage as plan_rational_opair; launches too small to give every wave of the chip a few tiles stay with the
// workgroup form (its static round-robin deal and smaller grid start sooner).
bool plan_rational_owave(const TypeKey &tk, const PolyArgs &a, int num_cus, PairArgs *out)
{
    if (!owave_env("MRHIP_OWAVE", 1)) return false;      // read per call: tests switch kernels at run time
    if (tk.x_f64 && !tk.r_f64) return false;
    if (tk.r_f64 && tk.complex_x) return false;
    const int nc = tk.complex_x ? 2 : 1;
    const int es = (tk.x_f64 ? 8 : 4) * nc;               // bytes per input sample
#ifdef MRHIP_PS_FAST_BUILD
    if (a.T != 24 && !(!tk.r_f64 && (a.T == 36 || a.T == 48))) return false;
#endif
    if (a.T < 1 || a.T > (tk.r_f64 ? 32 : 48)) return false;
    if (a.L < 2 || a.M < 2 || a.zero_start_below > 0) return false;
    if (!(2LL * a.M > a.L && a.M < 2LL * a.L)) return false;
    if (a.n_out * a.nch < static_cast<long long>(owave_env("MRHIP_OWAVE_MIN_OUT", 1 << 22))) return false;
    const int smin = a.M > a.L ? 1 : 0;
    // c even (a lane owns two outputs; the run base keeps its parity from step to step), P = c*L <= 1024 (32-bit lane
    // arithmetic, at most 8 types): the c whose last type is fullest, the largest such c.
    int best_c = 0;
    double best = -1.0;
    for (int c = 2; static_cast<long long>(c) * a.L <= 1024; c += 2) {
        const int P = static_cast<int>(c * a.L);
        const int nt = (P + 127) / 128;
        const double score = static_cast<double>(P) / (128.0 * nt);
        if (score >= best - 1e-9) { best = std::max(best, score); best_c = c; }
    }
    if (const int env_c = owave_env("MRHIP_OWAVE_C", 0); env_c > 0 && env_c % 2 == 0 && static_cast<long long>(env_c) * a.L <= 1024) best_c = env_c;
    if (!best_c) return false;
    const int c = best_c;
    const long long cM = static_cast<long long>(c) * a.M, cL = static_cast<long long>(c) * a.L;
    const int lanes = static_cast<int>(cL / 2);
    const int nt = (lanes + 63) / 64;
    if (nt > kOwMaxTypes) return false;
    // chunks of one step's slice: from the first sample of the type's first lane (rounded down to a chunk) to the end of
    // the last lane's run (the ring reads whole pairs: 2 * NPR samples)
    const int epc = 16 / es;
    const int W = a.T + smin + 2, npr = (W + 1) / 2;
    int cps = 1;
    for (int k = 0; k < nt; ++k) {
        const long long l_last = std::min<long long>(64LL * k + 63, lanes - 1);
        const long long lo = (((a.u0 + 128LL * k * a.M) / a.L) & ~1LL) & ~static_cast<long long>(epc - 1);
        const long long hi = (((a.u0 + 2 * l_last * a.M) / a.L) & ~1LL) + 2 * npr;
        cps = std::max<int>(cps, static_cast<int>((hi - lo + epc - 1) / epc));
    }
    PairArgs pa{};
    pa.c = c; pa.P = static_cast<int>(cL); pa.cM = static_cast<int>(cM);
    pa.nt = nt; pa.cps = cps;
    pa.cps_magic = static_cast<unsigned>(((1ULL << 32) + cps - 1) / static_cast<unsigned long long>(cps));
    pa.ns = 2;
    pa.nc = nc;
    pa.x_f64 = tk.x_f64 ? 1 : 0; pa.r_f64 = tk.r_f64 ? 1 : 0;
    pa.o0 = a.d0 - a.T;                      // x index of LDS sample 0 of a channel's first step (negative => history)
    const long long spc = (a.n_out + cL - 1) / cL;
    if (a.n_out >= (1LL << 31) - 64 * cL || spc * a.nch >= (1LL << 31) - 65536) return false;   // 32-bit step walk
    pa.steps_per_channel = static_cast<unsigned>(spc);
    pa.total_steps = static_cast<unsigned>(spc * a.nch);
    pa.spc_magic = spc == 1 ? 0xffffffffu : static_cast<unsigned>((1ULL << 32) / static_cast<unsigned long long>(spc));
    pa.bank_off = -1;
    *out = pa;
    return true;
}

// registers per lane -> waves per CU -> LDS per wave -> steps per tile
bool owave_finish_plan(PairArgs *pa, int regs, int es, int rs, const PolyArgs &a, size_t *lds, int *wg_per_cu)
{
    (void)es;
    const int alloc = (regs + 7) / 8 * 8;
    int waves_per_simd = std::min(8, 512 / std::max(alloc, 8));
    if (const int e = owave_env("MRHIP_OWAVE_WPS", 0); e > 0) waves_per_simd = e;
    const int wgs = waves_per_simd;                           // four-wave workgroups, one wave per SIMD each
    const long long budget = (160 * 1024 - 1024) / wgs / 4;   // LDS bytes per wave
    const long long step_bytes = static_cast<long long>(pa->cps) * 16;
    long long J = budget / 2 / step_bytes;
    if (const int e = owave_env("MRHIP_OWAVE_J", 0); e > 0) J = e;
    if (J > 16) J = 16;
    while (J > 1 && (J * pa->cps + 63) / 64 > 24) --J;        // LDS-DMA operations of a tile (vmcnt counts to 63)
    if (J < 1) return false;
    const long long stage_bytes = J * step_bytes;
    if (2 * stage_bytes * 4 > 160 * 1024) return false;
    pa->J = static_cast<int>(J);
    pa->stage_bytes = static_cast<int>(stage_bytes);
    pa->tile_in = J * pa->cM;
    pa->tile_out = J * static_cast<long long>(pa->P);
    *lds = static_cast<size_t>(2 * stage_bytes * 4);
    *wg_per_cu = wgs;
    {   // the tap bank goes through the workgroup's LDS when it fits (MRHIP_OWAVE_BANK=0: gather from global memory)
        const long long pt = (a.T + 4) | 1;
        const bool fits = static_cast<size_t>(a.L * pt * rs) <= *lds;
        pa->bank_off = (fits && owave_env("MRHIP_OWAVE_BANK", 1)) ? 0 : -1;
    }
    return true;
}

hipError_t launch_rational_owave(bool fused, const PolyArgs &a, const PairArgs &pa_in, hipStream_t s, const char **kname, int num_cus,
                                 unsigned *counters)
{
    if (!counters) return hipErrorInvalidValue;
    PairArgs pa = pa_in;
    pa.counters = counters;
    *kname = "rational_owave_kernel";
    const bool up = a.L > a.M;                 // SMIN = 0
    if (pa.r_f64)
        return up ? launch_owave_wide_s0(pa.x_f64 != 0, fused, a.T, s, a, pa, num_cus)
                  : launch_owave_wide_s1(pa.x_f64 != 0, fused, a.T, s, a, pa, num_cus);
    return up ? launch_owave_f32_s0(pa.nc, fused, a.T, s, a, pa, num_cus)
              : launch_owave_f32_s1(pa.nc, fused, a.T, s, a, pa, num_cus);
}

}  // namespace mrhip
