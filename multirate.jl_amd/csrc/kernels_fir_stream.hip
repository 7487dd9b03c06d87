// kernels_fir_stream.hip -- FIRStandard (M = 1) and FIRDecimator (L = 1), 2 to 16384 taps (as long as a tile fits the LDS), with
// Float32 arithmetic (Float32 or ComplexF32 samples, Float32 taps) and Float64 arithmetic (Float64 or ComplexF64 samples, or
// Float32 / ComplexF32 samples with Float64 taps): BASELINE config 3b.  This file holds the planning and the dispatch.  Two
// kernels share the mapping below: fir_stream_kernel.inc, instantiated per decimation M = 1..11, 13, 15 by
// kernels_fir_stream_{f32,f64,mix}.hip, and fir_stream_rt_kernel (kernels_fir_stream_rt.hip), which takes M at run time
// and serves every other decimation whose step fits the LDS.
//
// reference: src/Filters.jl:450-473 (Standard), :598-631 (Decimator); dot: src/support.jl:33-55.
//
// (Rounds 1-3 also had fir_direct_kernel: a tile staged synchronously -- all threads load, barrier, compute, barrier --
// through a transposed LDS layout, ONE sample per LDS instruction, 30-44 % of the multiply+add rate.  Retired in round 4:
// nothing reaches it any more.)  Here the machinery of the rational kernel is reused (pair_loader.h): a loader wave streams tiles HBM -> LDS
// with LDS-DMA one tile ahead, steps are handed out dynamically, shiftin! is fused.
//
// Mapping.  A lane owns two adjacent outputs 2l and 2l+1 of a step; their windows start 2lM and 2lM + M samples into
// the step, so ONE run of T + M samples feeds both dots: sample j of the run meets tap j of the first output
// (0 <= j < T) and tap j - M of the second (M <= j < T + M) -- no per-lane variation at all, and the taps are the same
// for every lane: they are read with SCALAR loads through the constant address space and feed the VALU as SGPR
// operands (no tap registers, no broadcasts).  The run is fetched 16 bytes at a time (ds_read_b128: four Float32 or
// two ComplexF32 samples; 8 bytes for Float32 single-rate, whose lanes are 8 bytes apart).
//
// Bank conflicts.  Lanes start S = 2*M*sizeof(sample) bytes apart.  When S is not a multiple of 16 (Float32 samples, odd M)
// the run is read 8 bytes at a time: the 32 lanes of a ds_read_b64 group are 2M dwords apart, gcd(2M, 64) = 2, so they
// cover all 64 banks exactly once.  Otherwise the run is read 16 bytes at a time and lanes start CD = S/16 chunks apart:
// an odd CD is conflict-free as it is (the 16 lanes of a ds_read_b128 group land on 16 different chunk positions mod 16);
// for an even CD (S = 32, 64, 96, 128, ...) a linear tile would put them on 8, 4, ... distinct positions, and the loader
// therefore writes one 16-byte PAD chunk after every CD data chunks (LDS-DMA takes a per-lane source address, so the
// pad costs nothing but 1/(CD + 1) of the LDS and of the DMA instructions): lanes then start an odd number of chunks apart.
//
// Arithmetic: exactly the generic kernel's (STRICT: separately rounded multiply and add, oldest sample first, first
// product initialises the accumulator, the start-from-zero quirk of the Vector seam variant, support.jl:46; FUSED:
// explicit fma) => bit-identical results.
#include "fir_stream_kernel.inc"
#include "mrhip_filter.h"   // kTapPad

namespace mrhip {

// instantiation units
hipError_t launch_fir_stream_f32(int nc, bool fused, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus);
hipError_t launch_fir_stream_f64(int nc, bool fused, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus);
hipError_t launch_fir_stream_mix(int nc, bool fused, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus);
hipError_t launch_fir_stream_rt(bool fused, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const PairArgs &pa, int num_cus);   // kernels_fir_stream_rt.hip

// Covers L == 1, 1 <= T <= 16384, every sample type, every M whose step fits a stage: M * bytes per sample <= ~900 with one
// output per lane (the run-time kernel's second lane map; ComplexF64: M <= 32), half that with two.  Returns false otherwise
// (the caller goes on to poly_tiled_kernel).
bool plan_fir_stream(const TypeKey &tk, const PolyArgs &a, int num_cus, PairArgs *out, dim3 *block, size_t *lds)
{
    if (!stream_env_int("MRHIP_STREAM", 1)) return false;   // read per call: tests switch kernels at run time
    if (a.L != 1 || (tk.x_f64 && !tk.r_f64)) return false;
    if (tk.r_f64 && !stream_env_int("MRHIP_STREAM_F64", 1)) return false;
    const int nc = tk.complex_x ? 2 : 1;
    const long long es = (tk.x_f64 ? 8 : 4) * nc;
    int pad_every = 0, min_taps = 0;   // (min_taps: one block of reads; shorter filters take the kernel's per-sample loop)
    const bool have_ct = a.M <= 64 && (es == 16 ? stream_geometry<16>(static_cast<int>(a.M), &pad_every, &min_taps)
                                       : es == 8 ? stream_geometry<8>(static_cast<int>(a.M), &pad_every, &min_taps) : stream_geometry<4>(static_cast<int>(a.M), &pad_every, &min_taps));
    // MRHIP_STREAM_RT: 0 = only the per-M instantiations, 1 = the run-time-M kernel where M is not instantiated, 2 = always
    const int rt_mode = stream_env_int("MRHIP_STREAM_RT", 1);
    const bool rt = rt_mode == 2 || (rt_mode == 1 && !have_ct);
    if (!rt && !have_ct) return false;
    int rt_rd = 16;
    bool single = false;                                    // one output per lane (kernels_fir_stream_rt.hip)
    if (rt) {                                               // the geometry of StreamGeo, at run time
        if (a.M < 1 || a.M + 16 > kTapPad) return false;    // (its blocks of tap reads stay inside the pads of the tap vector)
        // Two outputs per lane share most of a run, but a lane then owns 2 * M samples of LDS and a CU holds half the waves; one
        // output per lane reads T instead of (T + M) / 2 samples per output from LDS and has no mixed block in the middle of a
        // run.  Measured over M = 8..120, 24 and 128 taps, every sample type (profiles/r04/stream_rt_single_vs_pair.txt): one
        // output per lane wins (up to 1.8x) except where M is a whole number of blocks AND the pair still fills its waves
        // (128 <= M * es <= 192: 1//32 and 1//48 Float32, 1//16 ComplexF32 / Float64).  Its lane stride M * es must keep the
        // reads 8-byte aligned (Float32 with an odd M: pairs).  MRHIP_STREAM_RT_SINGLE=0|1 forces either.
        const int sm = stream_env_int("MRHIP_STREAM_RT_SINGLE", -1);
        const long long bs = es == 16 ? 8 : 16;             // the kernel's block of samples
        const bool pair_pref = a.M % bs == 0 && a.M * es >= 128 && a.M * es <= 192;
        single = (a.M * es) % 8 == 0 && (sm == 1 || (sm < 0 && !pair_pref));
        // ComplexF64 beyond 1//32: poly_tiled_kernel is ahead (1//36, 128 taps: 0.32 vs 0.68 ms)
        if (es == 16 && a.M > 32 && stream_env_int("MRHIP_STREAM_RT_ONE_WG", 0) == 0) return false;
        const long long S = (single ? 1 : 2) * a.M * es;
        rt_rd = S % 16 ? 8 : 16;
        const long long cd = rt_rd == 16 ? S / 16 : 0;
        pad_every = cd >= 2 && cd % 2 == 0 ? static_cast<int>(cd) : 0;
    }
    const long long opw = single ? 64 : 128;                // outputs per compute wave and step
    if (a.T < 1 || a.T > 16384) return false;   // (a tile must hold T samples: checked below)
    // compute waves: 3 (+ loader = a 256-thread workgroup) unless overridden; a step is 2 outputs per lane
    int ncw = stream_env_int("MRHIP_STREAM_WAVES", 3);
    if (ncw < 1) ncw = 1;
    if (ncw > 7) ncw = 7;
    while (ncw > 1 && opw * ncw * a.M * es > 24 * 1024) --ncw;   // a step of wide samples at a large decimation must leave room for two stages
    const long long P = opw * ncw, cM = P * a.M;
    const long long tail = a.T + 16;                            // run overhang T - M beyond the step, + the rounding of the last read
    // (the head block reads a whole block: BS <= T samples, inside the run)
    const int wg_per_cu = std::max(1, std::min(6, 32 / (ncw + 1)));
    const int ns = 2;
    // stage size: as many steps as the LDS share allows, at most 60 DMA slots
    auto lds_chunks = [&](long long j) {
        const long long nchunks = ((j * cM + tail) * es + 15) / 16;
        return pad_every > 0 ? (nchunks + pad_every - 1) / pad_every * (pad_every + 1) : nchunks;
    };
    const long long budget = (150LL * 1024 / wg_per_cu - 64) / ns;
    long long J = stream_env_int("MRHIP_STREAM_J", 0);
    if (J <= 0) {
        J = 1;
        while (J < 64 && (lds_chunks(J + 1) + 63) / 64 * 1024 <= budget && (lds_chunks(J + 1) + 63) / 64 <= 60) ++J;
        const long long want_tiles = 4LL * num_cus;            // small problems: enough tiles for every CU
        while (J > 1 && ((a.n_out + J * P - 1) / (J * P)) * a.nch < want_tiles) J = (J + 1) / 2;
    }
    const long long nslots = (lds_chunks(J) + 63) / 64;
    if (nslots > 60 || nslots * 1024 * ns > 150 * 1024) return false;
    // Pairs with one workgroup (= one compute wave at these sizes) per CU: 8- and 16-byte samples are then faster on poly_tiled_kernel
    // (measured, profiles/r04/stream_rt_vs_per_m.txt: 1//38 ComplexF32 0.31 vs 0.25 ms); Float32 samples are not (1//80: 0.22 vs 0.46)
    if (rt && !single && es >= 8 && nslots * 1024 * ns + 64 > 78 * 1024 && stream_env_int("MRHIP_STREAM_RT_ONE_WG", 0) == 0) return false;
    const size_t stage_bytes = static_cast<size_t>(nslots) * 1024;
    PairArgs pa{};
    pa.c = ncw; pa.P = static_cast<int>(P); pa.cM = static_cast<int>(cM);
    pa.J = static_cast<int>(J);
    pa.tail = static_cast<int>(tail);
    pa.tile_len = static_cast<int>(J * cM + tail);
    pa.dma_rounds = static_cast<int>(nslots);
    pa.stage_bytes = static_cast<int>(stage_bytes);
    pa.ns = ns;
    pa.nc = nc;
    pa.x_f64 = tk.x_f64 ? 1 : 0; pa.r_f64 = tk.r_f64 ? 1 : 0;
    pa.pad_every = pad_every;
    pa.rt = rt ? (single ? 2 : 1) : 0; pa.rt_rd = rt_rd;
    pa.bank_off = -1;
    pa.o0 = a.d0 - a.T;                      // x index of LDS sample 0 of a channel's first tile (negative => history)
    pa.tile_in = J * cM;
    pa.tile_out = J * P;
    pa.tiles_per_channel = (a.n_out + pa.tile_out - 1) / pa.tile_out;
    pa.total_tiles = pa.tiles_per_channel * a.nch;
    if (a.n_out >= (1LL << 31) - pa.tile_out || pa.total_tiles >= (1LL << 31) - 65536) return false;   // 32-bit tile walk
    if (a.x_len >= (1LL << 31) - cM) return false;
    const long long spc = (a.n_out + P - 1) / P;
    if (spc * a.nch >= (1LL << 31)) return false;
    pa.steps_per_channel = static_cast<unsigned>(spc);
    pa.total_steps = static_cast<unsigned>(spc * a.nch);
    pa.spc_magic = spc == 1 ? 0xffffffffu : static_cast<unsigned>((1ULL << 32) / static_cast<unsigned long long>(spc));
    pa.flags_off = static_cast<int>(ns * stage_bytes);
    *lds = ns * stage_bytes + 8 * ns;
#if MRHIP_STREAM_TAPS_LDS
    if (!rt) {                               // the per-M instantiations read their taps from LDS (fir_stream_kernel.inc)
        pa.bank_off = static_cast<int>((*lds + 15) / 16 * 16);
        *lds = static_cast<size_t>(pa.bank_off) + static_cast<size_t>(a.T) * (tk.r_f64 ? 8 : 4) + 64;
    }
#endif
    *out = pa;
    *block = dim3(static_cast<unsigned>(64 * (ncw + 1)));
    return true;
}

hipError_t launch_fir_stream(bool fused, const PolyArgs &a, const PairArgs &pa_in, dim3 block, size_t lds, hipStream_t s,
                             const char **kname, int num_cus, unsigned *counters)
{
    if (!counters) return hipErrorInvalidValue;
    PairArgs pa = pa_in;
    pa.counters = counters;
    if (pa.rt) {
        *kname = "fir_stream_rt_kernel";
        return launch_fir_stream_rt(fused, block, lds, s, a, pa, num_cus);
    }
    *kname = "fir_stream_kernel";
    if (pa.r_f64)
        return pa.x_f64 ? launch_fir_stream_f64(pa.nc, fused, block, lds, s, a, pa, num_cus) : launch_fir_stream_mix(pa.nc, fused, block, lds, s, a, pa, num_cus);
    return launch_fir_stream_f32(pa.nc, fused, block, lds, s, a, pa, num_cus);
}

}  // namespace mrhip
