// kernels_fir_stream.hip -- FIRStandard (M = 1) and FIRDecimator (L = 1, M = 2..16) with Float32 arithmetic
// (Float32 or ComplexF32 samples, Float32 taps), 16 (or one block of reads) to 512 taps: the streaming form of
// kernels_fir_direct.hip (BASELINE config 3b).
//
// reference: src/Filters.jl:450-473 (Standard), :598-631 (Decimator); dot: src/support.jl:33-55.
//
// What was wrong with the direct kernel: it stages a tile synchronously (all threads load, barrier, compute, barrier)
// through a transposed LDS layout and reads ONE sample per LDS instruction; it sat at 30-44 % of the multiply+add
// rate.  Here the machinery of the rational kernel is reused (pair_loader.h): a loader wave streams tiles HBM -> LDS
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
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "mrhip_internal.h"
#include "pair_device.h"
#include "pair_loader.h"

#pragma clang fp contract(off)

namespace mrhip {
namespace {

using namespace dev;

constexpr int kSMaxThreads = 512;
constexpr int kSGroups = 32;

inline int stream_env_int(const char *name, int dflt)
{
    const char *v = std::getenv(name);
    return v && *v ? std::atoi(v) : dflt;
}

// geometry of one (NC, M) instantiation
template <int NC, int M>
struct StreamGeo {
    static constexpr int ES = 4 * NC;                       // bytes per sample
    static constexpr int S = 2 * M * ES;                    // bytes between the runs of adjacent lanes
    static constexpr int RD = S % 16 ? 8 : 16;              // bytes per LDS read
    static constexpr int SPR = RD / ES;                     // samples per read
    static constexpr int CD = RD == 16 ? S / 16 : 0;        // data chunks per lane stride (16-byte reads only)
    static constexpr bool PAD = CD >= 2 && CD % 2 == 0;
    // reads per block: a multiple of CD when padded (a block must advance a whole number of pad periods), and enough
    // samples to reach past the second output's start
    static constexpr int CPB = PAD ? (CD > 4 ? CD : 4) : (4 * SPR > M ? 4 : 8);
    static constexpr int BS = CPB * SPR;                    // samples per block
    static_assert(BS > M && BS % SPR == 0 && (!PAD || CPB % CD == 0), "block geometry");
    // byte offset of read i of a lane's run (i = block * CPB + ii): pads after every CD chunks
    static constexpr int read_off(int i) { return PAD ? 16 * (i + i / CD) : RD * i; }
    static constexpr int block_bytes = PAD ? 16 * (CPB + CPB / CD) : RD * CPB;   // LDS bytes a block of reads advances
    static constexpr int lane_bytes = PAD ? 16 * (CD + 1) : S;                   // LDS bytes between adjacent lanes
};

template <int NC, int M, bool FUSED>
__global__ __launch_bounds__(kSMaxThreads + 64)
void fir_stream_kernel(PolyArgs a, PairArgs pa)
{
    using G = StreamGeo<NC, M>;
    constexpr int BS = G::BS, SPR = G::SPR, CPB = G::CPB;
    using read_t = std::conditional_t<G::RD == 8, v2u_t, v4u_t>;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int ncw = (blockDim.x >> 6) - 1;      // compute waves; the last wave is the loader

    if (wave == ncw) {
        pair_loader_wave<NC>(a, pa, smem, lane);
        return;
    }
    volatile unsigned *const tile_flag = reinterpret_cast<volatile unsigned *>(smem + pa.flags_off);
    // the tap vector is never written while a filter exists: through the constant address space the compiler uses
    // scalar loads for the wave-uniform tap indices below (it cannot prove that for a global pointer next to the y stores)
    typedef const __attribute__((address_space(4))) float *const_taps_t;
    // (the tap vector is a hipMalloc allocation: 256-byte aligned, which lets the tap loads of a block merge into
    //  s_load_dwordx4/x8/x16)
    const const_taps_t tc = (const_taps_t)(static_cast<const float *>(__builtin_assume_aligned(a.taps, 64)));
    const int T = a.T;
    const int NB = T / BS;                      // whole blocks of the first output's window (>= 1: the plan requires T >= BS)
    const int n_out = static_cast<int>(a.n_out);
    const int lanes = pa.P >> 1;                // lanes that own an output pair

    for (int s = 0;; s = (s + 1 == pa.ns ? 0 : s + 1)) {
        // One barrier per tile and no memory wait (see opair_kernel.inc): the loader arrives only after this tile's
        // data has landed and its descriptor is in LDS.
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned tg = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(tile_flag[2 * s])));
        const unsigned tj = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(tile_flag[2 * s + 1])));
        if (tj == 0u) break;                      // end marker
        const TileAt ta = pair_tile_at(pa, tg, tj);
        const int J = ta.jt;
        float *__restrict__ yc = static_cast<float *>(a.y) + (static_cast<long long>(ta.ch) * a.y_stride + static_cast<long long>(ta.st) * pa.P) * NC;
        const int first_out = ta.st * pa.P;                               // channel-relative index of the tile's first output
        const int remaining = n_out - first_out;
        // start-from-zero quirk (support.jl:46): outputs whose newest-sample index n = d0 + k*M is below the threshold
        const bool tile_has_zs = a.d0 + static_cast<long long>(first_out) * M < a.zero_start_below;   // wave-uniform
        const unsigned char *const stage = smem + static_cast<size_t>(s) * pa.stage_bytes;
        if (tid < lanes) {
#pragma unroll 1
            for (int j = 0; j < J; ++j) {
                const int k0 = j * pa.P + 2 * tid;                        // tile-relative index of this lane's first output
                if (k0 >= remaining) break;
                // the lane's run: step j starts lanes * lane_bytes further on (a step is `lanes` lane strides of samples)
                const unsigned char *const run = stage + (static_cast<size_t>(j) * lanes + tid) * G::lane_bytes;
                float acc0[NC], acc1[NC];
                auto load_block = [&](int b, float (&w)[BS][NC], auto nreads_tag) {
                    constexpr int NR = decltype(nreads_tag)::value;
                    const unsigned char *const p = run + static_cast<size_t>(b) * G::block_bytes;
#pragma unroll
                    for (int ii = 0; ii < NR; ++ii) {
                        const read_t v = *reinterpret_cast<const read_t *>(p + G::read_off(ii));
                        unsigned u[4];
                        if constexpr (G::RD == 8) { u[0] = v.x; u[1] = v.y; u[2] = u[3] = 0u; }
                        else { u[0] = v.x; u[1] = v.y; u[2] = v.z; u[3] = v.w; }
#pragma unroll
                        for (int e = 0; e < SPR; ++e)
#pragma unroll
                            for (int cc = 0; cc < NC; ++cc) w[ii * SPR + e][cc] = __uint_as_float(u[e * NC + cc]);
                    }
                };
                // ComplexF32: (re, im) of a sample share the tap -> one packed multiply and one packed add per sample and
                // output, written out by hand (this file is compiled with -fno-slp-vectorize: left to itself the
                // vectoriser packs the second output ACROSS samples and pays two v_mov shuffles per packed operation)
                auto mac = [&](float (&acc)[NC], float t, const float (&w)[NC]) {
                    if constexpr (NC == 2) {
                        v2f_t av = {acc[0], acc[1]};
                        const v2f_t wv = {w[0], w[1]}, tv = {t, t};
                        if constexpr (FUSED) av = __builtin_elementwise_fma(tv, wv, av);
                        else { const v2f_t p = tv * wv; av = av + p; }
                        acc[0] = av.x; acc[1] = av.y;
                    } else {
                        if constexpr (FUSED) acc[0] = __builtin_fmaf(t, w[0], acc[0]);
                        else { const float p = t * w[0]; acc[0] = acc[0] + p; }
                    }
                };
                auto init = [&](float (&acc)[NC], float t, const float (&w)[NC], bool zs) {   // first product initialises (support.jl:35,46)
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) acc[cc] = t * w[cc];
                    if (zs) {
#pragma unroll
                        for (int cc = 0; cc < NC; ++cc) acc[cc] = 0.f + acc[cc];
                    }
                };
                bool zs0 = false, zs1 = false;
                if (tile_has_zs) {
                    const long long n0 = a.d0 + static_cast<long long>(first_out + k0) * M;
                    zs0 = n0 < a.zero_start_below; zs1 = n0 + M < a.zero_start_below;
                }
                float w[BS][NC];
                // head block: the second output starts at sample M
                load_block(0, w, std::integral_constant<int, CPB>{});
#pragma unroll
                for (int e = 0; e < BS; ++e) {
                    if (e == 0) init(acc0, tc[0], w[0], zs0); else mac(acc0, tc[e], w[e]);
                    if (e == M) init(acc1, tc[0], w[e], zs1); else if (e > M) mac(acc1, tc[e - M], w[e]);
                }
                // body blocks: both outputs over all BS samples; unrolled so that the compiler issues the LDS reads and
                // the scalar tap loads of the following blocks ahead of the arithmetic of the current one
#ifndef MRHIP_STREAM_UNROLL
#define MRHIP_STREAM_UNROLL 2
#endif
#pragma unroll MRHIP_STREAM_UNROLL
                for (int b = 1; b < NB; ++b) {
                    load_block(b, w, std::integral_constant<int, CPB>{});
                    const int jb = b * BS;
#pragma unroll
                    for (int e = 0; e < BS; ++e) {
                        mac(acc0, tc[jb + e], w[e]);
                        mac(acc1, tc[jb + e - M], w[e]);
                    }
                }
                // tail: what is left of the first output's window when T is not a whole number of blocks (both outputs),
                // then the M samples past it, which belong to the second output alone; one read at a time, the
                // conditions are wave-uniform (j and T are)
                for (int j0 = NB * BS; j0 < T + M; j0 += SPR) {
                    const int ri = j0 / SPR;                               // read index inside the lane's run
                    const unsigned char *const p = run + (G::PAD ? 16 * (ri + ri / (G::CD > 0 ? G::CD : 1)) : G::RD * ri);
                    const read_t v = *reinterpret_cast<const read_t *>(p);
                    unsigned u[4];
                    if constexpr (G::RD == 8) { u[0] = v.x; u[1] = v.y; u[2] = u[3] = 0u; }
                    else { u[0] = v.x; u[1] = v.y; u[2] = v.z; u[3] = v.w; }
#pragma unroll
                    for (int e = 0; e < SPR; ++e) {
                        float we[NC];
#pragma unroll
                        for (int cc = 0; cc < NC; ++cc) we[cc] = __uint_as_float(u[e * NC + cc]);
                        const int jj = j0 + e;
                        if (jj < T) mac(acc0, tc[jj], we);
                        if (jj < T + M) mac(acc1, tc[jj - M], we);
                    }
                }

                float *const dst = yc + static_cast<long long>(k0) * NC;
                if (k0 + 1 < remaining) {
                    float o2[2 * NC];
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) { o2[cc] = acc0[cc]; o2[NC + cc] = acc1[cc]; }
                    __builtin_memcpy(dst, o2, sizeof(o2));
                } else {
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) dst[cc] = acc0[cc];
                }
            }
        }
    }
}

template <int NC, int M>
hipError_t launch_stream_nm(bool fused, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, PairArgs pa, int num_cus)
{
    auto go = [&](auto kfn) -> hipError_t {
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
            if (e != hipSuccess) return e;
        }
        int per_cu = 0;
        hipError_t eo = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, static_cast<int>(block.x), lds);
        if (eo != hipSuccess) return eo;
        if (per_cu < 1) per_cu = 1;
        const int bpc = stream_env_int("MRHIP_STREAM_BPC", 0);
        if (bpc > 0) per_cu = bpc;
        long long g = static_cast<long long>(num_cus) * per_cu;
        if (g > static_cast<long long>(pa.total_steps)) g = pa.total_steps;
        if (g < 1) g = 1;
        pa.ngroups = static_cast<int>(g < kSGroups ? g : kSGroups);
        pa.steps_per_group = static_cast<unsigned>((pa.total_steps + pa.ngroups - 1) / pa.ngroups);
        pa.static_grabs = (static_cast<long long>(pa.total_steps) + pa.J - 1) / pa.J <= 3 * g;
        static int dbg = stream_env_int("MRHIP_DEBUG", 0);
        if (dbg == 1) {
            dbg = 0;
            hipFuncAttributes fa;
            (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kfn));
            std::fprintf(stderr, "[mrhip] fir_stream T=%d M=%d nc=%d grid=%lld block=%u lds=%zu occ/CU=%d regs=%d P=%d cM=%d J=%d ns=%d pad_every=%d\n",
                         a.T, M, NC, g, block.x, lds, per_cu, fa.numRegs, pa.P, pa.cM, pa.J, pa.ns, pa.pad_every);
        }
        launch_kernel(kfn, dim3(static_cast<unsigned>(g)), block, lds, s, a, pa);
        return hipGetLastError();
    };
    return fused ? go(fir_stream_kernel<NC, M, true>) : go(fir_stream_kernel<NC, M, false>);
}

// the decimations the kernel is instantiated for
#define MRHIP_STREAM_MS(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)

template <int NC>
bool stream_geometry(int M, int *pad_every, int *min_taps)      // false: M is not instantiated
{
    switch (M) {
#define MRHIP_X(MV) case MV: *pad_every = StreamGeo<NC, MV>::PAD ? StreamGeo<NC, MV>::CD : 0; *min_taps = StreamGeo<NC, MV>::BS; return true;
        MRHIP_STREAM_MS(MRHIP_X)
#undef MRHIP_X
    default: return false;
    }
}

}  // namespace

// Covers L == 1 with Float32 arithmetic, M <= 16, max(16, one block) <= T <= 512.  Returns false
// otherwise (the caller falls back to kernels_fir_direct.hip).
bool plan_fir_stream(const TypeKey &tk, const PolyArgs &a, int num_cus, PairArgs *out, dim3 *block, size_t *lds)
{
    if (!stream_env_int("MRHIP_STREAM", 1)) return false;   // read per call: tests switch kernels at run time
    if (tk.x_f64 || tk.r_f64 || a.L != 1) return false;
    const int nc = tk.complex_x ? 2 : 1;
    const long long es = 4 * nc;
    int pad_every = 0, min_taps = 0;
    if (a.M > 16 || !(nc == 2 ? stream_geometry<2>(static_cast<int>(a.M), &pad_every, &min_taps) : stream_geometry<1>(static_cast<int>(a.M), &pad_every, &min_taps))) return false;
    if (a.T < std::max(16, min_taps) || a.T > 512) return false;   // the head block is a whole block of taps
    // compute waves: 3 (+ loader = a 256-thread workgroup) unless overridden; a step is 2 outputs per lane
    int ncw = stream_env_int("MRHIP_STREAM_WAVES", 3);
    if (ncw < 1) ncw = 1;
    if (ncw > 7) ncw = 7;
    const long long P = 128LL * ncw, cM = P * a.M;
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
    pa.pad_every = pad_every;
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
    *out = pa;
    *block = dim3(static_cast<unsigned>(64 * (ncw + 1)));
    *lds = ns * stage_bytes + 8 * ns;
    return true;
}

hipError_t launch_fir_stream(bool fused, const PolyArgs &a, const PairArgs &pa_in, dim3 block, size_t lds, hipStream_t s,
                             const char **kname, int num_cus, unsigned *counters)
{
    if (!counters) return hipErrorInvalidValue;
    PairArgs pa = pa_in;
    pa.counters = counters;
    *kname = "fir_stream_kernel";
#define MRHIP_X2(MV) case MV: return launch_stream_nm<2, MV>(fused, block, lds, s, a, pa, num_cus);
#define MRHIP_X1(MV) case MV: return launch_stream_nm<1, MV>(fused, block, lds, s, a, pa, num_cus);
    if (pa.nc == 2) {
        switch (a.M) { MRHIP_STREAM_MS(MRHIP_X2) default: return hipErrorInvalidValue; }
    }
    switch (a.M) { MRHIP_STREAM_MS(MRHIP_X1) default: return hipErrorInvalidValue; }
#undef MRHIP_X1
#undef MRHIP_X2
}

}  // namespace mrhip
