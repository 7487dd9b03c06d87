// kernels_interp_pair.hip -- FIRInterpolator (L//1), Float32 taps, Float32 or ComplexF32 samples,
// tapsPerPhi <= 32: TWO PHASES per lane.
//
// reference: src/Filters.jl:489-517 (filt!), dot: src/support.jl:5-31 (matrix unsafedot, no start-from-zero).
//
// Every input position p produces the L outputs p*L .. p*L+L-1 (phases 1..L), all from the SAME window of
// tapsPerPhi samples ending at p.  A lane owns one position of the step and two adjacent phases (2q, 2q+1):
// it keeps two tap columns in VGPRs for its whole life (the phase of a lane never changes, whatever the
// step stride) and feeds both dot products from one fetch of the window -- half the LDS reads per output of
// the one-output-per-lane kernel (kernels_phase_stationary.hip), which is what bounded it: at 128 VALU
// operations and 32 LDS reads per ComplexF32 output neither pipe had room to hide the other.  The lanes of a
// position read identical LDS addresses (broadcast), consecutive positions consecutive samples: conflict-free.
// Consecutive lanes own consecutive output pairs, so results go straight from the accumulators to one
// dense 2-sample store per lane (no staging through LDS).
//
// Everything around the dot products is the machinery of the rational kernel (pair_loader.h): a loader wave streams
// tiles HBM -> LDS with LDS-DMA three stages deep, hands out the work in dynamically drawn grabs of J steps
// (32 XCD-local counters), publishes tiles through LDS, and performs shiftin! at the end of the launch; the
// compute waves fetch the window through a small register ring with compile-time wait counts.
//
// Arithmetic: exactly the generic kernel's (STRICT: separately rounded multiply and add, oldest sample first,
// first product initialises the accumulator; FUSED: explicit fma) => bit-identical results.
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

constexpr int kIntMaxThreads = 512;
constexpr int kIntGroups = 32;

inline int int_env_int(const char *name, int dflt)
{
    const char *v = std::getenv(name);
    return v && *v ? std::atoi(v) : dflt;
}
inline bool int_debug_once()
{
    static int state = -1;
    if (state < 0) { const char *v = std::getenv("MRHIP_DEBUG"); state = (v && v[0] == '1') ? 1 : 0; }
    if (state == 1) { state = 0; return true; }
    return false;
}

// PairArgs is reused with these meanings: cM = positions per step (CP), c = lanes per position (LP = ceil(L/2)),
// P = outputs per step (CP*L); the scheduling fields are identical.
template <int T, bool FUSED, int NC>
__global__ __launch_bounds__(kIntMaxThreads + 64)
void interp_pair_kernel(PolyArgs a, PairArgs pa)
{
    constexpr unsigned ES = 4u * NC;            // bytes per sample
    using samp_t = std::conditional_t<NC == 1, unsigned, v2u_t>;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int ncw = (blockDim.x >> 6) - 1;      // compute waves; the last wave of the workgroup is the loader
    const int CP = pa.cM, LP = pa.c, L = a.L;

    // ---- tile walk shared by both roles (see opair_kernel.inc): steps of CP positions, numbered
    // channel-major, drawn in grabs of J from 32 group counters by the loader wave, published through LDS
    auto tile_at = [&](unsigned g, unsigned jt) -> TileAt { return pair_tile_at(pa, g, jt); };
    volatile unsigned *const tile_flag = reinterpret_cast<volatile unsigned *>(smem + pa.flags_off);   // [ns][2]

    if (wave == ncw) {
        // ================= loader wave (pair_loader.h): cM = positions per step, tail = T - 1, o0 = -(T - 1) =================
        pair_loader_wave<NC>(a, pa, smem, lane);
        return;
    }

    // ================= compute waves =================
    // lane -> (position within the step, phase pair)
    const int posl = tid / LP, q = tid - posl * LP;
    const bool lane_act = posl < CP;
    const int ph0 = 2 * q, ph1 = 2 * q + 1;
    const bool has1 = ph1 < L;
    float taps[2][T];
    {
        const float *__restrict__ t0 = static_cast<const float *>(a.taps) + static_cast<long long>(ph0 < L ? ph0 : 0) * T;
        const float *__restrict__ t1 = static_cast<const float *>(a.taps) + static_cast<long long>(has1 ? ph1 : 0) * T;
#pragma unroll
        for (int i = 0; i < T; ++i) { taps[0][i] = t0[i]; taps[1][i] = t1[i]; }
    }
    const int x_len = static_cast<int>(a.x_len);
    // outputs of one position are contiguous: y[(p*L + ph)]; this lane's first output inside a step
    const unsigned out_in_step = static_cast<unsigned>(posl) * static_cast<unsigned>(L) + static_cast<unsigned>(ph0);

    for (int s = 0;; s = (s + 1 == pa.ns ? 0 : s + 1)) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned tg = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(tile_flag[2 * s])));
        const unsigned tj = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(tile_flag[2 * s + 1])));
        if (tj == 0u) break;
        const TileAt ta = tile_at(tg, tj);
        const int J = ta.jt;
        const int p0 = ta.st * CP;                                       // first position (0-based) of the tile
        float *__restrict__ yc = static_cast<float *>(a.y) + (static_cast<long long>(ta.ch) * a.y_stride + static_cast<long long>(p0) * L) * NC;
        char *const ybytes = reinterpret_cast<char *>(yc);
        const int npos = x_len - p0;                                     // positions of this channel from this tile on
        const unsigned wbase = lds_base + static_cast<unsigned>(s) * pa.stage_bytes + static_cast<unsigned>(posl) * ES;

        // Ring-pipelined window: sample i of the window is read K samples before it is consumed; a step is
        // padded to TV = a multiple of K virtual samples so that slot numbers repeat (no strip operations here:
        // ring_younger(.., strip_ops = 0)).
        constexpr int K = T < 6 ? T : 6;
        constexpr int TV = (T + K - 1) / K * K;
        samp_t ring[K];
        auto read_samp = [&](auto off_tag, unsigned addr) -> samp_t {
            constexpr int OFF = decltype(off_tag)::value;
            if constexpr (NC == 1) return lds_read_b32<OFF * 4>(addr);
            else return lds_read_b64<OFF * 8>(addr);
        };
        static_for<0, K>([&](auto I) { ring[decltype(I)::value] = read_samp(I, wbase); });
#pragma unroll 1
        for (int j = 0; j < J; ++j) {
            const int jn = j + 1 < J ? j + 1 : j;
            const unsigned wcur = wbase + static_cast<unsigned>(j) * CP * ES;
            const unsigned wnext = wbase + static_cast<unsigned>(jn) * CP * ES;
            float acc0[NC] = {}, acc1[NC] = {};
            static_for<0, TV>([&](auto I) {
                constexpr int i = decltype(I)::value;
                constexpr int slot = i % K;
                if constexpr (i < T) {
                    lgkm_wait<ring_younger(i, T, K, 0)>(ring[slot]);
                    float w[NC];
                    if constexpr (NC == 1) w[0] = __uint_as_float(ring[slot]);
                    else { w[0] = __uint_as_float(ring[slot].x); w[1] = __uint_as_float(ring[slot].y); }
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) {
                        if constexpr (i == 0) { acc0[cc] = taps[0][0] * w[cc]; acc1[cc] = taps[1][0] * w[cc]; }
                        else { acc0[cc] = macf<FUSED>(taps[0][i], w[cc], acc0[cc]); acc1[cc] = macf<FUSED>(taps[1][i], w[cc], acc1[cc]); }
                    }
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) { pin(acc0[cc]); pin(acc1[cc]); }
                }
                if constexpr (i + K < T) ring[slot] = read_samp(std::integral_constant<int, i + K>{}, wcur);
                else if constexpr (i + K >= TV) ring[slot] = read_samp(std::integral_constant<int, i + K - TV>{}, wnext);
            });
            // store: outputs (position j*CP + posl, phases ph0, ph1) are adjacent in y
            const int pi = j * CP + posl;
            if (lane_act && pi < npos && !(pa.ablate & 2)) {
                char *const dst = ybytes + (static_cast<unsigned>(j) * static_cast<unsigned>(pa.P) + out_in_step) * ES;
                if (has1) {
                    float o2[2 * NC];
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) { o2[cc] = acc0[cc]; o2[NC + cc] = acc1[cc]; }
                    __builtin_memcpy(dst, o2, 2 * ES);
                } else {
                    __builtin_memcpy(dst, acc0, ES);
                }
            } else if (pa.ablate & 2) {
                if (acc0[0] == 1.2345e30f) *reinterpret_cast<float *>(ybytes) = acc1[0];
            }
        }
        // retire the last step's unused look-ahead reads before the stage can be overwritten
        lgkm_wait<0>(ring[0]);
        static_for<1, K>([&](auto I) { pin(ring[decltype(I)::value]); });
    }
}

template <bool FUSED, int NC>
hipError_t launch_interp_T(int T, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, PairArgs pa, int num_cus)
{
#define MRHIP_CASE(TT)                                                                              \
    case TT: {                                                                                      \
        auto kfn = interp_pair_kernel<TT, FUSED, NC>;                                               \
        if (lds > 48 * 1024) {                                                                      \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),                 \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)); \
            if (e != hipSuccess) return e;                                                          \
        }                                                                                           \
        int per_cu = 0;                                                                             \
        hipError_t eo = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, static_cast<int>(block.x), lds); \
        if (eo != hipSuccess) return eo;                                                            \
        if (per_cu < 1) per_cu = 1;                                                                 \
        long long g = static_cast<long long>(num_cus) * per_cu;                                     \
        if (g > static_cast<long long>(pa.total_steps)) g = pa.total_steps;                         \
        if (g < 1) g = 1;                                                                           \
        pa.ngroups = static_cast<int>(g < kIntGroups ? g : kIntGroups);                             \
        pa.steps_per_group = static_cast<unsigned>((pa.total_steps + pa.ngroups - 1) / pa.ngroups); \
        pa.static_grabs = (static_cast<long long>(pa.total_steps) + pa.J - 1) / pa.J <= 3 * g;      \
        if (int_debug_once()) {                                                                     \
            hipFuncAttributes fa;                                                                   \
            (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kfn));                   \
            std::fprintf(stderr, "[mrhip] interp_pair T=%d L=%d grid=%lld block=%u lds=%zu occ/CU=%d regs=%d CP=%d LP=%d J=%d steps=%u\n", \
                         TT, a.L, g, block.x, lds, per_cu, fa.numRegs, pa.cM, pa.c, pa.J, pa.total_steps); \
        }                                                                                           \
        launch_kernel(kfn, dim3(static_cast<unsigned>(g)), block, lds, s, a, pa);              \
        return hipGetLastError();                                                                   \
    }
    switch (T) {
#ifdef MRHIP_PS_FAST_BUILD
        MRHIP_CASE(32)
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

// Covers FIRInterpolator (M == 1, L >= 2), Float32 taps, Float32 / ComplexF32 samples, tapsPerPhi <= 32.
bool plan_interp_pair(const TypeKey &tk, const PolyArgs &a, int num_cus, PairArgs *out, dim3 *block, size_t *lds)
{
    static const int enabled = int_env_int("MRHIP_INTERP", 1);
    if (!enabled) return false;
    if (tk.x_f64 || tk.r_f64) return false;
    if (a.M != 1 || a.L < 2 || a.T < 1 || a.T > 32 || a.zero_start_below > 0) return false;
#ifdef MRHIP_PS_FAST_BUILD
    if (a.T != 32) return false;
#endif
    if (a.x_len >= (1LL << 31) / (a.L + 1) || a.n_out >= (1LL << 31)) return false;   // 32-bit walk
    const int nc = tk.complex_x ? 2 : 1;
    const long long es = 4 * nc;
    const int LP = (a.L + 1) / 2;
    if (LP > 256) return false;
    static const int env_j = int_env_int("MRHIP_INTERP_J", 0), env_w = int_env_int("MRHIP_INTERP_WAVES", 0);
    const int ns = 3;
    // lanes: 4 compute waves by default; CP = positions per step
    const int want_lanes = (env_w > 0 ? env_w : 4) * 64;
    int CP = want_lanes / LP;
    if (CP < 1) CP = 1;
    const int lanes = CP * LP;
    const int padded = (lanes + 63) / 64 * 64;
    const int nwaves = padded / 64;
    // stage sized for four workgroups per CU
    const long long budget = (160 * 1024 / 4 - 64) / ns / 1024 * 1024;
    long long J = (budget / es - (a.T - 1) - 4) / CP;
    if (env_j > 0) J = env_j;
    if (J < 1) J = 1;
    if (J > 64) J = 64;
    const long long spc = (a.x_len + CP - 1) / CP;                       // steps per channel
    if (env_j <= 0) {
        const long long want_tiles = 4LL * num_cus;
        while (J > 1 && ((spc + J - 1) / J) * a.nch < want_tiles) J = (J + 1) / 2;
    }
    const long long tile_len = (J * CP + a.T - 1 + 3) / 4 * 4;
    const long long nslots = (tile_len * es / 16 + 63) / 64;
    const size_t stage_bytes = static_cast<size_t>(nslots) * 1024;
    if (nslots > 60 / (ns - 2) || ns * stage_bytes > 150 * 1024) return false;
    if (spc * a.nch >= (1LL << 31)) return false;
    PairArgs pa{};
    pa.c = LP; pa.cM = CP; pa.P = static_cast<int>(static_cast<long long>(CP) * a.L);
    pa.J = static_cast<int>(J);
    pa.tile_len = static_cast<int>(tile_len);
    pa.dma_rounds = static_cast<int>(nslots);
    pa.stage_bytes = static_cast<int>(stage_bytes);
    pa.ns = ns; pa.nc = nc;
    pa.o0 = -(a.T - 1);           // x index of LDS sample 0 of a channel's first tile: the oldest sample of its first position
    pa.tail = a.T - 1;            // a tile of j steps needs j*CP + T - 1 samples (pair_loader.h)
    pa.bank_off = -1; pa.pad_every = 0;
    static const int env_ablate = int_env_int("MRHIP_PS_ABLATE", 0);
    pa.ablate = env_ablate;
    pa.steps_per_channel = static_cast<unsigned>(spc);
    pa.total_steps = static_cast<unsigned>(spc * a.nch);
    pa.spc_magic = spc == 1 ? 0xffffffffu : static_cast<unsigned>((1ULL << 32) / static_cast<unsigned long long>(spc));
    pa.flags_off = static_cast<int>(ns * stage_bytes);
    *out = pa;
    *block = dim3(static_cast<unsigned>(nwaves * 64 + 64));   // + the loader wave
    *lds = ns * stage_bytes + 8 * ns;
    return true;
}

hipError_t launch_interp_pair(bool fused, const PolyArgs &a, const PairArgs &pa_in, dim3 block, size_t lds, hipStream_t s,
                              const char **kname, int num_cus, unsigned *counters)
{
    if (!counters) return hipErrorInvalidValue;
    PairArgs pa = pa_in;
    pa.counters = counters;
    pa.probe = nullptr;
    *kname = "interp_pair_kernel";
    if (pa.nc == 2)
        return fused ? launch_interp_T<true, 2>(a.T, block, lds, s, a, pa, num_cus)
                     : launch_interp_T<false, 2>(a.T, block, lds, s, a, pa, num_cus);
    return fused ? launch_interp_T<true, 1>(a.T, block, lds, s, a, pa, num_cus)
                 : launch_interp_T<false, 1>(a.T, block, lds, s, a, pa, num_cus);
}

}  // namespace mrhip
