// kernels_rational_opair.hip -- FIRRational with L > M (sample-rate increase by a ratio below two, e.g. 160//147,
// 3//2) and the M > L ratios the position-pair kernel does not take: Float32 arithmetic, tapsPerPhi <= 32.
//
// Why another mapping.  kernels_rational_pair.hip gives a lane two adjacent INPUT positions; with M > L a position
// produces at most one output, so the lane runs two chains.  With L > M a position produces one or two outputs and
// that mapping would need a third, mostly idle chain.  Here a lane owns two adjacent OUTPUTS A = 2l, B = 2l+1 of the
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
#include <type_traits>

#include "mrhip_internal.h"
#include "pair_device.h"
#include "pair_loader.h"

#pragma clang fp contract(off)

namespace mrhip {
namespace {

constexpr int kOMaxThreads = 512;
constexpr int kOGroups = 32;

inline int opair_env_int(const char *name, int dflt)
{
    const char *v = std::getenv(name);
    return v && *v ? std::atoi(v) : dflt;
}

using namespace dev;

// one multiply-accumulate of a slot; COND: the slot is outside the window for the lanes where `used` is false
template <bool FUSED, bool FIRST, bool COND>
__device__ __forceinline__ float slot_mac(float tap, float w, float acc, bool used)
{
    if constexpr (FIRST) {
        const float p = tap * w;                       // the first product initialises the accumulator
        if constexpr (COND) return used ? p : -0.0f;
        else return p;
    } else if constexpr (FUSED) {
        const float r = __builtin_fmaf(tap, w, acc);
        if constexpr (COND) return used ? r : acc;
        else return r;
    } else {
        float p = tap * w;
        if constexpr (COND) p = used ? p : -0.0f;      // acc + (-0.0) == acc, bit for bit
        return acc + p;
    }
}

// T taps per phase; NC components per sample (1 Float32, 2 ComplexF32); SMIN = floor(M/L) in {0, 1}
template <int T, bool FUSED, int NC, int SMIN>
__global__ __launch_bounds__(kOMaxThreads + 64)
void rational_opair_kernel(PolyArgs a, PairArgs pa)
{
    constexpr unsigned ES = 4u * NC;
    using pair_t = std::conditional_t<NC == 1, v2u_t, v4u_t>;
    constexpr int W = T + SMIN + 2;             // samples of the run a lane may touch
    constexpr int NPR = (W + 1) / 2;            // aligned pair reads per lane per step
    constexpr int NA = T + 1, NB = T + 2;       // slots of the two outputs (B's slot j' meets run sample SMIN + j')

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int ncw = (blockDim.x >> 6) - 1;      // compute waves; the last wave is the loader

    if (wave == ncw) {
        pair_loader_wave<NC>(a, pa, smem, lane);
        return;
    }
    volatile unsigned *const tile_flag = reinterpret_cast<volatile unsigned *>(smem + pa.flags_off);

    // ---- the lane's two outputs: window starts, offsets into the even-aligned run, shifted tap columns, masks ----
    // 32-bit: plan_rational_opair guarantees c*L <= 1024 and M < 2L, so every product below is < 2^22
    const int Li = static_cast<int>(a.L), Mi = static_cast<int>(a.M), u0 = static_cast<int>(a.u0);
    const bool act = 2 * tid + 1 < pa.P;                    // P = c*L is even
    const int uA = u0 + (act ? 2 * tid : 0) * Mi, uB = uA + Mi;
    const int qA = uA / Li, phA = uA - qA * Li;
    const int qB = uB / Li, phB = uB - qB * Li;
    const int R = qA & ~1;
    const int offA = qA - R;                                // 0 or 1
    const int dB = qB - R - SMIN;                           // 0, 1 or 2
    float tapA[NA], tapB[NB];
    {
        const float *__restrict__ ca = static_cast<const float *>(a.taps) + static_cast<long long>(phA) * T;
        const float *__restrict__ cb = static_cast<const float *>(a.taps) + static_cast<long long>(phB) * T;
        float colA[T], colB[T];
#pragma unroll
        for (int i = 0; i < T; ++i) { colA[i] = ca[i]; colB[i] = cb[i]; }
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const float t0 = j < T ? colA[j] : 0.f, t1 = j >= 1 ? colA[j - 1] : 0.f;
            tapA[j] = act ? (offA == 0 ? t0 : t1) : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float t0 = j < T ? colB[j] : 0.f, t1 = (j >= 1 && j - 1 < T) ? colB[j - 1] : 0.f, t2 = j >= 2 ? colB[j - 2] : 0.f;
            tapB[j] = act ? (dB == 0 ? t0 : dB == 1 ? t1 : t2) : 0.f;
        }
    }
    // which lanes have the end slots inside their window (lane masks; wave-uniform SGPR pairs in the loop)
    const bool useA0 = offA == 0, useAT = offA == 1;
    const bool useB0 = dB == 0, useB1 = dB <= 1, useBT = dB >= 1, useBT1 = dB == 2;

    const int n_out = static_cast<int>(a.n_out);
    for (int s = 0;; s = (s + 1 == pa.ns ? 0 : s + 1)) {
        // One barrier per tile and no memory wait: the loader arrives only after this tile's data has landed and its
        // descriptor is in LDS; all compute waves arriving proves the oldest stage is no longer read.
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned tg = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(tile_flag[2 * s])));
        const unsigned tj = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(tile_flag[2 * s + 1])));
        if (tj == 0u) break;                      // end marker
        const TileAt ta = pair_tile_at(pa, tg, tj);
        const int J = ta.jt;
        float *__restrict__ yc = static_cast<float *>(a.y) + (static_cast<long long>(ta.ch) * a.y_stride + static_cast<long long>(ta.st) * pa.P) * NC;
        const int remaining = n_out - ta.st * pa.P;                       // outputs of this channel from this tile on
        const bool full = remaining >= J * pa.P;                          // wave-uniform
        // lane-constant addresses are re-derived per tile from a fresh lane id (opaque to the compiler): kept live across
        // the tile loop they would cost registers the step loop never uses
        unsigned lane_t;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_t));
        const unsigned tid_t = static_cast<unsigned>(wave) * 64u + lane_t;
        const unsigned my_out = 2u * tid_t;                               // first of the lane's two outputs within a step
        const unsigned uA_t = static_cast<unsigned>(u0) + (2u * tid_t < static_cast<unsigned>(pa.P) ? 2u * tid_t : 0u) * static_cast<unsigned>(Mi);
        const unsigned R_t = (uA_t / static_cast<unsigned>(Li)) & ~1u;
        const unsigned wbase = lds_base + static_cast<unsigned>(s) * pa.stage_bytes + R_t * ES;

        auto run_steps = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            constexpr int KP = NPR < (NC == 1 ? 5 : 3) ? NPR : (NC == 1 ? 5 : 3);   // ring: 5 pairs of 8 B or 3 pairs of 16 B
            constexpr int NPRV = (NPR + KP - 1) / KP * KP;
            constexpr bool PAIRED_WAITS = KP >= 5;
            pair_t pring[KP];
            char *const ybytes = reinterpret_cast<char *>(yc);
            auto read_pair = [&](auto off_tag, unsigned addr) -> pair_t {
                constexpr int OFFP = decltype(off_tag)::value;
                if constexpr (NC == 1) return lds_read_b64<OFFP * 8>(addr);
                else return lds_read_b128<OFFP * 16>(addr);
            };
            static_for<0, KP>([&](auto I) { pring[decltype(I)::value] = read_pair(I, wbase); });
#pragma unroll 1
            for (int j = 0; j < J; ++j) {
                const int jn = j + 1 < J ? j + 1 : j;         // the last step re-reads its own window (never used)
                const unsigned wcur = wbase + static_cast<unsigned>(j) * pa.cM * ES;
                const unsigned wnext = wbase + static_cast<unsigned>(jn) * pa.cM * ES;
                float accA[NC] = {}, accB[NC] = {};
                static_for<0, NPRV>([&](auto I) {
                    constexpr int r = decltype(I)::value;
                    constexpr int slot = r % KP;
                    if constexpr (r < NPR) {
                        if constexpr (!PAIRED_WAITS) lgkm_wait<ring_younger(r, NPR, KP, 0)>(pring[slot]);
                        else if constexpr (r % 2 == 0) {
                            if constexpr (r + 1 < NPR) lgkm_wait2<ring_younger_pair(r, NPR, KP, 0)>(pring[slot], pring[(r + 1) % KP]);
                            else lgkm_wait<ring_younger(r, NPR, KP, 0)>(pring[slot]);
                        }
                        float wv[2][NC];                      // run samples 2r and 2r+1
                        if constexpr (NC == 1) { wv[0][0] = __uint_as_float(pring[slot].x); wv[1][0] = __uint_as_float(pring[slot].y); }
                        else {
                            wv[0][0] = __uint_as_float(pring[slot].x); wv[0][1] = __uint_as_float(pring[slot].y);
                            wv[1][0] = __uint_as_float(pring[slot].z); wv[1][1] = __uint_as_float(pring[slot].w);
                        }
                        static_for<0, 2>([&](auto H) {
                            constexpr int js = 2 * r + decltype(H)::value;    // run sample index == A's slot
                            constexpr int jb = js - SMIN;                      // B's slot
                            // slot j of an output is inside the window of a lane with offset d iff d <= j <= d + T - 1: the
                            // lower bound matters for the first slots (j < max offset), the upper one for the last (j > T - 1)
                            constexpr bool a_lo = js < 1, a_hi = js > T - 1;             // A: offsets 0..1
                            constexpr bool b_lo = jb < 2, b_hi = jb > T - 1;             // B: offsets 0..2 (relative to SMIN)
#pragma unroll
                            for (int cc = 0; cc < NC; ++cc) {
                                const float w = wv[decltype(H)::value][cc];
                                if constexpr (js < NA) {
                                    const bool used = (a_lo ? useA0 : true) && (a_hi ? useAT : true);
                                    accA[cc] = slot_mac<FUSED, js == 0, a_lo || a_hi>(tapA[js], w, accA[cc], used);
                                }
                                if constexpr (jb >= 0 && jb < NB) {
                                    const bool used = (b_lo ? (jb == 0 ? useB0 : useB1) : true) && (b_hi ? (jb == T ? useBT : useBT1) : true);
                                    accB[cc] = slot_mac<FUSED, jb == 0, b_lo || b_hi>(tapB[jb], w, accB[cc], used);
                                }
                            }
                        });
#pragma unroll
                        for (int cc = 0; cc < NC; ++cc) { pin(accA[cc]); pin(accB[cc]); }   // pair r is consumed before its slot is re-targeted
                    }
                    if constexpr (r + KP < NPR) pring[slot] = read_pair(std::integral_constant<int, r + KP>{}, wcur);
                    else if constexpr (r + KP >= NPRV) pring[slot] = read_pair(std::integral_constant<int, r + KP - NPRV>{}, wnext);
                });
                // the lane's two adjacent outputs: one dense 2*ES-byte store per lane, straight from the accumulators
                {
                    const unsigned kj = static_cast<unsigned>(j) * static_cast<unsigned>(pa.P);
                    char *const dst = ybytes + (kj + my_out) * ES;
                    float v[2 * NC];
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) { v[cc] = accA[cc]; v[NC + cc] = accB[cc]; }
                    if constexpr (FULL) {
                        if (my_out + 1 < static_cast<unsigned>(pa.P)) __builtin_memcpy(dst, v, 2 * ES);
                    } else {
                        const unsigned rem_j = static_cast<unsigned>(remaining) > kj ? static_cast<unsigned>(remaining) - kj : 0u;
                        const unsigned lim = rem_j < static_cast<unsigned>(pa.P) ? rem_j : static_cast<unsigned>(pa.P);
                        if (my_out + 1 < lim) __builtin_memcpy(dst, v, 2 * ES);
                        else if (my_out < lim) __builtin_memcpy(dst, v, ES);
                    }
                }
            }
            lgkm_wait<0>(pring[0]);                           // retires the last step's unused reads
            static_for<0, KP>([&](auto I) { pin(pring[decltype(I)::value]); });
        };
        if (full) run_steps(std::true_type{});
        else run_steps(std::false_type{});
    }
}

template <bool FUSED, int NC, int SMIN>
hipError_t launch_opair_T(int T, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, PairArgs pa, int num_cus)
{
#define MRHIP_CASE(TT)                                                                              \
    case TT: {                                                                                      \
        auto kfn = rational_opair_kernel<TT, FUSED, NC, SMIN>;                                       \
        if (lds > 48 * 1024) {                                                                      \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),                 \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)); \
            if (e != hipSuccess) return e;                                                          \
        }                                                                                           \
        int per_cu = 0;                                                                             \
        hipError_t eo = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, static_cast<int>(block.x), lds); \
        if (eo != hipSuccess) return eo;                                                            \
        if (per_cu < 1) per_cu = 1;                                                                 \
        {   /* six-wave workgroups: the fourth never fits next to three running ones (see kernels_rational_pair.hip) */ \
            const long long tiles = (static_cast<long long>(pa.total_steps) + pa.J - 1) / pa.J;     \
            if (block.x == 6 * 64 && per_cu > 3 && tiles > 3LL * num_cus * per_cu) per_cu = 3;      \
        }                                                                                           \
        static const int bpc = opair_env_int("MRHIP_OPAIR_BPC", 0);                                 \
        if (bpc > 0) per_cu = bpc;                                                                  \
        long long g = static_cast<long long>(num_cus) * per_cu;                                     \
        if (g > static_cast<long long>(pa.total_steps)) g = pa.total_steps;                         \
        if (g < 1) g = 1;                                                                           \
        pa.ngroups = static_cast<int>(g < kOGroups ? g : kOGroups);                                 \
        pa.steps_per_group = static_cast<unsigned>((pa.total_steps + pa.ngroups - 1) / pa.ngroups); \
        pa.static_grabs = (static_cast<long long>(pa.total_steps) + pa.J - 1) / pa.J <= 3 * g;      \
        static int dbg = opair_env_int("MRHIP_DEBUG", 0);                                           \
        if (dbg == 1) {                                                                             \
            dbg = 0;                                                                                \
            hipFuncAttributes fa;                                                                   \
            (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kfn));                   \
            std::fprintf(stderr, "[mrhip] rational_opair T=%d smin=%d grid=%lld block=%u lds=%zu occ/CU=%d regs=%d c=%d P=%d cM=%d J=%d ns=%d\n", \
                         TT, SMIN, g, block.x, lds, per_cu, fa.numRegs, pa.c, pa.P, pa.cM, pa.J, pa.ns); \
        }                                                                                           \
        launch_kernel(kfn, dim3(static_cast<unsigned>(g)), block, lds, s, a, pa);                   \
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

// Covers FIRRational with Float32 arithmetic (Float32 or ComplexF32 samples, Float32 taps), tapsPerPhi <= 32 and
// 1/2 < M/L < 2 (M != L): SMIN = floor(M/L).  Returns false otherwise (the caller tries the next kernel).
bool plan_rational_opair(const TypeKey &tk, const PolyArgs &a, int num_cus, PairArgs *out, dim3 *block, size_t *lds)
{
    if (!opair_env_int("MRHIP_OPAIR", 1)) return false;   // read per call: tests switch kernels at run time
    if (tk.x_f64 || tk.r_f64) return false;
    const int nc = tk.complex_x ? 2 : 1;
    const long long es = 4 * nc;
#ifdef MRHIP_PS_FAST_BUILD
    if (a.T != 24) return false;
#endif
    if (a.T < 1 || a.T > 32) return false;
    if (a.L < 2 || a.M < 2 || a.zero_start_below > 0) return false;
    if (!(2LL * a.M > a.L && a.M < 2LL * a.L)) return false;
    const int smin = a.M > a.L ? 1 : 0;
    const int env_c = opair_env_int("MRHIP_OPAIR_C", 0), env_j = opair_env_int("MRHIP_OPAIR_J", 0), env_ns = opair_env_int("MRHIP_OPAIR_NS", 0);
    // c: lanes = c*L/2 <= 512; c*L and c*M even (a lane owns two outputs; the run base keeps its parity from step to
    // step) => c even, L and M being coprime.  Among the sizes with 3..7 full-ish compute waves take the fullest.
    int best_c = 0;
    double best = -1.0;
    for (int pass = 0; pass < 2 && !best_c; ++pass)
        for (int c = 2; static_cast<long long>(c) * a.L / 2 <= kOMaxThreads && static_cast<long long>(c) * a.L <= 1024; c += 2) {
            const int lanes = static_cast<int>(static_cast<long long>(c) * a.L / 2);
            const int padded = (lanes + 63) / 64 * 64;
            if (pass == 0 && (padded < 192 || padded > (nc == 1 ? 320 : 256))) continue;
            const double score = static_cast<double>(lanes) / padded * (padded < 192 ? 0.5 + 0.5 * padded / 192.0 : 1.0);
            if (score > best + 1e-9) { best = score; best_c = c; }
        }
    if (env_c > 0 && env_c % 2 == 0 && static_cast<long long>(env_c) * a.L / 2 <= kOMaxThreads && static_cast<long long>(env_c) * a.L <= 1024) best_c = env_c;
    if (!best_c) return false;
    const int c = best_c;
    const long long cM = static_cast<long long>(c) * a.M, cL = static_cast<long long>(c) * a.L;
    const int lanes = static_cast<int>(cL / 2);
    const int padded = (lanes + 63) / 64 * 64;
    const int nwaves = padded / 64;
    const int tail = a.T + smin + 4;                    // run overhang beyond the period: offsets <= smin + 2, + even rounding
    const int wg_per_cu = (nc == 1 && nwaves + 1 != 6) ? 4 : 3;
    int ns = env_ns >= 2 && env_ns <= 8 ? env_ns : 3;
    auto j_for = [&](int stages) -> long long {
        const long long budget_kib = ((wg_per_cu == 3 ? 150 : 160) * 1024 / wg_per_cu - 64) / stages / 1024;
        const long long j = ((budget_kib > 1 ? budget_kib : 1) * 1024 / es - tail) / cM;
        return j < 1 ? 1 : j;
    };
    long long J = j_for(ns);
    if (env_ns <= 0 && env_j <= 0) {        // long launches: two stages of larger tiles (see plan_rational_pair)
        const long long j2 = std::min<long long>(j_for(2), 64);
        const long long nslots2 = (((j2 * cM + tail + 3) / 4 * 4) * es / 16 + 63) / 64;
        const long long tiles2 = ((a.n_out + j2 * cL - 1) / (j2 * cL)) * a.nch;
        if (j2 > J && nslots2 <= 60 && tiles2 >= 48LL * num_cus * wg_per_cu) { ns = 2; J = j2; }
    }
    if (env_j > 0) J = env_j;
    if (J > 64) J = 64;
    if (env_j <= 0) {                       // small problems: enough tiles to give every CU a few workgroups
        const long long want_tiles = 4LL * num_cus;
        while (J > 2 && ((a.n_out + J * cL - 1) / (J * cL)) * a.nch < want_tiles) J = (J + 1) / 2;
    }
    long long tile_len = (J * cM + tail + 3) / 4 * 4;
    const long long nslots = (tile_len * es / 16 + 63) / 64;
    const size_t stage_bytes = static_cast<size_t>(nslots) * 1024;
    if (nslots > 60 / (ns > 2 ? ns - 2 : 1) || ns * stage_bytes > 156 * 1024) return false;
    PairArgs pa{};
    pa.c = c; pa.P = static_cast<int>(cL); pa.cM = static_cast<int>(cM);
    pa.J = static_cast<int>(J);
    pa.tile_len = static_cast<int>(tile_len);
    pa.tail = tail;
    pa.dma_rounds = static_cast<int>(nslots);
    pa.stage_bytes = static_cast<int>(stage_bytes);
    pa.ns = ns;
    pa.nc = nc;
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
    *out = pa;
    *block = dim3(static_cast<unsigned>(padded + 64));   // + the loader wave
    *lds = ns * stage_bytes + 8 * ns;
    return true;
}

hipError_t launch_rational_opair(bool fused, const PolyArgs &a, const PairArgs &pa_in, dim3 block, size_t lds, hipStream_t s,
                                 const char **kname, int num_cus, unsigned *counters)
{
    if (!counters) return hipErrorInvalidValue;
    PairArgs pa = pa_in;
    pa.counters = counters;
    *kname = "rational_opair_kernel";
    const bool up = a.L > a.M;
#define MRHIP_GO(FU, NCV) (up ? launch_opair_T<FU, NCV, 0>(a.T, block, lds, s, a, pa, num_cus) : launch_opair_T<FU, NCV, 1>(a.T, block, lds, s, a, pa, num_cus))
    if (pa.nc == 2) return fused ? MRHIP_GO(true, 2) : MRHIP_GO(false, 2);
    return fused ? MRHIP_GO(true, 1) : MRHIP_GO(false, 1);
#undef MRHIP_GO
}

}  // namespace mrhip
