// kernels_phase_stationary.hip -- tuned gfx950 kernel for the polyphase (pfb) kernels:
// FIRRational, FIRInterpolator (and short FIRStandard / FIRDecimator), tapsPerPhi <= 32.
//
// Idea ("phase-stationary"): the phase of output k is (u0 + k*M) mod L, so outputs k and k + c*L
// use the same tap column.  A workgroup has P = c*L active lanes; lane t owns the outputs
// k = tile*P*J + j*P + t, j = 0..J-1, of every tile it is handed, and therefore ONE tap column for
// its whole life: the T taps sit in VGPRs, loaded once per workgroup (persistent grid).  The
// consecutive-lane -> consecutive-output map keeps the y stores fully coalesced and makes the LDS
// window reads of a wave walk forward by M/L samples per lane.
//
// Per tile the workgroup stages J*c*M + T (+ alignment slack) input samples of one channel into
// LDS with 16-byte global loads (the seam with the previous call's history is resolved here, once,
// instead of per output).  4-byte samples are stored twice, the second copy shifted by one sample
// and placed 32 banks away, so that every lane can fetch its T-sample window with 8-byte-aligned
// ds_read_b64 (256 B/clk/CU instead of 128 for ds_read_b32) whatever the parity of its window start.
// LDS traffic is T*sizeof(sample) per output (taps cost nothing), HBM traffic is the algorithmic
// sizeof(Tx) + (L/M)*sizeof(Tb) per input sample plus the (T-1)-sample halo per tile.
//
// Arithmetic: identical to the generic kernel (STRICT: separately rounded multiply and add, oldest
// sample first, first product initialises; FUSED: explicit fma), so results are bit-identical
// between the two kernels and the oracle.
#include <algorithm>
#include <type_traits>

#include "mrhip_internal.h"

#pragma clang fp contract(off)

namespace mrhip {
namespace {

constexpr int kMaxThreads = 512;

template <typename R, bool FUSED>
__device__ __forceinline__ R mac(R t, R x, R acc)
{
    if constexpr (FUSED) {
        if constexpr (sizeof(R) == 4) return __builtin_fmaf(t, x, acc);
        else return __builtin_fma(t, x, acc);
    } else {
        R p = t * x;
        return acc + p;
    }
}

typedef unsigned v2u_t __attribute__((ext_vector_type(2)));

// compile-time loop (inline-asm immediates need template constants)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ds_read_b64 / ds_read_b32 as inline asm: hipcc fuses neighbouring 8-byte LDS loads into
// ds_read2_b64, which moves half the bytes per LDS cycle of ds_read_b64 (128 vs 256 B/clk/CU,
// MI355X_MICROARCH.md LDS table).  The compiler does not see loads inside asm, so the matching
// s_waitcnt is issued by hand through lgkm_wait below.
template <int OFF>
__device__ __forceinline__ v2u_t lds_read_b64(unsigned byte_addr)
{
    v2u_t v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ unsigned lds_read_b32(unsigned byte_addr)
{
    unsigned v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return v;
}
// Wait until at most N LDS operations of this wave are outstanding.  LDS operations return in
// order, so after issuing reads r_0..r_{n-1} back to back, lgkm_wait<n-1-i>(r_i) guarantees r_i has
// landed (extra outstanding operations only make the wait more conservative).  The "+v" ties the
// register to the wait so the compiler cannot move a use of it above the wait.
template <int N, typename V>
__device__ __forceinline__ void lgkm_wait(V &reg)
{
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(reg) : "n"(N < 15 ? N : 15));
}

// One 16-byte chunk = VPS samples of SW 32-bit words each.
template <int SW>
__device__ __forceinline__ uint4 load_chunk_checked(const unsigned *__restrict__ xc, const unsigned *__restrict__ hc,
                                                   int H, long long g, long long x_len)
{
    constexpr int VPS = 4 / SW;
    unsigned w[4];
#pragma unroll
    for (int e = 0; e < VPS; ++e) {
        const long long gi = g + e;
        const unsigned *src = nullptr;
        if (gi >= 0) { if (gi < x_len) src = xc + gi * SW; }
        else if (gi >= -static_cast<long long>(H)) src = hc + (static_cast<long long>(H) + gi) * SW;
#pragma unroll
        for (int s = 0; s < SW; ++s) w[e * SW + s] = src ? src[s] : 0u;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

template <int T, typename TX, typename R, int NC, bool FUSED>
__global__ __launch_bounds__(kMaxThreads) void poly_phase_stationary_kernel(PolyArgs a, TileArgs ta)
{
    constexpr int SW = static_cast<int>(sizeof(TX)) * NC / 4;   // 32-bit words per sample: 1, 2 or 4
    constexpr int VPS = 4 / SW;                                  // samples per 16-byte chunk
    constexpr bool TWO_COPIES = SW == 1;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *const ldsA = reinterpret_cast<unsigned *>(smem);
    unsigned *const ldsB = reinterpret_cast<unsigned *>(smem + ta.copyB_offset_bytes);
    // LDS byte offset of smem[0] (low 32 bits of the flat address of an LDS object are its LDS offset)
    const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));

    const int t = threadIdx.x;
    const bool active = t < ta.P;

    // per-lane phase and input offset: u = u0 + t*M ; phi = u mod L ; q = u div L
    const long long u = a.u0 + static_cast<long long>(t) * a.M;
    const long long qll = u / a.L;
    const int phi = static_cast<int>(u - qll * a.L);
    const int q = static_cast<int>(qll);

    R taps[T];
    {
        const R *__restrict__ tp = static_cast<const R *>(a.taps) + static_cast<long long>(active ? phi : 0) * T;
#pragma unroll
        for (int i = 0; i < T; ++i) taps[i] = tp[i];
    }

    const int cM = ta.c * a.M;                 // input samples between a lane's consecutive outputs
    const long long tile_in = static_cast<long long>(ta.J) * cM;
    const long long tile_out = static_cast<long long>(ta.J) * ta.P;
    const int nchunks = ta.tile_len / VPS;

    for (long long tile = blockIdx.x; tile < ta.total_tiles; tile += gridDim.x) {
        const int ch = static_cast<int>(tile / ta.tiles_per_channel);
        const long long tau = tile - static_cast<long long>(ch) * ta.tiles_per_channel;
        const unsigned *__restrict__ xc = static_cast<const unsigned *>(a.x) + static_cast<long long>(ch) * a.x_stride * SW;
        const unsigned *__restrict__ hc = static_cast<const unsigned *>(a.hist) + static_cast<long long>(ch) * a.H * SW;
        R *__restrict__ yc = static_cast<R *>(a.y) + static_cast<long long>(ch) * a.y_stride * NC;

        // 0-based x index of the oldest sample any lane of this tile needs, aligned down to a chunk
        const long long o = a.d0 - T + tau * tile_in;
        const long long g0 = o & ~static_cast<long long>(VPS - 1);
        const int sh = static_cast<int>(o - g0);

        __syncthreads();   // previous tile's window reads are done
        for (int ci = t; ci < nchunks; ci += blockDim.x) {
            const long long g = g0 + static_cast<long long>(ci) * VPS;
            uint4 v;
            if (ta.x_aligned16 && g >= 0 && g + VPS <= a.x_len)
                v = *reinterpret_cast<const uint4 *>(xc + g * SW);
            else
                v = load_chunk_checked<SW>(xc, hc, a.H, g, a.x_len);
            *reinterpret_cast<uint4 *>(ldsA + ci * 4) = v;
            if constexpr (TWO_COPIES) {
                // B[i] = A[i+1]  =>  B[4ci-1 .. 4ci+2] = v.x .. v.w
                if (ci > 0) ldsB[ci * 4 - 1] = v.x;
                *reinterpret_cast<uint2 *>(ldsB + ci * 4) = make_uint2(v.y, v.z);
                ldsB[ci * 4 + 2] = v.w;
            }
        }
        __syncthreads();

        if (active) {
            const long long kbase = tau * tile_out + t;
            const long long nbase = a.d0 + q + tau * tile_in;     // 1-based newest-sample index, j = 0
#pragma unroll 1
            for (int j = 0; j < ta.J; ++j) {
                const long long k = kbase + static_cast<long long>(j) * ta.P;
                if (k >= a.n_out) break;
                const int start = sh + q + j * cM;                // LDS sample index of the oldest sample
                R acc[NC];
                const bool zero_start = nbase + static_cast<long long>(j) * cM < a.zero_start_below;   // support.jl:46
                auto step = [&](auto idx, const R (&xv)[NC]) {
                    constexpr int i = decltype(idx)::value;
                    if constexpr (i == 0) {
#pragma unroll
                        for (int c2 = 0; c2 < NC; ++c2) acc[c2] = taps[0] * xv[c2];
                        if (zero_start) {
#pragma unroll
                            for (int c2 = 0; c2 < NC; ++c2) acc[c2] = static_cast<R>(0) + acc[c2];
                        }
                    } else {
#pragma unroll
                        for (int c2 = 0; c2 < NC; ++c2) acc[c2] = mac<R, FUSED>(taps[i], xv[c2], acc[c2]);
                    }
                };
                if constexpr (TWO_COPIES) {
                    // 4-byte samples: T/2 aligned 8-byte reads from whichever copy makes the window start even
                    constexpr int NP = T / 2;
                    constexpr int NR = NP + (T & 1);
                    const unsigned waddr = lds_base + ((start & 1) ? ta.copyB_offset_bytes + (start - 1) * 4 : start * 4);
                    v2u_t pr[NP > 0 ? NP : 1];
                    unsigned last = 0;
                    static_for<0, NP>([&](auto I) { pr[decltype(I)::value] = lds_read_b64<decltype(I)::value * 8>(waddr); });
                    if constexpr (T & 1) last = lds_read_b32<(T - 1) * 4>(waddr);
                    static_for<0, NP>([&](auto I) {
                        constexpr int i = decltype(I)::value;
                        lgkm_wait<NR - 1 - i>(pr[i]);
                        const R x0[1] = {static_cast<R>(__uint_as_float(pr[i].x))};
                        const R x1[1] = {static_cast<R>(__uint_as_float(pr[i].y))};
                        step(std::integral_constant<int, 2 * i>{}, x0);
                        step(std::integral_constant<int, 2 * i + 1>{}, x1);
                    });
                    if constexpr (T & 1) {
                        lgkm_wait<0>(last);
                        const R xl[1] = {static_cast<R>(__uint_as_float(last))};
                        step(std::integral_constant<int, T - 1>{}, xl);
                    }
                } else if constexpr (SW == 2) {
                    // 8-byte samples (ComplexF32 or Float64): one naturally aligned ds_read_b64 per tap
                    const unsigned waddr = lds_base + start * 8;
                    v2u_t pr[T];
                    static_for<0, T>([&](auto I) { pr[decltype(I)::value] = lds_read_b64<decltype(I)::value * 8>(waddr); });
                    static_for<0, T>([&](auto I) {
                        constexpr int i = decltype(I)::value;
                        lgkm_wait<T - 1 - i>(pr[i]);
                        TX tmp[NC];
                        __builtin_memcpy(tmp, &pr[i], 8);
                        R xv[NC];
#pragma unroll
                        for (int c2 = 0; c2 < NC; ++c2) xv[c2] = static_cast<R>(tmp[c2]);
                        step(I, xv);
                    });
                } else {
                    // 16-byte samples (ComplexF64): ds_read_b128, left to the compiler
                    const uint4 *wp = reinterpret_cast<const uint4 *>(ldsA + start * 4);
                    static_for<0, T>([&](auto I) {
                        const uint4 v = wp[decltype(I)::value];
                        TX tmp[NC];
                        __builtin_memcpy(tmp, &v, 16);
                        R xv[NC];
#pragma unroll
                        for (int c2 = 0; c2 < NC; ++c2) xv[c2] = static_cast<R>(tmp[c2]);
                        step(I, xv);
                    });
                }
                if constexpr (NC == 1) {
                    yc[k] = acc[0];
                } else {
                    using RV = typename std::conditional<sizeof(R) == 4, float2, double2>::type;
                    RV o2;
                    o2.x = acc[0];
                    o2.y = acc[1];
                    reinterpret_cast<RV *>(yc)[k] = o2;
                }
            }
        }
    }
}

template <typename TX, typename R, int NC, bool FUSED>
hipError_t launch_T(int T, dim3 grid, dim3 block, size_t lds, hipStream_t s, const PolyArgs &a, const TileArgs &ta)
{
#define MRHIP_CASE(TT)                                                                              \
    case TT: {                                                                                      \
        auto kfn = poly_phase_stationary_kernel<TT, TX, R, NC, FUSED>;                              \
        if (lds > 48 * 1024) {                                                                      \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),                 \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)); \
            if (e != hipSuccess) return e;                                                          \
        }                                                                                           \
        hipLaunchKernelGGL(kfn, grid, block, lds, s, a, ta);                                        \
        return hipGetLastError();                                                                   \
    }
    switch (T) {
        MRHIP_CASE(1) MRHIP_CASE(2) MRHIP_CASE(3) MRHIP_CASE(4) MRHIP_CASE(5) MRHIP_CASE(6) MRHIP_CASE(7) MRHIP_CASE(8)
        MRHIP_CASE(9) MRHIP_CASE(10) MRHIP_CASE(11) MRHIP_CASE(12) MRHIP_CASE(13) MRHIP_CASE(14) MRHIP_CASE(15) MRHIP_CASE(16)
        MRHIP_CASE(17) MRHIP_CASE(18) MRHIP_CASE(19) MRHIP_CASE(20) MRHIP_CASE(21) MRHIP_CASE(22) MRHIP_CASE(23) MRHIP_CASE(24)
        MRHIP_CASE(25) MRHIP_CASE(26) MRHIP_CASE(27) MRHIP_CASE(28) MRHIP_CASE(29) MRHIP_CASE(30) MRHIP_CASE(31) MRHIP_CASE(32)
    default: return hipErrorInvalidValue;
    }
#undef MRHIP_CASE
}

}  // namespace

// Host-side planning: pick c (P = c*L lanes), J (outputs per lane per tile) and the grid.
// Returns false if the configuration is outside what this kernel covers (caller falls back).
bool plan_phase_stationary(const TypeKey &tk, const PolyArgs &a, int num_cus, TileArgs *out, dim3 *grid, dim3 *block, size_t *lds)
{
    if (a.T < 1 || a.T > 32) return false;
    if (a.L > kMaxThreads) return false;
    const int sw = (tk.x_f64 ? 2 : 1) * (tk.complex_x ? 2 : 1);
    const int vps = 4 / sw;
    // choose c: P = c*L as close as possible to a multiple of 64 within [<=1024], prefer ~256-512 lanes
    int best_c = 0;
    double best_score = -1.0;
    for (int c = 1; static_cast<long long>(c) * a.L <= kMaxThreads; ++c) {
        const int P = c * a.L;
        const int padded = (P + 63) / 64 * 64;
        double score = static_cast<double>(P) / padded;                  // lane utilisation
        if (P < 192) score *= 0.5 + 0.5 * P / 192.0;                      // too few waves per block
        if (P > 512) score *= 0.97;                                       // mild preference for <= 512
        if (score > best_score + 1e-9) { best_score = score; best_c = c; }
    }
    if (!best_c) return false;
    const int c = best_c, P = c * a.L;
    const long long cM = static_cast<long long>(c) * a.M;
    // J: keep the LDS tile around 24 KiB per copy-set so ~4 workgroups fit a CU
    const long long budget_samples = 24 * 1024 / (sw * 4);
    long long J = (budget_samples - a.T - 8) / (cM > 0 ? cM : 1);
    if (J > 16) J = 16;
    if (J < 1) J = 1;
    long long tile_len = J * cM + a.T + 1 + (vps - 1);
    tile_len = (tile_len + 3) / 4 * 4;                                    // whole chunks (and 16-byte multiple)
    if (tile_len % vps) tile_len += vps - tile_len % vps;
    const size_t bytesA = (static_cast<size_t>(tile_len) * sw * 4 + 255) / 256 * 256;
    const size_t total = sw == 1 ? bytesA + 128 + bytesA : bytesA;        // copy B sits 32 banks (128 B) off copy A
    if (total > 150 * 1024) return false;
    TileArgs ta{};
    ta.c = c; ta.P = P; ta.J = static_cast<int>(J);
    ta.tile_len = static_cast<int>(tile_len);
    ta.copyB_offset_bytes = static_cast<int>(bytesA + 128);
    const long long tile_out = J * P;
    ta.tiles_per_channel = (a.n_out + tile_out - 1) / tile_out;
    ta.total_tiles = ta.tiles_per_channel * a.nch;
    const uintptr_t xb = reinterpret_cast<uintptr_t>(a.x);
    ta.x_aligned16 = (xb % 16 == 0) && ((a.x_stride * sw * 4) % 16 == 0 || a.nch == 1);
    const int padded = (P + 63) / 64 * 64;
    // persistent grid: enough workgroups to fill every CU at the occupancy LDS/waves allow
    int per_cu = static_cast<int>(std::min<size_t>(160 * 1024 / std::max<size_t>(total, 1), static_cast<size_t>(2048 / padded)));
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    long long g = static_cast<long long>(num_cus) * per_cu;
    if (g > ta.total_tiles) g = ta.total_tiles;
    if (g < 1) g = 1;
    *out = ta;
    *grid = dim3(static_cast<unsigned>(g));
    *block = dim3(static_cast<unsigned>(padded));
    *lds = total;
    return true;
}

hipError_t launch_poly_phase_stationary(const TypeKey &tk, bool fused, const PolyArgs &a, const TileArgs &ta, dim3 grid,
                                        dim3 block, size_t lds, hipStream_t s, const char **kname)
{
    *kname = "poly_phase_stationary_kernel";
#define MRHIP_GO(TX, R, NC)                                                                          \
    return fused ? launch_T<TX, R, NC, true>(a.T, grid, block, lds, s, a, ta)                        \
                 : launch_T<TX, R, NC, false>(a.T, grid, block, lds, s, a, ta)
    if (!tk.x_f64 && !tk.r_f64) { if (tk.complex_x) { MRHIP_GO(float, float, 2); } else { MRHIP_GO(float, float, 1); } }
    if (!tk.x_f64 && tk.r_f64) { if (tk.complex_x) { MRHIP_GO(float, double, 2); } else { MRHIP_GO(float, double, 1); } }
    if (tk.x_f64 && tk.r_f64) { if (tk.complex_x) { MRHIP_GO(double, double, 2); } else { MRHIP_GO(double, double, 1); } }
#undef MRHIP_GO
    return hipErrorInvalidValue;
}

}  // namespace mrhip
