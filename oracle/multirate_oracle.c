/*
 * multirate_oracle.c -- CPU restatement of Multirate.jl's filt!/filt hot path.
 *
 * ==========================================================================
 *  TEST INFRASTRUCTURE.  NOT PRODUCT CODE.
 *  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 *  load this library.  The shipped path (multirate.jl_amd/csrc) never links,
 *  calls or falls back to anything in oracle/.
 * ==========================================================================
 *
 * What it restates (reference paths relative to /root/reference):
 *   src/support.jl:5-55     the four unsafedot methods (oracle_typed.inc)
 *   src/support.jl:61-80    shiftin!
 *   src/Filters.jl:15-117   kernel constructors (flipud, initial state)
 *   src/Filters.jl:151-198  FIRFilter constructors (kernel selection, historyLen)
 *   src/Filters.jl:284-298  taps2pfb
 *   src/Filters.jl:352-385  outputlength
 *   src/Filters.jl:396-422  inputlength
 *   src/Filters.jl:433-439  nextphase
 *   src/Filters.jl:450-752  the five filt!/filt state machines + update()
 *   src/Filters.jl:123-147, 764-846  FIRFarrow: constructor, tapsforphase!, update, filt!/filt
 *       (the polynomial fit itself, pfb2pnfb/polyfit :311-321 + support.jl:85-88, is restated in
 *        oracle/oracle.py with numpy's least squares and handed to mro_create_farrow)
 *
 * Pinning status.  The reference is Julia-0.3 source; no Julia exists in the
 * build image and no modern Julia parses it, so it cannot be executed here.
 * It ships no seeded golden vectors (its tests are unseeded random vs a naive
 * model with loose isapprox).  The oracle is therefore pinned against
 *   (1) the three deterministic known answers the reference does hold:
 *       taps2pfb([1:9],4) (Filters.jl:276-282), the README 3//17 streaming
 *       example incl. chunk boundaries (README.md:58-141), and the nextphase
 *       table of test/runtests.jl:423-438;
 *   (2) the reference's own test recipe: zero-stuff -> FIR -> keep every M-th
 *       (test/runtests.jl:60,123,190,270) and NaiveResamplers.naivefilt
 *       (src/NaiveResamplers.jl:24-49), restated independently in
 *       oracle/naive.py, to the reference's isapprox tolerance and far tighter.
 * Bit-level summation order cannot be pinned to the reference binary: its
 * @simd loops (support.jl:9,23,26,37,47,50) license compiler-dependent
 * reassociation.  The order implemented here is the order the source states
 * (sequential, oldest sample first, first product initialises the accumulator,
 * separately rounded multiply and add, accumulator type promote_type(Th,Tx)).
 * => bit-level parity vs. the reference executable: UNPINNED; algorithmic
 *    parity: pinned by (1) and (2).
 *
 * Documented deviations from reference behaviour (all are reference crashes):
 *   - hLen == 1 single-rate/decimator: reference throws BoundsError reading
 *     b[1] of an empty history (support.jl:46); the oracle computes the result.
 *   - FIRArbitrary: reference writes past `buffer` if outputlength()
 *     under-estimates (no check, Filters.jl:693-742); the oracle returns -1.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no -ffast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "multirate_oracle.h"

struct mro_filter {
    int kind;          /* mro_kind */
    int th, tx;        /* mro_dtype of taps / samples */
    int nc;            /* components per sample: 1 real, 2 complex */
    long hLen;         /* number of caller taps */
    long L, M;         /* interpolation / decimation (reduced ratio) */
    long Nphi;         /* number of polyphase branches */
    long tapsPerPhi;
    long historyLen;
    void *taps;        /* flipped h (Standard/Decimator) or pfb, column-major tapsPerPhi x Nphi */
    void *dtaps;       /* dpfb (Arbitrary) */
    double *pnfb;      /* Farrow: tapsPerPhi x (polyorder+1) coefficients, ascending powers */
    long polyorder;
    void *currentTaps; /* Farrow: tapsPerPhi taps of Th, Filters.jl:128 */
    void *history;     /* historyLen samples of Tx */
    /* streaming state */
    long phiIdx;       /* 1-based */
    long inputDeficit; /* 1-based */
    double rate, phiAccumulator, alpha, delta;
    long xIdx;
};

/* 0 = STRICT (default, the reference's arithmetic), 1 = FUSED: checker for the library's opt-in
 * MRHIP_NUMERICS_FUSED mode (same order, one fma per tap).  Process-wide, test use only. */
static int g_mro_fused = 0;
void mro_set_fused(int fused) { g_mro_fused = fused ? 1 : 0; }

/* mod(x, y) of update() (src/Filters.jl:668, :786) for x >= 0, y > 0.  Form 0 (default): the exact floating-point remainder,
 * what Julia >= 0.4 computes for these operands (its mod() is rem() plus a sign fix-up).  Form 1: rem(y + rem(x, y), y), the
 * expression older Base versions used for floats -- the reference is Julia-0.3 code and nothing in this tree says which one
 * its Base had.  The two differ only when y + rem(x, y) is not representable, i.e. by one rounding of that sum
 * (tests/test_oracle.py::test_mod_form_of_julia_0_3_quantified states how often and by how much). */
static int g_mro_mod_form = 0;
void mro_set_mod_form(int form) { g_mro_mod_form = form ? 1 : 0; }
static double mro_mod_pos(double x, double y)
{
    const double r = fmod(x, y);
    if (!g_mro_mod_form) return r;
    return fmod(y + r, y);
}

static size_t dtype_scalar_size(int dt) { return (dt == MRO_F32 || dt == MRO_C64) ? 4 : 8; }
static int dtype_is_complex(int dt) { return dt == MRO_C64 || dt == MRO_C128; }
static int dtype_is_f64(int dt) { return dt == MRO_F64 || dt == MRO_C128; }

/* promote_type(Th, Tx), Filters.jl:476,522,581,636,746 */
int mro_output_dtype(int th, int tx)
{
    int f64 = dtype_is_f64(th) || dtype_is_f64(tx);
    if (dtype_is_complex(tx)) return f64 ? MRO_C128 : MRO_C64;
    return f64 ? MRO_F64 : MRO_F32;
}

/* reference: src/support.jl:61-80  shiftin!(a, b): a <- last len(a) of [a; b] */
void mro_shiftin(void *a, long aLen, const void *b, long bLen, size_t elsize)
{
    char *pa = (char *)a;
    const char *pb = (const char *)b;
    if (bLen >= aLen) {
        memcpy(pa, pb + (size_t)(bLen - aLen) * elsize, (size_t)aLen * elsize);
    } else {
        memmove(pa, pa + (size_t)bLen * elsize, (size_t)(aLen - bLen) * elsize);
        memcpy(pa + (size_t)(aLen - bLen) * elsize, pb, (size_t)bLen * elsize);
    }
}

/* reference: src/Filters.jl:433-439  nextphase(currentphase, ratio) */
long mro_nextphase(long currentphase, long interpolation, long decimation)
{
    long phiStep = decimation % interpolation;
    long phiNext = currentphase + phiStep;
    return phiNext > interpolation ? phiNext - interpolation : phiNext;
}

/* reference: src/Filters.jl:352-357  outputlength(inputlength, ratio, initialphi)
 * Float64 division then iceil, as written. */
long mro_outputlength_ratio(long inputlength, long interpolation, long decimation, long initialPhi)
{
    double outLen = (double)((inputlength * interpolation) - initialPhi + 1) / (double)decimation;
    return (long)ceil(outLen);
}

/* reference: src/Filters.jl:396-401  inputlength(outputlength, ratio, initialphi) */
long mro_inputlength_ratio(long outputlength, long interpolation, long decimation, long initialPhi)
{
    double inLen = (double)(outputlength * decimation + initialPhi - 1) / (double)interpolation;
    return (long)ceil(inLen);
}

/* reference: src/Filters.jl:284-298  taps2pfb(h, Nphi)
 * out: column-major tapsPerPhi x Nphi, rows flipped, zero padded. Returns tapsPerPhi. */
long mro_taps2pfb(const void *h, long hLen, int th, long Nphi, void *out)
{
    size_t es = dtype_scalar_size(th);
    long tapsPerPhi = (hLen + Nphi - 1) / Nphi; /* iceil(hLen/Nphi) */
    if (!out) return tapsPerPhi;
    long hIdx = 1;
    for (long rowIdx = tapsPerPhi; rowIdx >= 1; --rowIdx) {
        for (long colIdx = 1; colIdx <= Nphi; ++colIdx) {
            char *dst = (char *)out + ((size_t)(colIdx - 1) * tapsPerPhi + (rowIdx - 1)) * es;
            if (hIdx > hLen) memset(dst, 0, es);
            else memcpy(dst, (const char *)h + (size_t)(hIdx - 1) * es, es);
            hIdx += 1;
        }
    }
    return tapsPerPhi;
}

static long gcd_l(long a, long b) { while (b) { long t = a % b; a = b; b = t; } return a < 0 ? -a : a; }

static void *flipud_copy(const void *h, long hLen, size_t es)
{
    char *o = (char *)malloc((size_t)(hLen > 0 ? hLen : 1) * es);
    for (long i = 0; i < hLen; ++i) memcpy(o + (size_t)i * es, (const char *)h + (size_t)(hLen - 1 - i) * es, es);
    return o;
}

/* reference: src/Filters.jl:158-180  FIRFilter(h, resampleRatio::Rational) and the kernel
 * constructors :20-24, :35-41, :52-58, :72-80.  Julia Rationals are always reduced. */
mro_filter *mro_create_rational(const void *h, long hLen, int th, long num, long den, int tx)
{
    if (hLen < 1 || num < 1 || den < 1) return NULL;
    if (dtype_is_complex(th)) return NULL; /* reference tests never use complex taps */
    long g = gcd_l(num, den);
    long interpolation = num / g, decimation = den / g;
    size_t es = dtype_scalar_size(th);
    mro_filter *f = (mro_filter *)calloc(1, sizeof *f);
    f->th = th; f->tx = tx; f->nc = dtype_is_complex(tx) ? 2 : 1;
    f->hLen = hLen; f->L = interpolation; f->M = decimation;
    f->phiIdx = 1; f->inputDeficit = 1;
    f->phiAccumulator = 1.0; f->alpha = 0.0; f->xIdx = 1;

    if (interpolation == 1 && decimation == 1) {          /* single-rate, :163-165 */
        f->kind = MRO_STANDARD;
        f->taps = flipud_copy(h, hLen, es);
        f->Nphi = 1; f->tapsPerPhi = hLen;
        f->historyLen = hLen - 1;
    } else if (interpolation == 1) {                      /* decimate, :166-168 */
        f->kind = MRO_DECIMATOR;
        f->taps = flipud_copy(h, hLen, es);
        f->Nphi = 1; f->tapsPerPhi = hLen;
        f->historyLen = hLen - 1;
    } else {                                              /* interpolate :169-171 / rational :172-174 */
        f->kind = decimation == 1 ? MRO_INTERPOLATOR : MRO_RATIONAL;
        f->Nphi = interpolation;
        f->tapsPerPhi = mro_taps2pfb(h, hLen, th, interpolation, NULL);
        f->taps = malloc((size_t)f->tapsPerPhi * f->Nphi * es);
        mro_taps2pfb(h, hLen, th, interpolation, f->taps);
        f->historyLen = f->tapsPerPhi - 1;
    }
    f->history = calloc((size_t)(f->historyLen > 0 ? f->historyLen : 1) * f->nc, dtype_scalar_size(tx));
    return f;
}

/* reference: src/Filters.jl:183-189 FIRFilter(h, rate::FloatingPoint, Nphi) and :105-117 FIRArbitrary */
mro_filter *mro_create_arbitrary(const void *h, long hLen, int th, double rate, long Nphi, int tx)
{
    if (!(rate > 0.0)) return NULL;                       /* "rate must be greater than 0" */
    if (hLen < 1 || Nphi < 1 || dtype_is_complex(th)) return NULL;
    size_t es = dtype_scalar_size(th);
    mro_filter *f = (mro_filter *)calloc(1, sizeof *f);
    f->kind = MRO_ARBITRARY;
    f->th = th; f->tx = tx; f->nc = dtype_is_complex(tx) ? 2 : 1;
    f->hLen = hLen; f->L = Nphi; f->M = 1; f->Nphi = Nphi; f->rate = rate;

    /* dh = [diff(h), 0]   (Filters.jl:106) -- computed in the tap type */
    void *dh = malloc((size_t)hLen * es);
    if (th == MRO_F32) {
        const float *hf = (const float *)h; float *d = (float *)dh;
        for (long i = 0; i + 1 < hLen; ++i) d[i] = hf[i + 1] - hf[i];
        d[hLen - 1] = 0.0f;
    } else {
        const double *hd = (const double *)h; double *d = (double *)dh;
        for (long i = 0; i + 1 < hLen; ++i) d[i] = hd[i + 1] - hd[i];
        d[hLen - 1] = 0.0;
    }
    f->tapsPerPhi = mro_taps2pfb(h, hLen, th, Nphi, NULL);
    f->taps = malloc((size_t)f->tapsPerPhi * Nphi * es);
    f->dtaps = malloc((size_t)f->tapsPerPhi * Nphi * es);
    mro_taps2pfb(h, hLen, th, Nphi, f->taps);
    mro_taps2pfb(dh, hLen, th, Nphi, f->dtaps);
    free(dh);

    f->phiAccumulator = 1.0; f->phiIdx = 1; f->alpha = 0.0;
    f->delta = (double)Nphi / rate;                       /* Δ = Nphi/rate, :113 */
    f->inputDeficit = 1; f->xIdx = 1;
    f->historyLen = f->tapsPerPhi - 1;
    f->history = calloc((size_t)(f->historyLen > 0 ? f->historyLen : 1) * f->nc, dtype_scalar_size(tx));
    return f;
}

/* Polynomials.jl polyval(p::Poly{T}, x::Number): R = promote_type(T, typeof(x)) = Float64 here;
 * y = p[end]; for i = end-1:-1:0  y = p[i] + x*y  (separately rounded multiply and add). */
double mro_polyval(const double *c, long polyorder, double x)
{
    double y = c[polyorder];
    for (long i = polyorder - 1; i >= 0; --i) {
        double t = x * y;
        y = c[i] + t;
    }
    return y;
}

/* currentTaps[tapIdx] = polyval(pnfb[tapIdx], phiIdx), stored into Vector{Th} (Filters.jl:144, :790-792) */
static void farrow_taps(mro_filter *f, double phase)
{
    for (long i = 0; i < f->tapsPerPhi; ++i) {
        double v = mro_polyval(f->pnfb + i * (f->polyorder + 1), f->polyorder, phase);
        if (f->th == MRO_F32) ((float *)f->currentTaps)[i] = (float)v;
        else ((double *)f->currentTaps)[i] = v;
    }
}

/* reference: src/Filters.jl:192-198 FIRFilter(h, rate, Nphi, polyorder) and :138-147 FIRFarrow(...) */
mro_filter *mro_create_farrow(long hLen, int th, double rate, long Nphi, long polyorder, const double *pnfb, int tx)
{
    if (!(rate > 0.0)) return NULL;                       /* "rate must be greater than 0", :193 */
    if (hLen < 1 || Nphi < 1 || polyorder < 0 || dtype_is_complex(th) || !pnfb) return NULL;
    mro_filter *f = (mro_filter *)calloc(1, sizeof *f);
    f->kind = MRO_FARROW;
    f->th = th; f->tx = tx; f->nc = dtype_is_complex(tx) ? 2 : 1;
    f->hLen = hLen; f->L = Nphi; f->M = 1; f->Nphi = Nphi; f->rate = rate;
    f->tapsPerPhi = (hLen + Nphi - 1) / Nphi;
    f->polyorder = polyorder;
    f->pnfb = (double *)malloc((size_t)f->tapsPerPhi * (polyorder + 1) * sizeof(double));
    memcpy(f->pnfb, pnfb, (size_t)f->tapsPerPhi * (polyorder + 1) * sizeof(double));
    f->currentTaps = calloc((size_t)f->tapsPerPhi, dtype_scalar_size(th));
    f->phiAccumulator = 1.0;                              /* 𝜙Idx = 1.0 (a Float64 for this kernel), :142 */
    f->phiIdx = 1; f->alpha = 0.0;
    f->delta = (double)Nphi / rate;                       /* :143 */
    f->inputDeficit = 1; f->xIdx = 1;
    farrow_taps(f, f->phiAccumulator);                    /* :146 */
    f->historyLen = f->tapsPerPhi - 1;
    f->history = calloc((size_t)(f->historyLen > 0 ? f->historyLen : 1) * f->nc, dtype_scalar_size(tx));
    return f;
}

/* reference: src/Filters.jl:780-793  update(kernel::FIRFarrow) */
void mro_update_farrow(mro_filter *k)
{
    double Nphi = (double)k->Nphi;
    k->phiAccumulator += k->delta;
    if (k->phiAccumulator > Nphi) {
        k->xIdx += (long)floor((k->phiAccumulator - 1.0) / Nphi);
        k->phiAccumulator = mro_mod_pos(k->phiAccumulator - 1.0, Nphi) + 1.0;
    }
    farrow_taps(k, k->phiAccumulator);
    k->phiIdx = (long)floor(k->phiAccumulator);           /* bookkeeping only (state snapshots) */
    k->alpha = k->phiAccumulator - (double)k->phiIdx;
}

void mro_get_current_taps(const mro_filter *f, void *out)
{
    if (f->currentTaps) memcpy(out, f->currentTaps, (size_t)f->tapsPerPhi * dtype_scalar_size(f->th));
}

void mro_destroy(mro_filter *f)
{
    if (!f) return;
    free(f->taps); free(f->dtaps); free(f->pnfb); free(f->currentTaps); free(f->history); free(f);
}

/* reference: src/Filters.jl:663-673  update(kernel::FIRArbitrary)
 * Julia's mod(x::Float64, y) for positive operands is the exact remainder; fmod is exact. */
void mro_update_arbitrary(mro_filter *k)
{
    double Nphi = (double)k->Nphi;
    k->phiAccumulator += k->delta;
    if (k->phiAccumulator > Nphi) {
        k->xIdx += (long)floor((k->phiAccumulator - 1.0) / Nphi);
        k->phiAccumulator = mro_mod_pos(k->phiAccumulator - 1.0, Nphi) + 1.0;
    }
    k->phiIdx = (long)floor(k->phiAccumulator);
    k->alpha = k->phiAccumulator - (double)k->phiIdx;
}

/* reference: src/Filters.jl:359-385  outputlength(kernel, inputlength) */
long mro_outputlength(const mro_filter *f, long inputlength)
{
    switch (f->kind) {
    case MRO_STANDARD: return inputlength;                                     /* :359 */
    case MRO_INTERPOLATOR: return f->L * inputlength;                           /* :363 */
    case MRO_DECIMATOR: return mro_outputlength_ratio(inputlength - f->inputDeficit + 1, 1, f->M, 1); /* :367 */
    case MRO_RATIONAL: return mro_outputlength_ratio(inputlength - f->inputDeficit + 1, f->L, f->M, f->phiIdx); /* :371 */
    case MRO_ARBITRARY: return (long)ceil((double)(inputlength - f->inputDeficit + 1) * f->rate); /* :375 */
    case MRO_FARROW: return (long)ceil((double)(inputlength - f->inputDeficit + 1) * f->rate);    /* :379 */
    }
    return -1;
}

/* reference: src/Filters.jl:403-422  inputlength(FIRFilter, outputlength).
 * The Decimator method reads a non-existent field (:415); restated with the
 * intent the Rational method (:418-422) shows: + inputDeficit - 1. */
long mro_inputlength(const mro_filter *f, long outputlength)
{
    switch (f->kind) {
    case MRO_STANDARD: return outputlength;
    case MRO_INTERPOLATOR: return mro_inputlength_ratio(outputlength, f->L, 1, 1);
    case MRO_DECIMATOR: return mro_inputlength_ratio(outputlength, 1, f->M, 1) + f->inputDeficit - 1;
    case MRO_RATIONAL: return mro_inputlength_ratio(outputlength, f->L, f->M, f->phiIdx) + f->inputDeficit - 1;
    default: return -1;
    }
}

/* ---- typed bodies ------------------------------------------------------ */
#define FN(name) name##_ff
#define TH float
#define TX float
#define R float
#include "oracle_typed.inc"
#undef FN
#undef TH
#undef TX
#undef R

#define FN(name) name##_fd
#define TH float
#define TX double
#define R double
#include "oracle_typed.inc"
#undef FN
#undef TH
#undef TX
#undef R

#define FN(name) name##_df
#define TH double
#define TX float
#define R double
#include "oracle_typed.inc"
#undef FN
#undef TH
#undef TX
#undef R

#define FN(name) name##_dd
#define TH double
#define TX double
#define R double
#include "oracle_typed.inc"
#undef FN
#undef TH
#undef TX
#undef R

/* reference: the `filt(self, x)` wrappers, Filters.jl:475,519,577,633,744.
 * y must hold promote_type(Th,Tx) elements; returns the number of output
 * samples written, -1 if ycap is too small, -2 on an unsafedot guard. */
long mro_filt_sched(mro_filter *f, const void *x, long xLen, void *y, long ycap, mro_sched *sched)
{
    int hd = dtype_is_f64(f->th), xd = dtype_is_f64(f->tx);
#define DISPATCH(SUF, TXT, RT)                                                                   \
    switch (f->kind) {                                                                           \
    case MRO_STANDARD: return filt_standard_##SUF(f, (const TXT *)x, xLen, (RT *)y, ycap);       \
    case MRO_INTERPOLATOR: return filt_interpolator_##SUF(f, (const TXT *)x, xLen, (RT *)y, ycap); \
    case MRO_RATIONAL: return filt_rational_##SUF(f, (const TXT *)x, xLen, (RT *)y, ycap);       \
    case MRO_DECIMATOR: return filt_decimator_##SUF(f, (const TXT *)x, xLen, (RT *)y, ycap);     \
    case MRO_ARBITRARY: return filt_arbitrary_##SUF(f, (const TXT *)x, xLen, (RT *)y, ycap, sched); \
    case MRO_FARROW: return filt_farrow_##SUF(f, (const TXT *)x, xLen, (RT *)y, ycap, sched);       \
    }
    if (!hd && !xd) { DISPATCH(ff, float, float) }
    else if (!hd && xd) { DISPATCH(fd, double, double) }
    else if (hd && !xd) { DISPATCH(df, float, double) }
    else { DISPATCH(dd, double, double) }
#undef DISPATCH
    return -3;
}

long mro_filt(mro_filter *f, const void *x, long xLen, void *y, long ycap)
{
    return mro_filt_sched(f, x, xLen, y, ycap, NULL);
}

/* ---- introspection for tests ------------------------------------------- */
void mro_get_state(const mro_filter *f, mro_state *s)
{
    s->kind = f->kind; s->phiIdx = f->phiIdx; s->inputDeficit = f->inputDeficit;
    s->phiAccumulator = f->phiAccumulator; s->alpha = f->alpha; s->delta = f->delta;
    s->xIdx = f->xIdx; s->tapsPerPhi = f->tapsPerPhi; s->Nphi = f->Nphi;
    s->historyLen = f->historyLen; s->L = f->L; s->M = f->M; s->hLen = f->hLen;
}

void mro_set_state(mro_filter *f, long phiIdx, long inputDeficit, double phiAccumulator)
{
    f->phiIdx = phiIdx; f->inputDeficit = inputDeficit;
    if (f->kind == MRO_ARBITRARY || f->kind == MRO_FARROW) {
        f->phiAccumulator = phiAccumulator;
        f->phiIdx = (long)floor(phiAccumulator);
        f->alpha = phiAccumulator - (double)f->phiIdx;
        if (f->kind == MRO_FARROW) farrow_taps(f, phiAccumulator);
    }
}

void mro_get_history(const mro_filter *f, void *out)
{
    memcpy(out, f->history, (size_t)f->historyLen * f->nc * dtype_scalar_size(f->tx));
}

void mro_set_history(mro_filter *f, const void *in)
{
    memcpy(f->history, in, (size_t)f->historyLen * f->nc * dtype_scalar_size(f->tx));
}

/* taps as stored by the kernel (flipped h or pfb / dpfb), in the tap dtype */
void mro_get_taps(const mro_filter *f, int which, void *out)
{
    const void *src = which ? f->dtaps : f->taps;
    if (!src) return;
    memcpy(out, src, (size_t)f->tapsPerPhi * f->Nphi * dtype_scalar_size(f->th));
}

/* reference: src/Filters.jl:244-260 reset(). FIRRational resets phiIdx but not
 * inputDeficit (:247) and reset(::FIRArbitrary) is broken (:250-253).  The
 * oracle restores the constructor state for every kind (what the tests need:
 * runtests.jl:79 calls reset between the two-chunk and piecewise runs). */
void mro_reset(mro_filter *f)
{
    memset(f->history, 0, (size_t)f->historyLen * f->nc * dtype_scalar_size(f->tx));
    f->phiIdx = 1; f->inputDeficit = 1; f->phiAccumulator = 1.0; f->alpha = 0.0; f->xIdx = 1;
    if (f->kind == MRO_FARROW) farrow_taps(f, 1.0);
}
