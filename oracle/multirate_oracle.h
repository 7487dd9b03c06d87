/*
 * multirate_oracle.h -- C interface of the CPU oracle (TEST INFRASTRUCTURE ONLY;
 * see the header of multirate_oracle.c for scope, citations and pinning status).
 */
#ifndef MULTIRATE_ORACLE_H
#define MULTIRATE_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { MRO_F32 = 0, MRO_F64 = 1, MRO_C64 = 2, MRO_C128 = 3 } mro_dtype;
typedef enum {
    MRO_STANDARD = 0, MRO_DECIMATOR = 1, MRO_INTERPOLATOR = 2, MRO_RATIONAL = 3, MRO_ARBITRARY = 4,
    MRO_FARROW = 5
} mro_kind;

typedef struct mro_filter mro_filter;

typedef struct {
    long xIdx;     /* 1-based input index within the call */
    long phiIdx;   /* 1-based polyphase branch */
    double alpha;
} mro_sched;

typedef struct {
    int kind;
    long phiIdx, inputDeficit;
    double phiAccumulator, alpha, delta;
    long xIdx, tapsPerPhi, Nphi, historyLen, L, M, hLen;
} mro_state;

int mro_output_dtype(int th, int tx);
void mro_set_mod_form(int form); /* 0: exact remainder (default); 1: rem(y + rem(x, y), y), see multirate_oracle.c */
void mro_set_fused(int fused);   /* 1: one fma per tap (checker for MRHIP_NUMERICS_FUSED); 0: the reference's arithmetic */
void mro_shiftin(void *a, long aLen, const void *b, long bLen, size_t elsize);
long mro_nextphase(long currentphase, long interpolation, long decimation);
long mro_outputlength_ratio(long inputlength, long interpolation, long decimation, long initialPhi);
long mro_inputlength_ratio(long outputlength, long interpolation, long decimation, long initialPhi);
long mro_taps2pfb(const void *h, long hLen, int th, long Nphi, void *out);

mro_filter *mro_create_rational(const void *h, long hLen, int th, long num, long den, int tx);
mro_filter *mro_create_arbitrary(const void *h, long hLen, int th, double rate, long Nphi, int tx);
/* FIRFarrow (src/Filters.jl:123-147): `pnfb` holds the polynomial filter bank the caller fitted
 * (pfb2pnfb/polyfit, :311-321, support.jl:85-88): tapsPerPhi polynomials of polyorder+1 coefficients,
 * ascending powers, each already rounded to the tap type.  tapsPerPhi is ceil(hLen/Nphi). */
mro_filter *mro_create_farrow(long hLen, int th, double rate, long Nphi, long polyorder, const double *pnfb, int tx);
void mro_update_farrow(mro_filter *k);
/* polyval(p::Poly{T}, x::Float64) as Polynomials.jl evaluates it: Horner from the highest power in Float64 */
double mro_polyval(const double *coeffs, long polyorder, double x);
void mro_get_current_taps(const mro_filter *f, void *out);
void mro_destroy(mro_filter *f);
void mro_update_arbitrary(mro_filter *k);

long mro_outputlength(const mro_filter *f, long inputlength);
long mro_inputlength(const mro_filter *f, long outputlength);
long mro_filt(mro_filter *f, const void *x, long xLen, void *y, long ycap);
long mro_filt_sched(mro_filter *f, const void *x, long xLen, void *y, long ycap, mro_sched *sched);

void mro_get_state(const mro_filter *f, mro_state *s);
void mro_set_state(mro_filter *f, long phiIdx, long inputDeficit, double phiAccumulator);
void mro_get_history(const mro_filter *f, void *out);
void mro_set_history(mro_filter *f, const void *in);
void mro_get_taps(const mro_filter *f, int which, void *out);
void mro_reset(mro_filter *f);

#ifdef __cplusplus
}
#endif
#endif
