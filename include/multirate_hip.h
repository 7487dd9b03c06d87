/*
 * multirate_hip.h -- C ABI of libmultirate_hip.so, the MI355X (gfx950) engine behind
 * Multirate.jl's FIRFilter / filt / filt! hot path.
 *
 * The reference (JayKickliter/Multirate.jl) has no FFI boundary: its seam is Julia
 * multiple dispatch on
 *     filt!(buffer::Vector{Tb}, self::FIRFilter{K{Th}}, x::Vector{Tx})
 * (src/Filters.jl:450,489,536,598,693) reached through filt(self, x)
 * (src/Filters.jl:475,519,577,633,744) and filt(h, x, ratio|rate)
 * (src/Filters.jl:858,864).  Each entry point below names the reference
 * interface it replaces.  The Julia-side binding is in
 * multirate.jl_amd/julia/MultirateHIP.jl and described in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross the boundary
 *   - every function returns an mrhip_status (0 = ok) unless noted;
 *     mrhip_last_error() returns the message of the calling thread's last failure
 *     (the reference raises Julia error("...") in the same places)
 *   - complex samples are interleaved (re, im) pairs (Julia Complex{T} layout)
 *   - a filter object carries `nchannels` independent streams that share taps and
 *     the data-independent state (phase index, input deficit, phase accumulator)
 *     and own one history vector each: the batched form of the reference's
 *     one-Vector-per-FIRFilter model (SURVEY.md 8b "Batch")
 *   - channel c of a signal buffer starts at base + c * stride (stride in SAMPLES,
 *     not bytes): planar layout, the layout of a Julia Matrix with one channel per column
 *   - lengths are per channel, in samples
 *   - a handle is a mutable stream object exactly like the reference's FIRFilter:
 *     not to be used from two threads at once
 */
#ifndef MULTIRATE_HIP_H
#define MULTIRATE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRHIP_ABI_VERSION 1

typedef enum {
    MRHIP_OK = 0,
    MRHIP_ERR_INVALID_ARG = 1,      /* reference: error("rate must be greater than 0") etc. */
    MRHIP_ERR_BUFFER_TOO_SMALL = 2, /* reference: error("buffer is too small"), Filters.jl:460,503,550 */
    MRHIP_ERR_HIP = 3,              /* a HIP runtime call failed; message holds hipGetErrorString */
    MRHIP_ERR_NO_DEVICE = 4,        /* no gfx950 device visible: the engine has no CPU fallback */
    MRHIP_ERR_UNSUPPORTED = 5       /* e.g. complex taps (the reference tests never use them) */
} mrhip_status;

/* element types: Th in {F32,F64}; Tx in {F32,F64,C64,C128}; Tb = promote_type(Th,Tx) */
typedef enum { MRHIP_F32 = 0, MRHIP_F64 = 1, MRHIP_C64 = 2, MRHIP_C128 = 3 } mrhip_dtype;

/* kernel kinds == the reference's FIRKernel subtypes, src/Filters.jl:15-117 */
typedef enum {
    MRHIP_FIR_STANDARD = 0,     /* FIRStandard     src/Filters.jl:15-24   */
    MRHIP_FIR_DECIMATOR = 1,    /* FIRDecimator    src/Filters.jl:45-58   */
    MRHIP_FIR_INTERPOLATOR = 2, /* FIRInterpolator src/Filters.jl:28-41   */
    MRHIP_FIR_RATIONAL = 3,     /* FIRRational     src/Filters.jl:62-80   */
    MRHIP_FIR_ARBITRARY = 4,    /* FIRArbitrary    src/Filters.jl:91-117  */
    MRHIP_FIR_FARROW = 5        /* FIRFarrow       src/Filters.jl:123-147 */
} mrhip_kind;

/* Arithmetic contract for the tap dot product (src/support.jl:5-55).
 *   STRICT: the order and roundings the reference source states -- oldest sample first,
 *           first product initialises the accumulator, separately rounded multiply and add
 *           in promote_type(Th,Tx) (no FMA; Julia 0.3 never fused).  Default.
 *   FUSED : same order, each step one fused multiply-add.  Faster where the kernel is
 *           VALU-bound; differs from STRICT by <= 1 rounding per tap. */
typedef enum { MRHIP_NUMERICS_STRICT = 0, MRHIP_NUMERICS_FUSED = 1 } mrhip_numerics;

typedef struct mrhip_filter mrhip_filter; /* opaque; replaces FIRFilter{Tk}, src/Filters.jl:151-155 */

/* Snapshot of the kernel fields of the reference structs (src/Filters.jl:15-117,151-155).
 * Indices are 1-based like the reference's. */
typedef struct {
    int32_t kind;            /* mrhip_kind */
    int32_t tap_dtype;       /* Th */
    int32_t sample_dtype;    /* Tx */
    int32_t output_dtype;    /* Tb = promote_type(Th,Tx) */
    int64_t nchannels;
    int64_t hLen;            /* length(h) as passed */
    int64_t interpolation;   /* L (reduced) ; Nphi for ARBITRARY */
    int64_t decimation;      /* M (reduced) */
    int64_t Nphi;            /* N𝜙 */
    int64_t tapsPerPhi;      /* tapsPer𝜙 (== hLen for STANDARD/DECIMATOR) */
    int64_t historyLen;
    int64_t phiIdx;          /* 𝜙Idx          FIRRational :68, FIRArbitrary :98 */
    int64_t inputDeficit;    /* inputDeficit  :49, :69, :101 */
    int64_t xIdx;            /* FIRArbitrary.xIdx :102 */
    double rate;             /* FIRArbitrary.rate :92 */
    double phiAccumulator;   /* 𝜙Accumulator :97 */
    double alpha;            /* α :99 */
    double delta;            /* Δ :100 */
} mrhip_state;

/* ---- library ------------------------------------------------------------------------- */
int mrhip_abi_version(void);
const char *mrhip_last_error(void);
/* number of visible HIP devices whose arch is gfx950 (0 => every create fails with NO_DEVICE) */
int mrhip_device_count(void);

/* ---- host-only helpers (no GPU needed) ------------------------------------------------ */
/* replaces taps2pfb(h, Nphi), src/Filters.jl:284-298.  `pfb` receives tapsPerPhi*Nphi elements,
 * column-major (column = phase, contiguous); pass pfb = NULL to query.  Returns tapsPerPhi. */
int64_t mrhip_taps2pfb(const void *h, int64_t hLen, int tap_dtype, int64_t Nphi, void *pfb);
/* replaces nextphase(currentphase, ratio), src/Filters.jl:433-439 */
int64_t mrhip_nextphase(int64_t currentphase, int64_t interpolation, int64_t decimation);
/* replaces outputlength(inputlength, ratio, initial𝜙), src/Filters.jl:352-357 */
int64_t mrhip_outputlength_ratio(int64_t inputlength, int64_t interpolation, int64_t decimation,
                                 int64_t initialPhi);
/* replaces inputlength(outputlength, ratio, initial𝜙), src/Filters.jl:396-401 */
int64_t mrhip_inputlength_ratio(int64_t outputlength, int64_t interpolation, int64_t decimation,
                                int64_t initialPhi);
/* replaces polyfit(y, polyorder), src/support.jl:85-88: least-squares polynomial through (x = 1..n, y[x]);
 * coef receives polyorder+1 coefficients, ascending powers (Poly.a).  Returns 0, or MRHIP_ERR_INVALID_ARG when
 * n < polyorder+1.  Float64 throughout (Julia's A \\ y promotes to Float64). */
int mrhip_polyfit(const double *y, int64_t n, int64_t polyorder, double *coef);
/* promote_type(Th, Tx) as used by every filt wrapper, e.g. src/Filters.jl:581 */
int mrhip_output_dtype(int tap_dtype, int sample_dtype);

/* ---- FIR design, host only (src/FIRDesign.jl) ----------------------------------------------- */
/* replaces @enum(FIRResponse, LOWPASS, BANDPASS, HIGHPASS, BANDSTOP), src/FIRDesign.jl:7 (values 0..3, src/enum.jl) */
typedef enum { MRHIP_LOWPASS = 0, MRHIP_BANDPASS = 1, MRHIP_HIGHPASS = 2, MRHIP_BANDSTOP = 3 } mrhip_fir_response;
/* replaces kaiserlength(transition, attenuation = 60; samplerate = 1.0), src/FIRDesign.jl:18-33: (numtaps, beta) */
int mrhip_kaiserlength(double transition, double attenuation, double samplerate, int64_t *numtaps, double *beta);
/* the window firdes multiplies with when windowfunction == kaiser (src/FIRDesign.jl:82-83; DSP.jl's kaiser in the
 * reference, beta taken as is like src/Window.jl:53-58): out receives n Float64 samples */
int mrhip_kaiser(int64_t n, double beta, double *out);
/* replaces firprototype(numtaps, F; response), src/FIRDesign.jl:47-66.  F holds nF cutoffs in cycles/sample (two for
 * BANDPASS / BANDSTOP).  Returns the length of the prototype (numtaps, or numtaps+1 for HIGHPASS with an even
 * numtaps, :55) or -1 (mrhip_last_error: "Not a valid FIR_TYPE", :62); out = NULL only queries the length. */
int64_t mrhip_firprototype(int64_t numtaps, const double *F, int nF, int response, double *out);
/* replaces firdes(numtaps, cutoff, windowfunction; response, samplerate, beta), src/FIRDesign.jl:76-88.
 * `window` = NULL selects the Kaiser window with `beta` (windowfunction == kaiser, :82-83); otherwise it points to
 * the caller's window evaluated at the prototype length (:85) -- query that length first with out = NULL.
 * Returns the number of taps written (Float64), or -1. */
int64_t mrhip_firdes(int64_t numtaps, const double *cutoff, int ncutoff, int response, double samplerate, double beta,
                     const double *window, double *out);
/* replaces firdes(cutoff, transitionwidth, stopbandAttenuation = 60; response, samplerate), src/FIRDesign.jl:90-95
 * (kaiserlength, then the method above).  out = NULL queries the length. */
int64_t mrhip_firdes_kaiser(const double *cutoff, int ncutoff, double transitionwidth, double stopbandAttenuation, int response,
                            double samplerate, double *out);

/* ---- construction --------------------------------------------------------------------- */
/* replaces FIRFilter(h::Vector, resampleRatio::Rational = 1//1), src/Filters.jl:158-180.
 * num//den is reduced like a Julia Rational; the kernel kind is chosen exactly as :163-175
 * (ratio == 1 -> STANDARD, L == 1 -> DECIMATOR, M == 1 -> INTERPOLATOR, else RATIONAL).
 * `h` is a host pointer to hLen taps of tap_dtype (F32 | F64).  `device` is a HIP ordinal. */
int mrhip_create_rational(const void *h, int64_t hLen, int tap_dtype, int64_t num, int64_t den,
                          int sample_dtype, int64_t nchannels, int device, mrhip_filter **out);
/* replaces FIRFilter(h::Vector, rate::FloatingPoint, Nphi::Integer = 32), src/Filters.jl:183-189
 * (+ FIRArbitrary(h, rate, Nphi), :105-117: dh = [diff(h), 0], two PFBs).  rate <= 0 is
 * MRHIP_ERR_INVALID_ARG ("rate must be greater than 0", :184). */
int mrhip_create_arbitrary(const void *h, int64_t hLen, int tap_dtype, double rate, int64_t Nphi,
                           int sample_dtype, int64_t nchannels, int device, mrhip_filter **out);
/* replaces FIRFilter(h::Vector, rate::FloatingPoint, Nphi::Integer, polyorder::Integer),
 * src/Filters.jl:192-198 (+ FIRFarrow(h, rate, Nphi, polyorder), :138-147, pfb2pnfb :311-321 and
 * polyfit, src/support.jl:85-88): every ROW of the tapsPerPhi x Nphi filter bank is replaced by its
 * least-squares polynomial of degree polyorder over x = 1..Nphi, stored in the tap type; the taps of an
 * output are those polynomials evaluated at its Float64 phase.  The fit (Julia: A \ y, LAPACK QR) is
 * done on the host in Float64 by a Householder QR. */
int mrhip_create_farrow(const void *h, int64_t hLen, int tap_dtype, double rate, int64_t Nphi, int64_t polyorder,
                        int sample_dtype, int64_t nchannels, int device, mrhip_filter **out);
/* same, with the polynomial filter bank supplied by the caller: pnfb[tapsPerPhi][polyorder+1] Float64,
 * ascending powers (the layout of Poly.a), tapsPerPhi = ceil(hLen/Nphi); values are rounded to the tap type
 * as the reference's Poly{T} storage does.  Lets a caller fit with its own least-squares routine (the
 * reference pins no bits of the fit) and lets tests hand oracle and GPU the same coefficients. */
int mrhip_create_farrow_pnfb(const double *pnfb, int64_t hLen, int tap_dtype, double rate, int64_t Nphi,
                             int64_t polyorder, int sample_dtype, int64_t nchannels, int device, mrhip_filter **out);
/* the polynomial filter bank in use, [tapsPerPhi][polyorder+1] Float64 (pfb2pnfb's result) */
int mrhip_get_pnfb(const mrhip_filter *f, double *host_out);
/* replaces tapsforphase(kernel::FIRFarrow, phase), src/Filters.jl:764-775: tapsPerPhi taps of tap_dtype
 * for a phase in [0, Nphi+1] (host evaluation; MRHIP_ERR_INVALID_ARG outside the range like :765) */
int mrhip_farrow_tapsforphase(const mrhip_filter *f, double phase, void *host_out);
/* replaces tapsforphase(kernel::FIRArbitrary, phase), src/Filters.jl:677-690: (alpha, phiIdx) = modf(phase);
 * taps[i] = pfb[i, phiIdx] + alpha * dpfb[i, phiIdx], evaluated in Float64 (alpha is a Float64 there) and stored in
 * tap_dtype.  phase outside [0, Nphi+1] is MRHIP_ERR_INVALID_ARG (:678); so is a phase whose integer part is not a
 * column of the bank (0 or Nphi+1: a BoundsError in the reference). */
int mrhip_arbitrary_tapsforphase(const mrhip_filter *f, double phase, void *host_out);
void mrhip_destroy(mrhip_filter *f);

/* ---- bookkeeping ---------------------------------------------------------------------- */
/* replaces outputlength(self::FIRFilter, inputlength), src/Filters.jl:359-385 (per channel).
 * For ARBITRARY this is the reference's ceil((n - deficit + 1) * rate) estimate (:375-377). */
int64_t mrhip_outputlength(const mrhip_filter *f, int64_t inputlength);
/* exact number of samples the next filt call with `inputlength` samples will write per channel
 * (== outputlength for the rational family when inputlength >= inputDeficit, else 0; for ARBITRARY it
 * runs the phase recurrence of update(), src/Filters.jl:663-673, without touching the state). */
int64_t mrhip_next_output_count(const mrhip_filter *f, int64_t inputlength);
/* Advance the stream state exactly as a filt! call over `inputlength` samples per channel would -- the state updates at
 * the end of every filt! (src/Filters.jl:472 Standard: none, :515-516 Interpolator, :571-572 Rational, :647-648 Decimator,
 * :731-735 Arbitrary, :836-838 Farrow) -- WITHOUT data and without touching the history: returns the number of outputs
 * that call would have written per channel, or -1.  The state machine is data independent, so a stream can be entered
 * at any sample: advance to it, mrhip_set_history with the tapsPerPhi-1 samples in front of it (support.jl:61-80 is
 * all a later call sees of the earlier ones), filt from there -- what sharding.py's TimeShardedFilter does per GPU. */
int64_t mrhip_advance_state(mrhip_filter *f, int64_t inputlength);
/* replaces inputlength(self::FIRFilter, outputlength), src/Filters.jl:403-422 */
int64_t mrhip_inputlength(const mrhip_filter *f, int64_t outputlength);
int mrhip_get_state(const mrhip_filter *f, mrhip_state *st);
/* restore a stream position (phiIdx, inputDeficit; phiAccumulator for ARBITRARY, from which
 * phiIdx and alpha are re-derived as update() does).  The reference has no such call; it is the
 * working counterpart of setphase/reset (src/Filters.jl:210-260, several of which are broken). */
int mrhip_set_state(mrhip_filter *f, int64_t phiIdx, int64_t inputDeficit, double phiAccumulator);
/* copy the history vectors (FIRFilter.history, src/Filters.jl:153) of all channels to / from a
 * host buffer laid out [nchannels][historyLen] in sample_dtype.  Synchronous. */
int mrhip_get_history(mrhip_filter *f, void *host_out);
int mrhip_set_history(mrhip_filter *f, const void *host_in);
/* the same from DEVICE memory on f's device, asynchronous on `stream` (ordered behind the filter's earlier calls and ahead of
 * its next one): a halo that arrived over RCCL goes into the filter without touching the host (sharding.py). */
int mrhip_set_history_device(mrhip_filter *f, const void *device_in, void *stream);
/* replaces reset(self::FIRFilter), src/Filters.jl:256-260: zero history; state back to the
 * constructor's (the reference resets 𝜙Idx only and is broken for FIRArbitrary, :247-253). */
int mrhip_reset(mrhip_filter *f);
int mrhip_set_numerics(mrhip_filter *f, int numerics);
/* FIRArbitrary / FIRFarrow: which floating-point mod() update() wraps the phase accumulator with (src/Filters.jl:668, :786).
 * form 0 (default): the exact remainder -- Julia >= 0.4.  form 1: rem(y + rem(x, y), y) -- how Julia's Base computed mod() for
 * floats before 0.4; the reference is Julia-0.3 code and nothing in its tree says which Base it ran on.  The two agree bit for bit
 * whenever Nphi is a power of two (BASELINE config 4: Nphi = 32); for other Nphi form 1 differs by one rounding of y + rem(x, y)
 * every few outputs (a random walk of ~1e-10 phase steps per 1e6 outputs).  With form 1 and such an Nphi the phase schedule is
 * evaluated by the host's serial loop (the device evaluation implements the exact remainder) and asynchronous / captured calls
 * are MRHIP_ERR_UNSUPPORTED.  The oracle has the same switch (oracle.set_mod_form). */
int mrhip_set_mod_form(mrhip_filter *f, int form);
/* the polyphase taps as stored on the device, converted back to tap_dtype (which = 0: h flipped or
 * pfb; which = 1: dpfb).  Column-major tapsPerPhi x Nphi.  For tests / tapsforphase. */
int mrhip_get_taps(mrhip_filter *f, int which, void *host_out);

/* ---- the hot path --------------------------------------------------------------------- */
/* replaces filt!(buffer, self, x) for all five kernels, src/Filters.jl:450,489,536,598,693,
 * on DEVICE memory: x and y are device pointers on f's device; channel c is read from
 * x + c*x_stride and written to y + c*y_stride (strides in samples).  y holds elements of
 * output_dtype.  y_capacity is the per-channel room in y; if it is smaller than the number of
 * samples the call will produce the call fails with MRHIP_ERR_BUFFER_TOO_SMALL and the filter
 * state is unchanged (reference: error() before the loop, :460,:503,:550).
 * *n_written receives the per-channel output count (the Int that the Rational/Decimator/
 * Arbitrary filt! return, :574,:630,:741; xLen resp. L*xLen for Standard/Interpolator).  It is a
 * pure function of (state, x_len) and is valid on return even though the kernels are only
 * enqueued: the call is asynchronous on `stream` (a hipStream_t, NULL = default stream).
 * A short input (x_len < inputDeficit) is not an error: history is shifted, inputDeficit
 * reduced, *n_written = 0 (:543-547, :638-643, :705-709).
 * On a stream that is being captured into a HIP graph the call takes the device-planned path of mrhip_filt_device_async
 * (same buffer rule); *n_written then is the count of the FIRST replay (rational family) or -1 (FIRArbitrary / FIRFarrow:
 * read it with mrhip_sync_state after a replay). */
int mrhip_filt_device(mrhip_filter *f, const void *x, int64_t x_len, int64_t x_stride, void *y,
                      int64_t y_capacity, int64_t y_stride, int64_t *n_written, void *stream);
/* filt!(buffer, self, x) with NOTHING returned to the host: the call is planned ON THE DEVICE.  The reference mutates
 * 𝜙Idx / inputDeficit / 𝜙Accumulator at the end of every filt! (src/Filters.jl:571-572, 627-628, 734-735, update() :663-673);
 * here that state has a device-resident record per filter that every call keeps current in stream order.  This entry reads
 * the record in a one-lane plan kernel in front of the filter kernel (FIRArbitrary / FIRFarrow: in the kernels that evaluate
 * the phase schedule), advances it there, and never makes the host wait -- so a loop of such calls only enqueues, and the
 * same calls captured into a HIP graph replay correctly at ANY fixed chunk size and for every kind (mrhip_filt_device on a
 * capturing stream takes this path by itself).  The count is data independent but state dependent, so:
 *   - y_capacity (and y_stride for nchannels > 1) must be at least mrhip_outputlength_bound(x_len), else
 *     MRHIP_ERR_BUFFER_TOO_SMALL and nothing is enqueued (the reference's filt sizes its buffer from outputlength the same
 *     way, src/Filters.jl:744-751);
 *   - *count_out, if not NULL, must be DEVICE-ACCESSIBLE memory (hipMalloc or hipHostMalloc): the per-channel output count
 *     is stored there in stream order; mrhip_sync_state returns the last call's count as well;
 *   - the host-side view of the state (mrhip_get_state, outputlength, ...) re-reads the record when next asked, which
 *     waits for the filter's stream (after a capture: for the device).
 * A call longer than one launch (2^30 samples; 2^24 / rate for FIRArbitrary) is MRHIP_ERR_UNSUPPORTED here. */
int mrhip_filt_device_async(mrhip_filter *f, const void *x, int64_t x_len, int64_t x_stride, void *y,
                            int64_t y_capacity, int64_t y_stride, int64_t *count_out, void *stream);
/* The second stage of a device-resident chain (filt(f2, filt(f1, x)) in the reference's terms: the caller passes one filter's
 * output Vector to the next, src/Filters.jl:744-751): `f`'s input is what `prev`'s latest asynchronous or captured call --
 * earlier on the same stream -- wrote, and its LENGTH is that call's count, which only the device knows.  x_len_bound =
 * mrhip_outputlength_bound(prev, prev's input length): the launch is sized for it, and y_capacity / y_stride must cover
 * mrhip_outputlength_bound(f, x_len_bound).  `f` and `prev`: any kind (FIRArbitrary / FIRFarrow lay their phase
 * schedule out for the device-side length in the kernels that evaluate it); same count_out / state rules
 * as mrhip_filt_device_async.  With it a chain of filters runs, and replays from a HIP graph, at any chunk size without
 * the host learning a single count. */
int mrhip_filt_device_chained(mrhip_filter *f, const mrhip_filter *prev, const void *x, int64_t x_len_bound, int64_t x_stride,
                              void *y, int64_t y_capacity, int64_t y_stride, int64_t *count_out, void *stream);
/* filt!(buffer_i, self_i, x_i) for i = 0..n-1 -- n INDEPENDENT FIRFilter objects, each with its own phase, input deficit,
 * history and call length, the reference's one-FIRFilter-per-signal streaming usage (README.md:87-141) -- issued as ONE
 * launch when the filters agree in kind (the rational family: FIRStandard, FIRDecimator, FIRInterpolator, FIRRational), ratio,
 * tapsPerPhi, dtypes, numerics and device:
 * every workgroup of the launch works for one of the streams.  x[i] / y[i] are device pointers, channel c of filter i at
 * x[i] + c*x_len[i] and y[i] + c*y_capacity[i]; n_written[i] (optional) receives filter i's per-channel count.  Results,
 * states and histories are exactly those of n mrhip_filt_device calls, which is also what runs when the filters do not agree
 * (and inside a HIP-graph capture).  MRHIP_ERR_BUFFER_TOO_SMALL before anything is enqueued. */
int mrhip_filt_device_multi(mrhip_filter *const *filters, int n, const void *const *x, const int64_t *x_len, void *const *y,
                            const int64_t *y_capacity, int64_t *n_written, void *stream);
/* the largest per-channel output count a call of `inputlength` samples can have whatever the stream state: outputlength
 * (src/Filters.jl:352-385) evaluated for 𝜙Idx = inputDeficit = 1 (+ 2 for FIRArbitrary / FIRFarrow, whose outputlength is
 * an estimate, :375-381) */
int64_t mrhip_outputlength_bound(const mrhip_filter *f, int64_t inputlength);
/* wait for everything the filter has enqueued (after a HIP-graph capture of one of its calls: for the device -- replays run
 * on streams the library never saw) and take the device-resident stream state over into the host object;
 * *last_n_written (optional) receives the per-channel count of the last call.  Returns the status of a device-planned call
 * that failed since the last mrhip_sync_state, once.  A device-wide wait invalidates a HIP-graph capture that is going on on
 * another stream: while a stream the library has been called on is still capturing, the wait of a filter whose calls were captured
 * once -- here and in every entry point that needs the host-side state (mrhip_next_output_count, mrhip_outputlength,
 * mrhip_get_state, a synchronous mrhip_filt_*) -- is refused with MRHIP_ERR_UNSUPPORTED (-1 from the counting functions). */
int mrhip_sync_state(mrhip_filter *f, int64_t *last_n_written);
/* Streaming helper: exactly the sequence of filt!(buffer, self, x[a:a+chunk]) calls a caller would make
 * over consecutive `chunk`-sample pieces of a device-resident signal, issued back to back by the library (one host
 * call instead of x_len/chunk).  Output k of piece i lands right after the outputs of piece i-1 in y; *n_written is the
 * total per channel.  State and history are carried from piece to piece on the device, bit-identical to the caller's own loop.
 * Like that loop, an error (e.g. MRHIP_ERR_BUFFER_TOO_SMALL for a later piece) leaves the pieces already issued applied:
 * *n_written then holds the outputs produced so far and the filter stands at the start of the failing piece. */
int mrhip_filt_device_chunked(mrhip_filter *f, const void *x, int64_t x_len, int64_t x_stride, int64_t chunk, void *y,
                              int64_t y_capacity, int64_t y_stride, int64_t *n_written, void *stream);
/* same contract with HOST pointers: copies x in, runs mrhip_filt_device, copies y out,
 * synchronises.  Replaces filt!(buffer, self, x) for a caller whose data lives in host memory. */
int mrhip_filt_host(mrhip_filter *f, const void *x, int64_t x_len, int64_t x_stride, void *y,
                    int64_t y_capacity, int64_t y_stride, int64_t *n_written);
/* block until everything enqueued on the filter's behalf on `stream` has finished */
int mrhip_synchronize(mrhip_filter *f, void *stream);

/* ---- cascades (the reference chains filt calls by hand) ------------------------ */
/* A chain of filters run back to back on one stream: y = filt(stage[n-1], ... filt(stage[0], x)).  The stages are
 * ordinary stateful filters (borrowed, not owned: destroy them after the cascade); stage i+1's sample dtype must be
 * stage i's output dtype, all stages share nchannels and the device.  Intermediate signals live in device buffers
 * owned by the cascade and never leave HBM; every count is closed-form on the host, so nothing is read back between
 * stages and chunked calls continue the stream exactly like calling the stages by hand. */
typedef struct mrhip_cascade mrhip_cascade;
int mrhip_cascade_create(mrhip_filter *const *stages, int nstages, mrhip_cascade **out);
void mrhip_cascade_destroy(mrhip_cascade *c);
/* outputlength of the chain (each stage's outputlength applied in turn: an estimate where a stage's is, :375) */
int64_t mrhip_cascade_outputlength(const mrhip_cascade *c, int64_t inputlength);
/* exact per-channel output count of the next mrhip_cascade_filt_device call with `inputlength` samples */
int64_t mrhip_cascade_next_output_count(const mrhip_cascade *c, int64_t inputlength);
/* same contract as mrhip_filt_device for the whole chain; MRHIP_ERR_BUFFER_TOO_SMALL leaves every stage untouched.
 * Under HIP-graph capture the lengths planned at capture time are what every replay runs with (a stage's input length is part
 * of the next stage's launch), so every stage must map its input length to the same count on every replay: rational-family
 * stages with inputlength * L a multiple of M; anything else (and FIRArbitrary / FIRFarrow stages) is MRHIP_ERR_UNSUPPORTED
 * before anything is launched.  y needs room for mrhip_outputlength_bound of the last stage, and one plain call of the same
 * size must have run before the capture (it allocates the buffers between the stages). */
int mrhip_cascade_filt_device(mrhip_cascade *c, const void *x, int64_t x_len, int64_t x_stride, void *y, int64_t y_capacity,
                              int64_t y_stride, int64_t *n_written, void *stream);
/* The chain with nothing returned to the host (asynchronous, or under HIP-graph capture at ANY chunk size): the first stage is
 * mrhip_filt_device_async, every later one mrhip_filt_device_chained on the previous stage's count.  y needs room for the
 * last stage's bound of the bounds; *count_out (device-accessible, may be NULL) receives the chain's per-channel output
 * count.  One plain call of the same size must have run before a capture (it allocates the buffers between the stages);
 * every kind of stage in every position. */
int mrhip_cascade_filt_device_async(mrhip_cascade *c, const void *x, int64_t x_len, int64_t x_stride, void *y, int64_t y_capacity,
                                    int64_t y_stride, int64_t *count_out, void *stream);
int mrhip_cascade_reset(mrhip_cascade *c);

/* ---- a ring of arriving chunks: ONE resident kernel instead of a launch per chunk ------------------------------ */
/* replaces the reference's streaming usage -- a loop of  y_i = filt(self, x_i)  over the chunks of a signal as they arrive, on one
 * stateful FIRFilter (README.md:87-141; the state carried from call to call: src/Filters.jl:571-572, history: support.jl:61-80).
 * mrhip_ring_open starts a kernel that stays resident on the filter's device and consumes chunk descriptors; mrhip_ring_push is
 * filt!(y, self, x) for the next chunk: it plans the call on the host (the state recurrence in closed form, so *n_written -- the
 * Int the reference's filt! returns -- is valid on return), writes one descriptor into pinned memory and returns; the kernel
 * filters the chunk beside the chunks before it (several are in flight at once, the history travels between them on the
 * device) and raises a flag in pinned memory when its outputs are complete: mrhip_ring_wait / mrhip_ring_drain.  Results, counts,
 * end state and history are bit for bit those of the same calls made through mrhip_filt_device.
 *   - x must be COMPLETE in device memory when it is pushed (the resident kernel is not ordered behind any stream: synchronise
 *     the producer of x first) and x and y must stay untouched until the chunk's flag is up;
 *   - at most 63 chunks are in flight; a push into a full ring waits for the oldest;
 *   - while a ring is open the filter's other entry points (filt, reset, set_state, ...) return MRHIP_ERR_INVALID_ARG, and no call
 *     that waits for the whole DEVICE (hipDeviceSynchronize, hipFree, ...) returns before mrhip_ring_close: the kernel leaves
 *     only when the ring is closed, or by itself after MRHIP_RING_IDLE_MS (default 2000) without a push -- the next push then
 *     starts a new one;
 *   - FIRRational / FIRInterpolator with 24 or 32 taps per phase and M < 2L (the BASELINE shapes; STRICT numerics) run on the
 *     resident kernel; every other filter takes the same interface as stream-ordered launches, one per chunk, on the ring's
 *     stream (mrhip_ring_info tells which);
 *   - a chunk is "complete" when a later kernel on any stream of the device, or a copy, reads its outputs: small chunks are stored
 *     write-through, chunks of at least MRHIP_RING_FLUSH_MIN_MB (32) of outputs plainly with one L2 write-back per XCD in front of the flag;
 *   - the resident kernel's launch is checked: if not all of its workgroups are on the chip within 3 ms (the dealing of the work needs
 *     every one of them) it is launched again with as many as were. */
typedef struct mrhip_ring mrhip_ring;
int mrhip_ring_open(mrhip_filter *f, mrhip_ring **out);
/* filt!(y, self, x) for the next arriving chunk (same argument meaning and errors as mrhip_filt_device); *seq (optional)
 * receives the chunk's number for mrhip_ring_wait */
int mrhip_ring_push(mrhip_ring *r, const void *x, int64_t x_len, int64_t x_stride, void *y, int64_t y_capacity, int64_t y_stride,
                    int64_t *n_written, uint64_t *seq);
/* the library's chunk loop over a device-resident signal (mrhip_filt_device_chunked's contract): one mrhip_ring_push per `chunk`
 * samples, outputs back to back in y; *n_written the total, *last_seq the last chunk's number */
int mrhip_ring_push_chunks(mrhip_ring *r, const void *x, int64_t x_len, int64_t x_stride, int64_t chunk, void *y, int64_t y_capacity,
                           int64_t y_stride, int64_t *n_written, uint64_t *last_seq);
/* block until chunk `seq` is complete (its outputs are in device memory); chunks may complete out of order */
int mrhip_ring_wait(mrhip_ring *r, uint64_t seq);
/* block until every pushed chunk is complete */
int mrhip_ring_drain(mrhip_ring *r);
/* drain, end the resident kernel, hand the stream (state behind the last chunk, history, device record) back to the filter,
 * free the ring.  (A filter destroyed while it still feeds a ring -- finalizers run in any order -- shuts the ring down itself;
 * the ring's entry points then fail and mrhip_ring_close only frees the handle.) */
int mrhip_ring_close(mrhip_ring *r);
/* info[0..n): [0] 1 = resident kernel, 0 = one launch per chunk; [1] ring depth; [2] chunks pushed; [3] kernels restarted after an
 * idle deadline; [4] steps per grab; [5] outputs per step; [6] launches repeated with fewer workgroups (not all started in time: a
 * co-tenant kernel held part of the chip); [7] workgroups of the resident kernel (0: as many as the chip holds); [8] XCDs of the
 * device (other than 8: every chunk completes by write-through stores) */
int mrhip_ring_info(const mrhip_ring *r, int64_t *info, int n);

/* ---- one FIRFilter whose channels are split over several GPUs (BASELINE.json config 5) ------------------------ */
/* replaces filt(self, x) (src/Filters.jl:577-587 and its siblings :475,519,633,744,841) for a multi-channel signal -- in the
 * reference: one FIRFilter per channel, nothing shared but the read-only taps -- split by CHANNEL over `ndevices` GPUs of one
 * process: shard i is an ordinary filter on devices[i] with channels [start_i, start_i + count_i) (contiguous; the first
 * nchannels % ndevices shards take one channel more), a stream of its own, and no exchange with the others during compute.  The
 * only traffic between devices is the final gather of the outputs (mrhip_sharded_gather: peer copies over xGMI on the shards'
 * streams).  The same device may be named more than once (two shards on one GPU).
 * ctor selects the reference constructor: 0 = FIRFilter(h, num//den) (:158), 1 = FIRFilter(h, rate, Nphi) (:183),
 * 2 = FIRFilter(h, rate, Nphi, polyorder) (:192); arguments a constructor does not take are ignored. */
typedef struct mrhip_sharded mrhip_sharded;
int mrhip_sharded_create(int ctor, const void *h, int64_t hLen, int tap_dtype, int64_t num, int64_t den, double rate, int64_t Nphi,
                         int64_t polyorder, int sample_dtype, int64_t nchannels, const int *devices, int ndevices, mrhip_sharded **out);
void mrhip_sharded_destroy(mrhip_sharded *s);
int mrhip_sharded_nshards(const mrhip_sharded *s);
/* shard i: first channel, channel count, device, and the shard's filter itself (borrowed: state, history, timing ... through the
 * ordinary entry points; NULL for a shard without channels); any of the four may be NULL */
int mrhip_sharded_shard(const mrhip_sharded *s, int i, int64_t *start, int64_t *count, int *device, mrhip_filter **filter);
/* outputlength / the exact count of the next call (the state machine is data independent: every shard agrees) / reset */
int64_t mrhip_sharded_outputlength(const mrhip_sharded *s, int64_t inputlength);
int64_t mrhip_sharded_next_output_count(const mrhip_sharded *s, int64_t inputlength);
int mrhip_sharded_reset(mrhip_sharded *s);
/* filt!(buffer, self, x) with device-resident data: x[i] / y[i] are device pointers ON SHARD i's DEVICE holding its count_i channels
 * (channel c at + c * stride[i]; NULL strides: x_len / y_capacity); same contract as mrhip_filt_device per shard, each on the
 * shard's own stream (asynchronous: mrhip_sharded_synchronize waits for all); *n_written = the per-channel count. */
int mrhip_sharded_filt_device(mrhip_sharded *s, const void *const *x, int64_t x_len, const int64_t *x_stride, void *const *y, int64_t y_capacity,
                              const int64_t *y_stride, int64_t *n_written);
/* filt!(buffer, self, X) for a HOST matrix with one channel per column (Julia's layout; x_stride / y_stride in samples): the columns
 * are split over the shards, one host thread per shard runs its copies and kernels (mrhip_filt_host), all devices at once; blocks
 * until y holds every channel's outputs.  What the Julia shim's filt(::ShardedFIRFilter, X::Matrix) calls. */
int mrhip_sharded_filt_host(mrhip_sharded *s, const void *x, int64_t x_len, int64_t x_stride, void *y, int64_t y_capacity, int64_t y_stride,
                            int64_t *n_written);
/* the final gather: rows [start_i, start_i + count_i) of a (nchannels x n_out) result at `dst` on `dst_device` (row stride dst_stride
 * samples) from y[i] on shard i's device, asynchronous on the shards' streams behind their filter kernels */
int mrhip_sharded_gather(mrhip_sharded *s, const void *const *y, int64_t n_out, const int64_t *y_stride, void *dst, int64_t dst_stride, int dst_device);
/* ordering against a CALLER's stream on shard i's device (the shards run on private streams): _wait_stream puts the shard's stream
 * behind what `stream` holds so far (inputs the caller's kernels still write, buffers its allocator handed out in stream order),
 * _signal_stream puts `stream` behind what the shard's stream holds so far (the caller's later kernels see the outputs).  No
 * reference counterpart: the reference is synchronous (src/Filters.jl:577-587). */
int mrhip_sharded_wait_stream(mrhip_sharded *s, int shard, void *stream);
int mrhip_sharded_signal_stream(mrhip_sharded *s, int shard, void *stream);
int mrhip_sharded_synchronize(mrhip_sharded *s);

/* replaces the stateless filt(h, x, ratio), src/Filters.jl:858-861, and
 * filt(h, x, rate, Nphi), :864-867, for host data, one channel: construct, filter once,
 * destroy.  rate <= 0 selects the rational form with num//den; rate > 0 the arbitrary form. */
int mrhip_filt_once(const void *h, int64_t hLen, int tap_dtype, int64_t num, int64_t den, double rate,
                    int64_t Nphi, const void *x, int64_t x_len, int sample_dtype, void *y,
                    int64_t y_capacity, int64_t *n_written, int device);

/* ---- measurement ---------------------------------------------------------------------- */
/* With timing enabled every compute-kernel launch made by mrhip_filt_device on this filter is
 * bracketed by a pair of HIP events recorded on the launch stream (the history-shift kernel and
 * the copies are outside the bracket).  Nothing synchronises until mrhip_timing_read, which waits
 * for the recorded events, returns how many launches were bracketed since the last read and the sum
 * of their durations in milliseconds, and clears the log.  bench.py uses it for roofline.achieved.
 * enabled = n > 1 brackets every n-th compute launch only (the two event records cost a few microseconds of
 * stream time per launch, which a throughput measurement of back-to-back launches would otherwise include).
 * enabled = -n < -1 puts ONE bracket around every n consecutive compute launches: mrhip_timing_read then returns the
 * launches covered by complete groups and the groups' total duration, gaps between the launches included. */
int mrhip_set_timing(mrhip_filter *f, int enabled);
int mrhip_timing_read(mrhip_filter *f, int64_t *n_launches, double *total_ms);
/* name of the device kernel the last filt call dispatched (for profiles / logs) */
const char *mrhip_last_kernel_name(const mrhip_filter *f);
/* How the phase schedule of a FIRArbitrary / FIRFarrow filter -- update(), src/Filters.jl:663-673 and :780-792 -- has
 * been evaluated so far.  info[0..n) receives, as far as n reaches:
 *   [0] 1 if the device evaluation covers this (rate, Nphi), else 0 (the host's serial loop runs)
 *   [1] candidate values per congruence class G/Umin   [2] candidates per segment
 *   [3] period of the accumulator's cycle, 0 if none was found (a cycle makes the schedule a closed form)
 *   [4] outputs scheduled by the host loop             [5] outputs scheduled by the closed form of a cycle
 *   [6] pieces scheduled and verified on the device    [7] pieces whose verification failed (redone by the host loop)
 *   [8] calls that reused the schedule of an identical earlier call (same accumulator, inputDeficit and length)
 * (diagnostics and tests; results never depend on the path taken) */
int mrhip_schedule_info(const mrhip_filter *f, int64_t *info, int n);

#ifdef __cplusplus
}
#endif
#endif /* MULTIRATE_HIP_H */
