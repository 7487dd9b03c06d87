// design.cpp -- windowed-sinc FIR design behind the C ABI: the reference's src/FIRDesign.jl:7-95
// (FIRResponse :7, kaiserlength :18-33, firprototype :47-66, both firdes methods :76-95).
//
// Host only, O(taps), Float64 like the reference (its comprehensions produce Vector{Float64}).  The Kaiser window
// takes beta directly, w[k] = I0(beta*sqrt(1 - (2k/(n-1) - 1)^2)) / I0(beta) (the reference's in-tree
// src/Window.jl:53-58 convention; the window it really calls lives in the un-vendored DSP.jl of 2014, so tap VALUES
// are "parity unpinned" against a historical run -- SURVEY.md 8c; the engine and the oracle are always fed the same taps).
#include <cmath>
#include <cstdint>
#include <vector>

#include "mrhip_internal.h"

namespace mrhip {
namespace {

// modified Bessel function of the first kind, order 0: power series sum_k ((x/2)^k / k!)^2 (all terms positive,
// converges in < 40 terms for the beta range of window design; relative error a few ulp)
double bessel_i0(double x)
{
    const double q = 0.25 * x * x;
    double term = 1.0, sum = 1.0;
    for (int k = 1; k < 500; ++k) {
        term *= q / (static_cast<double>(k) * static_cast<double>(k));
        sum += term;
        if (term < sum * 1e-17) break;
    }
    return sum;
}

// sinc(x) = sin(pi x) / (pi x), sinc(0) = 1 (Julia Base.sinc)
double sinc(double x)
{
    if (x == 0.0) return 1.0;
    const double px = M_PI * x;
    return std::sin(px) / px;
}

}  // namespace
}  // namespace mrhip

using namespace mrhip;

extern "C" {

int mrhip_kaiser(int64_t n, double beta, double *out)
{
    if (n < 1 || !out) return fail(MRHIP_ERR_INVALID_ARG, "kaiser: n must be >= 1 and out non-NULL");
    if (n == 1) { out[0] = 1.0; return MRHIP_OK; }
    const double i0b = bessel_i0(beta);
    for (int64_t k = 0; k < n; ++k) {
        const double r = 2.0 * static_cast<double>(k) / static_cast<double>(n - 1) - 1.0;
        const double a = 1.0 - r * r;
        out[k] = bessel_i0(beta * std::sqrt(a > 0.0 ? a : 0.0)) / i0b;
    }
    return MRHIP_OK;
}

int mrhip_kaiserlength(double transition, double attenuation, double samplerate, int64_t *numtaps, double *beta)
{
    if (!numtaps || !beta) return fail(MRHIP_ERR_INVALID_ARG, "NULL argument");
    if (!(transition > 0.0) || !(samplerate > 0.0)) return fail(MRHIP_ERR_INVALID_ARG, "transition and samplerate must be > 0");
    transition = transition / samplerate;                                                   // FIRDesign.jl:20
    *numtaps = static_cast<int64_t>(std::ceil((attenuation - 7.95) / (2.0 * M_PI * 2.285 * transition)));   // :21
    if (attenuation > 50.0) *beta = 0.1102 * (attenuation - 8.7);                           // :23-24
    else if (attenuation >= 21.0) *beta = 0.5842 * std::pow(attenuation - 21.0, 0.4) + 0.07886 * (attenuation - 21.0);   // :25-26
    else *beta = 0.0;                                                                       // :28
    return MRHIP_OK;
}

int64_t mrhip_firprototype(int64_t numtaps, const double *F, int nF, int response, double *out)
{
    if (numtaps < 1 || !F) { set_error("firprototype: numtaps must be >= 1 and F non-NULL"); return -1; }
    const bool two = response == MRHIP_BANDPASS || response == MRHIP_BANDSTOP;
    if (response < MRHIP_LOWPASS || response > MRHIP_BANDSTOP) { set_error("Not a valid FIR_TYPE"); return -1; }   // :62
    if (nF < (two ? 2 : 1)) { set_error("firprototype: band responses take two cutoff frequencies"); return -1; }
    int64_t Mo = numtaps - 1;                                                                // :48
    if (response == MRHIP_HIGHPASS && (Mo & 1)) Mo += 1;                                     // :55 (type 1: even order)
    if (!out) return Mo + 1;
    const double half = static_cast<double>(Mo) / 2.0;
    for (int64_t n = 0; n <= Mo; ++n) {
        const double t = static_cast<double>(n) - half;
        switch (response) {
        case MRHIP_LOWPASS:  out[n] = 2.0 * F[0] * sinc(2.0 * F[0] * t); break;                                           // :50
        case MRHIP_BANDPASS: out[n] = 2.0 * (F[0] * sinc(2.0 * F[0] * t) - F[1] * sinc(2.0 * F[1] * t)); break;           // :52
        case MRHIP_HIGHPASS: out[n] = sinc(t) - 2.0 * F[0] * sinc(2.0 * F[0] * t); break;                                 // :56
        default:             out[n] = 2.0 * (F[1] * sinc(2.0 * F[1] * t) - F[0] * sinc(2.0 * F[0] * t)); break;           // :58
        }
    }
    return Mo + 1;
}

int64_t mrhip_firdes(int64_t numtaps, const double *cutoff, int ncutoff, int response, double samplerate, double beta,
                     const double *window, double *out)
{
    if (!(samplerate > 0.0) || !cutoff || ncutoff < 1 || ncutoff > 2) { set_error("firdes: bad cutoff / samplerate"); return -1; }
    double F[2] = {cutoff[0] / samplerate, ncutoff > 1 ? cutoff[1] / samplerate : 0.0};      // :78
    const int64_t n = mrhip_firprototype(numtaps, F, ncutoff, response, nullptr);            // :79-80
    if (n < 0 || !out) return n;
    if (mrhip_firprototype(numtaps, F, ncutoff, response, out) != n) return -1;
    if (window) {                                                                            // :85 prototype .* windowfunction(numtaps)
        for (int64_t i = 0; i < n; ++i) out[i] *= window[i];
    } else {                                                                                 // :83 prototype .* kaiser(numtaps, beta)
        std::vector<double> w(static_cast<size_t>(n));
        if (mrhip_kaiser(n, beta, w.data()) != MRHIP_OK) return -1;
        for (int64_t i = 0; i < n; ++i) out[i] *= w[static_cast<size_t>(i)];
    }
    return n;
}

int64_t mrhip_firdes_kaiser(const double *cutoff, int ncutoff, double transitionwidth, double stopbandAttenuation, int response,
                            double samplerate, double *out)
{
    int64_t numtaps = 0;
    double beta = 0.0;
    if (mrhip_kaiserlength(transitionwidth, stopbandAttenuation, samplerate, &numtaps, &beta) != MRHIP_OK) return -1;   // :92
    if (numtaps < 1) { set_error("firdes: kaiserlength gives no taps for this transition width / attenuation"); return -1; }
    return mrhip_firdes(numtaps, cutoff, ncutoff, response, samplerate, beta, nullptr, out);                            // :93
}

}  // extern "C"
