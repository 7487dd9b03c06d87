// host_logic.cpp -- data-independent bookkeeping of the filt! hot path.
//
// Everything here depends only on lengths and on the (phase, deficit) state, never on sample
// values, so the host can size buffers, pick grids and advance the stream state without ever
// reading back from the device.
#include <cmath>
#include <cstring>
#include <vector>

#include "mrhip_internal.h"

namespace mrhip {

// Polyphase decomposition of the prototype filter.
// Behaviour to match: src/Filters.jl:284-298 -- tapsPerPhi = ceil(hLen/Nphi); the matrix is
// tapsPerPhi x Nphi, column-major; column p holds h[p], h[p+Nphi], ... REVERSED (so that a dot
// with a forward-running sample window is a convolution), zero padded past hLen.
// Formulation here: element (row r, col c), 0-based, is h[(T-1-r)*Nphi + c].
int64_t taps2pfb(const void *h, int64_t hLen, int th, int64_t Nphi, void *out)
{
    const int64_t T = (hLen + Nphi - 1) / Nphi;
    if (!out) return T;
    const size_t es = dtype_scalar_size(th);
    auto *dst = static_cast<unsigned char *>(out);
    const auto *src = static_cast<const unsigned char *>(h);
    for (int64_t c = 0; c < Nphi; ++c) {
        for (int64_t r = 0; r < T; ++r) {
            const int64_t hi = (T - 1 - r) * Nphi + c;
            unsigned char *d = dst + static_cast<size_t>(c * T + r) * es;
            if (hi < hLen) std::memcpy(d, src + static_cast<size_t>(hi) * es, es);
            else std::memset(d, 0, es);
        }
    }
    return T;
}

// src/Filters.jl:433-439
int64_t nextphase(int64_t phase, int64_t L, int64_t M)
{
    const int64_t next = phase + M % L;
    return next > L ? next - L : next;
}

// src/Filters.jl:352-357.  The reference divides in Float64 and takes iceil; for every length
// that fits a double exactly (< 2^53) that equals the integer ceiling computed here.
int64_t outputlength_ratio(int64_t inputlength, int64_t L, int64_t M, int64_t initialPhi)
{
    const int64_t a = inputlength * L - initialPhi + 1;
    return a >= 0 ? (a + M - 1) / M : -((-a) / M);
}

// src/Filters.jl:396-401
int64_t inputlength_ratio(int64_t outputlength, int64_t L, int64_t M, int64_t initialPhi)
{
    const int64_t a = outputlength * M + initialPhi - 1;
    return a >= 0 ? (a + L - 1) / L : -((-a) / L);
}

// Closed form of the rational loop (src/Filters.jl:558-571): with u_k = (phi0-1) + k*M,
// output k uses phase u_k mod L and input index d0 + u_k div L, and exists while that index
// is <= xLen.  STANDARD and DECIMATOR are L == 1; INTERPOLATOR restarts at (phi, idx) = (1, 1)
// on every call (src/Filters.jl:500-501) and never carries a deficit.
CallPlan plan_rational(int kind, int64_t L, int64_t M, int64_t phiIdx, int64_t inputDeficit, int64_t xLen)
{
    CallPlan p;
    switch (kind) {
    case MRHIP_FIR_STANDARD:
        p.phi0 = 1; p.d0 = 1; p.n_out = xLen; p.phi_end = 1; p.d_end = 1;
        return p;
    case MRHIP_FIR_INTERPOLATOR:
        p.phi0 = 1; p.d0 = 1; p.n_out = L * xLen; p.phi_end = 1; p.d_end = 1;
        return p;
    default:
        break;
    }
    p.phi0 = (kind == MRHIP_FIR_DECIMATOR) ? 1 : phiIdx;
    p.d0 = inputDeficit;
    if (xLen < inputDeficit) {  // not enough input for a single output
        p.short_input = true;
        p.n_out = 0;
        p.phi_end = phiIdx;
        p.d_end = inputDeficit - xLen;
        return p;
    }
    p.n_out = outputlength_ratio(xLen - inputDeficit + 1, L, M, p.phi0);
    const int64_t u_end = (p.phi0 - 1) + p.n_out * M;
    p.phi_end = u_end % L + 1;
    p.d_end = p.d0 + u_end / L - xLen;
    return p;
}

// FIRArbitrary: the phase accumulator is a serial Float64 recurrence whose roundings the
// reference's outputs depend on (src/Filters.jl:663-673), so it is evaluated here, in order,
// with the same IEEE operations, once per call; every channel shares the result.
int64_t run_arbitrary_schedule(ArbState &st, double delta, int64_t Nphi, int64_t xLen,
                               std::vector<int32_t> *n_idx, std::vector<double> *acc_out)
{
    if (xLen < st.inputDeficit) {          // src/Filters.jl:705-709
        st.inputDeficit -= xLen;
        return 0;
    }
    const double N = static_cast<double>(Nphi);
    const bool n_pow2 = (Nphi & (Nphi - 1)) == 0;
    const double invN = 1.0 / N;               // exact when Nphi is a power of two
    double acc = st.acc;
    int64_t xIdx = st.inputDeficit;        // :715
    int64_t count = 0;
    if (n_idx) n_idx->clear();
    if (acc_out) acc_out->clear();
    {   // one allocation up front: about (xLen - deficit + 1) * rate outputs (rate = Nphi / delta)
        const double est = static_cast<double>(xLen - xIdx + 1) * (N / delta) + 16.0;
        if (n_idx && est < 4e9) n_idx->reserve(static_cast<size_t>(est));
        if (acc_out && est < 4e9) acc_out->reserve(static_cast<size_t>(est));
    }
    while (xIdx <= xLen) {                 // :717
        if (n_idx) n_idx->push_back(static_cast<int32_t>(xIdx));
        if (acc_out) acc_out->push_back(acc);
        ++count;
        acc += delta;                      // update(), :664
        if (acc > N) {                     // :666-669
            const double am1 = acc - 1.0;
            // xIdx += ifloor(am1 / N): the reference rounds the quotient before flooring; a power-of-two
            // N makes the division an exact scaling, otherwise a true division is kept for fidelity
            const double qd = n_pow2 ? am1 * invN : am1 / N;
            xIdx += static_cast<int64_t>(std::floor(qd));
            // acc = mod(am1, N) + 1: for positive operands the remainder is exact; am1 - k*N by repeated
            // subtraction is exact at every step (the difference of a multiple of ulp(am1) and an integer
            // that is no larger than am1) and is the same number fmod returns, at a fraction of its cost
            double r = am1;
            if (am1 < 4.0 * N) { while (r >= N) r -= N; }
            else r = std::fmod(am1, N);        // very low rates: many periods per step
            acc = r + 1.0;
        }
    }
    st.acc = acc;
    st.phiIdx = static_cast<int64_t>(std::floor(acc));   // :671
    st.alpha = acc - static_cast<double>(st.phiIdx);     // :672
    st.xIdx = xIdx;
    st.inputDeficit = xIdx - xLen;         // :734
    return count;
}

// The same recurrence, resumable: continues from (st.acc, st.xIdx) and writes at most `max_outputs` schedule
// entries into raw buffers (the caller's pinned staging memory); *done is set once xIdx has passed xLen, at which
// point st holds the call-end state exactly as run_arbitrary_schedule leaves it.  Lets filt! overlap the serial
// host recurrence of one long call with the kernels of the pieces already scheduled.
int64_t run_arbitrary_schedule_piece(ArbState &st, double delta, int64_t Nphi, int64_t xLen, int32_t *n_idx, double *acc_out,
                                     int64_t max_outputs, bool *done)
{
    const double N = static_cast<double>(Nphi);
    const bool n_pow2 = (Nphi & (Nphi - 1)) == 0;
    const double invN = 1.0 / N;
    double acc = st.acc;
    int64_t xIdx = st.xIdx;
    int64_t count = 0;
    while (xIdx <= xLen && count < max_outputs) {
        n_idx[count] = static_cast<int32_t>(xIdx);
        acc_out[count] = acc;
        ++count;
        acc += delta;
        if (acc > N) {
            const double am1 = acc - 1.0;
            const double qd = n_pow2 ? am1 * invN : am1 / N;
            xIdx += static_cast<int64_t>(std::floor(qd));
            double r = am1;
            if (am1 < 4.0 * N) { while (r >= N) r -= N; }
            else r = std::fmod(am1, N);
            acc = r + 1.0;
        }
    }
    st.acc = acc;
    st.xIdx = xIdx;
    *done = xIdx > xLen;
    if (*done) {
        st.phiIdx = static_cast<int64_t>(std::floor(acc));
        st.alpha = acc - static_cast<double>(st.phiIdx);
        st.inputDeficit = xIdx - xLen;
    }
    return count;
}

// polyfit(y, polyorder), src/support.jl:85-88: A = [x^p for x in 1:n, p = 0:polyorder]; coefficients = A \ y,
// i.e. the least-squares solution Julia computes by a QR factorisation of A in Float64.  Restated as a
// Householder QR (the reference pins no bits here: SURVEY.md 8c -- LAPACK's blocked QR and this one agree
// to rounding).  Returns false when the system is rank deficient (n < polyorder+1).
bool polyfit_rows(const double *y, int64_t n, int polyorder, double *coef)
{
    const int m = polyorder + 1;
    if (n < m || m < 1) return false;
    std::vector<double> A(static_cast<size_t>(n) * m), b(y, y + n);
    for (int64_t r = 0; r < n; ++r) {
        double v = 1.0;
        for (int c = 0; c < m; ++c) { A[static_cast<size_t>(r) * m + c] = v; v *= static_cast<double>(r + 1); }
    }
    for (int c = 0; c < m; ++c) {                     // Householder reflections, column by column
        double norm = 0.0;
        for (int64_t r = c; r < n; ++r) norm += A[static_cast<size_t>(r) * m + c] * A[static_cast<size_t>(r) * m + c];
        norm = std::sqrt(norm);
        if (norm == 0.0) return false;
        const double akk = A[static_cast<size_t>(c) * m + c];
        const double alpha = akk > 0 ? -norm : norm;
        std::vector<double> v(static_cast<size_t>(n - c));
        v[0] = akk - alpha;
        for (int64_t r = c + 1; r < n; ++r) v[static_cast<size_t>(r - c)] = A[static_cast<size_t>(r) * m + c];
        double vnorm2 = 0.0;
        for (double t : v) vnorm2 += t * t;
        if (vnorm2 == 0.0) continue;
        for (int cc = c; cc < m; ++cc) {
            double dot = 0.0;
            for (int64_t r = c; r < n; ++r) dot += v[static_cast<size_t>(r - c)] * A[static_cast<size_t>(r) * m + cc];
            const double f = 2.0 * dot / vnorm2;
            for (int64_t r = c; r < n; ++r) A[static_cast<size_t>(r) * m + cc] -= f * v[static_cast<size_t>(r - c)];
        }
        double dot = 0.0;
        for (int64_t r = c; r < n; ++r) dot += v[static_cast<size_t>(r - c)] * b[static_cast<size_t>(r)];
        const double f = 2.0 * dot / vnorm2;
        for (int64_t r = c; r < n; ++r) b[static_cast<size_t>(r)] -= f * v[static_cast<size_t>(r - c)];
    }
    for (int c = m - 1; c >= 0; --c) {                // back substitution on the upper-triangular R
        double sacc = b[static_cast<size_t>(c)];
        for (int cc = c + 1; cc < m; ++cc) sacc -= A[static_cast<size_t>(c) * m + cc] * coef[cc];
        const double d = A[static_cast<size_t>(c) * m + c];
        if (d == 0.0) return false;
        coef[c] = sacc / d;
    }
    return true;
}

}  // namespace mrhip
