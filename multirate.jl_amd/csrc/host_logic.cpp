// host_logic.cpp -- data-independent bookkeeping of the filt! hot path.
//
// Everything here depends only on lengths and on the (phase, deficit) state, never on sample
// values, so the host can size buffers, pick grids and advance the stream state without ever
// reading back from the device.
#include <algorithm>
#include <cmath>
#include <type_traits>
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

// FIRArbitrary / FIRFarrow: the phase accumulator is a serial Float64 recurrence whose roundings the reference's
// outputs depend on (src/Filters.jl:663-673), so it is evaluated here, in order, once per call; every channel shares
// the result.  One step of update():  acc += delta; if acc > N: xIdx += ifloor((acc-1)/N); acc = mod(acc-1, N) + 1.
//
// The loop is bound by the LATENCY of that dependent Float64 chain (add, -1, -N, +1), so the chain is shortened
// without changing a bit: with a1 = fl(acc + delta) and k = floor((a1-1)/N),
//     fl(fl(fl(a1 - 1) - k*N) + 1) == a1 - k*N   exactly,
// because 1 and N are multiples of ulp(a1) (a1 < 2^53), every intermediate is a multiple of ulp(a1) no larger in
// magnitude than a1, hence representable, hence each of the three operations is exact; and k = #{j >= 1 : a1 >= j*N + 1}
// (a1 - 1 >= j*N  <=>  a1 >= j*N + 1, both sides exact).  So the new accumulator is ONE exact subtraction away from a1,
// selected among a1, a1-N, a1-2N, a1-3N by comparisons.  The index step keeps the reference's expression (quotient
// rounded before the floor; for a power-of-two N it is exactly k).  Rates so low that a step spans four periods or
// more take the reference's expressions one by one (slow_step), OUTSIDE the hot loop: a call in the loop body makes
// the compiler keep the loop constants (and, through reference parameters, the accumulator) in memory -- 3x slower.
// Checked: schedule-dependent outputs and end state == oracle (a plain restatement) bit for bit
// (tests/test_gpu_parity.py::test_arbitrary_phase_recurrence_many_rates; 8e6 steps of a Python model of both forms).
namespace {

struct ArbConsts {
    double delta, N, invN;
    bool n_pow2;
    bool old_mod;      // mrhip_set_mod_form(f, 1) on a N𝜙 that is not a power of two: the plain loop below with mod() as older Julia Base versions computed it
    ArbConsts(double delta_, int64_t Nphi, int mod_form = 0) : delta(delta_), N(static_cast<double>(Nphi)), invN(1.0 / static_cast<double>(Nphi)),
                                             n_pow2((Nphi & (Nphi - 1)) == 0), old_mod(mod_form != 0 && (Nphi & (Nphi - 1)) != 0) {}
};

// update() (src/Filters.jl:663-673) one expression at a time, with mod(x, y) = rem(y + rem(x, y), y): the form Julia's Base used for
// floats before 0.4 -- the reference is Julia-0.3 code and nothing in its tree says which Base it ran on.  Identical to the exact
// remainder whenever N𝜙 is a power of two, so only other N𝜙 come here.
int64_t arb_run_old_mod(const ArbConsts &c, double &acc, int64_t &xIdx, int64_t xLen, int32_t *n_idx, double *acc_out, int64_t max_outputs)
{
    int64_t count = 0;
    while (xIdx <= xLen && count < max_outputs) {          // :717
        if (n_idx) n_idx[count] = static_cast<int32_t>(xIdx);
        if (acc_out) acc_out[count] = acc;
        ++count;
        acc += c.delta;                                      // :664
        if (acc > c.N) {
            const double am1 = acc - 1.0;
            xIdx += static_cast<int64_t>(std::floor(am1 / c.N));              // :667
            acc = std::fmod(c.N + std::fmod(am1, c.N), c.N) + 1.0;           // :668, old Base
        }
    }
    return count;
}

struct SlowStep { double acc; int64_t dx; };
__attribute__((noinline)) SlowStep arb_slow_step(double a1, double N, double invN, bool n_pow2)
{
    const double am1 = a1 - 1.0;
    const double qd = n_pow2 ? am1 * invN : am1 / N;   // exact scaling for a power of two; else the reference's division
    return SlowStep{std::fmod(am1, N) + 1.0,             // exact remainder of positive operands = mod()
                    static_cast<int64_t>(std::floor(qd))};
}

// Runs until xIdx > xLen, `max_outputs` entries were written, or a step needs the slow path (returned in *slow_a1,
// with the accumulator NOT yet advanced).  n_idx / acc_out may be null (count only).
template <bool POW2>
inline int64_t arb_hot_loop(const ArbConsts &c, double &acc_io, int64_t &xIdx_io, int64_t xLen, int32_t *n_idx, double *acc_out,
                            int64_t max_outputs, bool *need_slow, double *slow_a1)
{
    const double delta = c.delta, N = c.N, N2 = 2.0 * N, N3 = 3.0 * N;
    const double Np1 = N + 1.0, N2p1 = N2 + 1.0, N3p1 = N3 + 1.0, fast_limit = 4.0 * N + 1.0;
    double acc = acc_io;
    int64_t xIdx = xIdx_io, count = 0;
    *need_slow = false;
    while (xIdx <= xLen && count < max_outputs) {          // :717
        if (n_idx) n_idx[count] = static_cast<int32_t>(xIdx);
        if (acc_out) acc_out[count] = acc;
        ++count;
        const double a1 = acc + delta;                       // update(), :664
        if (__builtin_expect(!(a1 < fast_limit), 0)) { *need_slow = true; *slow_a1 = a1; break; }
        const double s1 = a1 - N, s2 = a1 - N2, s3 = a1 - N3;
        const bool w1 = a1 >= Np1, w2 = a1 >= N2p1, w3 = a1 >= N3p1;
        double nacc = a1;
        nacc = w1 ? s1 : nacc;
        nacc = w2 ? s2 : nacc;
        nacc = w3 ? s3 : nacc;
        if constexpr (POW2) xIdx += static_cast<int64_t>(w1) + static_cast<int64_t>(w2) + static_cast<int64_t>(w3);
        else if (a1 > N) xIdx += static_cast<int64_t>((a1 - 1.0) / N);   // :667: quotient rounded first; positive, so the cast floors
        acc = nacc;
    }
    acc_io = acc;
    xIdx_io = xIdx;
    return count;
}

// up to max_outputs schedule entries from (acc, xIdx); returns the number written
int64_t arb_run(const ArbConsts &c, double &acc, int64_t &xIdx, int64_t xLen, int32_t *n_idx, double *acc_out, int64_t max_outputs)
{
    if (c.old_mod) return arb_run_old_mod(c, acc, xIdx, xLen, n_idx, acc_out, max_outputs);
    int64_t count = 0;
    while (xIdx <= xLen && count < max_outputs) {
        bool need_slow = false;
        double a1 = 0.0;
        int32_t *pn = n_idx ? n_idx + count : nullptr;
        double *pa = acc_out ? acc_out + count : nullptr;
        count += c.n_pow2 ? arb_hot_loop<true>(c, acc, xIdx, xLen, pn, pa, max_outputs - count, &need_slow, &a1)
                          : arb_hot_loop<false>(c, acc, xIdx, xLen, pn, pa, max_outputs - count, &need_slow, &a1);
        if (need_slow) {                                     // the entry was written; finish its update() the slow way
            const SlowStep r = arb_slow_step(a1, c.N, c.invN, c.n_pow2);
            acc = r.acc;
            xIdx += r.dx;
        }
    }
    return count;
}

}  // namespace

int64_t run_arbitrary_schedule(ArbState &st, double delta, int64_t Nphi, int64_t xLen,
                               std::vector<int32_t> *n_idx, std::vector<double> *acc_out, int mod_form)
{
    if (xLen < st.inputDeficit) {          // src/Filters.jl:705-709
        st.inputDeficit -= xLen;
        return 0;
    }
    const ArbConsts c(delta, Nphi, mod_form);
    double acc = st.acc;
    int64_t xIdx = st.inputDeficit;        // :715
    int64_t count = 0;
    if (n_idx) n_idx->clear();
    if (acc_out) acc_out->clear();
    const int64_t chunk = 1 << 16;
    while (xIdx <= xLen) {
        if (n_idx) n_idx->resize(static_cast<size_t>(count + chunk));
        if (acc_out) acc_out->resize(static_cast<size_t>(count + chunk));
        count += arb_run(c, acc, xIdx, xLen, n_idx ? n_idx->data() + count : nullptr, acc_out ? acc_out->data() + count : nullptr, chunk);
    }
    if (n_idx) n_idx->resize(static_cast<size_t>(count));
    if (acc_out) acc_out->resize(static_cast<size_t>(count));
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
                                     int64_t max_outputs, bool *done, int mod_form)
{
    const ArbConsts c(delta, Nphi, mod_form);
    double acc = st.acc;
    int64_t xIdx = st.xIdx;
    const int64_t count = arb_run(c, acc, xIdx, xLen, n_idx, acc_out, max_outputs);
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

// Un-rounded phase after k steps of the recurrence from acc_p (double-double product, folded onto [1, N+1)): what the
// device schedule (kernels_schedule.hip: sched_anchor) predicts segment starts with.  The host uses it to measure how far
// the true accumulator drifted from it over a run it evaluated itself.
double sched_anchor_host(const SchedPlan &c, double acc_p, double k)
{
    const double hi = k * c.delta;
    const double lo = std::fma(k, c.delta, -hi);
    const double w = std::floor(((acc_p - 1.0) + hi) / c.N);
    double r = ((acc_p - 1.0) + (hi - w * c.N)) + lo;
    if (r < 0.0) r += c.N;
    else if (r >= c.N) r -= c.N;
    return r + 1.0;
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
