// host_abi_asan.cpp -- NOT part of libmultirate_hip.so.  `make asan` links it with host_logic.cpp and design.cpp (the library's
// data-independent host logic: taps2pfb, outputlength, the closed form of the rational state recurrence, the FIRArbitrary / FIRFarrow
// phase schedule in both mod() forms, polyfit, FIR design) into libmultirate_host_asan.so, built by gcc with
// -fsanitize=address,undefined: tests/test_sanitizers.py drives every entry below from a child process and compares the results
// with the oracle's.  (GPU sanitizers are not available on this pool; this is the CPU-side leg SURVEY.md section 5 asks for.)
#include <cstring>
#include <string>

#include "mrhip_internal.h"

namespace mrhip {
static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
int fail(int code, const std::string &msg) { g_err = msg; return code; }
}  // namespace mrhip

using namespace mrhip;

extern "C" {

const char *mrhip_last_error(void) { return g_err.c_str(); }
int64_t mrhip_taps2pfb(const void *h, int64_t hLen, int tap_dtype, int64_t Nphi, void *pfb)
{
    if (!h || hLen < 1 || Nphi < 1) return -1;
    return taps2pfb(h, hLen, tap_dtype, Nphi, pfb);
}
int64_t mrhip_nextphase(int64_t p, int64_t L, int64_t M) { return nextphase(p, L, M); }
int64_t mrhip_outputlength_ratio(int64_t n, int64_t L, int64_t M, int64_t phi) { return outputlength_ratio(n, L, M, phi); }
int64_t mrhip_inputlength_ratio(int64_t n, int64_t L, int64_t M, int64_t phi) { return inputlength_ratio(n, L, M, phi); }
int mrhip_polyfit(const double *y, int64_t n, int64_t polyorder, double *coef)
{
    if (!y || !coef || polyorder < 0 || polyorder > 64) return fail(MRHIP_ERR_INVALID_ARG, "bad polyfit arguments");
    return polyfit_rows(y, n, static_cast<int>(polyorder), coef) ? MRHIP_OK : fail(MRHIP_ERR_INVALID_ARG, "polynomial fit is rank deficient");
}
// the closed form every rational-family call is planned with (host_logic.cpp: plan_rational); out[5] = n_out, phi0, d0, phi_end, d_end
int mrhip_host_plan_rational(int kind, int64_t L, int64_t M, int64_t phiIdx, int64_t inputDeficit, int64_t xLen, int64_t *out)
{
    const CallPlan p = plan_rational(kind, L, M, phiIdx, inputDeficit, xLen);
    out[0] = p.n_out; out[1] = p.phi0; out[2] = p.d0; out[3] = p.phi_end; out[4] = p.d_end;
    return p.short_input ? 1 : 0;
}
// one call of the FIRArbitrary / FIRFarrow phase schedule (update(), Filters.jl:663-673) by the host's loop: entries into n_idx / acc
// (room for `cap`), state in / out through st[4] = {acc, inputDeficit} -> {acc, inputDeficit, phiIdx, xIdx}; returns the count or -1
int64_t mrhip_host_arbitrary_schedule(double *st, double delta, int64_t Nphi, int64_t xLen, int mod_form, int32_t *n_idx, double *acc, int64_t cap)
{
    ArbState s{st[0], 1, 0.0, 1, static_cast<int64_t>(st[1])};
    std::vector<int32_t> n;
    std::vector<double> a;
    const int64_t cnt = run_arbitrary_schedule(s, delta, Nphi, xLen, &n, &a, mod_form);
    if (cnt > cap) return -1;
    if (cnt > 0) { std::memcpy(n_idx, n.data(), static_cast<size_t>(cnt) * sizeof(int32_t)); std::memcpy(acc, a.data(), static_cast<size_t>(cnt) * sizeof(double)); }
    st[0] = s.acc; st[1] = static_cast<double>(s.inputDeficit); st[2] = static_cast<double>(s.phiIdx); st[3] = static_cast<double>(s.xIdx);
    // ... and the resumable form in ragged pieces must give the same entries (what filt! pipelines with its launches)
    ArbState s2{a.empty() ? st[0] : a[0], 1, 0.0, n.empty() ? 1 : n[0], 1};
    if (cnt > 0) {
        std::vector<int32_t> n2(static_cast<size_t>(cnt) + 8);
        std::vector<double> a2(static_cast<size_t>(cnt) + 8);
        int64_t k = 0;
        bool done = false;
        int64_t piece = 7;
        while (!done) {
            if (k + piece > static_cast<int64_t>(n2.size())) piece = static_cast<int64_t>(n2.size()) - k;
            if (piece <= 0) return -2;
            k += run_arbitrary_schedule_piece(s2, delta, Nphi, xLen, n2.data() + k, a2.data() + k, piece, &done, mod_form);
            piece = piece * 3 + 1;
        }
        if (k != cnt || std::memcmp(n2.data(), n.data(), static_cast<size_t>(cnt) * 4) != 0 || std::memcmp(a2.data(), a.data(), static_cast<size_t>(cnt) * 8) != 0) return -3;
        if (s2.acc != s.acc || s2.inputDeficit != s.inputDeficit) return -4;
    }
    return cnt;
}

}  // extern "C"
