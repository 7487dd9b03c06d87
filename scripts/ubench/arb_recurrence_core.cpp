#include <cstdint>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <chrono>
#include <type_traits>
namespace {
struct ArbConsts {
    double delta, N, invN;
    bool n_pow2;
    ArbConsts(double delta_, int64_t Nphi) : delta(delta_), N(static_cast<double>(Nphi)), invN(1.0 / static_cast<double>(Nphi)),
                                             n_pow2((Nphi & (Nphi - 1)) == 0) {}
};

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

}
int main(int argc, char **argv) {
    const int64_t Nphi = argc > 1 ? atoll(argv[1]) : 32; const double rate = argc > 2 ? atof(argv[2]) : 3.14159265358979323846/3; const double delta = Nphi / rate;
    std::vector<int32_t> n(20000000); std::vector<double> a(20000000);
    for (int rep = 0; rep < 3; ++rep) {
        double acc = 1.0; int64_t x = 1; ArbConsts c(delta, Nphi);
        auto t0 = std::chrono::steady_clock::now();
        int64_t cnt = 0;
        while (x <= 10000000) cnt += arb_run(c, acc, x, 10000000, n.data() + cnt, a.data() + cnt, 262144);
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("Nphi=%lld rate=%g: %lld outputs %.2f ns/output acc=%.17g x=%lld\n", (long long)Nphi, rate, (long long)cnt, dt / cnt * 1e9, acc, (long long)x);
    }
}
