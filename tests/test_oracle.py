"""CPU tests of the oracle: pinned against every deterministic known answer the reference holds,
cross-checked against the reference's own naive test model, and frozen by the golden vectors."""
import math
from fractions import Fraction

import numpy as np
import pytest

from conftest import assert_bit_equal
from oracle import naive as N


def test_taps2pfb_known_answer(O, known_answers):
    ka = known_answers["taps2pfb"]                       # src/Filters.jl:276-282
    pfb = O.taps2pfb(np.array(ka["h"], dtype=np.float64), ka["Nphi"])
    assert np.array_equal(pfb, np.array(ka["pfb_rows"], dtype=np.float64))


def test_readme_streaming_known_answer(O, known_answers):
    ka = known_answers["readme_stream"]                  # README.md:58-141
    x = np.arange(ka["x_first"], ka["x_last"] + 1, dtype=np.float64)
    h = np.array(ka["h"], dtype=np.float64)
    f = O.FIRFilter(h, Fraction(*ka["ratio"]), tx=np.float64)
    assert np.array_equal(f.taps(), np.array(ka["pfb_rows"], dtype=np.float64))
    st = f.state
    for k, v in ka["kernel_fields"].items():
        assert getattr(st, k) == v
    ys = []
    for (a, b), want in zip(ka["chunks"], ka["y"]):
        y = f.filt(x[a - 1:b])
        assert y.tolist() == want
        ys.append(y)
    y = np.concatenate(ys)
    assert len(y) == ka["total_outputs"]
    assert np.sum(y - O.filt(h, x, Fraction(*ka["ratio"]))) == ka["sum_diff_vs_stateless"]


def test_notebook_farrow_output_length(O, known_answers):
    """doc/Polyphase Filtering Explained.ipynb, last code cell: filt(FIRFilter(h, float64(pi), 32, 4), x) over 40 samples
    returns 126 outputs (the cell's stored output is the 126-element time vector, last value 125/pi - 5).  The count is
    a property of the phase recurrence alone, which FIRFarrow and FIRArbitrary share."""
    ka = known_answers["notebook_farrow"]
    rate = math.pi
    t = np.arange(ka["Nx"])
    x = np.cos(2 * np.pi * ka["xf1"] * t) + 0.5 * np.sin(2 * np.pi * ka["xf2"] * t * np.pi)
    hLen = ka["tapsPerPhi"] * ka["Nphi"]
    k = np.arange(hLen)
    F = min(0.45 / ka["Nphi"], rate / ka["Nphi"])
    h = 2 * F * np.sinc(2 * F * (k - (hLen - 1) / 2)) * np.kaiser(hLen, 7.8562) * ka["Nphi"]
    y_farrow = O.FIRFilter(h, rate, ka["Nphi"], tx=np.float64, polyorder=ka["polyorder"]).filt(x)
    y_arb = O.FIRFilter(h, rate, ka["Nphi"], tx=np.float64).filt(x)
    assert len(y_farrow) == ka["len_y"] and len(y_arb) == ka["len_y"]
    ty = np.arange(len(y_farrow)) / rate - ka["tapsPerPhi"] / 2
    assert ty[0] == ka["ty_first"] and round(ty[-1], 4) == ka["ty_last_printed"]
    # the two kernels resample the same signal: they agree to the accuracy of the degree-4 fit
    assert np.abs(y_farrow - y_arb).max() < 2e-2 * np.abs(y_arb).max()


def test_nextphase_table(O, known_answers):
    lo, hi = known_answers["nextphase"]["L_range"]       # test/runtests.jl:423-438
    for L0 in range(lo, hi + 1):
        for M0 in range(lo, hi + 1):
            g = math.gcd(L0, M0)
            L, M = L0 // g, M0 // g
            ref = np.tile(np.arange(1, L + 1), M)[::M].tolist()
            got = [1]
            for _ in range(2, L + 1):
                got.append(O.nextphase(got[-1], L, M))
            assert got == ref


def test_shiftin(O):
    a = np.arange(5, dtype=np.float64)
    assert O.shiftin(a, np.array([10., 11.])).tolist() == [2, 3, 4, 10, 11]
    assert O.shiftin(a, np.arange(20., 30.)).tolist() == [25, 26, 27, 28, 29]
    assert O.shiftin(a, np.array([])).tolist() == a.tolist()


def _rand(rng, n, tx):
    if np.issubdtype(tx, np.complexfloating):
        return (rng.random(n) + 1j * rng.random(n)).astype(tx)
    return rng.random(n).astype(tx)


@pytest.mark.parametrize("seed", range(4))
def test_rational_family_vs_naive_model(O, seed):
    """The reference's own test recipe (runtests.jl:60,123,190,270): zero-stuff, FIR, keep every
    M-th.  Tolerance: far inside Julia-0.3 isapprox (rtol cbrt(eps)); we require 8*T*eps."""
    rng = np.random.default_rng(100 + seed)
    for _ in range(60):
        L, M = int(rng.integers(1, 33)), int(rng.integers(1, 33))
        th = rng.choice([np.float32, np.float64])
        tx = rng.choice([np.float32, np.float64, np.complex64, np.complex128])
        h = rng.random(int(rng.integers(16, 129))).astype(th)
        n = int(rng.integers(200, 301))
        x = _rand(rng, n, tx)
        fr = Fraction(L, M)
        ref = N.naive_rational(h, x, fr.numerator, fr.denominator)
        y = O.filt(h, x, fr)
        assert len(y) == len(ref)
        eps = np.finfo(y.real.dtype).eps
        scale = np.sum(np.abs(h.astype(np.float64))) * 1.5
        assert np.max(np.abs(y - ref)) <= 8 * len(h) * eps * scale
        # stateful: pivot split and one-sample pieces are bit-identical to the stateless run
        f = O.FIRFilter(h, fr, tx=tx)
        p = min(int(rng.integers(50, 151)), n // 4)
        assert_bit_equal(np.concatenate([f.filt(x[:p]), f.filt(x[p:])]), y, "pivot")
        f = O.FIRFilter(h, fr, tx=tx)
        assert_bit_equal(np.concatenate([f.filt(x[i:i + 1]) for i in range(n)]), y, "piecewise")


def test_arbitrary_vs_naive_model(O):
    """src/NaiveResamplers.jl:24-49.  Tolerance: |h[end]|*max|x| + float
    noise -- the reference's dh[end] = 0 (Filters.jl:106) is the only systematic difference, and it
    only shows at phiIdx == Nphi."""
    k = np.arange(1024)
    h = (2 * (0.45 / 32) * np.sinc(2 * (0.45 / 32) * (k - 1023 / 2)) * np.kaiser(1024, 7.8562)) * 32
    rng = np.random.default_rng(5)
    for tx, tol in ((np.float64, 1e-12), (np.float32, 2e-6), (np.complex64, 2e-6)):
        for rate in (math.pi / 3, 0.731, 1.9):
            x = _rand(rng, 3000, tx)
            hh = h.astype(np.float64 if tx == np.float64 else np.float32)
            y, sc = O.FIRFilter(hh, float(rate), 32, tx=tx).filt(x, return_schedule=True)
            ref = N.naive_arbitrary(h, x, float(rate), 32)
            n = min(len(y), len(ref))
            assert abs(len(y) - len(ref)) <= 1
            d = np.abs(y[:n] - ref[:n])
            assert d.max() <= abs(h[-1]) * 1.5 + tol
            assert d[sc["phiIdx"][:n] != 32].max() <= tol
            f = O.FIRFilter(hh, float(rate), 32, tx=tx)
            assert_bit_equal(np.concatenate([f.filt(x[i:i + 7]) for i in range(0, 3000, 7)]), y, "chunked")


def test_f32_ulp_histogram_vs_extended(O):
    """How far the stated-order Float32 sum sits from the exactly rounded result on the C1 shape
    (SURVEY.md hard part 1): documents the +-ULP band a reassociating @simd build may land in."""
    k = np.arange(3528)
    h = (2 * (0.5 / 147) * np.sinc(2 * (0.5 / 147) * (k - 3527 / 2)) * np.kaiser(3528, 7.8562)).astype(np.float32)
    x = np.random.default_rng(1).random(4000).astype(np.float32)
    y = O.filt(h, x, Fraction(147, 160))
    ref = N.naive_rational(h, x, 147, 160)
    scale = N.naive_rational(np.abs(h), np.abs(x), 147, 160)[: len(y)]   # sum |tap*sample| per output
    err = np.abs(y.astype(np.longdouble) - ref[: len(y)])
    # forward error bound of a length-24 recursive sum: (T+1) * eps/2 * sum|terms|
    assert np.all(err <= 25 * (np.finfo(np.float32).eps / 2) * np.maximum(scale, 1e-30) * 1.01)


def test_short_inputs_and_empty(O):
    h = np.arange(1, 40, dtype=np.float64)
    f = O.FIRFilter(h, Fraction(1, 8), tx=np.float64)
    assert len(f.filt(np.ones(1))) == 1            # first input always produces an output
    assert f.state.inputDeficit == 8
    assert len(f.filt(np.ones(3))) == 0            # short input: no output, deficit reduced
    assert f.state.inputDeficit == 5
    assert len(f.filt(np.ones(0))) == 0
    assert f.state.inputDeficit == 5
    assert len(f.filt(np.ones(5))) == 1
    f = O.FIRFilter(np.ones(1), Fraction(1, 1), tx=np.float32)   # hLen == 1 (reference throws)
    assert f.filt(np.array([2., 3.], dtype=np.float32)).tolist() == [2., 3.]


def test_golden_vectors_frozen(O, golden):
    """The committed fixtures are exactly what the oracle produces today."""
    meta, data = golden
    for m in meta:
        k = m["id"]
        h, x, sizes = data[k + "_h"], data[k + "_x"], data[k + "_sizes"]
        ratio = Fraction(m["L"], m["M"]) if m["kind"] == "rational" else float(m["rate"])
        f = O.FIRFilter(h, ratio, m.get("Nphi", 32), tx=x.dtype)
        outs, pos = [], 0
        for s in sizes:
            outs.append(f.filt(x[pos:pos + s]))
            pos += s
        assert_bit_equal(np.concatenate(outs), data[k + "_y"], k)
        assert [len(o) for o in outs] == data[k + "_counts"].tolist()
        st = f.state
        assert [st.phiIdx, st.inputDeficit] == data[k + "_state"].tolist()
        assert_bit_equal(f.history, data[k + "_hist"], k + " history")


def test_farrow_oracle_properties(O, pkg):
    """FIRFarrow restatement (src/Filters.jl:123-147, 764-846).  The reference holds no asserted test for it
    (test/farrowtest.jl only prints), so the oracle is checked through properties: polyfit is the
    least-squares fit numpy.polyfit also finds; polyval is Horner; chunked == unchunked bit for bit;
    counts and state follow FIRArbitrary's (same recurrence); a high-order Farrow filter approaches
    FIRArbitrary's output (both interpolate the same filter bank)."""
    rng = np.random.default_rng(5)
    y = rng.standard_normal(32)
    for order in (0, 1, 3, 4, 7):
        c = O.polyfit(y, order)
        ref = np.polyfit(np.arange(1, 33), y, order)[::-1]
        # the Vandermonde matrix over 1..32 is ill-conditioned at high order: compare the FITTED VALUES
        A = np.vander(np.arange(1.0, 33.0), order + 1, increasing=True)
        assert np.abs(A @ c - A @ ref).max() <= 1e-5 * np.abs(y).max()      # noise data: error ~ cond^2 * eps * |residual|
        x0 = 3.25
        assert abs(O.polyval(c, x0) - np.polyval(c[::-1], x0)) <= 1e-12 * max(1.0, abs(np.polyval(c[::-1], x0)))
        # the library's own Householder fit agrees with LAPACK's to rounding
        assert np.abs(A @ pkg.polyfit(y, order) - A @ c).max() <= 1e-6 * np.abs(y).max()
    h = (pkg.firdes(32 * 32, 0.45 / 32, beta=7.8562) * 32)
    rate = float(np.pi / 3)
    for th, tx in ((np.float64, np.float64), (np.float32, np.float32), (np.float32, np.complex64), (np.float64, np.float32)):
        x = _rand(rng, 3000, tx)
        ht = h.astype(th)
        fa = O.FIRFilter(ht, rate, 32, tx=tx)
        ff = O.FIRFilter(ht, rate, 32, tx=tx, polyorder=6)
        ya, yf = fa.filt(x), ff.filt(x)
        assert len(ya) == len(yf) and yf.dtype == ya.dtype
        assert np.abs(ya - yf).max() < 2e-3 * np.abs(ya).max()
        assert ff.state.inputDeficit == fa.state.inputDeficit and ff.state.phiAccumulator == fa.state.phiAccumulator
        f2 = O.FIRFilter(ht, rate, 32, tx=tx, polyorder=6)
        yc = np.concatenate([f2.filt(x[i:i + 7]) for i in range(0, len(x), 7)])
        assert np.array_equal(yc, yf)
        # initial taps are the polynomials at phase 1.0 (Filters.jl:142-146)
        t1 = np.array([O.polyval(r, 1.0) for r in ff.pnfb]).astype(th)
        f3 = O.FIRFilter(ht, rate, 32, tx=tx, polyorder=6)
        assert np.array_equal(f3.current_taps(), t1)
    with pytest.raises(ValueError):
        O.FIRFilter(h, -0.5, 32, polyorder=3)


def _farrow_golden():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_farrow_v1.npz"))


def test_farrow_golden_vectors_frozen(O):
    """The committed FIRFarrow fixtures (tests/golden/make_golden_farrow.py) freeze the oracle's bits."""
    g = _farrow_golden()
    for k in g["names"]:
        Nphi, order = (int(v) for v in g[k + "_par"])
        x = g[k + "_x"]
        f = O.FIRFilter(g[k + "_h"], float(g[k + "_rate"]), Nphi, tx=x.dtype, polyorder=order, pnfb=g[k + "_pnfb"])
        outs, pos = [], 0
        for s in g[k + "_sizes"]:
            outs.append(f.filt(x[pos:pos + int(s)]))
            pos += int(s)
        assert_bit_equal(np.concatenate(outs), g[k + "_y"], k)
        assert [len(o) for o in outs] == g[k + "_counts"].tolist()
        assert f.state.inputDeficit == int(g[k + "_state"][0]) and f.state.phiAccumulator == float(g[k + "_acc"])
        assert_bit_equal(f.history, g[k + "_hist"], k + " history")


def test_phase_recurrence_shortened_chain_is_exact():
    """host_logic.cpp evaluates update() (src/Filters.jl:663-669) as  a1 = fl(acc + delta);  acc' = a1 - k*N  with
    k = #{j >= 1 : a1 >= j*N + 1}  instead of  fl(mod(fl(a1 - 1), N) + 1): a Python model of both forms (IEEE doubles)
    must agree bit for bit, step by step, for power-of-two and other N, rates from 1/100 to 30.  (The C++ itself is
    checked on the GPU box against the oracle: test_arbitrary_phase_recurrence_many_rates.)"""
    import random
    import struct

    def ref_step(acc, x, delta, N):
        acc = acc + delta
        if acc > N:
            am1 = acc - 1.0
            x += int(math.floor(am1 / N))
            acc = math.fmod(am1, N) + 1.0
        return acc, x

    def new_step(acc, x, delta, N, pow2):
        a1 = acc + delta
        if a1 < 4.0 * N + 1.0:
            w1, w2, w3 = a1 >= N + 1.0, a1 >= 2.0 * N + 1.0, a1 >= 3.0 * N + 1.0
            n = a1 - 3.0 * N if w3 else (a1 - 2.0 * N if w2 else (a1 - N if w1 else a1))
            if pow2:
                x += int(w1) + int(w2) + int(w3)
            elif a1 > N:
                x += int((a1 - 1.0) / N)
            return n, x
        am1 = a1 - 1.0
        return math.fmod(am1, N) + 1.0, x + int(math.floor(am1 / N))

    rnd = random.Random(1)
    for trial in range(60):
        nphi = rnd.choice([32, 32, 16, 64, 7, 10, 33, 100, 3, 1, 2, 128])
        rate = rnd.choice([math.pi / 3, rnd.uniform(0.05, 8.0), rnd.uniform(0.9, 1.1), 10 ** rnd.uniform(-2, 1.5), 1.0, 0.5, 2.0, 1 / 3])
        delta, N = nphi / rate, float(nphi)
        a = b = 1.0
        xa = xb = 1
        for i in range(4000):
            a, xa = ref_step(a, xa, delta, N)
            b, xb = new_step(b, xb, delta, N, (nphi & (nphi - 1)) == 0)
            assert struct.pack("d", a) == struct.pack("d", b) and xa == xb, (nphi, rate, i)


def test_mod_form_of_julia_0_3_quantified(O):
    """update(::FIRArbitrary) wraps the accumulator with mod(acc - 1, N𝜙) (src/Filters.jl:668).  The oracle takes it as the
    exact remainder (Julia >= 0.4); older Base versions computed rem(y + rem(x, y), y), which rounds once more when
    y + rem(x, y) is not representable.  The reference is Julia-0.3 code, so: how far apart are the two?  A one-tap-per-phase
    filter with h = 0, 1, 2, ... fed with ones outputs acc - 1 exactly (yLower = 𝜙Idx - 1, yUpper = 1, combine in Float64),
    i.e. the accumulator of every output.

    Result (also DESIGN.md 3): for a power-of-two N𝜙 -- BASELINE config 4 has N𝜙 = 32 -- the two forms are IDENTICAL over
    1e7 outputs: rem(x, y) is then a multiple of ulp(acc + Δ) >= ulp(N𝜙 + rem), so the extra sum is exact.  For other N𝜙
    the older form drifts by one rounding of that sum per wrap in which the phase stays in (N𝜙, N𝜙 + 1)."""
    def accs(rate, Nphi, n_out, julia03):
        O.set_mod_form(julia03)
        try:
            h = np.arange(Nphi, dtype=np.float64)
            y = O.FIRFilter(h, rate, Nphi, tx=np.float64).filt(np.ones(int(n_out / rate)))
        finally:
            O.set_mod_form(False)
        return y

    rate = float(np.pi / 3)
    a, b = accs(rate, 32, 10_000_000, False), accs(rate, 32, 10_000_000, True)
    assert len(a) == len(b) >= 9_999_000
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), "config 4 (N𝜙 = 32): the two mod forms must agree bit for bit"
    for r, N in ((0.5, 64), (2.9, 8), (1 / 2.123456789, 32)):          # other powers of two, rates on either side of 1
        a, b = accs(r, N, 1_000_000, False), accs(r, N, 1_000_000, True)
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), (r, N)
    report = []
    for r, N in ((rate, 10), (0.37, 12), (1.7, 7), (2.9, 24)):
        a, b = accs(r, N, 1_000_000, False), accs(r, N, 1_000_000, True)
        m = min(len(a), len(b))
        nd = int(np.count_nonzero(a[:m] != b[:m]))
        first = int(np.flatnonzero(a[:m] != b[:m])[0]) if nd else -1
        # (phase distance, folded onto the circle: after the first difference the two streams are different streams)
        d = np.abs(a[:m] - b[:m])
        d = np.minimum(d, N - d)
        report.append((r, N, len(a) - len(b), nd, first, float(d.max() / np.spacing(float(N)))))
        assert abs(len(a) - len(b)) <= 1
        assert d.max() <= 1e-6, report[-1]                              # roundings of 1 ulp(2 N𝜙) each: a slow random walk, not a jump
    print("mod forms, (rate, Nphi, count difference, outputs that differ of 1e6, first, max phase distance in ulp(Nphi)):", report)
