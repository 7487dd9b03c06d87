"""The C oracle (oracle/) against a SECOND restatement of the reference's hot path that shares nothing with it
(tests/strict_restatement.py: the Julia source loop by loop in NumPy scalar arithmetic, nextphase stepping instead of the
closed form, 1-based seam indices as written): outputs, end state and history bit for bit on 240 small cases -- every
kernel kind, Float32 / Float64 taps, the four sample types, whole / two-chunk / one-sample-at-a-time / ragged chunkings
(the reference's own three ways of feeding, test/runtests.jl:49-51, 81-83), chunks shorter than the history, signed zeros.

What this buys: the bit-level ORDER of operations (oldest sample first over the logical window, first product
initialises, the Vector seam variant's start from zero, the Float64 combine of FIRArbitrary) no longer rests on one
author's single formulation of support.jl:5-55."""
from fractions import Fraction

import numpy as np
import pytest

from conftest import assert_bit_equal
from strict_restatement import Restated


def _rand(rng, n, tx):
    x = rng.standard_normal(n)
    if np.dtype(tx).kind == "c":
        x = x + 1j * rng.standard_normal(n)
    return x.astype(tx)


def _chunkings(rng, n):
    pivot = int(rng.integers(1, max(n - 1, 2)))
    cuts = sorted(set(rng.integers(0, n + 1, size=4).tolist()))
    return {"whole": [n], "pivot": [pivot, n - pivot], "ones": [1] * n, "ragged": np.diff([0] + cuts + [n]).tolist()}


@pytest.mark.parametrize("seed", range(6))
def test_oracle_equals_the_second_restatement(O, seed):
    rng = np.random.default_rng(4200 + seed)
    ncases = 0
    for _ in range(10):
        L, M = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        th = [np.float32, np.float64][int(rng.integers(0, 2))]
        tx = [np.float32, np.float64, np.complex64, np.complex128][int(rng.integers(0, 4))]
        hLen = int(rng.integers(1 if L > 1 or M > 1 else 2, 40))
        if Fraction(L, M).numerator == 1 and hLen < 2:
            hLen = 2                                      # (hLen == 1 single-rate / decimator: the reference reads b[1] of an empty history)
        h = rng.standard_normal(hLen).astype(th)
        n = int(rng.integers(20, 61))
        x = _rand(rng, n, tx)
        if seed % 2:                                      # signed zeros and a run of them: the seam's start-from-zero is visible only there
            x[rng.integers(0, n, size=n // 3)] = -0.0
            h[rng.integers(0, hLen, size=max(hLen // 4, 1))] = -0.0
        arb = rng.random() < 0.3
        ratio = float(rng.uniform(0.3, 3.0)) if arb else Fraction(L, M)
        Nphi = int(rng.integers(1, 9))
        for name, sizes in _chunkings(rng, n).items():
            fo = O.FIRFilter(h, ratio, Nphi, tx=tx)
            fr = Restated(h, ratio, Nphi, tx=tx)
            pos = 0
            for s in sizes:
                yo, yr = fo.filt(x[pos:pos + s]), fr.filt(x[pos:pos + s])
                assert_bit_equal(yo, yr, f"{ratio} Nphi={Nphi} th={th.__name__} tx={np.dtype(tx)} hLen={hLen} {name} at {pos}")
                pos += s
            so = fo.state
            if fr.kind in ("rational", "decimator", "arbitrary"):
                assert so.inputDeficit == fr.inputDeficit
            if fr.kind == "rational":
                assert so.phiIdx == fr.phiIdx
            if fr.kind == "arbitrary":
                assert (so.phiAccumulator, so.phiIdx, so.alpha) == (fr.acc, fr.phiIdx, fr.alpha)
            assert_bit_equal(np.asarray(fo.history), fr.history_array(), "history")
            ncases += 1
    assert ncases == 40


def test_negative_zero_seam_quirk_in_both(O):
    """support.jl:46: the Vector seam variant starts from zero(...): an all-(-0) sum comes out +0 inside the first hLen
    samples of a call and -0 past them -- in both restatements."""
    h = np.ones(3, dtype=np.float32)
    x = np.full(8, -0.0, dtype=np.float32)
    for ratio in (Fraction(1, 1), Fraction(1, 2)):
        yo, yr = O.FIRFilter(h, ratio, tx=np.float32).filt(x), Restated(h, ratio, tx=np.float32).filt(x)
        assert_bit_equal(yo, yr, str(ratio))
        assert np.signbit(yo).any() and (~np.signbit(yo)).any()
