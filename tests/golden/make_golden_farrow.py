#!/usr/bin/env python3
"""Generates tests/golden/golden_farrow_v1.npz from the CPU oracle (seeded; run from the repo root).

FIRFarrow (src/Filters.jl:123-147, 764-846).  The reference has no asserted Farrow test
(test/farrowtest.jl prints), cannot be executed here, and pins no bits of its least-squares fit, so each
case stores the polynomial bank it was produced with: the fixture freezes everything downstream of the
fit (Float64 Horner, rounding to the tap type, the Vector dot incl. the seam rule, the phase recurrence).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def firdes(n, F, beta):
    k = np.arange(n, dtype=np.float64)
    return 2 * F * np.sinc(2 * F * (k - (n - 1) / 2)) * np.kaiser(n, beta)


def main():
    rng = np.random.default_rng(20141004)
    arrays, names = {}, []
    cases = [(np.float32, np.float32, 32, 1024, 4, np.pi / 3, "whole"), (np.float64, np.float64, 32, 1000, 6, 0.731, "sevens"),
             (np.float32, np.complex64, 8, 90, 2, 2.5, "ones"), (np.float64, np.float32, 12, 300, 3, 1 / 2.123456789, "pivot"),
             (np.float64, np.complex128, 32, 640, 5, 31.7, "whole"), (np.float32, np.float64, 10, 77, 0, 1.0, "sevens")]
    for i, (th, tx, Nphi, hlen, order, rate, chunking) in enumerate(cases):
        h = (firdes(hlen, 0.45 / Nphi, 7.8562) * Nphi).astype(th)
        n = 400
        x = (rng.random(n) + 1j * rng.random(n)).astype(tx) if np.issubdtype(tx, np.complexfloating) else rng.random(n).astype(tx)
        pn = O.pfb2pnfb(O.taps2pfb(h, Nphi), order)
        sizes = {"whole": [n], "sevens": [7] * (n // 7) + [n % 7], "ones": [1] * n, "pivot": [113, n - 113]}[chunking]
        f = O.FIRFilter(h, float(rate), Nphi, tx=tx, polyorder=order, pnfb=pn)
        outs, pos = [], 0
        for s in sizes:
            outs.append(f.filt(x[pos:pos + s]))
            pos += s
        k = f"c{i}"
        names.append(k)
        arrays[k + "_h"], arrays[k + "_x"], arrays[k + "_pnfb"] = h, x, pn
        arrays[k + "_y"] = np.concatenate(outs)
        arrays[k + "_counts"] = np.array([len(o) for o in outs], dtype=np.int64)
        arrays[k + "_sizes"] = np.array(sizes, dtype=np.int64)
        arrays[k + "_par"] = np.array([Nphi, order], dtype=np.int64)
        arrays[k + "_rate"] = np.float64(rate)
        st = f.state
        arrays[k + "_state"] = np.array([st.inputDeficit], dtype=np.int64)
        arrays[k + "_acc"] = np.float64(st.phiAccumulator)
        arrays[k + "_hist"] = f.history
    arrays["names"] = np.array(names)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "golden_farrow_v1.npz"), **arrays)
    print("wrote", len(names), "cases")


if __name__ == "__main__":
    main()
