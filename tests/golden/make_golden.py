#!/usr/bin/env python3
"""Generates tests/golden/golden_v1.npz from the CPU oracle (seeded; run from the repo root).

The reference cannot be executed (Julia 0.3 source, no Julia in the image) and ships no seeded
vectors, so these fixtures are produced by oracle/multirate_oracle.c AFTER it has been pinned to the
reference's three deterministic known answers (reference_known_answers.json) and cross-checked
against the naive zero-stuff/filter/decimate model (oracle/naive.py).  They freeze the oracle's
bit-exact output so that (a) the oracle cannot drift silently and (b) the GPU path can be compared
on the GPU box without regenerating anything.

Case recipe follows test/runtests.jl:389-421: hLen in 16..128, xLen in 200..300, L and M in 1..32,
Th in {Float32, Float64}, Tx in {Float32, Float64, Complex64, Complex128}; chunkings: whole, pivot
split (runtests.jl:49-51), 1-sample pieces (runtests.jl:81-83) and 7-sample pieces; plus one
C1-shaped case (L=147, M=160, 3528 taps) and FIRArbitrary cases (runtests.jl:334-341 tap recipe).
"""
import json
import os
import sys
from fractions import Fraction

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def firdes(n, F, beta):
    M = n - 1
    k = np.arange(n, dtype=np.float64)
    return 2 * F * np.sinc(2 * F * (k - M / 2)) * np.kaiser(n, beta)


def rand_x(rng, n, tx):
    if np.issubdtype(tx, np.complexfloating):
        return (rng.random(n) + 1j * rng.random(n)).astype(tx)
    return rng.random(n).astype(tx)


def chunk_sizes(name, n, rng):
    if name == "whole":
        return [n]
    if name == "pivot":
        p = min(int(rng.integers(50, 151)), n // 4)
        return [p, n - p]
    if name == "ones":
        return [1] * n
    if name == "sevens":
        return [7] * (n // 7) + ([n % 7] if n % 7 else [])
    raise ValueError(name)


def run_case(h, x, ratio, Nphi, sizes):
    f = O.FIRFilter(h, ratio, Nphi, tx=x.dtype)
    outs, pos = [], 0
    for s in sizes:
        outs.append(f.filt(x[pos:pos + s]))
        pos += s
    st = f.state
    return (np.concatenate(outs), np.array([len(o) for o in outs], dtype=np.int64),
            np.array([st.phiIdx, st.inputDeficit], dtype=np.int64), np.float64(st.phiAccumulator), f.history)


def main():
    rng = np.random.default_rng(20141003)
    arrays, meta = {}, []
    ratios = [(1, 1), (1, 4), (1, 7), (4, 1), (5, 1), (3, 17), (7, 3), (13, 32), (32, 9), (2, 3), (147, 160)]
    combos = [(np.float32, np.float32), (np.float32, np.complex64), (np.float64, np.float32),
              (np.float64, np.float64), (np.float32, np.float64), (np.float64, np.complex128)]
    cid = 0
    for (L, M) in ratios:
        for ci, (th, tx) in enumerate(combos):
            if (L, M) == (147, 160):
                if ci > 1:
                    continue
                h = firdes(24 * 147, 0.5 / 147, 7.8562).astype(th)
                n = 5000
            else:
                if (cid + ci) % 2 and ci > 2:   # keep the file small: thin out the wide-type combos
                    continue
                h = rng.random(int(rng.integers(16, 129))).astype(th)
                n = int(rng.integers(200, 301))
                n -= n % M
            x = rand_x(rng, n, tx)
            for chunking in (("whole", "pivot", "sevens") if n > 1000 else ("whole", "pivot", "ones", "sevens")):
                sizes = chunk_sizes(chunking, n, rng)
                y, counts, state, acc, hist = run_case(h, x, Fraction(L, M), 32, sizes)
                key = f"c{cid:03d}"
                arrays[key + "_h"], arrays[key + "_x"], arrays[key + "_y"] = h, x, y
                arrays[key + "_sizes"] = np.array(sizes, dtype=np.int64)
                arrays[key + "_counts"], arrays[key + "_state"], arrays[key + "_hist"] = counts, state, hist
                meta.append({"id": key, "kind": "rational", "L": L, "M": M, "chunking": chunking,
                             "th": np.dtype(th).name, "tx": np.dtype(tx).name})
                cid += 1
    # FIRArbitrary
    for rate in (float(np.pi / 3), 0.4709, 2.123456789, 1.0):
        for (th, tx) in [(np.float64, np.float64), (np.float32, np.float32), (np.float32, np.complex64)]:
            h = (firdes(32 * 32, 0.45 / 32, 7.8562) * 32).astype(th)
            n = 600
            x = rand_x(rng, n, tx)
            for chunking in ("whole", "pivot", "ones", "sevens"):
                sizes = chunk_sizes(chunking, n, rng)
                y, counts, state, acc, hist = run_case(h, x, rate, 32, sizes)
                key = f"c{cid:03d}"
                arrays[key + "_h"], arrays[key + "_x"], arrays[key + "_y"] = h, x, y
                arrays[key + "_sizes"] = np.array(sizes, dtype=np.int64)
                arrays[key + "_counts"], arrays[key + "_state"], arrays[key + "_hist"] = counts, state, hist
                arrays[key + "_acc"] = np.array([acc])
                meta.append({"id": key, "kind": "arbitrary", "rate": rate, "Nphi": 32, "chunking": chunking,
                             "th": np.dtype(th).name, "tx": np.dtype(tx).name})
                cid += 1
    out = os.path.join(ROOT, "tests", "golden", "golden_v1.npz")
    np.savez_compressed(out, **arrays)
    with open(os.path.join(ROOT, "tests", "golden", "golden_v1.json"), "w") as fh:
        json.dump(meta, fh, indent=0)
    print(f"wrote {out}: {cid} cases, {os.path.getsize(out) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
