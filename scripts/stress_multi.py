#!/usr/bin/env python3
"""Randomised parity stress of mrhip_filt_device_multi (n independent FIRFilter objects -- each its own phase, deficit, history and call length --
in ONE launch): random kinds of the rational family, ratios, sample types, stream counts and ragged call lengths over several rounds, against the
same calls made one by one on a second set of filters, bit for bit, states and histories included.
    python scripts/stress_multi.py [--cases 100] [--seed 1] [--seconds 200]"""
import argparse, os, sys, time
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=100); ap.add_argument("--seed", type=int, default=1); ap.add_argument("--seconds", type=float, default=200.0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t0, bad, done = time.time(), 0, 0
for case in range(a.cases):
    if time.time() - t0 > a.seconds: break
    L, M, T = [(147, 160, 24), (3, 2, 32), (1, 4, 40), (4, 1, 32), (1, 1, 33), (7, 9, 24), (625, 512, 8), (160, 147, 24)][rng.integers(8)]
    tx = [np.float32, np.complex64, np.float64][rng.integers(3)]
    th = np.float64 if tx == np.float64 else np.float32
    h = (rng.standard_normal(T * L) / 4).astype(th)
    n = int(rng.integers(1, 40)); nch = int(rng.choice([1, 1, 2]))
    fs = [pkg.FIRFilter(h, Fraction(L, M)).bind(tx, nch) for _ in range(n)]
    gs = [pkg.FIRFilter(h, Fraction(L, M)).bind(tx, nch) for _ in range(n)]
    ok, why = True, ""
    try:
        for rnd in range(int(rng.integers(1, 4))):
            lens = [int(rng.integers(1, 60_000)) if rng.random() > 0.1 else int(rng.integers(1, 30)) for _ in range(n)]
            xs = []
            for ln in lens:
                x = rng.standard_normal((nch, ln)).astype(np.float32)
                if np.dtype(tx).kind == "c": x = x + 1j * rng.standard_normal(x.shape).astype(np.float32)
                xs.append(torch.from_numpy(x.astype(tx)).cuda())
            ys = pkg.filt_multi(fs, xs)
            for i in range(n):
                want = gs[i].filt(xs[i]).reshape(nch, -1); got = ys[i].reshape(nch, -1)
                if got.shape != want.shape or not torch.equal(got.contiguous().view(torch.uint8), want.contiguous().view(torch.uint8)):
                    ok = False; why = f"round {rnd} stream {i}: {tuple(got.shape)} vs {tuple(want.shape)}"
            if not ok: break
        for i in range(n):
            if ok and ((fs[i].state.phiIdx, fs[i].state.inputDeficit) != (gs[i].state.phiIdx, gs[i].state.inputDeficit) or
                       not np.array_equal(np.asarray(fs[i].history).view(np.uint8), np.asarray(gs[i].history).view(np.uint8))):
                ok = False; why = f"state / history of stream {i}"
    except Exception as e:
        ok = False; why = "exception " + str(e)[:200]
    if not ok:
        bad += 1
        print("MISMATCH", dict(case=case, L=L, M=M, T=T, tx=np.dtype(tx).name, n=n, nch=nch, why=why), flush=True)
    done += 1
    for f in fs + gs: f.close()
print(f"multi-stream stress: cases {done} mismatches {bad} in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
