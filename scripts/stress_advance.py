#!/usr/bin/env python3
"""Randomised stress of mrhip_advance_state (time sharding: enter a stream at sample n without filtering the samples before it) mixed with plain,
asynchronous, reset and set_history calls: the state after advance_state(n) must be the state after filt over n samples (outputs counted the same),
and the stream must go on identically -- every kind, against a second filter that filtered everything.
    python scripts/stress_advance.py [--cases 100] [--seed 1] [--seconds 200]"""
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
    kind = ["rational", "decimator", "arbitrary", "farrow", "arbitrary"][rng.integers(5)]
    po = None
    if kind == "rational": h, ratio = (rng.standard_normal(24 * 7) / 4).astype(np.float32), Fraction(7, 9)
    elif kind == "decimator": h, ratio = rng.standard_normal(40).astype(np.float32), Fraction(1, 5)
    else: h, ratio, po = (pkg.firdes(32 * 6, 0.45 / 32, beta=7.0) * 32).astype(np.float32), float([0.8123, 1 / 2.123456789, 1.25, 3.0, np.pi / 3][rng.integers(5)]), (3 if kind == "farrow" else None)
    nch = int(rng.choice([1, 2]))
    f = pkg.FIRFilter(h, ratio, 32, po).bind(np.float32, nch)
    g = pkg.FIRFilter(h, ratio, 32, po, pnfb=f.pnfb() if po else None).bind(np.float32, nch)
    ok, why = True, ""
    try:
        for step in range(int(rng.integers(2, 7))):
            n = int(rng.integers(1, 900_000)) if rng.random() > 0.15 else int(rng.integers(1, 40))
            x = torch.from_numpy(rng.standard_normal((nch, n)).astype(np.float32)).cuda()
            op = rng.integers(4)
            if op == 0:                                   # skip n samples on f, filter them on g: counts and states must agree
                want = g.filt(x).reshape(nch, -1)
                cnt = f.advance_state(n)
                f.set_history(g.history)                   # (advance_state leaves the history alone: the time-sharded caller supplies it)
                if cnt != want.shape[1]: ok = False; why = f"step {step}: advance_state count {cnt} vs {want.shape[1]}"
            elif op == 1:
                y = torch.empty((nch, f.outputlength_bound(n)), dtype=torch.float64 if po is not None or kind == "arbitrary" else torch.float32, device="cuda")
                y = torch.empty((nch, f.outputlength_bound(n)), dtype={np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}[np.dtype(f.output_dtype)], device="cuda")
                f.filt_into_async(y, x); k = f.sync_state()
                want = g.filt(x).reshape(nch, -1)
                if k != want.shape[1] or not torch.equal(y[:, :k].contiguous().view(torch.uint8), want.contiguous().view(torch.uint8)): ok = False; why = f"step {step}: asynchronous call"
            elif op == 2:
                got = f.filt(x).reshape(nch, -1); want = g.filt(x).reshape(nch, -1)
                if got.shape != want.shape or not torch.equal(got.contiguous().view(torch.uint8), want.contiguous().view(torch.uint8)): ok = False; why = f"step {step}: plain call"
            else:
                f.reset(); g.reset()
            if not ok: break
            sf, sg = f.state, g.state
            if (sf.phiIdx, sf.inputDeficit, sf.phiAccumulator) != (sg.phiIdx, sg.inputDeficit, sg.phiAccumulator): ok = False; why = f"step {step} (op {op}): state {sf} vs {sg}"; break
    except Exception as e:
        ok = False; why = "exception " + str(e)[:200]
    if not ok:
        bad += 1
        print("MISMATCH", dict(case=case, kind=kind, ratio=str(ratio), nch=nch, why=why), flush=True)
    done += 1
    f.close(); g.close()
print(f"advance-state stress: cases {done} mismatches {bad} in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
