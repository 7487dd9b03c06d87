#!/usr/bin/env python3
"""Randomised parity stress of device-resident cascades (mrhip_cascade_*; chained device-planned calls): two or three random stages (decimator,
rational, interpolator, FIRArbitrary, FIRFarrow), random channel counts and ragged chunkings, every call either plain (`filt`) or
asynchronous (`filt_into_async`, counts on the device), against the same stages called one by one on separate FIRFilter objects, bit for bit.
    python scripts/stress_cascade.py [--cases 80] [--seed 1] [--seconds 200]"""
import argparse, os, sys, time
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
TD = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64, np.dtype(np.complex64): torch.complex64, np.dtype(np.complex128): torch.complex128}
ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=80); ap.add_argument("--seed", type=int, default=1); ap.add_argument("--seconds", type=float, default=200.0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t0, bad, done = time.time(), 0, 0

def stage(rng):
    k = ["decimator", "rational", "interpolator", "arbitrary", "farrow"][rng.integers(5)]
    if k == "decimator": return k, rng.standard_normal(40).astype(np.float32), Fraction(1, int(rng.integers(2, 6))), None
    if k == "rational":
        L, M = [(7, 9), (3, 2), (147, 160), (5, 3)][rng.integers(4)]
        return k, (rng.standard_normal(24 * L) / 4).astype(np.float32), Fraction(L, M), None
    if k == "interpolator": return k, rng.standard_normal(32 * 3).astype(np.float32), Fraction(3, 1), None
    h = (pkg.firdes(32 * 6, 0.45 / 32, beta=7.0) * 32).astype(np.float32)
    return k, h, float([0.8123, 1 / 2.123456789, 1.25][rng.integers(3)]), (3 if k == "farrow" else None)

for case in range(a.cases):
    if time.time() - t0 > a.seconds: break
    nst = int(rng.integers(2, 4))
    spec = [stage(rng) for _ in range(nst)]
    tx = [np.float32, np.complex64][rng.integers(2)]
    nch = int(rng.choice([1, 1, 2, 4]))
    nchunks = int(rng.integers(2, 7))
    sizes = [int(rng.integers(500, 60_000)) for _ in range(nchunks)]
    x = rng.standard_normal((nch, sum(sizes))).astype(np.float32)
    if np.dtype(tx).kind == "c": x = x + 1j * rng.standard_normal(x.shape).astype(np.float32)
    x = x.astype(tx)
    xd = torch.from_numpy(x).cuda()
    cas = pkg.FilterCascade(*[pkg.FIRFilter(h, r, 32, po) for (_, h, r, po) in spec])
    one = [pkg.FIRFilter(h, r, 32, po) for (_, h, r, po) in spec]
    ok, pos, why = True, 0, ""
    try:
        for i, s_ in enumerate(sizes):
            xs = xd[:, pos:pos + s_]; pos += s_
            want = xs
            for f in one: want = f.filt(want)
            want = want.reshape(nch, -1)
            if i > 0 and rng.random() < 0.5:          # (the first call of a size is a plain one: it allocates the buffers between the stages)
                buf = torch.empty((nch, max(cas.outputlength_bound(s_), 1)), dtype=TD[np.dtype(cas.output_dtype)], device="cuda")
                cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
                cas.filt(xd[:, pos - s_:pos][:, :0]) if False else None
                cas.filt_into_async(buf, xs, cnt)
                torch.cuda.synchronize()
                got = buf[:, :int(cnt.item())]
            else:
                got = cas.filt(xs).reshape(nch, -1)
            if got.shape != want.shape or not torch.equal(torch.view_as_real(got.contiguous()) if got.is_complex() else got.contiguous(), torch.view_as_real(want.contiguous()) if want.is_complex() else want.contiguous()):
                ok = False; why = f"call {i}: shapes {tuple(got.shape)} vs {tuple(want.shape)}"
                break
    except Exception as e:
        ok = False; why = "exception " + str(e)[:160]
    if not ok:
        bad += 1
        print("MISMATCH", dict(case=case, stages=[(k, str(r)) for (k, _, r, _) in spec], tx=np.dtype(tx).name, nch=nch, sizes=sizes, why=why), flush=True)
    done += 1
print(f"cascade stress: cases {done} mismatches {bad} in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
