#!/usr/bin/env python3
"""Randomised parity stress of the host-pointer path (mrhip_filt_host: pinned double-buffered H2D / kernel / D2H pipeline cut into pieces; run it with
MRHIP_HOST_PIECE_KB=64 so that every call is many pieces) and of the library's chunk loop (mrhip_filt_device_chunked): numpy inputs of random length
through FIRFilter.filt against the same filter fed device tensors, every kind, bit for bit, over several calls of one stream.
    MRHIP_HOST_PIECE_KB=64 python scripts/stress_host.py [--cases 100] [--seed 1] [--seconds 200]"""
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
    kind = ["rational", "decimator", "interpolator", "standard", "arbitrary", "farrow", "bigL"][rng.integers(7)]
    tx = [np.float32, np.complex64, np.float64][rng.integers(3)]
    po = None
    if kind == "rational": h, ratio = (rng.standard_normal(24 * 7) / 4).astype(np.float32), Fraction(7, 9)
    elif kind == "decimator": h, ratio = rng.standard_normal(40).astype(np.float32), Fraction(1, int(rng.integers(2, 40)))
    elif kind == "interpolator": h, ratio = rng.standard_normal(32 * 3).astype(np.float32), Fraction(3, 1)
    elif kind == "standard": h, ratio = rng.standard_normal(33).astype(np.float32), Fraction(1, 1)
    elif kind == "bigL": h, ratio = (rng.standard_normal(8 * 625) / 4).astype(np.float32), Fraction(625, 512)
    else: h, ratio, po = (pkg.firdes(32 * 6, 0.45 / 32, beta=7.0) * 32).astype(np.float32), float([0.8123, 1 / 2.123456789, 1.25][rng.integers(3)]), (3 if kind == "farrow" else None)
    nch = int(rng.choice([1, 1, 2, 3]))
    f = pkg.FIRFilter(h, ratio, 32, po)
    g = pkg.FIRFilter(h, ratio, 32, po, pnfb=None)
    ok, why = True, ""
    try:
        for call in range(int(rng.integers(1, 4))):
            n = int(rng.integers(1, 400_000)) if rng.random() > 0.1 else int(rng.integers(1, 50))
            x = rng.standard_normal((nch, n)).astype(np.float32)
            if np.dtype(tx).kind == "c": x = x + 1j * rng.standard_normal(x.shape).astype(np.float32)
            x = x.astype(tx)
            xin = x if nch > 1 else x[0]
            got = np.asarray(f.filt(xin)).reshape(nch, -1)                      # host path
            if call == 0 and po:                                                # (the same fitted polynomial bank in the second filter)
                g.close(); g = pkg.FIRFilter(h, ratio, 32, po, pnfb=f.pnfb())
            want = g.filt(torch.from_numpy(x).cuda()).cpu().numpy().reshape(nch, -1)   # device path
            if got.shape != want.shape or not np.array_equal(got.view(np.uint8), np.ascontiguousarray(want).view(np.uint8)):
                ok = False; why = f"call {call}: {got.shape} vs {want.shape}"; break
    except Exception as e:
        ok = False; why = "exception " + str(e)[:200]
    if not ok:
        bad += 1
        print("MISMATCH", dict(case=case, kind=kind, ratio=str(ratio), tx=np.dtype(tx).name, nch=nch, why=why), flush=True)
    done += 1
    f.close(); g.close()
print(f"host-path stress: cases {done} mismatches {bad} in {time.time() - t0:.0f} s, piece_kb={os.environ.get('MRHIP_HOST_PIECE_KB', 'default')}")
sys.exit(1 if bad else 0)
