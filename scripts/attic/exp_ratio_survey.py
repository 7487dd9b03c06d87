#!/usr/bin/env python3
"""Which kernel serves which resampling ratio, and how fast: Float32, 64 channels x 2e6 samples, 24 taps per phase
(or --taps-per-phase N), one filt! call per pass.  Prints kernel name, ms per pass, % of the HBM roofline (algorithmic bytes).
    python scripts/exp_ratio_survey.py [L/M ...]
"""
import os, sys
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
args = sys.argv[1:]
T = 24
dt = "float32"
if "--dtype" in args:
    i = args.index("--dtype"); dt = args[i + 1]; del args[i:i + 2]
if "--taps-per-phase" in args:
    i = args.index("--taps-per-phase"); T = int(args[i + 1]); del args[i:i + 2]
ratios = [Fraction(a) for a in args] or [Fraction(147, 160), Fraction(160, 147), Fraction(3, 2), Fraction(2, 3), Fraction(441, 160), Fraction(160, 441),
                                         Fraction(80, 441), Fraction(3, 17), Fraction(5, 4), Fraction(4, 5), Fraction(7, 3), Fraction(3, 7), Fraction(1, 2), Fraction(2, 1),
                                         Fraction(1, 3), Fraction(3, 1), Fraction(5, 2), Fraction(2, 5), Fraction(25, 12), Fraction(12, 25)]
nch, n = 64, 2_000_000
tdt = getattr(torch, dt)
es = {"float32": 4, "float64": 8, "complex64": 8, "complex128": 16}[dt]
x = torch.view_as_complex(torch.rand((nch, n, 2), device="cuda", dtype=torch.float32 if dt == "complex64" else torch.float64)) if dt.startswith("complex") else torch.rand((nch, n), device="cuda", dtype=tdt)
for r in ratios:
    L, M = r.numerator, r.denominator
    h = pkg.firdes(T * L, 0.5 / max(L, M), beta=7.8562).astype(np.float64 if dt in ("float64", "complex128") else np.float32)
    f = pkg.FIRFilter(h, r)
    y = f.filt(x)
    f.set_timing(True)
    for _ in range(4):
        f.reset(); f.filt_into(y, x)
    torch.cuda.synchronize()
    nl, ms = f.timing_read()
    per = ms / 4
    bytes_ = nch * n * es * (1 + L / M)
    print(f"{L:4d}//{M:<4d} M/L={M / L:5.2f}  {f.last_kernel_name():32s} launches/pass {nl // 4}  {per:8.4f} ms  {bytes_ / (per * 1e-3) / 8e12 * 100:5.1f} % HBM  {nch * n / per / 1e3:10.0f} Msamples/s in", flush=True)
    f.close()
