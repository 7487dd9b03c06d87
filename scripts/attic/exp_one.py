#!/usr/bin/env python3
"""One configuration, a few launches: the program to put behind `rocprofv3 --pmc ... --` for A/B counter passes
(the library and the kernel switches come from the environment: MRHIP_LIB_PATH, MRHIP_OPAIR_C, ...).
    python3 scripts/exp_one.py [--long N] [--ratio 147/160] [--dtype float32] [--taps64 0] [--reps 3]
"""
import os, sys, time
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
args = sys.argv[1:]
ratio, nlong, nch, dt, taps64, reps, zeros = Fraction(147, 160), 20_000_000, 64, "float32", 0, 3, 0
while args:
    k = args.pop(0); v = args.pop(0)
    if k == "--ratio": ratio = Fraction(v)
    elif k == "--long": nlong = int(v)
    elif k == "--channels": nch = int(v)
    elif k == "--dtype": dt = v
    elif k == "--taps64": taps64 = int(v)
    elif k == "--reps": reps = int(v)
    elif k == "--zeros": zeros = int(v)
L, M = ratio.numerator, ratio.denominator
h = pkg.firdes(24 * L, 0.5 / max(L, M), beta=7.8562).astype(np.float64 if (taps64 or dt == "float64") else np.float32)
tdt = getattr(torch, dt)
if dt == "complex64": x = torch.view_as_complex(torch.rand((nch, nlong, 2), device="cuda"))
else: x = torch.rand((nch, nlong), device="cuda", dtype=tdt)
if zeros: x.zero_()
f = pkg.FIRFilter(h, ratio)
y = f.filt(x)
f.set_timing(1)
for rep in range(reps):
    f.reset(); f.filt_into(y, x)
torch.cuda.synchronize()
nl, ms = f.timing_read()
print(f"kernel {f.last_kernel_name()}  launches {nl}  {ms / max(nl, 1):.4f} ms per launch")
