#!/usr/bin/env python3
"""BASELINE config 1 (147//160, 1 channel x 1e6 samples): wall time per call of a stream that is reset before every call (the bench row), of a
stream that goes on, and of asynchronous calls -- what the 24-25 us of the row are made of."""
import os, sys, time
from fractions import Fraction
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
h = pkg.firdes(3528, 0.45 / 160, beta=7.8562).astype(np.float32)
n = 1_000_000
x = torch.rand((1, n), dtype=torch.float32, device="cuda")
f = pkg.FIRFilter(h, Fraction(147, 160)).bind(np.float32, 1)
y = torch.empty((1, f.outputlength_bound(n) + 8), dtype=torch.float32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
def timed(label, body, reps=2000):
    for _ in range(50): body()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): body()
    torch.cuda.synchronize()
    print(f"{label:58s} {(time.perf_counter() - t0) / reps * 1e6:7.2f} us per call", flush=True)
timed("reset() + filt_into()  (the bench row)", lambda: (f.reset(), f.filt_into(y, x)))
timed("filt_into() on a stream that goes on", lambda: f.filt_into(y, x))
timed("filt_into_async() (the count stays on the device)", lambda: f.filt_into_async(y, x, cnt))
timed("reset() alone", lambda: f.reset())
f.set_timing(True)
for _ in range(200): f.filt_into(y, x)
nl, ms = f.timing_read()
print(f"kernel alone (HIP events): {ms / max(nl, 1) * 1e3:.2f} us", flush=True)
