#!/usr/bin/env python3
"""BASELINE config 2 through the library's chunk loop (one launch per pass of 100 chunks), 16 passes in a row: kernel time per pass.
Why did the row read 0.16 ms per chunk after two untimed passes and 0.22 after four?

    python scripts/exp_c2_drift.py
"""
import os
import sys
import time
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge

pkg = ge.load_package()
h = pkg.firdes(3528, 0.45 / 160, beta=7.8562).astype(np.float32) if hasattr(pkg, "firdes") else None
n, chunk = 100_000_000, 1_000_000
x = torch.rand((1, n), dtype=torch.float32, device="cuda")
for mode in ("sync after every pass", "no sync between passes", "sync after every pass, 30 ms idle"):
    f = pkg.FIRFilter(h, Fraction(147, 160))
    f.filt(x[:, :chunk])
    f.reset()
    y = torch.empty((1, f.outputlength(n) + 8), dtype=torch.float32, device="cuda")
    f.set_timing(True)
    times = []
    for i in range(16):
        f.reset()
        f.filt_into_chunked(y, x, chunk)
        if not mode.startswith("no sync"):
            nl, ms = f.timing_read()
            times.append(ms)
            if "idle" in mode:
                time.sleep(0.03)
    if mode.startswith("no sync"):
        nl, ms = f.timing_read()
        times = [ms / 16]
    print(f"{mode:36s} kernel={f.last_kernel_name()} ms per pass: " + " ".join(f"{t:.2f}" for t in times), flush=True)
    f.close()

# the bench harness' order of events: k untimed passes (a device synchronisation after each), timing on, three timed passes
for k in (2, 3, 4, 5, 2, 4):
    f = pkg.FIRFilter(h, Fraction(147, 160))
    f.filt(x[:, :chunk])
    f.reset()
    y = torch.empty((1, f.outputlength(n) + 8), dtype=torch.float32, device="cuda")
    for _ in range(k):
        f.reset()
        f.filt_into_chunked(y, x, chunk)
        torch.cuda.synchronize()
    f.set_timing(True)
    torch.cuda.synchronize()
    for _ in range(3):
        f.reset()
        f.filt_into_chunked(y, x, chunk)
    torch.cuda.synchronize()
    nl, ms = f.timing_read()
    print(f"{k} untimed passes, then 3 timed: launches={nl} ms per launch={ms / max(nl, 1):.4f} (ms total {ms:.2f})", flush=True)
    f.close()
