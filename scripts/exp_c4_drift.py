#!/usr/bin/env python3
"""BASELINE config 4, the same call 40 times in a row: the filter kernel's time per call (HIP events around the launch), on a continuing
stream (every call's schedule is evaluated beside the call before's filter kernel) and as reset + the same block (the schedule's memo).
Is the row's run-to-run spread (3.95 ... 4.6 ms) a settling curve, or the schedule beside the kernel?

    python scripts/exp_c4_drift.py [nch] [n]
"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MRHIP_ENV_DYNAMIC", "1")
import numpy as np
import torch
import __graft_entry__ as ge

pkg = ge.load_package()
nch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
harb = pkg.firdes(32 * 32, 0.45 / 32, beta=7.8562) * 32
x = torch.rand((nch, n), dtype=torch.float64, device="cuda")
for mode in ("continuing", "reset+memo", "continuing", "sleep 50 ms between calls"):
    f = pkg.FIRFilter(harb, float(math.pi / 3), 32).bind(np.float64, nch)
    y = torch.empty((nch, f.outputlength_bound(n) + 8), dtype=torch.float64, device="cuda")
    f.filt_into(y, x)
    torch.cuda.synchronize()
    f.set_timing(True)
    times = []
    t0 = time.perf_counter()
    for i in range(40):
        if mode == "reset+memo":
            f.reset()
        f.filt_into(y, x)
        if mode.startswith("sleep"):
            torch.cuda.synchronize()
            time.sleep(0.05)
        nl, ms = f.timing_read()          # (waits for the call; the sum of the launches since the last read)
        times.append(ms)
    wall = (time.perf_counter() - t0) * 1e3 / 40
    print(f"{mode:28s} kernel={f.last_kernel_name()} wall/call={wall:.3f} ms; kernel ms per call: " + " ".join(f"{t:.2f}" for t in times), flush=True)
    f.close()
