"""Launch-gap experiment: whole-pass time of 100 x 1e6-sample chunks with and without the per-launch HIP events."""
import os, sys, time
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
L, M = 147, 160
h = pkg.firdes(24 * L, 0.5 / L, beta=7.8562).astype(np.float32)
nch, n, chunk = 64, 50_000_000, 1_000_000
x = torch.rand((nch, n), dtype=torch.float32, device="cuda")
y = torch.empty((nch, n * L // M + 8), dtype=torch.float32, device="cuda")
f = pkg.FIRFilter(h, Fraction(L, M)); f.bind(np.float32, nch)
def one_pass():
    f.reset(); k = 0
    for a in range(0, n, chunk):
        cnt = f.next_output_count(chunk)
        f.filt_into(y[:, k:k + cnt], x[:, a:a + chunk]); k += cnt
for timing in (False, True, False, True):
    one_pass(); torch.cuda.synchronize()
    f.set_timing(timing)
    t0 = time.perf_counter()
    for _ in range(3): one_pass()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    extra = ""
    if timing:
        nl, ms = f.timing_read(); extra = f" kernel avg {ms / nl * 1e3:.1f} us"
    f.set_timing(False)
    print(f"timing={timing}: {dt / (n // chunk) * 1e6:.1f} us per chunk wall{extra}")
