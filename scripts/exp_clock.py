"""Is the headline kernel clock/power limited?  Times single launches after idle gaps vs back-to-back."""
import os, sys, time
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
L, M = 147, 160
h = pkg.firdes(24 * L, 0.5 / L, beta=7.8562).astype(np.float32)
nch, n = 64, 1_000_000
dev = torch.device("cuda", 0)
zero = len(sys.argv) > 1 and sys.argv[1] == "zero"
x = torch.zeros((nch, n), dtype=torch.float32, device=dev) if zero else torch.rand((nch, n), dtype=torch.float32, device=dev)
y = torch.empty((nch, n * L // M + 8), dtype=torch.float32, device=dev)
f = pkg.FIRFilter(h, Fraction(L, M), device=0); f.bind(np.float32, nch)
def launch(k):
    f.reset(); cnt = f.next_output_count(n)
    f.set_timing(True)
    for _ in range(k):
        f.filt_into(y[:, :cnt], x)
        f.reset()
    torch.cuda.synchronize()
    nl, ms = f.timing_read(); f.set_timing(False)
    return ms / nl * 1e3
launch(3)
for gap in (0.5, 0.5, 0.5):
    time.sleep(gap); print(f"single launch after {gap}s idle: {launch(1):.1f} us")
for k in (2, 5, 10, 50, 200, 1000, 3000):
    time.sleep(0.5); print(f"{k} back-to-back: avg {launch(k):.1f} us")
