import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from fractions import Fraction
import __graft_entry__ as ge
pkg = ge.load_package()
h = pkg.firdes(24 * 147, 0.5 / 147, beta=7.8562).astype(np.float32)
f = pkg.FIRFilter(h, Fraction(147, 160), device=0).bind(np.float32, 1)
x = torch.rand((1, 1_000_000), device="cuda"); y = torch.empty((1, 918_760), device="cuda")
torch.cuda.synchronize()
for rep in range(6):
    f.reset(); torch.cuda.synchronize()
    t0 = time.perf_counter(); ring = f.open_ring(); t1 = time.perf_counter()
    ring.push(y, x); ring.drain(); t2 = time.perf_counter()
    ring.close(); t3 = time.perf_counter()
    print(f"open {1e6*(t1-t0):.0f} us  push+drain {1e6*(t2-t1):.0f} us  close {1e6*(t3-t2):.0f} us")
