import os, sys, time, json
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
h = pkg.firdes(24 * 147, 0.5 / 147, beta=7.8562).astype(np.float64)
nch = int(os.environ.get("EXP_NCH", "16")); n, chunk = 64_000_000 // nch, 1_000_000
x = torch.rand((nch, n), device="cuda", dtype=torch.float64)
f = pkg.FIRFilter(h, Fraction(147, 160), device=0).bind(np.float64, nch)
y = torch.empty((nch, f.outputlength(n) + 8), device="cuda", dtype=torch.float64)
ts = []
for rep in range(5):
    f.reset(); torch.cuda.current_stream().synchronize()
    ring = f.open_ring()
    t1 = time.perf_counter(); ring.push_chunks(y, x, chunk); ring.drain(); t2 = time.perf_counter()
    info = ring.info(); ring.close(); ts.append(t2 - t1)
ms = 1e3 * sorted(ts[1:])[len(ts[1:]) // 2]
print(json.dumps({"f64 ring nch": nch, "ms": round(ms, 4), "frac": round(nch * n * 15.35 / (ms * 1e-3) / 8e12, 4), "info": info}))
