"""160//147 (the headline's mirror: 32 lanes of an LDS read span 29 words, no bank conflicts) at the headline's scale, for PMC passes:
is the headline's 44 % of conflict cycles what holds its VALU at 76 %?  usage: [rocprofv3 --pmc ... --] python3 scripts/exp_160_147_long.py [L M]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fractions import Fraction
import __graft_entry__ as ge
pkg = ge.load_package()
L, M = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (160, 147)
h = pkg.firdes(24 * L, 0.5 / max(L, M), beta=7.8562).astype(np.float32)
nch, n = 64, 50_000_000
x = torch.rand((nch, n), device="cuda", dtype=torch.float32)
f = pkg.FIRFilter(h, Fraction(L, M)).bind(np.float32, nch)
y = torch.empty((nch, f.outputlength(n) + 2), dtype=torch.float32, device="cuda")
f.filt_into(y, x); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    f.reset(); f.filt_into(y, x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
gb = nch * n * (4 + 4 * L / M) / 1e9
print(f"{L}//{M} 64ch x {n}: {ms:.3f} ms  {gb / ms * 1e3:.0f} GB/s  {gb / ms * 1e3 / 8000:.3f} of HBM  kernel={f.last_kernel_name()}")
