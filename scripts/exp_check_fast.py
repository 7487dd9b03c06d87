#!/usr/bin/env python3
"""Quick bit-exactness check of the FAST developer build (tapsPerPhi = 24 only) against the universal kernel:
147//160 and 160//147, Float32 and ComplexF32, launches long enough for dynamic scheduling + tapered tail, short
ones for the static path, chunked.  MRHIP_LIB_PATH must point at libmultirate_hip_fast.so."""
import os, sys
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
ok = True
TAPS = [int(a) for a in sys.argv[1:]] or [24]         # tapsPerPhi values to check (FAST build: 24, 36, 48)
RATIOS = [tuple(int(v) for v in r.split("/")) for r in os.environ.get("RATIOS", "147/160,160/147,2/3,3/2").split(",")]   # L/M list
for (L, M, T) in [(l, m, t) for t in TAPS for (l, m) in RATIOS]:
    if T > 32 and False:
        continue
    h32 = pkg.firdes(T * L, 0.5 / max(L, M), beta=7.8562).astype(np.float32)
    for dt, nch, n, th in ((torch.float32, 64, 1_000_000, np.float32), (torch.float32, 3, 40_000, np.float32), (torch.complex64, 32, 700_000, np.float32),
                           (torch.float32, 1, 3_000_000, np.float32), (torch.float64, 64, 500_000, np.float64), (torch.float64, 2, 30_011, np.float64),
                           (torch.float32, 64, 500_000, np.float64), (torch.float32, 1, 1_000_000, np.float64), (torch.complex64, 32, 400_000, np.float64), (torch.complex128, 24, 300_000, np.float64)):
        h = h32.astype(th)
        if th == np.float64 and (T > 48 or (T > 32 and dt.is_complex)):
            continue                                  # Float64 arithmetic: tapsPerPhi <= 48 on real samples, <= 32 on complex ones
        if dt == torch.float64:
            x = torch.rand((nch, n), device="cuda", dtype=torch.float64) - 0.5
        elif dt == torch.complex64:
            x = torch.view_as_complex(torch.rand((nch, n, 2), device="cuda") - 0.5)
        elif dt == torch.complex128:
            x = torch.view_as_complex(torch.rand((nch, n, 2), device="cuda", dtype=torch.float64) - 0.5)
        else:
            x = torch.rand((nch, n), device="cuda") - 0.5
        sizes = [n // 2 + 7, 1, n - n // 2 - 8]
        os.environ.pop("MRHIP_FORCE_GENERIC", None)
        f = pkg.FIRFilter(h, Fraction(L, M))
        ys, pos = [], 0
        for s in sizes:
            ys.append(f.filt(x[:, pos:pos + s])); pos += s
        y = torch.cat(ys, dim=1); kn = f.last_kernel_name()
        os.environ["MRHIP_FORCE_GENERIC"] = "1"
        g = pkg.FIRFilter(h, Fraction(L, M))
        ys, pos = [], 0
        for s in sizes:
            ys.append(g.filt(x[:, pos:pos + s])); pos += s
        yg = torch.cat(ys, dim=1)
        os.environ.pop("MRHIP_FORCE_GENERIC", None)
        a = torch.view_as_real(y) if dt.is_complex else y
        b = torch.view_as_real(yg) if dt.is_complex else yg
        same = torch.equal(a.view(torch.int32), b.view(torch.int32)) and np.array_equal(f.history.view(np.uint32), g.history.view(np.uint32))
        print(f"{L}//{M} T={T} {dt} taps={np.dtype(th)} nch={nch} n={n} kernel={kn} vs {g.last_kernel_name()}: {'OK' if same else 'MISMATCH'}", flush=True)
        ok = ok and same
print("ALL OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
