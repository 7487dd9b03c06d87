#!/usr/bin/env python3
"""poly_phase_stationary_kernel timed on three shapes it serves (MRHIP_PS_ABLATE from the environment: 16 = every DMA round waited for)."""
import os, sys
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
rng = np.random.default_rng(3)
for (L, M, taps, nch, n, dt) in ((4, 33, 19, 64, 4_000_000, np.float32), (3, 40, 60, 64, 4_000_000, np.float32), (2, 25, 40, 64, 2_000_000, np.complex64), (5, 64, 100, 32, 4_000_000, np.float32)):
    h = rng.standard_normal(taps).astype(np.float32)
    x = torch.rand((nch, n), dtype=torch.float32, device="cuda") if dt == np.float32 else torch.view_as_complex(torch.rand((nch, n, 2), dtype=torch.float32, device="cuda"))
    f = pkg.FIRFilter(h, Fraction(L, M)).bind(dt, nch)
    y = torch.empty((nch, f.outputlength_bound(n) + 8), dtype=x.dtype, device="cuda")
    for _ in range(3):
        f.reset(); f.filt_into(y, x)
    f.set_timing(True)
    for _ in range(10):
        f.reset(); f.filt_into(y, x)
    nl, ms = f.timing_read()
    print(f"{L}//{M} {taps} taps {np.dtype(dt).name} {nch} ch x {n}: kernel={f.last_kernel_name()} {ms / 10:.4f} ms", flush=True)
    f.close()
