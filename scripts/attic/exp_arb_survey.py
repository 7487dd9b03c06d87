import os, sys, math, time
sys.path.insert(0, ".")
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
for (th, dt, nch, n) in ((np.float32, torch.float32, 64, 10_000_000), (np.float32, torch.complex64, 64, 5_000_000), (np.float64, torch.float64, 64, 10_000_000),
                         (np.float32, torch.float32, 512, 2_000_000), (np.float64, torch.float64, 256, 2_000_000)):
    h = (pkg.firdes(32 * 32, 0.45 / 32, beta=7.8562) * 32).astype(th)
    if dt.is_complex: x = torch.view_as_complex(torch.rand((nch, n, 2), device="cuda"))
    else: x = torch.rand((nch, n), device="cuda", dtype=dt)
    for rate in (math.pi / 3, 0.37):
        f = pkg.FIRFilter(h, float(rate), 32)
        y = f.filt(x); f.set_timing(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2): f.reset(); y = f.filt(x)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 2
        nl, ms = f.timing_read(); per = ms / 2
        es = x.element_size(); ob = y.element_size()
        b = nch * n * (es + ob * rate)
        print(f"arbitrary rate={rate:.3f} {str(dt):16s} taps={np.dtype(th).name} nch={nch} n={n}: {f.last_kernel_name()} kernel {per:.3f} ms ({b / (per * 1e-3) / 8e12 * 100:.1f} % HBM) wall {wall * 1e3:.3f} ms ({b / wall / 8e12 * 100:.1f} %) launches {nl // 2}", flush=True)
        f.close()
