import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
haf = pkg.firdes(320, 0.45 / 32, beta=7.8562) * 32
for nch in (1, 2, 4):
    for n in (300_000, 1_000_000, 3_000_000, 10_000_000):
        f = pkg.FIRFilter(haf, 1 / 2.123456789, 32).bind(np.float32, nch)
        x = torch.rand((nch, n), device="cuda")
        y = torch.empty((nch, f.outputlength_bound(n)), device="cuda", dtype=torch.float64)
        f.filt_into(y, x); f.set_timing(True)
        for _ in range(5): f.filt_into(y, x)
        torch.cuda.synchronize(); nl, ms = f.timing_read()
        print(nch, n, f.last_kernel_name(), round(ms / 5 * 1e3, 1), "us", flush=True)
        f.close()
