"""Host-pointer path (mrhip_filt_host: what a Julia caller with ordinary Arrays uses): throughput vs PCIe."""
import os, sys, time
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
h = pkg.firdes(24 * 147, 0.5 / 147, beta=7.8562).astype(np.float32)
for nch, n in ((1, 10_000_000), (1, 100_000_000), (64, 1_000_000), (64, 10_000_000)):
    x = np.random.default_rng(0).random((nch, n), dtype=np.float32)
    f = pkg.FIRFilter(h, Fraction(147, 160))
    y = f.filt(x); f.reset()
    ts = []
    for _ in range(3):
        f.reset(); t0 = time.perf_counter(); y = f.filt(x); ts.append(time.perf_counter() - t0)
    t = min(ts)
    gb = (x.nbytes + y.nbytes) / 1e9
    print(f"host path {nch} ch x {n}: {t*1e3:.1f} ms, {nch*n/t/1e6:.0f} Msamples/s, {gb/t:.1f} GB/s over PCIe (in+out)", flush=True)
    f.close()
