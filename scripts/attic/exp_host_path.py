"""Host-pointer path (mrhip_filt_host: what a Julia caller with ordinary Arrays uses): throughput vs PCIe, for pageable
numpy arrays and for page-locked buffers (torch pinned memory viewed as numpy; hipHostRegister'ed Julia Arrays behave the
same), with the library's double-buffered copy/compute pipeline and with it switched off (one piece per call)."""
import os, sys, time
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
h = pkg.firdes(24 * 147, 0.5 / 147, beta=7.8562).astype(np.float32)
for nch, n in ((1, 100_000_000), (64, 4_000_000)):
    src = np.random.default_rng(0).random((nch, n), dtype=np.float32)
    n_out = (n * 147 + 159) // 160
    for pinned in (False, True):
        if pinned:
            xt = torch.empty((nch, n), dtype=torch.float32).pin_memory(); xt.numpy()[:] = src
            yt = torch.empty((nch, n_out), dtype=torch.float32).pin_memory()
            x, y = xt.numpy(), yt.numpy()
        else:
            x, y = src, np.empty((nch, n_out), dtype=np.float32)
        for piece_kb in (None, 1 << 30):          # default pieces (64 MiB) vs one piece (no overlap)
            if piece_kb:
                os.environ["MRHIP_HOST_PIECE_KB"] = str(piece_kb)
            else:
                os.environ.pop("MRHIP_HOST_PIECE_KB", None)
            f = pkg.FIRFilter(h, Fraction(147, 160))
            f.filt_into(y, x); ts = []
            for _ in range(3):
                f.reset(); t0 = time.perf_counter(); got = f.filt_into(y, x); ts.append(time.perf_counter() - t0)
            assert got == n_out
            t = min(ts)
            gb = (x.nbytes + y.nbytes) / 1e9
            print(f"host path {nch} ch x {n}, {'pinned' if pinned else 'pageable'}, {'pipelined 64 MiB pieces' if not piece_kb else 'one piece'}: "
                  f"{t*1e3:.1f} ms, {nch*n/t/1e6:.0f} Msamples/s, {gb/t:.1f} GB/s over PCIe (in+out)", flush=True)
            f.close()
