import os, sys, time
from fractions import Fraction
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MRHIP_ENV_DYNAMIC", "1")
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
h = pkg.firdes(128, 0.5 / 4, beta=7.8562).astype(np.float32)
for nch, n in ((256, 4_000_000), (256, 8_000_000), (64, 16_000_000)):
    x = torch.view_as_complex(torch.rand((nch, n, 2), dtype=torch.float32, device="cuda"))
    y = torch.empty((nch, n // 4 + 8), dtype=torch.complex64, device="cuda")
    for label, env in (("fir_stream", {"MRHIP_DECIM_LANE": "0"}), ("decim_lane", {"MRHIP_DECIM_LANE": "2"}), ("fir_stream", {"MRHIP_DECIM_LANE": "0"}), ("decim_lane", {"MRHIP_DECIM_LANE": "2"})):
        os.environ.update(env)
        f = pkg.FIRFilter(h, Fraction(1, 4)).bind(np.complex64, nch)
        for _ in range(6):
            f.reset(); f.filt_into(y, x)
        f.set_timing(True)
        for _ in range(6):
            f.reset(); f.filt_into(y, x)
        nl, ms = f.timing_read()
        print(f"{nch} ch x {n}: {label:10s} kernel={f.last_kernel_name():18s} {ms / 6:.3f} ms", flush=True)
        f.close()
    del x, y
    torch.cuda.empty_cache()
