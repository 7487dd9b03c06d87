#!/usr/bin/env python3
"""Knob sweep on the headline shape (147//160 Float32, 64 channels): for every environment setting given on the command
line as K=V,K=V items, time (a) one call over --long samples per channel and (b) the same signal streamed per call in
1e6-sample chunks.  Kernel time from the library's HIP events (every launch for (a), every 4th for (b)).
    python scripts/exp_sweep.py [--ratio 147/160] [--long 20000000] "MRHIP_PAIR=0,MRHIP_OPAIR_C=6" "MRHIP_PAIR_J=6" ...
"""
import json, os, sys, time
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
args = sys.argv[1:]
ratio, nlong, nch, dtype = Fraction(147, 160), 20_000_000, 64, torch.float32
while args and args[0].startswith("--"):
    k = args.pop(0)
    v = args.pop(0)
    if k == "--ratio": ratio = Fraction(v)
    elif k == "--long": nlong = int(v)
    elif k == "--channels": nch = int(v)
    elif k == "--dtype": dtype = getattr(torch, v)
L, M = ratio.numerator, ratio.denominator
h = pkg.firdes(24 * L, 0.5 / max(L, M), beta=7.8562).astype(np.float32)
es = 8 if dtype == torch.complex64 else 4
bpi = es * (1 + L / M)
if dtype == torch.complex64:
    x = torch.view_as_complex(torch.rand((nch, nlong, 2), device="cuda", dtype=torch.float32))
else:
    x = torch.rand((nch, nlong), device="cuda", dtype=dtype)
y = torch.empty((nch, nlong * L // M + 16), device="cuda", dtype=dtype)
base_env = dict(os.environ)
for spec in args or [""]:
    os.environ.clear(); os.environ.update(base_env)
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("="); os.environ[k] = v
    f = pkg.FIRFilter(h, ratio).bind(np.complex64 if dtype == torch.complex64 else np.float32, nch)
    res = {"env": spec}
    for mode in ("one_call", "streamed_1e6"):
        chunk = nlong if mode == "one_call" else 1_000_000
        os.environ["MRHIP_CHUNKED_PER_CALL"] = "1"
        for rep in range(3):
            if rep == 2:
                f.set_timing(1 if mode == "one_call" else 4); torch.cuda.synchronize(); t0 = time.perf_counter()
            f.reset(); f.filt_into_chunked(y, x, chunk)
        torch.cuda.synchronize(); wall = time.perf_counter() - t0
        nl, ms = f.timing_read(); f.set_timing(False)
        per = ms / max(nl, 1)
        res[mode] = {"kernel": f.last_kernel_name(), "launch_ms": round(per, 4), "frac": round(nch * chunk * bpi / (per * 1e-3) / 8e12, 4),
                     "wall_frac": round(nch * nlong * bpi / wall / 8e12, 4)}
    print(json.dumps(res), flush=True)
    f.close()
