#!/usr/bin/env python3
"""Ring experiments: 1 channel x 1e8 in 1e6-sample chunks through mrhip_ring_push_chunks under MRHIP_RING_OPTS variants, same box."""
import os, sys, time, json
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
os.environ["MRHIP_ENV_DYNAMIC"] = "1"
h = pkg.firdes(24 * 147, 0.5 / 147, beta=7.8562).astype(np.float32)
nch = int(os.environ.get("EXP_NCH", "1"))
n, chunk = int(os.environ.get("EXP_N", "100000000")) // nch, int(os.environ.get("EXP_CHUNK", "1000000"))
x = torch.rand((nch, n), device="cuda")
f = pkg.FIRFilter(h, Fraction(147, 160), device=0).bind(np.float32, nch)
y = torch.empty((nch, f.outputlength(n) + 8), device="cuda")
for opts in [int(v) for v in (sys.argv[1:] or ["0", "1", "2", "3", "0"])]:
    os.environ["MRHIP_RING_OPTS"] = str(opts)
    ts, tps = [], []
    for rep in range(5):
        f.reset(); torch.cuda.synchronize()
        ring = f.open_ring()
        t1 = time.perf_counter()
        ring.push_chunks(y, x, chunk)
        tp = time.perf_counter()
        ring.drain()
        t2 = time.perf_counter()
        ring.close()
        ts.append(t2 - t1); tps.append(tp - t1)
    ts = sorted(ts[1:])
    ms = 1e3 * ts[len(ts) // 2]
    print(json.dumps({"opts": opts, "nch": nch, "chunk": chunk, "ms": round(ms, 4), "us_per_chunk": round(1e3 * ms / (n // chunk), 3), "frac": round(nch * n * 7.675 / (ms * 1e-3) / 8e12, 4), "all_ms": [round(1e3 * t, 3) for t in ts], "push_returned_ms": [round(1e3 * t, 3) for t in tps[1:]]}), flush=True)
