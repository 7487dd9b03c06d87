"""FIRArbitrary / FIRFarrow on a stream that does NOT repeat: bench_configs.py resets the filter and filters the same block
again, so after the first pass the phase schedule comes from the memo of the identical earlier call.  Here the calls continue
the stream (new state every call: the schedule is evaluated every time) -- wall per call against the filter kernel alone,
synchronous and asynchronous calls.
Usage: python scripts/exp_arb_continuing.py"""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda", 0)
TD = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64, np.dtype(np.complex64): torch.complex64, np.dtype(np.complex128): torch.complex128}


def row(name, h, rate, polyorder, nch, n, dtype, ncalls=6):
    x = torch.rand((nch, n), device=dev, dtype=dtype)
    f = pkg.FIRFilter(h, rate, 32, polyorder)
    f.bind(np.float64 if dtype == torch.float64 else np.float32, nch)
    y = torch.empty((nch, f.outputlength_bound(n)), dtype=TD[np.dtype(f.output_dtype)], device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    out = {"row": name, "channels": nch, "samples_per_call": n}
    for mode in ("memo (reset + the same block)", "continuing, synchronous", "continuing, asynchronous"):
        f.reset()
        f.filt_into(y, x); f.filt_into(y, x)
        torch.cuda.synchronize()
        f.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(ncalls):
            if mode.startswith("memo"):
                f.reset()
                f.filt_into(y, x)
            elif mode.endswith("asynchronous"):
                f.filt_into_async(y, x, cnt)
            else:
                f.filt_into(y, x)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3 / ncalls
        if mode.endswith("asynchronous"):
            f.sync_state()
        nl, ms = f.timing_read()
        f.set_timing(False)
        out[mode] = {"wall_ms_per_call": round(wall, 4), "filter_kernel_ms_per_call": round(ms / max(nl, 1) * (nl / ncalls), 4), "kernel": f.last_kernel_name()}
    print(json.dumps(out), flush=True)
    f.close()
    del x, y
    torch.cuda.empty_cache()


def main():
    global row
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    if only:                       # row names containing any of the words given
        row_all = row
        row = lambda name, *a, **k: row_all(name, *a, **k) if any(w in name for w in only) else None
    harb = pkg.firdes(32 * 32, 0.45 / 32, beta=7.8562) * 32
    haf = pkg.firdes(320, 0.45 / 32, beta=7.8562) * 32
    row("C4 FIRArbitrary pi/3 32x32 f64 64ch x 1e7", harb, float(math.pi / 3), None, 64, 10_000_000, torch.float64)
    row("C4f FIRFarrow pi/3 32x32 polyorder 4 f64 64ch x 1e7", harb, float(math.pi / 3), 4, 64, 10_000_000, torch.float64)
    row("C4 shape, 64ch x 1e6", harb, float(math.pi / 3), None, 64, 1_000_000, torch.float64, ncalls=20)
    row("AF FIRArbitrary 1/2.123456789 f32 1ch x 1e7", haf, 1 / 2.123456789, None, 1, 10_000_000, torch.float32)
    row("AF FIRFarrow 1/2.123456789 f32 1ch x 1e7", haf, 1 / 2.123456789, 4, 1, 10_000_000, torch.float32)
    row("AF FIRArbitrary 1/2.123456789 f32 1ch x 1e6", haf, 1 / 2.123456789, None, 1, 1_000_000, torch.float32, ncalls=20)


if __name__ == "__main__":
    main()
