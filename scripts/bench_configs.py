#!/usr/bin/env python3
"""Secondary measurements: BASELINE.json configs 1-5 on one MI355X (kernel-only GB/s from the library's
HIP-event log).  Not the headline bench (that is bench.py); used for DESIGN.md's per-kernel table."""
import json
import math
import os
import sys
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge

pkg = ge.load_package()
dev = torch.device("cuda", 0)


def rand(shape, dtype):
    if dtype.is_complex:
        return torch.view_as_complex(torch.rand(shape + (2,), device=dev, dtype=torch.float32 if dtype == torch.complex64 else torch.float64))
    return torch.rand(shape, device=dev, dtype=dtype)


def run(name, h, ratio, nphi, nch, n, dtype, bytes_per_in, reps=5, chunk=None, polyorder=None):
    x = rand((nch, n), dtype)
    f = pkg.FIRFilter(h, ratio, nphi, polyorder, device=0)
    chunk = chunk or n
    f.filt(x[:, :chunk])                       # warm-up + bind
    f.reset()
    f.set_timing(True)
    import time
    torch.cuda.synchronize()
    t_wall = time.perf_counter()
    ychunked = None
    if chunk != n and os.environ.get("MRHIP_BENCH_CHUNKED", "1") == "1":      # streaming through the library's chunk loop
        ychunked = torch.empty((nch, f.outputlength(n) + 8), dtype=torch.float32 if dtype == torch.float32 else dtype, device=dev)
    for _ in range(reps):
        f.reset()
        if ychunked is not None:
            f.filt_into_chunked(ychunked, x, chunk)
            continue
        for a in range(0, n, chunk):
            f.filt(x[:, a:a + chunk])
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t_wall) * 1e3 / reps
    nl, ms = f.timing_read()
    per_pass_ms = ms / reps
    gbps = nch * n * bytes_per_in / (per_pass_ms * 1e-3) / 1e9
    out = {"config": name, "kernel": f.last_kernel_name(), "channels": nch, "samples_per_channel": n,
           "kernel_ms_per_pass": round(per_pass_ms, 4), "wall_ms_per_pass_incl_host": round(wall_ms, 3), "launches_per_pass": nl // reps,
           "Msamples_per_s_in": round(nch * n / (per_pass_ms * 1e-3) / 1e6, 1),
           "algorithmic_GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / 8000, 4)}
    print(json.dumps(out), flush=True)
    f.close()
    del x
    torch.cuda.empty_cache()


h147 = pkg.firdes(24 * 147, 0.5 / 147, beta=7.8562).astype(np.float32)
h128 = pkg.firdes(128, 0.5 / 4, beta=7.8562).astype(np.float32)
harb = pkg.firdes(32 * 32, 0.45 / 32, beta=7.8562) * 32

which = sys.argv[1:] or ["c1", "c2", "c3a", "c3b", "c4", "c4f", "c5"]
if "c1" in which:
    run("C1 rational 147//160 f32 1ch x 1e6 (one call)", h147, Fraction(147, 160), 32, 1, 1_000_000, torch.float32, 7.675)
if "c2" in which:
    run("C2 rational 147//160 f32 1ch x 1e8 in 1e6 chunks", h147, Fraction(147, 160), 32, 1, 100_000_000, torch.float32, 7.675, reps=2, chunk=1_000_000)
if "c3a" in which:
    run("C3a interpolator 4//1 128 taps c64 256ch x 1e6", h128, Fraction(4, 1), 32, 256, 1_000_000, torch.complex64, 40.0)
if "c3b" in which:
    run("C3b decimator 1//4 128 taps c64 256ch x 1e6", h128, Fraction(1, 4), 32, 256, 1_000_000, torch.complex64, 10.0)
if "c4" in which:
    run("C4 arbitrary pi/3 32x32 taps f64 64ch x 1e7", harb, float(math.pi / 3), 32, 64, 10_000_000, torch.float64, 8 + 8 * math.pi / 3, reps=2)
if "c4f" in which:
    run("C4f farrow pi/3 32x32 taps polyorder 4 f64 64ch x 1e7", harb, float(math.pi / 3), 32, 64, 10_000_000, torch.float64, 8 + 8 * math.pi / 3, reps=2, polyorder=4)
if "c5" in which:
    run("C5 rational 147//160 c64 512ch x 1e6 (one GPU's shard of 4096)", h147, Fraction(147, 160), 32, 512, 1_000_000, torch.complex64, 15.35)

# extra shapes outside BASELINE.json (where the non-headline kernels stand)
if "x160" in which:
    h160 = pkg.firdes(24 * 160, 0.5 / 160, beta=7.8562).astype(np.float32)
    run("X rational 160//147 (44.1k->48k) f32 64ch x 1e6", h160, Fraction(160, 147), 32, 64, 1_000_000, torch.float32, 4 + 4 * 160 / 147)
if "xf64" in which:
    run("X rational 147//160 f64 64ch x 1e6", h147.astype(np.float64), Fraction(147, 160), 32, 64, 1_000_000, torch.float64, 2 * 7.675)
if "xstd" in which:
    run("X standard 1//1 128 taps f32 64ch x 4e6", h128, Fraction(1, 1), 32, 64, 4_000_000, torch.float32, 8.0)
if "x32" in which:
    h32 = pkg.firdes(24 * 3, 0.5 / 3, beta=7.8562).astype(np.float32)
    run("X rational 3//2 f32 64ch x 1e6", h32, Fraction(3, 2), 32, 64, 1_000_000, torch.float32, 4 + 6)
    run("X rational 2//3 f32 64ch x 1e6", h32, Fraction(2, 3), 32, 64, 1_000_000, torch.float32, 4 + 8 / 3)
if "xc32" in which:   # the headline's launch size (491 MB) on the ComplexF32 kernel: 32 complex channels x 1e6 per launch
    os.environ["MRHIP_BENCH_CHUNKED"] = "0"
    run("X rational 147//160 c64 32ch x 2e7 in 1e6 chunks", h147, Fraction(147, 160), 32, 32, 20_000_000, torch.complex64, 15.35, reps=3, chunk=1_000_000)
    run("X rational 147//160 f32 64ch x 2e7 in 1e6 chunks", h147, Fraction(147, 160), 32, 64, 20_000_000, torch.float32, 7.675, reps=3, chunk=1_000_000)
