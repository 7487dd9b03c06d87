#!/usr/bin/env python3
"""Randomised soak of the device-side phase schedule (GPU box): FIRArbitrary / FIRFarrow streams with random rates
(irrational, rational, on and next to the wrap thresholds), N-phi, piece limits and call lengths, a few calls per stream so
that state, drift estimate and cycle detection carry over -- the device-evaluated schedule against the serial host loop
(MRHIP_SCHED_DEVICE=0), outputs and end state bit for bit.  Three taps per phase and one channel: every schedule entry
decides an output.

    python scripts/stress_schedule.py [--cases 150] [--seed 1] [--seconds 300]
Prints one line per failure and a tally of the paths taken; exit code 1 on any mismatch."""
import argparse
import math
import os
import sys
import time
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MRHIP_ENV_DYNAMIC"] = "1"          # the knobs below change between filters
import numpy as np
import torch
import __graft_entry__ as ge

pkg = ge.load_package()


def pick_rate(rng):
    k = rng.integers(0, 6)
    if k == 0:
        return float(rng.choice([math.pi / 3, math.e / 3, math.sqrt(2), 1 / 2.123456789, 48000 / 44100, 44100 / 48000, 0.9991, 1.0009]))
    if k == 1:                                    # small rationals: phases that sit ON thresholds
        return float(Fraction(int(rng.integers(1, 40)), int(rng.integers(1, 40))))
    if k == 2:
        return float(rng.choice([1.0, 2.0, 3.0, 0.5, 0.25, 11 / 7, 56 / 37, 2.5, 1.5]))
    if k == 3:                                    # next to a rational
        return float(Fraction(int(rng.integers(1, 20)), int(rng.integers(1, 20)))) * (1 + float(rng.choice([-1, 1])) * 2.0 ** -int(rng.integers(20, 52)))
    return float(np.exp(rng.uniform(math.log(0.08), math.log(9.0))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--show-fallbacks", action="store_true", help="one line per case in which a piece did not verify (redone serially: exact, slow)")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    gen = torch.Generator(device="cuda").manual_seed(args.seed)
    tally = {"device_pieces": 0, "periodic_steps": 0, "host_steps": 0, "fallback_pieces": 0, "cases": 0, "farrow": 0}
    bad, t0 = 0, time.time()
    for case in range(args.cases):
        if time.time() - t0 > args.seconds:
            break
        rate = pick_rate(rng)
        nphi = int(rng.choice([8, 10, 16, 32, 32, 48]))
        farrow = rng.random() < 0.25
        tx = rng.choice([torch.float32, torch.float64])
        h = np.random.default_rng(case).standard_normal(3 * nphi).astype(np.float32)
        knobs = {}
        if rng.random() < 0.5:
            knobs["MRHIP_SCHED_PMAX"] = str(int(rng.choice([1 << 16, 1 << 18, 1 << 20, 1 << 22])))
        if rng.random() < 0.3:
            knobs["MRHIP_SCHED_PREFIX"] = str(int(rng.choice([4096, 16384, 65536])))
        if rng.random() < 0.2:
            knobs["MRHIP_SCHED_CYCLE"] = "0"
        n_out_total = int(rng.choice([300_000, 1_000_000, 3_000_000, 8_000_000]) * (0.5 + rng.random()))
        n_in = max(1000, int(n_out_total / rate))
        if n_in > 40_000_000:
            n_in = 40_000_000
        x = torch.rand(n_in, generator=gen, device="cuda", dtype=tx) - 0.5
        ncalls = int(rng.integers(1, 4))
        cuts = sorted(set(int(c) for c in rng.integers(0, n_in + 1, size=ncalls - 1)))
        pieces = [x[a:b] for a, b in zip([0] + cuts, cuts + [n_in])]
        for k, v in knobs.items():
            os.environ[k] = v
        try:
            os.environ["MRHIP_SCHED_DEVICE"] = "0"
            fh = pkg.FIRFilter(h, rate, nphi, 3) if farrow else pkg.FIRFilter(h, rate, nphi)
            yh = [fh.filt(p) for p in pieces]
            sh = fh.state
            del os.environ["MRHIP_SCHED_DEVICE"]
            fd = pkg.FIRFilter(h, rate, nphi, 3, pnfb=fh.pnfb()) if farrow else pkg.FIRFilter(h, rate, nphi)
            yd = [fd.filt(p) for p in pieces]
            sd = fd.state
            info = fd.schedule_info()
            ok = all(a.shape == b.shape and torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(yd, yh))
            ok = ok and (sd.phiAccumulator, sd.inputDeficit, sd.phiIdx) == (sh.phiAccumulator, sh.inputDeficit, sh.phiIdx)
            if rng.random() < 0.5:                # the same stream again after reset(): the drift estimate is kept
                fd.reset()
                yd2 = [fd.filt(p) for p in pieces]
                ok = ok and all(a.shape == b.shape and torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(yd2, yh))
                sd = fd.state
                ok = ok and (sd.phiAccumulator, sd.inputDeficit, sd.phiIdx) == (sh.phiAccumulator, sh.inputDeficit, sh.phiIdx)
                info = fd.schedule_info()
            if not ok:
                bad += 1
                print(f"MISMATCH case {case}: rate={rate!r} nphi={nphi} farrow={farrow} {tx} n_in={n_in} cuts={cuts} knobs={knobs} info={info}", flush=True)
            if args.show_fallbacks and info["fallback_pieces"]:
                print(f"fallback case {case}: rate={rate!r} nphi={nphi} knobs={knobs} info={info}", flush=True)
            for k in ("device_pieces", "periodic_steps", "host_steps", "fallback_pieces"):
                tally[k] += info[k]
            tally["cases"] += 1
            tally["farrow"] += int(farrow)
            fd.close(); fh.close()
        finally:
            os.environ.pop("MRHIP_SCHED_DEVICE", None)
            for k in knobs:
                os.environ.pop(k, None)
    print(f"mismatches {bad} {tally} in {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
