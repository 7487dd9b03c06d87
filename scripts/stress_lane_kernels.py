#!/usr/bin/env python3
"""Randomised parity stress of the three one-wave-per-stretch lane kernels -- interp_lane_kernel (FIRInterpolator 4//1, 32 taps per phase,
ComplexF32; the default for long calls, forced here whatever the length), decim_lane_kernel (FIRDecimator 1//4, 128 taps, ComplexF32) and
arb_window_kernel (FIRArbitrary, Float64, 32 taps per phase, rate >= 1), the latter two behind their switches: random channel counts
(partial last groups), random chunkings (one-sample calls, calls too short for the kernels, ragged blocks), x and y as views at odd sample
offsets, STRICT / FUSED, special values, EVERY SIGNAL FRESHLY UPLOADED and the tuned kernel called first (profiles/r06/experiments.md K) --
outputs, end state and history against the universal kernel, bit for bit, and two channels against the oracle.

    python scripts/stress_lane_kernels.py [cases] [seed]
"""
import math
import os
import sys
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MRHIP_ENV_DYNAMIC", "1")
os.environ["MRHIP_ARB_SMALL_MAX"] = "0"
import numpy as np
import torch
import __graft_entry__ as ge

pkg = ge.load_package()
from oracle import oracle as O   # the checker


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20261006
    rng = np.random.default_rng(seed)
    bad, tally = 0, {}
    for case in range(cases):
        which = rng.choice(["interp", "decim", "window"])
        nch = int(rng.choice([48, 50, 63, 64, 64, 65 + 47, 127, 128, 192, 256, 300]))
        numerics = int(rng.choice([pkg.NUMERICS_STRICT, pkg.NUMERICS_FUSED]))
        n = int(rng.integers(3_000, 40_000 if which != "decim" else 120_000))
        if which == "window":
            nphi = int(rng.choice([2, 8, 10, 32, 33]))
            rate = float(rng.choice([1.0, 1.0 + 10 ** rng.uniform(-9, -1), rng.uniform(1.0, 2.0), rng.uniform(2.0, 12.0), math.pi / 3, 7.25]))
            h = rng.standard_normal(nphi * 32)
            x = rng.random((nch, n)) - 0.5
            mk = lambda: pkg.FIRFilter(h, rate, nphi, numerics=numerics)
            mko = lambda: O.FIRFilter(h, rate, nphi, tx=np.float64)
            env, want, tdt = {"MRHIP_ARB_WINDOW": "2"}, "arb_window_kernel", torch.float64
        else:
            ratio = Fraction(4, 1) if which == "interp" else Fraction(1, 4)
            h = (rng.standard_normal(128) / 4).astype(np.float32)
            x = (rng.standard_normal((nch, n)) + 1j * rng.standard_normal((nch, n))).astype(np.complex64)
            mk = lambda: pkg.FIRFilter(h, ratio, numerics=numerics)
            mko = lambda: O.FIRFilter(h, ratio, tx=np.complex64)
            env = {"MRHIP_INTERP_LANE": "2"} if which == "interp" else {"MRHIP_DECIM_LANE": "2"}
            want, tdt = ("interp_lane_kernel" if which == "interp" else "decim_lane_kernel"), torch.complex64
        flat = x.view(np.float32 if x.dtype == np.complex64 else np.float64)
        if rng.random() < 0.4:
            flat[rng.integers(nch), rng.integers(flat.shape[1])] = rng.choice([np.inf, -np.inf, np.nan, -0.0])
        if rng.random() < 0.3:
            c0, a0 = int(rng.integers(nch)), int(rng.integers(0, flat.shape[1] - 700))
            flat[c0, a0:a0 + 700] = -0.0
        sizes, left = [], n
        while left > 0 and len(sizes) < 8:
            k = int(rng.choice([1, 2, 15, 17, 31, 63, 64, 65, 100])) if rng.random() < 0.35 else int(rng.integers(1, max(2, left)))
            k = min(k, left)
            sizes.append(k); left -= k
        off = int(rng.integers(0, 4))
        xt = torch.zeros((nch, n + 7), dtype=tdt)
        xt[:, off:off + n] = torch.from_numpy(x)
        res = {}
        for mode, e in (("lane", env), ("generic", {"MRHIP_FORCE_GENERIC": "1"})):
            os.environ.update(e)
            xd = xt.cuda()[:, off:off + n]                       # a fresh upload for each filter: the tuned kernel sees it first
            f = mk()
            outs, pos, kern = [], 0, set()
            for sz in sizes:
                outs.append(f.filt(xd[:, pos:pos + sz]))
                kern.add(f.last_kernel_name())
                pos += sz
            st = f.state
            res[mode] = (torch.cat(outs, dim=-1).cpu().numpy(), kern, (st.phiIdx, st.inputDeficit) + ((st.phiAccumulator, st.alpha) if which == "window" else ()), np.array(f.history))
            f.close()
            for k in e:
                os.environ.pop(k)
        bits = (lambda a_: np.ascontiguousarray(a_).view(np.uint32)) if which != "window" else (lambda a_: np.ascontiguousarray(a_).view(np.uint64))
        ok = np.array_equal(bits(res["lane"][0]), bits(res["generic"][0])) and res["lane"][2] == res["generic"][2] and np.array_equal(bits(res["lane"][3]), bits(res["generic"][3]))
        if ok and numerics == pkg.NUMERICS_STRICT:
            for c in (0, nch - 1):
                fo = mko()
                yo = np.concatenate([fo.filt(x[c, a:a + sz]) for a, sz in zip(np.cumsum([0] + sizes[:-1]), sizes)])
                ok = ok and np.array_equal(bits(res["lane"][0][c]), bits(yo))
        served = want in res["lane"][1]
        tally[which] = tally.get(which, 0) + int(served)
        bad += not ok
        if not ok or case % 25 == 0:
            print(("ok  " if ok else "BAD ") + f"case {case}: {which} nch={nch} numerics={numerics} n={n} sizes={sizes} offset={off} kernels={sorted(res['lane'][1])}", flush=True)
    print(f"stress_lane_kernels: {cases} cases, through the lane kernels {tally}, {bad} mismatches", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
