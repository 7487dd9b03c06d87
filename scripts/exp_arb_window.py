#!/usr/bin/env python3
"""arb_window_kernel (kernels_arb_window.hip: FIRArbitrary, Float64 x Float64, 32 taps per phase, a lane per channel, one wave per stretch,
the window in registers) against the universal kernel, arb_lane_kernel and the oracle, bit for bit, then timed on BASELINE config 4's
shape beside arb_lane_kernel and arb_pipe_kernel.

    python scripts/exp_arb_window.py [check] [time]
"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MRHIP_ENV_DYNAMIC", "1")
import numpy as np
import torch
import __graft_entry__ as ge

pkg = ge.load_package()
from oracle import oracle as O   # the checker


def chunks(f, x, sizes):
    outs, pos = [], 0
    for s in sizes:
        outs.append(f.filt(x[..., pos:pos + s]))
        pos += s
    return outs


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


def check():
    rng = np.random.default_rng(606)
    bad = 0
    cases = [(32, 32, 64, math.pi / 3), (32, 32, 64, 1.0), (32, 32, 128, 1.5), (32, 32, 50, 2.7), (32, 32, 64, 10.3), (32, 32, 64, 1.0000001),
             (8, 32, 64, 1.9), (10, 32, 113, math.e / 2), (32, 32, 192, 7.25)]
    for (nphi, T, nch, rate) in cases:
        for th in (np.float64,):
            h = rng.standard_normal(nphi * T).astype(th)
            n = 40_000 if nch <= 64 else 21_000
            x = rng.random((nch, n)) - 0.5
            x[0, 5] = -0.0; x[0, 1000] = np.inf; x[1, 1001] = -np.inf; x[2, 17_000] = np.nan
            xd = torch.from_numpy(x).cuda()
            sizes = [9_000, 1, 17, n - 9_018 - 5_003, 5_003]
            for numerics in (pkg.NUMERICS_STRICT, pkg.NUMERICS_FUSED):
                ys = {}
                for mode, env in (("lane", {"MRHIP_ARB_SMALL_MAX": "0", "MRHIP_ARB_WINDOW": "2"}), ("pipe", {"MRHIP_ARB_WINDOW": "0", "MRHIP_ARB_SMALL_MAX": "0"}), ("generic", {"MRHIP_FORCE_GENERIC": "1"})):
                    os.environ.update(env)
                    f = pkg.FIRFilter(h, float(rate), nphi, numerics=numerics)
                    ys[mode] = (torch.cat(chunks(f, xd, sizes), dim=-1).cpu().numpy(), f.last_kernel_name(), (lambda st: (st.phiIdx, st.inputDeficit, st.phiAccumulator, st.alpha))(f.state), np.array(f.history))
                    f.close()
                    for k in env:
                        os.environ.pop(k)
                tag = f"Nphi={nphi} T={T} nch={nch} rate={rate:.6f} taps={np.dtype(th)} numerics={numerics}"
                ok = ys["lane"][1] == "arb_window_kernel" and ys["pipe"][1] == "arb_lane_kernel"
                same_g = np.array_equal(bits(ys["lane"][0]), bits(ys["generic"][0]))
                same_p = np.array_equal(bits(ys["lane"][0]), bits(ys["pipe"][0]))
                same_o = True
                if numerics == pkg.NUMERICS_STRICT:
                    for c in (0, 1, 2, nch - 1):
                        fo = O.FIRFilter(h, float(rate), nphi, tx=np.float64)
                        yo = np.concatenate(chunks(fo, x[c], sizes))
                        same_o &= np.array_equal(bits(ys["lane"][0][c]), bits(yo))
                st_ok = ys["lane"][2] == ys["generic"][2] and (ys["lane"][3] is None or np.array_equal(bits(ys["lane"][3]), bits(ys["generic"][3])))
                good = ok and same_g and same_p and same_o and st_ok
                bad += not good
                print(("ok  " if good else "BAD ") + tag + f" kernel={ys['lane'][1]}/{ys['pipe'][1]} generic={same_g} pipe={same_p} oracle={same_o} state={st_ok} outputs={ys['lane'][0].shape[1]}", flush=True)
                if not good and ok:
                    d = np.argwhere(bits(ys["lane"][0]) != bits(ys["generic"][0]))
                    print("    first mismatches (channel, output):", d[:8].tolist(), "of", len(d), flush=True)
    print("MISMATCHES" if bad else "ALL OK", bad, flush=True)
    return bad


def timed():
    harb = pkg.firdes(32 * 32, 0.45 / 32, beta=7.8562) * 32
    nch, n = 64, 10_000_000
    x = torch.rand((nch, n), dtype=torch.float64, device="cuda")
    W = "arb_window_kernel"
    for label, env in (("arb_pipe_kernel", {"MRHIP_ARB_LANE": "0", "MRHIP_ARB_WINDOW": "0"}), ("arb_lane_kernel", {"MRHIP_ARB_WINDOW": "0"}), (W, {}), (W + " stretch=64", {"MRHIP_ARB_WINDOW_STRETCH": "64"}),
                       (W + " stretch=256", {"MRHIP_ARB_WINDOW_STRETCH": "256"}), (W + " stretch=512", {"MRHIP_ARB_WINDOW_STRETCH": "512"}), ("arb_lane_kernel", {"MRHIP_ARB_WINDOW": "0"}), (W, {})):
        os.environ.update(env)
        f = pkg.FIRFilter(harb, float(math.pi / 3), 32).bind(np.float64, nch)
        y = torch.empty((nch, f.outputlength_bound(n)), dtype=torch.float64, device="cuda")
        for _ in range(2):
            f.reset(); f.filt_into(y, x)
        f.set_timing(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 4
        for _ in range(reps):
            f.reset(); f.filt_into(y, x)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        nl, ms = f.timing_read()
        gb = nch * n * (8 + 8 * math.pi / 3) / 1e9
        print(f"{label:32s} kernel={f.last_kernel_name():18s} kernel_ms={ms / reps:.4f} wall_ms={wall:.4f} frac_hbm={gb / (ms / reps * 1e-3) / 8000:.4f}", flush=True)
        f.close()
        for k in env:
            os.environ.pop(k)


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "time"]
    rc = 0
    if "check" in what:
        rc = check()
    if "time" in what and not rc:
        timed()
    sys.exit(1 if rc else 0)
