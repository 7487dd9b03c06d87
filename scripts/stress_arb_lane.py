#!/usr/bin/env python3
"""Randomised parity stress of arb_lane_kernel (kernels_arb_lane.hip): random channel counts (partial last groups), rates >= 1 on both
sides of where consecutive windows coincide, tapsPerPhi 16 / 32, random Nphi, Float64 / Float32 taps, STRICT / FUSED, random chunkings
(one-sample calls, calls shorter than the history, calls of a few hundred thousand samples), synchronous / asynchronous calls --
outputs, end state and history against the universal kernel, bit for bit, and a few channels against the oracle.

    python scripts/stress_arb_lane.py [cases] [seed]
"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MRHIP_ENV_DYNAMIC", "1")
os.environ["MRHIP_ARB_SMALL_MAX"] = "0"
import numpy as np
import torch
import __graft_entry__ as ge

pkg = ge.load_package()
from oracle import oracle as O   # the checker


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20261005
    rng = np.random.default_rng(seed)
    bad = lane_runs = 0
    for case in range(cases):
        T = int(rng.choice([16, 32]))
        nphi = int(rng.choice([2, 5, 8, 10, 16, 32, 33, 64]))
        nch = int(rng.choice([48, 50, 63, 64, 64, 64, 65 + 47, 127, 128, 192, 200]))
        rate = float(rng.choice([1.0, 1.0 + 10 ** rng.uniform(-9, -1), rng.uniform(1.0, 2.0), rng.uniform(2.0, 12.0), math.pi / 3, 1.5, 2.0, 7.25]))
        th = rng.choice([np.float64, np.float32])
        numerics = int(rng.choice([pkg.NUMERICS_STRICT, pkg.NUMERICS_FUSED]))
        n = int(rng.integers(3_000, 60_000))
        h = rng.standard_normal(nphi * T).astype(th)
        x = rng.random((nch, n)) - 0.5
        if rng.random() < 0.3:
            x[rng.integers(nch), rng.integers(n)] = rng.choice([np.inf, -np.inf, np.nan, -0.0])
        # chunking: a few big pieces with small ones thrown in
        sizes, left = [], n
        while left > 0 and len(sizes) < 8:
            k = int(rng.choice([1, 2, 15, 17, 31, 64, 100])) if rng.random() < 0.35 else int(rng.integers(1, max(2, left)))
            k = min(k, left)
            sizes.append(k); left -= k
        xd = torch.from_numpy(x).cuda()
        use_async = rng.random() < 0.3
        res = {}
        for mode, env in (("lane", {}), ("generic", {"MRHIP_FORCE_GENERIC": "1"})):
            os.environ.update(env)
            f = pkg.FIRFilter(h, rate, nphi, numerics=numerics)
            outs, pos, kern = [], 0, set()
            for sz in sizes:
                piece = xd[:, pos:pos + sz]
                if use_async and mode == "lane":
                    f.bind(np.float64, nch)
                    y = torch.empty((nch, max(f.outputlength_bound(sz), 1)), dtype=torch.float64, device="cuda")
                    f.filt_into_async(y, piece)
                    outs.append(y[:, :f.sync_state()].clone())
                else:
                    outs.append(f.filt(piece))
                kern.add(f.last_kernel_name())
                pos += sz
            st = f.state
            res[mode] = (torch.cat(outs, dim=-1).cpu().numpy(), kern, (st.phiIdx, st.inputDeficit, st.phiAccumulator, st.alpha), np.array(f.history))
            f.close()
            for k in env:
                os.environ.pop(k)
        ok = np.array_equal(bits(res["lane"][0]), bits(res["generic"][0])) and res["lane"][2] == res["generic"][2] and np.array_equal(bits(res["lane"][3]), bits(res["generic"][3]))
        if ok and numerics == pkg.NUMERICS_STRICT:
            for c in (0, nch - 1):
                fo = O.FIRFilter(h, rate, nphi, tx=np.float64)
                yo = np.concatenate([fo.filt(x[c, a:a + sz]) for a, sz in zip(np.cumsum([0] + sizes[:-1]), sizes)])
                ok = ok and np.array_equal(bits(res["lane"][0][c]), bits(yo))
        lane_runs += "arb_lane_kernel" in res["lane"][1]
        bad += not ok
        print(("ok  " if ok else "BAD ") + f"case {case}: T={T} Nphi={nphi} nch={nch} rate={rate!r} taps={np.dtype(th)} numerics={numerics} n={n} sizes={sizes} async={use_async} kernels={sorted(res['lane'][1])}", flush=True)
    print(f"stress_arb_lane: {cases} cases, {lane_runs} through arb_lane_kernel, {bad} mismatches", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
