#!/usr/bin/env python3
"""Randomised parity stress (GPU box): random kinds / ratios / tap counts / dtypes / channel counts / chunkings,
every tuned kernel against the universal kernel (all channels) and against the CPU oracle (two channels), bit for bit.

    python tests/stress_random.py [--cases 300] [--seed 1] [--seconds 240]
Prints one line per failure and a per-kernel tally; exit code 1 on any mismatch."""
import argparse
import math
import os
import sys
import time
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (tests/ -> repo root)
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
from oracle import oracle as O

pkg = ge.load_package()


def rand(rng, shape, dt):
    if np.dtype(dt).kind == "c":
        base = np.float32 if dt == np.complex64 else np.float64
        return (rng.random(shape, dtype=base) - 0.5 + 1j * (rng.random(shape, dtype=base) - 0.5)).astype(dt)
    return (rng.random(shape, dtype=dt) - 0.5).astype(dt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--arb", type=float, default=None, help="fraction of FIRArbitrary / FIRFarrow cases (default 0.15)")
    ap.add_argument("--async-mix", type=float, default=0.0, help="fraction of the tuned filter's calls issued through filt_into_async "
                    "(planned on the device from the device-resident stream state; counts collected once per case)")
    ap.add_argument("--aligned", action="store_true", help="sample counts (and with them the channel rows) are multiples of 16 bytes: every case takes "
                    "the LDS-DMA paths of the kernels that have one (profiles/r06/experiments.md K)")
    ap.add_argument("--big", action="store_true", help="long launches: 32-96 channels x 0.5-4e6 samples, ratios the pair kernels take "
                    "(dynamic scheduling, two-stage tiles)")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    tally, bad, t0 = {}, 0, time.time()
    for case in range(args.cases):
        if time.time() - t0 > args.seconds:
            break
        arbitrary = rng.random() < (args.arb if args.arb is not None else 0.15)
        th = rng.choice([np.float32, np.float32, np.float64])
        tx = rng.choice([np.float32, np.float32, np.complex64, np.float64, np.complex128])
        if th == np.float64 and tx in (np.float32, np.complex64) and rng.random() < 0.5:
            tx = np.float64
        nch = int(rng.choice([1, 2, 3, 5, 8, 31, 32, 33, 64, 70]))
        n = int(rng.choice([1, 7, 300, 5_000, 40_000, 150_000]))
        n = max(1, int(n * (0.5 + rng.random())))
        if args.aligned:
            n = max(4, n // 4 * 4)
        if args.big:
            arbitrary = False
            th = np.float32
            tx = rng.choice([np.float32, np.float32, np.complex64])
            nch = int(rng.choice([32, 48, 64, 96]))
            n = int(rng.choice([500_000, 1_500_000, 4_000_000]) * (0.6 + 0.4 * rng.random()))
            if tx == np.complex64:
                n //= 2
        cuts = sorted(set(int(c) for c in rng.integers(0, n + 1, size=int(rng.integers(0, 4)))))
        sizes = np.diff([0] + cuts + [n]).tolist()
        if arbitrary:
            nphi = int(rng.choice([32, 8, 10]))
            T = int(rng.integers(1, 40))
            h = rng.standard_normal(T * nphi - int(rng.integers(0, nphi)) if T > 1 else nphi).astype(th)
            ratio = float(rng.choice([math.pi / 3, 0.37, 1.0, 2.5, 7.7, 0.011]))
            mk = lambda: pkg.FIRFilter(h, ratio, nphi)
            mko = lambda: O.FIRFilter(h, ratio, nphi, tx=tx)
            desc = f"arbitrary rate={ratio} Nphi={nphi} hLen={len(h)}"
            if rng.random() < 0.5 and T > 1:      # FIRFarrow: one polynomial bank (fitted by the oracle) for all three filters
                order = int(rng.integers(0, 6))
                pn = O.pfb2pnfb(O.taps2pfb(h, nphi), order)
                mk = lambda: pkg.FIRFilter(h, ratio, nphi, order, pnfb=pn)
                mko = lambda: O.FIRFilter(h, ratio, nphi, tx=tx, polyorder=order, pnfb=pn)
                desc = f"farrow rate={ratio} Nphi={nphi} hLen={len(h)} order={order}"
        else:
            kind = rng.choice(["rational", "rational", "near1", "interp", "decim", "standard", "h147"])
            if args.big:
                kind = rng.choice(["near1", "near1", "h147", "interp", "wide", "wide", "decim16"])
            if kind == "h147":
                L, M = 147, 160
            elif kind == "near1":
                M = int(rng.integers(4, 200)); L = max(1, M - int(rng.integers(1, max(2, M // 4))))
            elif kind == "interp":
                L, M = int(rng.integers(2, 40)), 1
            elif kind == "decim":
                L, M = 1, int(rng.integers(2, 120))
            elif kind == "decim16":
                L, M = 1, int(rng.integers(1, 17))
            elif kind == "wide":                 # ratios beyond (1/2, 2): L >= 2M and 2 <= M/L < 6
                while True:
                    L, M = int(rng.integers(2, 30)), int(rng.integers(1, 60))
                    if math.gcd(L, M) == 1 and (L >= 2 * M or (M >= 2 * L and M < 6 * L)):
                        break
            elif kind == "standard":
                L, M = 1, 1
            else:
                L, M = int(rng.integers(1, 40)), int(rng.integers(1, 40))
            fr = Fraction(L, M)
            L, M = fr.numerator, fr.denominator
            tmax = (33 if args.big else 44) if L > 1 else 700
            hl = max(2, int(rng.integers(1, tmax)) * L - int(rng.integers(0, L)))
            if L == 1:
                hl = int(rng.integers(2, 700)) if kind != "decim16" else int(rng.integers(16, 400))
            h = rng.standard_normal(hl).astype(th)
            mk = lambda: pkg.FIRFilter(h, Fraction(L, M))
            mko = lambda: O.FIRFilter(h, Fraction(L, M), tx=tx)
            desc = f"{L}//{M} hLen={hl}"
        desc += f" th={np.dtype(th).name} tx={np.dtype(tx).name} nch={nch} sizes={sizes}"
        x = rand(rng, (nch, n), tx)
        xd = torch.from_numpy(x).cuda()
        try:
            os.environ.pop("MRHIP_FORCE_GENERIC", None)
            f = mk()
            outs, pos = [], 0
            if args.async_mix > 0.0 and n > 0:
                # a mix of plain calls and calls nobody waits for: the latter leave their counts in a device array; the
                # host-side view of the state is stale in between and must be re-read by the next plain call
                f.bind(np.dtype(tx), nch)
                cnt = torch.full((len(sizes),), -1, dtype=torch.int64, device="cuda")
                pend = []
                tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
                       np.dtype(np.complex64): torch.complex64, np.dtype(np.complex128): torch.complex128}[np.dtype(f.output_dtype)]
                for i_, s_ in enumerate(sizes):
                    if rng.random() < args.async_mix:
                        yb = torch.empty((nch, f.outputlength_bound(s_)), dtype=tdt, device="cuda")
                        f.filt_into_async(yb, xd[:, pos:pos + s_], cnt[i_:i_ + 1])
                        pend.append((len(outs), i_, yb))
                        outs.append(None)
                    else:
                        outs.append(f.filt(xd[:, pos:pos + s_]))
                    pos += s_
                f.sync_state()
                c_ = cnt.cpu().tolist()
                for slot, i_, yb in pend:
                    outs[slot] = yb[:, :max(c_[i_], 0)] if s_ >= 0 else yb[:, :0]
                    if sizes[i_] == 0:
                        outs[slot] = yb[:, :0]
            else:
                for s_ in sizes:
                    outs.append(f.filt(xd[:, pos:pos + s_])); pos += s_
            y = torch.cat(outs, dim=-1).cpu().numpy()
            kname = f.last_kernel_name()
            os.environ["MRHIP_FORCE_GENERIC"] = "1"
            g = mk()
            outs, pos = [], 0
            for s_ in sizes:
                outs.append(g.filt(xd[:, pos:pos + s_])); pos += s_
            yg = torch.cat(outs, dim=-1).cpu().numpy()
            os.environ.pop("MRHIP_FORCE_GENERIC", None)
            ok = y.shape == yg.shape and y.tobytes() == yg.tobytes() and np.asarray(f.history).tobytes() == np.asarray(g.history).tobytes()
            for c in sorted({0, nch - 1}):
                fo = mko()
                yo = np.concatenate([fo.filt(p) for p in np.split(x[c], np.cumsum(sizes)[:-1])]) if n else np.zeros(0, y.dtype)
                ok = ok and yo.shape == y[c].shape and yo.tobytes() == y[c].tobytes()
            f.close(); g.close()
        except Exception as e:      # constructor/argument errors must agree with the oracle's: report
            ok, kname = False, f"EXC {type(e).__name__}: {e}"
        tally[kname] = tally.get(kname, 0) + 1
        if not ok:
            bad += 1
            print("MISMATCH", desc, kname, flush=True)
    print("cases", sum(tally.values()), "mismatches", bad, "kernels", tally, flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
