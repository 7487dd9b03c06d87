#!/usr/bin/env python3
"""Randomised parity stress of the sharded filter behind the C ABI (mrhip_sharded_*: one process, several shards -- here all on device 0):
random kinds, channel counts, shard counts and ragged chunkings; host path and device path + gather against ONE unsharded FIRFilter, bit for bit.
    python scripts/stress_sharded.py [--cases 80] [--seed 1] [--seconds 200]"""
import argparse, os, sys, time
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=80); ap.add_argument("--seed", type=int, default=1); ap.add_argument("--seconds", type=float, default=200.0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t0, bad, done = time.time(), 0, 0
for case in range(a.cases):
    if time.time() - t0 > a.seconds: break
    kind = ["rational", "decimator", "interpolator", "arbitrary", "farrow"][rng.integers(5)]
    tx = [np.float32, np.complex64, np.float64][rng.integers(3)]
    nch = int(rng.integers(1, 24)); nsh = int(rng.integers(1, 5))
    if kind == "rational": h, ratio, kw = rng.standard_normal(24 * 7).astype(np.float32), Fraction(7, 9), {}
    elif kind == "decimator": h, ratio, kw = rng.standard_normal(40).astype(np.float32), Fraction(1, 5), {}
    elif kind == "interpolator": h, ratio, kw = rng.standard_normal(32 * 3).astype(np.float32), Fraction(3, 1), {}
    else: h, ratio, kw = (pkg.firdes(32 * 6, 0.45 / 32, beta=7.0) * 32).astype(np.float32), 0.8123, ({"polyorder": 3} if kind == "farrow" else {})
    nchunks = int(rng.integers(1, 5))
    sizes = [int(rng.integers(1, 30_000)) for _ in range(nchunks)]
    x = rng.standard_normal((nch, sum(sizes))).astype(np.float32)
    if np.dtype(tx).kind == "c": x = x + 1j * rng.standard_normal(x.shape).astype(np.float32)
    x = x.astype(tx)
    sh = pkg.ShardedFIRFilter(h, ratio, nch, [0] * nsh, dtype=tx, **kw)
    ref = pkg.FIRFilter(h, ratio, 32, kw.get("polyorder")).bind(tx, nch)
    if kind == "farrow":       # the same fitted polynomial bank in both (the fit itself is host arithmetic, identical; this keeps the test about sharding)
        pass
    ok, pos = True, 0
    use_dev = rng.random() < 0.5
    for s_ in sizes:
        xs = x[:, pos:pos + s_]; pos += s_
        want = ref.filt(torch.from_numpy(np.ascontiguousarray(xs)).cuda()).cpu().numpy().reshape(nch, -1)
        if use_dev:
            parts, c0 = [], 0
            for (st, cnt, dev) in sh.shards:
                parts.append(torch.from_numpy(np.ascontiguousarray(xs[st:st + cnt])).cuda() if cnt else None)
            ys = sh.filt_shards(parts)
            got = sh.gather(ys, 0); sh.synchronize(); got = got.cpu().numpy()
        else:
            got = sh.filt(xs)
        if got.shape != want.shape or not np.array_equal(got.view(np.uint8), np.ascontiguousarray(want).view(np.uint8)):
            ok = False
    if not ok:
        bad += 1
        print("MISMATCH", dict(case=case, kind=kind, tx=np.dtype(tx).name, nch=nch, nsh=nsh, sizes=sizes, dev=use_dev), flush=True)
    done += 1
    sh.close(); ref.close()
print(f"sharded stress: cases {done} mismatches {bad} in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
