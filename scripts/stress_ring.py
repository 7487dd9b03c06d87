#!/usr/bin/env python3
"""Randomised parity stress of the ring of arriving chunks (GPU box): random eligible shapes (24 / 32 taps per phase, M < 2L), sample types,
channel counts and ragged chunkings through ONE resident kernel, every chunk against the oracle's chunk loop, bit for bit.  Run it under both
completion protocols:  MRHIP_RING_FLUSH_MIN_MB=0 (every chunk by L2 write-backs) and the default (small chunks write-through).
    python scripts/stress_ring.py [--cases 60] [--seed 1] [--seconds 200]"""
import argparse, os, sys, time
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as O
pkg = ge.load_package()
TD = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64, np.dtype(np.complex64): torch.complex64, np.dtype(np.complex128): torch.complex128}
ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60); ap.add_argument("--many", action="store_true", help="80-200 small chunks per case (more chunks than ring slots) and now and then a pause longer than the idle deadline (set MRHIP_RING_IDLE_MS=30)"); ap.add_argument("--seed", type=int, default=1); ap.add_argument("--seconds", type=float, default=200.0)
a = ap.parse_args()
os.environ.setdefault("MRHIP_RING_IDLE_MS", "500")
rng = np.random.default_rng(a.seed)
t0, bad, done, resident = time.time(), 0, 0, 0
RATIOS = [Fraction(147, 160), Fraction(160, 147), Fraction(3, 2), Fraction(4, 1), Fraction(7, 9), Fraction(5, 3), Fraction(2, 3), Fraction(16, 15)]
for case in range(a.cases):
    if time.time() - t0 > a.seconds: break
    ratio = RATIOS[rng.integers(len(RATIOS))]
    T = int(rng.choice([24, 32])); L = ratio.numerator
    th, tx = [(np.float32, np.float32), (np.float32, np.complex64), (np.float64, np.float32), (np.float64, np.float64)][rng.integers(4)]
    nch = int(rng.choice([1, 1, 2, 3, 5, 8, 17]))
    nchunks = int(rng.integers(80, 200)) if a.many else int(rng.integers(3, 40))
    big = rng.random() < 0.2 and not a.many
    sizes = [int(rng.integers(1, 400_000 if big else (12_000 if a.many else 60_000))) if rng.random() > 0.15 else int(rng.integers(1, 40)) for _ in range(nchunks)]
    n = sum(sizes)
    h = (pkg.firdes(T * L, 0.45 / max(L, ratio.denominator), beta=7.0) * L).astype(th)
    x = rng.standard_normal((nch, n)).astype(np.float32)
    if np.dtype(tx).kind == "c": x = x + 1j * rng.standard_normal((nch, n)).astype(np.float32)
    x = x.astype(tx)
    if os.environ.get('STRESS_VERBOSE'): print('case', case, str(ratio), T, np.dtype(th).name, np.dtype(tx).name, nch, sizes, flush=True)
    f = pkg.FIRFilter(h, ratio, device=0).bind(tx, nch)
    fos = [O.FIRFilter(h, ratio, tx=tx) for _ in range(nch)]
    xd = torch.from_numpy(x).cuda()
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    bound = max(f.outputlength_bound(s) for s in sizes)
    ys = torch.zeros((nchunks, nch, max(bound, 1)), dtype=TD[np.dtype(f.output_dtype)], device="cuda")
    torch.cuda.current_stream().synchronize()
    counts = []
    with f.open_ring() as ring:
        resident += int(ring.info()["resident"])
        for i in range(nchunks):
            cnt, seq = ring.push(ys[i], xd[:, cuts[i]:cuts[i + 1]]); counts.append(cnt)
            if a.many and rng.random() < 0.01: time.sleep(0.08)        # longer than MRHIP_RING_IDLE_MS=30: the kernel leaves and is restarted by the next push
        ring.drain()
    got = ys.cpu().numpy()
    ok = True
    for c in range(nch):
        for i in range(nchunks):
            r = fos[c].filt(x[c, cuts[i]:cuts[i + 1]])
            if counts[i] != len(r) or not np.array_equal(got[i, c, :counts[i]].view(np.uint8), np.ascontiguousarray(r).view(np.uint8)):
                ok = False
    hist = np.asarray(f.history).reshape(nch, -1)
    ok = ok and all(np.array_equal(hist[c].view(np.uint8), np.ascontiguousarray(fos[c].history).view(np.uint8)) for c in range(nch))
    if not ok:
        bad += 1
        print("MISMATCH", dict(case=case, ratio=str(ratio), T=T, th=np.dtype(th).name, tx=np.dtype(tx).name, nch=nch, sizes=sizes[:8]), flush=True)
    done += 1
    f.close()
print(f"ring stress: cases {done} (resident kernel in {resident}) mismatches {bad} in {time.time() - t0:.0f} s, flush_min_mb={os.environ.get('MRHIP_RING_FLUSH_MIN_MB', 'default')}")
sys.exit(1 if bad else 0)
