#!/usr/bin/env python3
"""Randomised parity stress of the period-block mode of the output-pair kernel (512 < L <= 4096; kernels_rational_opair.hip:
plan_rational_opair_blocks): random L, M, taps per phase, types, channel counts and ragged chunkings; the tuned path against the universal
kernel (all channels) and the CPU oracle (channel 0), bit for bit, plus end state and history.
    python scripts/stress_blocks.py [--cases 120] [--seed 1] [--seconds 200]"""
import argparse, math, os, sys, time
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as O
pkg = ge.load_package()
os.environ["MRHIP_ENV_DYNAMIC"] = "1"
ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=120); ap.add_argument("--seed", type=int, default=1); ap.add_argument("--seconds", type=float, default=200.0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t0, bad, done, kernels = time.time(), 0, 0, {}

def run(f, xd, sizes):
    outs, pos = [], 0
    for s_ in sizes:
        outs.append(f.filt(xd[..., pos:pos + s_])); pos += s_
    return torch.cat(outs, dim=-1).cpu().numpy()

for case in range(a.cases):
    if time.time() - t0 > a.seconds: break
    while True:
        L = int(rng.integers(513, 4097)); M = int(rng.integers(1, 2 * L))
        if math.gcd(L, M) == 1: break
    T = int(rng.choice([3, 7, 8, 12, 16, 24, 24, 32]))
    th, tx = [(np.float32, np.float32), (np.float32, np.complex64), (np.float64, np.float32), (np.float64, np.float64), (np.float64, np.complex128)][rng.integers(5)]
    nch = int(rng.choice([1, 2, 3, 7]))
    n = int(rng.integers(20_000, 150_000))
    k = int(rng.integers(1, 6))
    cuts = sorted(set([0, n] + [int(v) for v in rng.integers(1, n, size=k)] + ([1] if rng.random() < 0.3 else [])))
    sizes = [b - a_ for a_, b in zip(cuts[:-1], cuts[1:])]
    h = (rng.standard_normal(T * L) / 8).astype(th)
    x = (rng.random((nch, n)) - 0.5).astype(np.float32 if np.dtype(tx).itemsize // (2 if np.dtype(tx).kind == "c" else 1) == 4 else np.float64)
    if np.dtype(tx).kind == "c": x = (x + 1j * (rng.random((nch, n)) - 0.5)).astype(tx)
    x = x.astype(tx)
    xd = torch.from_numpy(x).cuda()
    os.environ.pop("MRHIP_FORCE_GENERIC", None)
    f = pkg.FIRFilter(h, Fraction(L, M))
    y_t = run(f, xd, sizes)
    kn = f.last_kernel_name(); kernels[kn] = kernels.get(kn, 0) + 1
    os.environ["MRHIP_FORCE_GENERIC"] = "1"
    g = pkg.FIRFilter(h, Fraction(L, M))
    y_g = run(g, xd, sizes)
    os.environ.pop("MRHIP_FORCE_GENERIC", None)
    fo = O.FIRFilter(h, Fraction(L, M), tx=tx)
    pos, ref = 0, []
    for s_ in sizes:
        ref.append(fo.filt(x[0, pos:pos + s_])); pos += s_
    want = np.concatenate(ref)
    ok = np.array_equal(y_t.view(np.uint8), y_g.view(np.uint8)) and np.array_equal(np.ascontiguousarray(y_t.reshape(nch, -1)[0]).view(np.uint8), want.view(np.uint8))
    ok = ok and np.array_equal(np.asarray(f.history).view(np.uint8), np.asarray(g.history).view(np.uint8)) and (f.state.phiIdx, f.state.inputDeficit) == (g.state.phiIdx, g.state.inputDeficit) == (fo.state.phiIdx, fo.state.inputDeficit)
    if not ok:
        bad += 1
        print("MISMATCH", dict(case=case, L=L, M=M, T=T, th=np.dtype(th).name, tx=np.dtype(tx).name, nch=nch, sizes=sizes, kernel=kn), flush=True)
    done += 1
    f.close(); g.close()
print(f"period-block stress: cases {done} mismatches {bad} kernels {kernels} in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
