#!/usr/bin/env python3
"""Randomised parity stress of captured FIRArbitrary / FIRFarrow / rational calls (HIP graphs replayed with fresh input): random rates, chunk sizes
(also shorter than the history: the filter kernel then does not write the history itself), channel counts, calls per graph and replay counts,
against a loop of plain calls on a second FIRFilter, bit for bit, counts and end state included.
    python scripts/stress_graph.py [--cases 60] [--seed 1] [--seconds 200]"""
import argparse, os, sys, time
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
TD = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64, np.dtype(np.complex64): torch.complex64, np.dtype(np.complex128): torch.complex128}
ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60); ap.add_argument("--seed", type=int, default=1); ap.add_argument("--seconds", type=float, default=200.0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t0, bad, done = time.time(), 0, 0
for case in range(a.cases):
    if time.time() - t0 > a.seconds: break
    kind = ["arbitrary", "farrow", "rational", "decimator"][rng.integers(4)]
    tx = [np.float32, np.complex64, np.float64][rng.integers(3)]
    nch = int(rng.choice([1, 1, 2, 3]))
    if kind == "rational": h, ratio, po = (rng.standard_normal(24 * 7) / 4).astype(np.float32), Fraction(7, 9), None
    elif kind == "decimator": h, ratio, po = rng.standard_normal(40).astype(np.float32), Fraction(1, 5), None
    else: h, ratio, po = (pkg.firdes(32 * 6, 0.45 / 32, beta=7.0) * 32).astype(np.float32), float([0.8123, 1 / 2.123456789, 1.25, 3.0][rng.integers(4)]), (3 if kind == "farrow" else None)
    chunk = int(rng.choice([3, 5, 100, 4_099, 20_011, 70_001, 100_003]))
    ncalls = int(rng.integers(1, 4)); nrep = int(rng.integers(3, 12))
    f = pkg.FIRFilter(h, ratio, 32, po).bind(tx, nch)
    g = pkg.FIRFilter(h, ratio, 32, po, pnfb=f.pnfb() if po else None).bind(tx, nch)
    n = chunk * ncalls * (nrep + 1)
    x = rng.standard_normal((nch, n)).astype(np.float32)
    if np.dtype(tx).kind == "c": x = x + 1j * rng.standard_normal(x.shape).astype(np.float32)
    x = x.astype(tx)
    xd = torch.from_numpy(x).cuda()
    ok, why = True, ""
    try:
        bound = f.outputlength_bound(chunk)
        # warm-up: one plain and one asynchronous call per captured size (allocations cannot be captured)
        pos = 0
        for w in range(ncalls):
            ya = f.filt(xd[:, pos:pos + chunk]); yb = g.filt(xd[:, pos:pos + chunk]); pos += chunk
            ok = ok and torch.equal(torch.view_as_real(ya.contiguous()) if ya.is_complex() else ya, torch.view_as_real(yb.contiguous()) if yb.is_complex() else yb)
        tmp = torch.empty((nch, max(bound, 1)), dtype=TD[np.dtype(f.output_dtype)], device="cuda")
        f.filt_into_async(tmp, xd[:, pos:pos + chunk]); k0 = f.sync_state()
        yb = g.filt(xd[:, pos:pos + chunk]).reshape(nch, -1); pos += chunk
        ok = ok and k0 == yb.shape[1] and torch.equal(tmp[:, :k0].contiguous().view(torch.uint8), yb.contiguous().view(torch.uint8))
        xs = torch.zeros((nch, chunk * ncalls), dtype=xd.dtype, device="cuda")
        ys = torch.zeros((ncalls, nch, max(bound, 1)), dtype=TD[np.dtype(f.output_dtype)], device="cuda")
        cnt = torch.zeros(ncalls, dtype=torch.int64, device="cuda")
        gr = torch.cuda.CUDAGraph(); st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(gr, stream=st):
            for i in range(ncalls):
                f.filt_into_async(ys[i], xs[:, i * chunk:(i + 1) * chunk], cnt[i:i + 1])
        for rep in range(nrep - 1):
            if pos + chunk * ncalls > n: break
            xs.copy_(xd[:, pos:pos + chunk * ncalls])
            gr.replay(); torch.cuda.synchronize()
            c = cnt.cpu().tolist()
            for i in range(ncalls):
                yb = g.filt(xd[:, pos:pos + chunk]).reshape(nch, -1); pos += chunk
                if c[i] != yb.shape[1] or not torch.equal(ys[i, :, :c[i]].contiguous().view(torch.uint8), yb.contiguous().view(torch.uint8)):
                    ok = False; why = f"replay {rep} call {i}: count {c[i]} vs {yb.shape[1]}"
            if not ok: break
        f.sync_state()
        sf, sg = f.state, g.state
        if ok and (sf.phiIdx, sf.inputDeficit, sf.phiAccumulator) != (sg.phiIdx, sg.inputDeficit, sg.phiAccumulator): ok = False; why = "end state"
        if ok and not np.array_equal(np.asarray(f.history).view(np.uint8), np.asarray(g.history).view(np.uint8)): ok = False; why = "history"
    except Exception as e:
        ok = False; why = "exception " + str(e)[:200]
    if not ok:
        bad += 1
        print("MISMATCH", dict(case=case, kind=kind, ratio=str(ratio), tx=np.dtype(tx).name, nch=nch, chunk=chunk, ncalls=ncalls, nrep=nrep, why=why), flush=True)
    done += 1
    f.close(); g.close()
print(f"graph stress: cases {done} mismatches {bad} in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
