#!/usr/bin/env python3
"""Every instantiation of the hand-scheduled kernels (inline-assembly LDS reads with counted waits) that uses scratch or AccVGPRs,
read from the built objects, run on the GPU against the universal kernel: a register parked between the issue of a read and its
wait would hold stale data, deterministically -- so one bit-exact run per such instantiation settles it.
    python scripts/check_spilling_instantiations.py        (GPU box; exit code 1 on any mismatch)"""
import glob
import importlib.util
import os
import re
import sys
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MRHIP_ENV_DYNAMIC"] = "1"
import numpy as np
import torch
import __graft_entry__ as ge

spec = importlib.util.spec_from_file_location("tbp", os.path.join(ROOT, "tests", "test_build_properties.py"))
tbp = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tbp)
pkg = ge.load_package()
DT = {"f": np.float32, "d": np.float64}


def run(mk, xd, fused):
    f = mk(pkg.NUMERICS_FUSED if fused else pkg.NUMERICS_STRICT)
    y = torch.cat([f.filt(xd[:, :70001]), f.filt(xd[:, 70001:])], dim=-1).cpu().numpy()
    kn = f.last_kernel_name()
    f.close()
    return y, kn


def main():
    rng = np.random.default_rng(11)
    cases = {}
    for obj in sorted(glob.glob(os.path.join(ROOT, "multirate.jl_amd", "csrc", "build", "kernels_*.hip.o"))):
        try:
            ks = tbp._kernel_scratch(obj)
        except AssertionError:
            continue
        for name, sz in ks.items():
            if not sz:
                continue
            m = re.search(r"fir_stream_kernelI([fd])([fd])Li(\d)ELi(\d+)ELb([01])E", name)
            if m:
                cases[("stream", m.group(1), m.group(2), int(m.group(3)), int(m.group(4)), int(m.group(5)))] = sz
            m = re.search(r"rational_opair_kernelILi(\d+)ELb([01])ELi(\d)ELi(\d)E([fd])([fd])E", name)
            if m:
                cases[("opair", m.group(5), m.group(6), int(m.group(3)), (int(m.group(1)), int(m.group(4))), int(m.group(2)))] = sz
    bad = 0
    for key, sz in sorted(cases.items(), key=str):
        kind, txs, r, nc, shape, fused = key
        tx = {("f", 1): np.float32, ("f", 2): np.complex64, ("d", 1): np.float64, ("d", 2): np.complex128}[(txs, nc)]
        th = DT[r]                                   # arithmetic type = promote(taps, samples): Float64 arithmetic needs Float64 taps here
        if kind == "stream":
            M = shape
            ratio, hl, want = Fraction(1, M), 6 * M + 5, "fir_stream_kernel"
        else:
            T, smin = shape
            ratio = Fraction(3, 2) if smin == 0 else Fraction(5, 7)
            hl, want = T * ratio.numerator, "rational_opair_kernel"
        hls = [hl] if kind != "stream" else [hl, shape + 3, 2 * shape + 1, 96, 16]     # (which filter lengths reach the instantiation differs)
        nch = 40
        x = (rng.standard_normal((nch, 200_000)) + (1j * rng.standard_normal((nch, 200_000)) if nc == 2 else 0)).astype(tx)
        xd = torch.from_numpy(x).cuda()
        ok, hit, kn, kg = True, False, "", ""
        for hl_try in hls:
            h = rng.standard_normal(hl_try).astype(th)
            mk = lambda numerics: pkg.FIRFilter(h, ratio, numerics=numerics)
            os.environ.pop("MRHIP_FORCE_GENERIC", None)
            y, kn = run(mk, xd, fused)
            os.environ["MRHIP_FORCE_GENERIC"] = "1"
            yg, kg = run(mk, xd, fused)
            os.environ.pop("MRHIP_FORCE_GENERIC", None)
            ok = ok and y.shape == yg.shape and y.tobytes() == yg.tobytes()
            hit = kn == want
            if hit:
                break
        bad += 0 if ok else 1
        print(("ok  " if ok else "BAD ") + f"{key} scratch/agpr={sz} ran {kn}{'' if hit else ' (instantiation not reached by this shape)'} vs {kg}", flush=True)
    print("instantiations with scratch:", len(cases), "mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
