#!/usr/bin/env python3
"""Variations of the 4//33 x 19 taps x 33 channels x 132 808 samples case (scripts/attic/repro_ps_4_33.py): which ingredient matters?"""
import os
import sys
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MRHIP_ENV_DYNAMIC", "1")
import numpy as np
import torch
import __graft_entry__ as ge

pkg = ge.load_package()


def run(label, nch, n, env=None, offset=0, reps=3, sync=False, fresh=False, pinned=False, generic_first=False):
    rng = np.random.default_rng(5)
    h = rng.standard_normal(19).astype(np.float32)
    x = (rng.random((nch, n + offset), dtype=np.float32) - 0.5)
    xt = torch.from_numpy(x)
    if pinned:
        xt = xt.pin_memory()
    xd = xt.cuda()[:, offset:]
    if sync:
        torch.cuda.synchronize()
    res = []
    for r_ in range(reps):
        if fresh and r_:
            xd = xt.cuda()[:, offset:]
        if generic_first:
            os.environ["MRHIP_FORCE_GENERIC"] = "1"
            f = pkg.FIRFilter(h, Fraction(4, 33)); f.filt(xd); f.close()
            os.environ.pop("MRHIP_FORCE_GENERIC")
        ys = {}
        for mode, e in (("tuned", env or {}), ("generic", {"MRHIP_FORCE_GENERIC": "1"})):
            os.environ.update(e)
            f = pkg.FIRFilter(h, Fraction(4, 33))
            ys[mode] = (f.filt(xd).cpu().numpy(), f.last_kernel_name())
            f.close()
            for k in e:
                os.environ.pop(k)
        a, b = ys["tuned"][0].view(np.uint32), ys["generic"][0].view(np.uint32)
        res.append(int((a != b).sum()))
    print(f"{label:44s} kernel={ys['tuned'][1]:30s} mismatching outputs in {reps} runs: {res}", flush=True)


run("fresh uploads", 33, 132_808, fresh=True, reps=4)
run("fresh uploads, every DMA waited for before the next (ablate 4)", 33, 132_808, fresh=True, reps=4, env={"MRHIP_PS_ABLATE": "4"})
run("fresh uploads", 33, 132_808, fresh=True, reps=4)
