#!/usr/bin/env python3
"""Bit-exactness of fir_stream_kernel (FIRStandard / FIRDecimator, Float32 arithmetic) against the universal kernel and
the direct kernel: M in {1, 2, 4, 8}, Float32 and ComplexF32, several tap counts, chunked, multi-channel."""
import os, sys
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
ok = True
rng = np.random.default_rng(5)
for M in (1, 2, 4, 8):
    for T in (32, 33, 47, 127, 128, 129, 208, 500, 512):
        for dt, nch, n in ((torch.float32, 5, 300_000), (torch.complex64, 3, 200_000), (torch.float32, 64, 1_000_000 if T == 128 else 50_000)):
            h = rng.standard_normal(T).astype(np.float32)
            if dt == torch.complex64:
                x = torch.view_as_complex(torch.rand((nch, n, 2), device="cuda") - 0.5)
            else:
                x = torch.rand((nch, n), device="cuda") - 0.5
            sizes = [n // 2 + 7, 1, 3, n - n // 2 - 11]
            outs = {}
            for mode in ("stream", "generic"):
                os.environ.pop("MRHIP_FORCE_GENERIC", None)
                if mode == "generic":
                    os.environ["MRHIP_FORCE_GENERIC"] = "1"
                f = pkg.FIRFilter(h, Fraction(1, M))
                ys, pos = [], 0
                for s in sizes:
                    ys.append(f.filt(x[:, pos:pos + s])); pos += s
                y = torch.cat(ys, dim=1)
                outs[mode] = (torch.view_as_real(y) if dt == torch.complex64 else y, f.history.copy(), f.last_kernel_name(), (f.state.phiIdx, f.state.inputDeficit))
            os.environ.pop("MRHIP_FORCE_GENERIC", None)
            a, b = outs["stream"], outs["generic"]
            same = a[0].shape == b[0].shape and torch.equal(a[0].view(torch.int32), b[0].view(torch.int32)) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and a[3] == b[3]
            print(f"1//{M} T={T} {dt} nch={nch} n={n}: {a[2]} vs {b[2]}: {'OK' if same else 'MISMATCH'}", flush=True)
            ok = ok and same and a[2] == "fir_stream_kernel"
print("ALL OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
