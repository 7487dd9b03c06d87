"""poly_phase_stationary_kernel against poly_tiled_kernel on the ratios that reach it (the output-pair kernel refuses them:
M / L >= 6, L > 512, Float64 arithmetic with M / L >= 2): is the 4.8 MB kernel still worth its place?
Usage: python scripts/exp_ps_vs_tiled.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MRHIP_ENV_DYNAMIC"] = "1"
from fractions import Fraction

import numpy as np
import torch

import __graft_entry__ as ge
pkg = ge.load_package()


def run(h, ratio, x, env, reps=5):
    for k in ("MRHIP_PS", "MRHIP_OPAIR", "MRHIP_FORCE_GENERIC"):
        os.environ.pop(k, None)
    os.environ.update(env)
    f = pkg.FIRFilter(h, ratio)
    y = torch.empty((x.shape[0], f.outputlength(x.shape[1]) + 2), dtype=torch.promote_types(x.dtype, torch.from_numpy(h).dtype), device="cuda")
    f.filt_into(y, x)
    name = f.last_kernel_name()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            f.filt_into(y, x)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    nbytes = x.numel() * x.element_size() + x.shape[0] * f.outputlength(x.shape[1]) * y.element_size()
    f.close()
    return best, name, nbytes


def opair_vs_ps():
    """M / L in [2, 6): the output-pair kernel's SMIN = 2..5 instantiations (11 MB of code objects) against the phase-stationary kernel"""
    rng = np.random.default_rng(1)
    for L, M in ((2, 5), (3, 7), (3, 10), (2, 9), (3, 14), (2, 11), (5, 28)):
        for tx in (np.float32, np.complex64):
            for T in (8, 24, 32):
                h = rng.standard_normal(T * L).astype(np.float32)
                x = torch.from_numpy(rng.standard_normal((16, 4_000_000)).astype(np.float32)).cuda()
                if tx == np.complex64:
                    x = torch.view_as_complex(torch.stack([x, x.flip(1)], dim=-1).contiguous())
                out = []
                for label, env in (("default", {}), ("no-opair", {"MRHIP_OPAIR": "0"})):
                    ms, name, nbytes = run(h, Fraction(L, M), x, env)
                    out.append(f"{label}: {ms:7.3f} ms {nbytes / ms / 1e6:6.0f} GB/s {name[:28]:28s}")
                print(f"{L:4d}//{M:<4d} T={T:2d} float32 x {np.dtype(tx).name:9s} " + " | ".join(out), flush=True)


def main():
    if "--opair" in sys.argv:
        return opair_vs_ps()
    rng = np.random.default_rng(0)
    cases = [(2, 13, np.float32, np.float32), (3, 20, np.float32, np.float32), (5, 64, np.float32, np.complex64), (625, 512, np.float32, np.float32),
             (1000, 999, np.float32, np.float32), (640, 441, np.float32, np.complex64), (3, 7, np.float64, np.float64), (2, 5, np.float64, np.complex64),
             (2, 5, np.float64, np.float64), (3, 10, np.float64, np.float32), (1, 32, np.float32, np.float32)]
    for L, M, th, tx in cases:
        for T in (24, 32) if L > 1 else (1,):
            h = rng.standard_normal(T * L).astype(th)
            n = 2_000_000 if L < 100 else 1_000_000
            x = torch.from_numpy(rng.standard_normal((16, n)).astype(np.float32)).cuda()
            if tx == np.complex64:
                x = torch.view_as_complex(torch.stack([x, x.flip(1)], dim=-1).contiguous())
            elif tx == np.float64:
                x = x.double()
            out = []
            for label, env in (("default", {}), ("no-PS", {"MRHIP_PS": "0"})):
                ms, name, nbytes = run(h, Fraction(L, M), x, env)
                out.append(f"{label}: {ms:7.3f} ms {nbytes / ms / 1e6:6.0f} GB/s {name[:28]:28s}")
            print(f"{L:4d}//{M:<4d} T={T:2d} {np.dtype(th).name:7s} x {np.dtype(tx).name:9s} " + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
