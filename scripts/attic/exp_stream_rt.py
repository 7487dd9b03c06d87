"""fir_stream_rt_kernel (run-time decimation) against the per-M instantiations of fir_stream_kernel and against
fir_direct_kernel / the universal kernel, per decimation and sample type: kernel time (torch events around `reps` calls on
resident data) and the HBM rate it corresponds to.  Usage: MRHIP_ENV_DYNAMIC=1 python scripts/exp_stream_rt.py [--quick]"""
import os
import sys

os.environ.setdefault("MRHIP_ENV_DYNAMIC", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractions import Fraction

import numpy as np
import torch

import __graft_entry__ as ge
pkg = ge.load_package()


def run(h, M, x, env, reps=5):
    for k in ("MRHIP_STREAM_RT", "MRHIP_STREAM", "MRHIP_FORCE_GENERIC"):
        os.environ.pop(k, None)
    os.environ.update(env)
    f = pkg.FIRFilter(h, Fraction(1, M))
    y = torch.empty((x.shape[0], f.outputlength(x.shape[1]) + 2), dtype=torch.promote_types(x.dtype, torch.from_numpy(h).dtype), device="cuda")
    f.reset(); f.filt_into(y, x)
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
    f.close()
    return best, name, x.numel() * x.element_size() + y.numel() * y.element_size()


def single_vs_pair():
    """one output per lane against two (MRHIP_STREAM_RT_SINGLE), run-time-M kernel forced; the outputs are compared bit for bit"""
    rng = np.random.default_rng(0)
    for tx in (np.float32, np.complex64, np.float64, np.complex128):
        for M in (8, 12, 16, 20, 24, 32, 36, 48, 64, 80, 100, 120):
            for T in (24, 128):
                es = np.dtype(tx).itemsize
                if (M * es) % 8:
                    continue
                n = 8_000_000 if es <= 8 else 4_000_000
                th = np.float64 if tx in (np.float64, np.complex128) else np.float32
                h = rng.standard_normal(T).astype(th)
                x = torch.from_numpy(rng.standard_normal((16, n)).astype(np.float32)).cuda()
                if tx == np.complex64:
                    x = torch.view_as_complex(torch.stack([x, x.flip(1)], dim=-1).contiguous())
                elif tx == np.float64:
                    x = x.double()
                elif tx == np.complex128:
                    x = torch.view_as_complex(torch.stack([x.double(), x.flip(1).double()], dim=-1).contiguous())
                out, ys = {}, {}
                for label, env in (("pair", {"MRHIP_STREAM_RT": "2", "MRHIP_STREAM_RT_SINGLE": "0", "MRHIP_STREAM_RT_ONE_WG": "1"}),
                                   ("single", {"MRHIP_STREAM_RT": "2", "MRHIP_STREAM_RT_SINGLE": "1", "MRHIP_STREAM_RT_ONE_WG": "1"}),
                                   ("tiled", {"MRHIP_STREAM": "0"})):
                    for k in ("MRHIP_STREAM_RT_SINGLE", "MRHIP_STREAM_RT_ONE_WG"):
                        os.environ.pop(k, None)
                    ms, name, nbytes = run(h, M, x, env)
                    out[label] = (ms, name, nbytes / ms / 1e6)
                    f = pkg.FIRFilter(h, Fraction(1, M)); ys[label] = f.filt(x[:, :200_000]); f.close()
                same = all(torch.equal(torch.view_as_real(ys["pair"]) if ys["pair"].is_complex() else ys["pair"], torch.view_as_real(v) if v.is_complex() else v) for v in ys.values())
                print(f"{np.dtype(tx).name:10s} M={M:3d} T={T:3d} " + " | ".join(f"{k}: {v[0]:7.3f} ms {v[2]:6.0f} GB/s {v[1][:14]:14s}" for k, v in out.items()) + f" | same bits: {same}", flush=True)


def main():
    if "--single" in sys.argv:
        return single_vs_pair()
    quick = "--quick" in sys.argv
    rng = np.random.default_rng(0)
    rows = []
    cases = []
    small = "--small-m" in sys.argv
    for tx in (np.float32, np.complex64, np.float64):
        for M in (tuple(range(1, 16)) if small else (1, 2, 3, 4, 5, 8, 10, 16, 20, 32, 33, 36, 38, 48, 64, 65, 80, 100) if not quick else (1, 3, 10, 16, 36, 80)):
            for T in ((24, 128) if not quick else (128,)):
                cases.append((tx, M, T))
    for tx, M, T in cases:
        es = np.dtype(tx).itemsize
        n = min(40_000_000 // es * 4 // 4, 8_000_000)
        nch = 16
        th = np.float64 if tx == np.float64 else np.float32
        h = rng.standard_normal(T).astype(th)
        xs = rng.standard_normal((nch, n)).astype(np.float32)
        x = torch.from_numpy(xs).cuda()
        if tx == np.complex64:
            x = torch.view_as_complex(torch.stack([x, x.flip(1)], dim=-1).contiguous())
        elif tx == np.float64:
            x = x.double()
        out = {}
        for label, env in (("ct", {"MRHIP_STREAM_RT": "0"}), ("rt", {"MRHIP_STREAM_RT": "2"}), ("nostream", {"MRHIP_STREAM": "0"})):
            try:
                ms, name, nbytes = run(h, M, x, env)
                out[label] = (ms, name, nbytes / ms / 1e6)
            except Exception as e:      # noqa: BLE001
                out[label] = (float("nan"), type(e).__name__, 0.0)
        line = f"{np.dtype(tx).name:10s} M={M:3d} T={T:3d} " + " | ".join(f"{k}: {v[0]:7.3f} ms {v[2]:6.0f} GB/s {v[1][:18]:18s}" for k, v in out.items())
        print(line, flush=True)
        rows.append(line)


if __name__ == "__main__":
    main()
