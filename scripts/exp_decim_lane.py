#!/usr/bin/env python3
"""decim_lane_kernel (kernels_decim_lane.hip: FIRDecimator 1//4, 128 taps, ComplexF32, a lane per channel in transposed form) against the
universal kernel, fir_stream_kernel and the oracle, bit for bit, then timed on BASELINE config 3b's shape beside fir_stream_kernel.

    python scripts/exp_decim_lane.py [check] [time] [sweep]
"""
import os
import sys
import time
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MRHIP_ENV_DYNAMIC", "1")
import numpy as np
import torch
import __graft_entry__ as ge

pkg = ge.load_package()
from oracle import oracle as O   # the checker


def chunks(f, x, sizes):
    outs, pos = [], 0
    for s in sizes:
        outs.append(f.filt(x[..., pos:pos + s]))
        pos += s
    return outs


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def check():
    rng = np.random.default_rng(707)
    bad = 0
    for nch in (64, 256, 50, 113, 192):
        h = (rng.standard_normal(128) / 4).astype(np.float32)
        n = 30_000 if nch <= 64 else 9_000
        x = (rng.standard_normal((nch, n)) + 1j * rng.standard_normal((nch, n))).astype(np.complex64)
        flat = x.view(np.float32)
        flat[0, 10] = -0.0; flat[0, 2001] = np.inf; flat[1, 2003] = -np.inf; flat[2, 5001] = np.nan
        xd = torch.from_numpy(x).cuda()
        sizes = [3_001, 1, 17, 64, 65, 2, 3, n - 3_153 - 1_003, 1_003]
        for numerics in (pkg.NUMERICS_STRICT, pkg.NUMERICS_FUSED):
            ys = {}
            for mode, env in (("lane", {"MRHIP_DECIM_LANE": "2"}), ("stream", {"MRHIP_DECIM_LANE": "0"}), ("generic", {"MRHIP_FORCE_GENERIC": "1"})):
                os.environ.update(env)
                f = pkg.FIRFilter(h, Fraction(1, 4), numerics=numerics)
                outs = chunks(f, xd, sizes)
                ys[mode] = (torch.cat(outs, dim=-1).cpu().numpy(), f.last_kernel_name(), np.array(f.history))
                f.close()
                for k in env:
                    os.environ.pop(k)
            tag = f"nch={nch} numerics={numerics}"
            same_g = np.array_equal(bits(ys["lane"][0]), bits(ys["generic"][0]))
            same_s = np.array_equal(bits(ys["lane"][0]), bits(ys["stream"][0]))
            same_h = np.array_equal(bits(ys["lane"][2]), bits(ys["generic"][2]))
            same_o = True
            if numerics == pkg.NUMERICS_STRICT:
                for c in (0, 1, 2, nch - 1):
                    fo = O.FIRFilter(h, Fraction(1, 4), tx=np.complex64)
                    same_o &= np.array_equal(bits(ys["lane"][0][c]), bits(np.concatenate(chunks(fo, x[c], sizes))))
            good = same_g and same_s and same_o and same_h
            bad += not good
            print(("ok  " if good else "BAD ") + tag + f" kernels={ys['lane'][1]}/{ys['stream'][1]}/{ys['generic'][1]} generic={same_g} stream={same_s} oracle={same_o} history={same_h}", flush=True)
            if not good:
                d = np.argwhere(bits(ys["lane"][0]) != bits(ys["generic"][0]))
                print("    first mismatches (channel, word):", d[:8].tolist(), "of", len(d), flush=True)
    print("MISMATCHES" if bad else "ALL OK", bad, flush=True)
    return bad


def timed():
    h = pkg.firdes(128, 0.5 / 4, beta=7.8562).astype(np.float32)
    nch, n = 256, 1_000_000
    x = torch.view_as_complex(torch.rand((nch, n, 2), dtype=torch.float32, device="cuda"))
    L = "decim_lane_kernel"
    for label, env in (("fir_stream_kernel", {"MRHIP_DECIM_LANE": "0"}), (L, {"MRHIP_DECIM_LANE": "1"})) + tuple((L + f" stretch<={st}", {"MRHIP_DECIM_LANE": "1", "MRHIP_DECIM_STRETCH": str(st)}) for st in (512, 176, 112)) + (("fir_stream_kernel", {"MRHIP_DECIM_LANE": "0"}), (L, {"MRHIP_DECIM_LANE": "1"})):
        for numerics in (pkg.NUMERICS_STRICT, pkg.NUMERICS_FUSED):
            os.environ.update(env)
            f = pkg.FIRFilter(h, Fraction(1, 4), numerics=numerics).bind(np.complex64, nch)
            y = torch.empty((nch, n // 4 + 8), dtype=torch.complex64, device="cuda")
            for _ in range(2):
                f.reset(); f.filt_into(y, x)
            f.set_timing(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                f.reset(); f.filt_into(y, x)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / reps * 1e3
            nl, ms = f.timing_read()
            gb = nch * n * 10.0 / 1e9
            print(f"{label:34s} numerics={numerics} kernel={f.last_kernel_name():22s} kernel_ms={ms / reps:.4f} wall_ms={wall:.4f} frac_hbm={gb / (ms / reps * 1e-3) / 8000:.4f}", flush=True)
            f.close()
            for k in env:
                os.environ.pop(k)


def sweep():
    """where the one-wave-per-stretch kernel stops paying: short calls leave most of the chip's wave slots empty"""
    h = pkg.firdes(128, 0.5 / 4, beta=7.8562).astype(np.float32)
    for nch in (256, 64):
        for n in (2_000, 10_000, 30_000, 100_000, 300_000, 1_000_000):
            x = torch.view_as_complex(torch.rand((nch, n, 2), dtype=torch.float32, device="cuda"))
            y = torch.empty((nch, n // 4 + 8), dtype=torch.complex64, device="cuda")
            row = []
            for label, env in (("stream", {"MRHIP_DECIM_LANE": "0"}), ("lane", {"MRHIP_DECIM_LANE": "2"})):
                os.environ.update(env)
                f = pkg.FIRFilter(h, Fraction(1, 4)).bind(np.complex64, nch)
                for _ in range(2):
                    f.reset(); f.filt_into(y, x)
                f.set_timing(True)
                reps = 5
                for _ in range(reps):
                    f.reset(); f.filt_into(y, x)
                torch.cuda.synchronize()
                nl, ms = f.timing_read()
                row.append(f"{label}={ms / reps * 1e3:.1f}us")
                f.close()
                for k in env:
                    os.environ.pop(k)
            print(f"nch={nch} n={n}: " + "  ".join(row), flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "time"]
    if "sweep" in what:
        sweep()
    rc = 0
    if "check" in what:
        rc = check()
    if "time" in what and not rc:
        timed()
    sys.exit(1 if rc else 0)
