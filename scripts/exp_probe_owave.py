#!/usr/bin/env python3
"""Per-wave records of one launch of the wave-autonomous kernel (developer build: make -C multirate.jl_amd/csrc PROBE=1):
total cycles, cycles waiting for the wave's own tiles (counted vmcnt), tiles.
    python scripts/exp_probe_owave.py [--long N] [--ratio 147/160] [--dtype float32|float64|complex64] [--taps64 1]
"""
import collections, os, sys
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MRHIP_LIB_PATH", os.path.join(ROOT, "multirate.jl_amd", "libmultirate_hip_probe.so"))
out = os.path.join(ROOT, "gpurun_out", "probe_owave.bin")
os.makedirs(os.path.dirname(out), exist_ok=True)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
args = sys.argv[1:]
ratio, nlong, nch, dt, taps64 = Fraction(147, 160), 20_000_000, 64, "float32", 0
while args:
    k = args.pop(0); v = args.pop(0)
    if k == "--ratio": ratio = Fraction(v)
    elif k == "--long": nlong = int(v)
    elif k == "--channels": nch = int(v)
    elif k == "--dtype": dt = v
    elif k == "--taps64": taps64 = int(v)
L, M = ratio.numerator, ratio.denominator
h = pkg.firdes(24 * L, 0.5 / max(L, M), beta=7.8562).astype(np.float64 if (taps64 or dt == "float64") else np.float32)
tdt = getattr(torch, dt)
x = torch.rand((nch, nlong), device="cuda", dtype=tdt)
f = pkg.FIRFilter(h, ratio)
for rep in range(3):
    if rep == 2: os.environ["MRHIP_PROBE_OUT"] = out
    f.reset(); y = f.filt(x)
torch.cuda.synchronize()
os.environ.pop("MRHIP_PROBE_OUT")
print("kernel:", f.last_kernel_name())
rec = np.fromfile(out, dtype=np.uint64).reshape(-1, 4)
rec = rec[rec[:, 1] > 0]
hw = rec[:, 0]
simd = (hw >> np.uint64(4)) & np.uint64(3)
tot, wait, tiles = rec[:, 1].astype(float), rec[:, 2].astype(float), (rec[:, 3] & np.uint64(0xffffffff)).astype(float)
wall = (rec[:, 3] >> np.uint64(32)).astype(float)
busy = tiles > 0
print(f'in-kernel clock (busy waves): {np.median(tot[busy] / wall[busy]) * 100:.0f} MHz; wall of the longest wave {wall.max() / 100:.1f} us; busy waves {busy.sum()}')
print(f"waves {len(rec)}  total cycles mean {tot.mean():.0f} (min {tot.min():.0f} max {tot.max():.0f})  tiles per wave mean {tiles.mean():.1f} (min {tiles.min():.0f} max {tiles.max():.0f})")
print(f"share of a wave's time waiting for its tile: mean {np.mean(wait / tot):.3f}  p10 {np.percentile(wait / tot, 10):.3f}  p90 {np.percentile(wait / tot, 90):.3f}")
print(f"cycles per tile: mean {np.mean(tot / np.maximum(tiles, 1)):.0f}; of which waiting {np.mean(wait / np.maximum(tiles, 1)):.0f}")
for s in range(4):
    m = simd == s
    print(f"  SIMD {s}: waves {m.sum()}, tiles per wave {tiles[m].mean():.1f}, wait share {np.mean(wait[m] / tot[m]):.3f}")
