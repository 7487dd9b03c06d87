#!/usr/bin/env python3
"""Per-wave records of one launch of the output-pair kernel (developer build: make -C multirate.jl_amd/csrc PROBE=1).
For every wave: hardware id (XCC, SE, CU, SIMD), total cycles, cycles spent at the tile barrier, tiles.  Prints how
the compute and loader waves of the resident workgroups are spread over the SIMDs of a CU and how the time at the
barrier depends on it.
    MRHIP_LIB_PATH=multirate.jl_amd/libmultirate_hip_probe.so python scripts/exp_probe.py [--long N] [--ratio 147/160]
"""
import collections, json, os, sys
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MRHIP_LIB_PATH", os.path.join(ROOT, "multirate.jl_amd", "libmultirate_hip_probe.so"))
out = os.path.join(ROOT, "gpurun_out", "probe.bin")
os.makedirs(os.path.dirname(out), exist_ok=True)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
args = sys.argv[1:]
ratio, nlong, nch = Fraction(147, 160), 20_000_000, 64
while args:
    k = args.pop(0); v = args.pop(0)
    if k == "--ratio": ratio = Fraction(v)
    elif k == "--long": nlong = int(v)
    elif k == "--channels": nch = int(v)
L, M = ratio.numerator, ratio.denominator
h = pkg.firdes(24 * L, 0.5 / max(L, M), beta=7.8562).astype(np.float32)
x = torch.rand((nch, nlong), device="cuda", dtype=torch.float32)
y = torch.empty((nch, nlong * L // M + 16), device="cuda", dtype=torch.float32)
f = pkg.FIRFilter(h, ratio).bind(np.float32, nch)
for rep in range(3):
    if rep == 2: os.environ["MRHIP_PROBE_OUT"] = out
    f.reset(); f.filt_into(y, x)
torch.cuda.synchronize()
os.environ.pop("MRHIP_PROBE_OUT")
rec = np.fromfile(out, dtype=np.uint64).reshape(-1, 8, 4)
hw = rec[:, :, 0]
valid = rec[:, :, 1] > 0
loader = (hw >> np.uint64(63)) & np.uint64(1)
xcc = (hw >> np.uint64(32)) & np.uint64(0xF)
simd = (hw >> np.uint64(4)) & np.uint64(3)
cu = (hw >> np.uint64(8)) & np.uint64(0xF)
sh = (hw >> np.uint64(12)) & np.uint64(1)
se = (hw >> np.uint64(13)) & np.uint64(7)
cukey = ((xcc * 8 + se) * 2 + sh) * 16 + cu
tot, wait = rec[:, :, 1].astype(np.float64), rec[:, :, 2].astype(np.float64)
wallt = (rec[:, :7, 3] >> np.uint64(32)).astype(np.float64)
rec[:, :7, 3] &= np.uint64(0xffffffff)
print(f'in-kernel clock: {np.median(tot[:, :7] / wallt) * 100:.0f} MHz; wall of the longest wave {wallt.max() / 100:.1f} us')
# compute waves per (CU, SIMD)
per_simd = collections.Counter(); per_simd_ld = collections.Counter(); wg_per_cu = collections.Counter()
for b in range(rec.shape[0]):
    seen = set()
    for w in range(8):
        if not valid[b, w]: continue
        k = (int(cukey[b, w]), int(simd[b, w]))
        if loader[b, w]: per_simd_ld[k] += 1
        else: per_simd[k] += 1
        seen.add(int(cukey[b, w]))
    for c in seen: wg_per_cu[c] += 1
print("workgroups:", rec.shape[0], "CUs seen:", len(wg_per_cu), "workgroups per CU:", dict(collections.Counter(wg_per_cu.values())))
pat = collections.Counter()
for c in wg_per_cu:
    pat[tuple(sorted(per_simd[(c, s)] for s in range(4)))] += 1
print("compute waves per SIMD (sorted) -> CUs:", dict(pat))
patl = collections.Counter()
for c in wg_per_cu:
    patl[tuple(sorted(per_simd_ld[(c, s)] for s in range(4)))] += 1
print("loader waves per SIMD (sorted) -> CUs:", dict(patl))
# barrier share of a compute wave by how many compute waves share its SIMD
by_n = collections.defaultdict(list)
for b in range(rec.shape[0]):
    for w in range(8):
        if valid[b, w] and not loader[b, w]:
            n = per_simd[(int(cukey[b, w]), int(simd[b, w]))]
            by_n[n].append((wait[b, w] / tot[b, w], tot[b, w], rec[b, w, 3]))
for n in sorted(by_n):
    a = np.array(by_n[n])
    print(f"compute waves on a SIMD holding {n} compute waves: {len(a)} waves, barrier share mean {a[:,0].mean():.3f} (min {a[:,0].min():.3f} max {a[:,0].max():.3f}), total cycles {a[:,1].mean():.0f}, tiles {a[:,2].mean():.1f}")
ld = [(wait[b, 7] / tot[b, 7], rec[b, 7, 3] / tot[b, 7]) for b in range(rec.shape[0]) if valid[b, 7] and loader[b, 7]]
ld = np.array(ld)
print(f"loader waves: barrier share {ld[:,0].mean():.3f}, vmcnt share {ld[:,1].mean():.3f}")
# per-workgroup: slowest wave decides; spread of barrier share inside a workgroup
spread = []
for b in range(rec.shape[0]):
    ws = [wait[b, w] / tot[b, w] for w in range(8) if valid[b, w] and not loader[b, w]]
    if ws: spread.append((min(ws), max(ws)))
spread = np.array(spread)
print(f"inside a workgroup: least-waiting wave {spread[:,0].mean():.3f}, most-waiting wave {spread[:,1].mean():.3f} of its time at the barrier")
