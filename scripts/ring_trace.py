#!/usr/bin/env python3
"""Reads the per-workgroup trace the ring's resident kernel leaves under MRHIP_RING_OPTS=256 (MRHIP_RING_TRACE=<file>): when each
workgroup staged its real tiles, which tickets, how many idle tiles between them; row 0 is the feeder."""
import sys, numpy as np
ROWS, TILES = 512, 32
tr = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(ROWS, TILES, 3)
fe = tr[0]
nf = int((fe[:, 0] != 0).sum())
t0 = int(fe[0, 0]) if nf else int(tr[1:, :, 0][tr[1:, :, 0] != 0].min())
us = lambda v: (int(v) - t0) / 100.0
print("feeder batches (seen head move at, published at, head):", [(round(us(fe[k, 0]), 1), round(us(fe[k, 1]), 1), int(fe[k, 2])) for k in range(nf)])
dur, gap, first, last, cnt, parts = [], [], [], [], [], []
def split(v):
    v = int(v)
    return [((v >> s) & 0xffff) / 100.0 for s in (0, 16, 32, 48)]
for w in range(1, ROWS):
    n = int((tr[w, :, 0] != 0).sum())
    if not n: continue
    b = np.array([us(v) for v in tr[w, :n, 0]]); pp = np.array([split(v) for v in tr[w, :n, 1]]); e = b + pp.sum(axis=1)
    parts += list(pp[1:])
    dur += list(e - b); gap += list(b[1:] - e[:-1]); first.append(b[0]); last.append(e[-1]); cnt.append(n)
dur, gap, parts = np.array(dur), np.array(gap), np.array(parts)
pc = lambda a: [round(float(np.percentile(a, p)), 2) for p in (0, 10, 50, 90, 99, 100)]
print("workgroups", len(cnt), "real tiles", int(np.sum(cnt)), "per workgroup", min(cnt), max(cnt))
print("produce() of a real tile, us (min p10 p50 p90 p99 max):", pc(dur))
for k, name in enumerate(["chunk found", "DMA issued", "descriptor published", "prefetch + landing"]):
    print("   ...", name, pc(parts[:, k]))
print("from the end of produce() to the next real tile's (barrier, compute waves), us:", pc(gap))
print("first real tile began at:", pc(np.array(first)), " last published at:", pc(np.array(last)))
for w in [int(v) for v in sys.argv[2:]] or [1, 2, 200, 510]:
    n = int((tr[w, :, 0] != 0).sum())
    print("workgroup", w, [(round(us(tr[w, k, 0]), 1), split(tr[w, k, 1]), int(tr[w, k, 2]) >> 16, int(tr[w, k, 2]) & 0xffff) for k in range(n)])
