"""N>1 path on CPU: world_size-2 gloo.  The shard/gather logic is the product's
(multirate.jl_amd/sharding.py); the per-rank compute is injected (oracle-backed) because the HIP
engine needs a GPU."""
import os
import socket
import sys
from fractions import Fraction

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_channels_partition(pkg):
    for nch in (1, 2, 7, 64, 4096, 4097):
        for ws in (1, 2, 3, 8):
            spans = [pkg.shard_channels(nch, ws, r) for r in range(ws)]
            assert sum(c for _, c in spans) == nch
            pos = 0
            for s, c in spans:
                assert s == pos
                pos += c
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nch, q, cplx=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    from oracle import oracle as O
    pkg = ge.load_package()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(3)
        # cplx: Float64 taps x ComplexF32 samples -> ComplexF64 outputs (the promoted type, also for an empty shard)
        h = rng.random(77).astype(np.float64 if cplx else np.float32)
        x = rng.random((nch, 400)).astype(np.float32)
        if cplx:
            x = (x + 1j * rng.random((nch, 400)).astype(np.float32)).astype(np.complex64)
        tx = np.complex64 if cplx else np.float32
        ratio = Fraction(3, 5)

        class OracleBatch:   # FIRFilter-shaped stand-in for the HIP filter (CPU test only)
            def __init__(self):
                self.f = None

            def filt(self, xl):
                xl = xl.numpy()
                if self.f is None:
                    self.f = [O.FIRFilter(h, ratio, tx=tx) for _ in range(xl.shape[0])]
                return torch.from_numpy(np.stack([f.filt(r) for f, r in zip(self.f, xl)]))

        sf = pkg.ChannelShardedFilter(h, ratio, nch, filter_factory=OracleBatch)
        xl = torch.from_numpy(sf.local_slice(x).copy())
        outs = []
        for a, b in ((0, 150), (150, 400)):          # streaming: state carried per shard
            outs.append(sf.filt(xl[:, a:b]))
        y_local = torch.cat(outs, dim=1)
        full = sf.gather(y_local, dst=0)
        allg = sf.all_gather(y_local)
        ref = np.stack([O.filt(h, x[c], ratio) for c in range(nch)])
        ok_all = allg.numpy().dtype == ref.dtype and np.array_equal(allg.numpy(), ref)
        ok_root = (full is None) if rank != 0 else (full.numpy().dtype == ref.dtype and np.array_equal(full.numpy(), ref))
        if sf.count == 0:     # a rank without channels returns an empty shard of the promoted output type
            ok_all = ok_all and y_local.shape[0] == 0 and y_local.numpy().dtype == ref.dtype
        q.put((rank, bool(ok_all), bool(ok_root), sf.start, sf.count))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nch,cplx", [(5, False), (8, False), (1, False), (4, True), (1, True)])
def test_sharded_filter_world2_gloo(pkg, O, nch, cplx):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nch, q, cplx)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res.sort()
    assert all(r[1] and r[2] for r in res), res
    assert res[0][3] == 0 and res[1][3] == res[0][4] and res[0][4] + res[1][4] == nch


# ---- time-axis sharding: one long stream split over the ranks, a tapsPerPhi-1 halo from the neighbour -----------------

def test_shard_time_partition(pkg):
    for n in (0, 1, 7, 1000, 1001):
        for ws in (1, 2, 3, 8):
            for mult in (1, 4, 160):
                spans = [pkg.shard_time(n, ws, r, mult) for r in range(ws)]
                assert sum(c for _, c in spans) == n
                pos = 0
                for s_, c in spans:
                    assert s_ == pos and (s_ % mult == 0 or c == 0)
                    pos += c


def _time_worker(rank, world, port, case, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    from oracle import oracle as O
    pkg = ge.load_package()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ratio, hl, cplx, nch, n = case
        rng = np.random.default_rng(5)
        h = rng.standard_normal(hl).astype(np.float32)
        tx = np.complex64 if cplx else np.float32
        x = rng.standard_normal((nch, n)).astype(np.float32)
        if cplx:
            x = (x + 1j * rng.standard_normal((nch, n)).astype(np.float32)).astype(np.complex64)
        mk = (lambda: O.FIRFilter(h, ratio, 32, tx=tx)) if isinstance(ratio, float) else (lambda: O.FIRFilter(h, ratio, tx=tx))

        class OracleStream:   # FIRFilter-shaped stand-in for the HIP filter (CPU test only): one oracle filter per channel
            def __init__(self):
                self.f = [mk() for _ in range(nch)]
                self.historyLen = len(self.f[0].history)

            def reset(self):
                for f in self.f:
                    f.reset()

            def advance_state(self, m):   # the state machine is data independent: run it over zeros, then drop the history
                for f in self.f:
                    f.filt(np.zeros(m, dtype=tx))
                    f.set_history(np.zeros(self.historyLen, dtype=tx))

            def set_history(self, hist):
                for f, hrow in zip(self.f, np.asarray(hist).reshape(nch, -1)):
                    f.set_history(hrow.astype(tx))

            def filt(self, xl):
                rows = [f.filt(r) for f, r in zip(self.f, xl.numpy())]
                return torch.from_numpy(np.stack(rows))

        ts = pkg.TimeShardedFilter(h, ratio, n, filter_factory=OracleStream)
        y_local = ts.filt(torch.from_numpy(ts.local_slice(x).copy()))
        full = ts.gather(y_local, dst=0)
        ok = True
        if rank == 0:
            # the reference for a time-sharded run is the caller's chunk loop with the same boundaries
            ref_rows = []
            for c in range(nch):
                fo = mk()
                ref_rows.append(np.concatenate([fo.filt(x[c, a:a + m]) for a, m in ts.slices]))
            ref = np.stack(ref_rows)
            ok = full is not None and full.numpy().dtype == ref.dtype and full.numpy().shape == ref.shape and \
                np.array_equal(full.numpy().view(np.uint8), ref.view(np.uint8))
        else:
            ok = full is None
        q.put((rank, bool(ok), ts.start, ts.count))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", [(Fraction(3, 5), 77, False, 3, 1000), (Fraction(147, 160), 147 * 4, True, 1, 2500), (Fraction(1, 4), 64, False, 2, 999),
                                  (Fraction(4, 1), 64, True, 2, 500), (Fraction(1, 1), 33, False, 1, 700), (float(np.pi / 3), 96, False, 2, 1200)])
def test_time_sharded_filter_world2_gloo(pkg, O, case):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_time_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res.sort()
    assert all(r[1] for r in res), res
    assert res[0][2] == 0 and res[1][2] == res[0][3] and res[0][3] + res[1][3] == case[4]
