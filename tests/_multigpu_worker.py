"""One rank of the multi-GPU parity test (started by tests/test_multigpu.py, one process per rank).

Environment: RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT, MRHIP_TEST_BACKEND (nccl: one GPU per rank, RCCL;
gloo: ranks share the visible GPUs round-robin and the collectives run on host copies -- the plumbing check for a
1-GPU box).  The filter is the real HIP engine behind ChannelShardedFilter; the oracle is the checker."""
import os
import sys
from fractions import Fraction

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    from oracle import oracle as O

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("MRHIP_TEST_BACKEND", "nccl")
    nch, n = int(os.environ.get("MRHIP_TEST_CHANNELS", "37")), int(os.environ.get("MRHIP_TEST_SAMPLES", "50000"))
    dev_index = rank if backend == "nccl" else rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = ge.load_package()

    if os.environ.get("MRHIP_TEST_MODE", "channels") == "time":
        return time_sharded(pkg, O, dist, torch, rank, world, backend, dev_index, dev)

    L, M = 147, 160
    h = pkg.firdes(24 * L, 0.5 / L, beta=7.8562).astype(np.float32)
    rng = np.random.default_rng(1234)                 # every rank draws the same global signal
    x = (rng.random((nch, n), dtype=np.float32) + 1j * rng.random((nch, n), dtype=np.float32)).astype(np.complex64)
    sf = pkg.ChannelShardedFilter(h, Fraction(L, M), nch, device=dev_index)
    assert (sf.rank, sf.world_size) == (rank, world)
    xl = torch.from_numpy(np.ascontiguousarray(sf.local_slice(x))).to(dev)
    cut = 17_003                                      # two calls: phase state and history carried per shard
    y_local = torch.cat([sf.filt(xl[:, :cut].contiguous()), sf.filt(xl[:, cut:].contiguous())], dim=1)
    if sf.filter is not None:
        assert sf.filter.last_kernel_name() != ""
    yc = y_local if backend == "nccl" else y_local.cpu()
    full = sf.gather(yc, dst=0)
    allg = sf.all_gather(yc)
    n_out = (n * L + M - 1) // M
    assert allg.shape == (nch, n_out), (allg.shape, nch, n_out)
    allg = allg.cpu().numpy()
    # oracle on a few channels: the first, the last, and the first channel of every shard boundary this rank sees
    check = sorted(c for c in {0, nch - 1, sf.start, sf.start + sf.count - 1, nch // 2} if 0 <= c < nch)
    for c in check:
        fo = O.FIRFilter(h, Fraction(L, M), tx=np.complex64)
        ref = np.concatenate([fo.filt(x[c, :cut]), fo.filt(x[c, cut:])])
        assert ref.shape == (n_out,)
        assert np.array_equal(allg[c].view(np.uint32), ref.view(np.uint32)), f"rank {rank}: all_gather channel {c} differs from the oracle"
        if rank == 0:
            assert np.array_equal(full[c].cpu().numpy().view(np.uint32), ref.view(np.uint32)), f"root gather channel {c} differs"
    if rank != 0:
        assert full is None
    formed = torch.ones(1, dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(formed)
    assert int(formed.item()) == world
    if rank == 0:
        print(f"MULTIGPU_OK ranks={int(formed.item())} backend={backend} channels={nch} shard0={sf.count}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


def time_sharded(pkg, O, dist, torch, rank, world, backend, dev_index, dev):
    """ONE stream per channel split along time over the ranks (TimeShardedFilter): state entered with mrhip_advance_state,
    a tapsPerPhi-1 halo from the left neighbour; the result must be the chunk loop's with the same boundaries."""
    import math
    rng = np.random.default_rng(99)
    n = int(os.environ.get("MRHIP_TEST_SAMPLES", "200000"))
    cases = [(Fraction(147, 160), pkg.firdes(24 * 147, 0.5 / 147, beta=7.8562).astype(np.float32), np.float32, 3, 160),
             (Fraction(1, 4), pkg.firdes(128, 0.125, beta=7.0).astype(np.float32), np.complex64, 2, 4),
             (Fraction(4, 1), pkg.firdes(128, 0.125, beta=7.0).astype(np.float32), np.float32, 2, 1),
             (Fraction(1, 1), pkg.firdes(65, 0.2, beta=7.0).astype(np.float64), np.float64, 1, 1),
             (float(math.pi / 3), (pkg.firdes(32 * 12, 0.45 / 32, beta=7.0) * 32).astype(np.float32), np.float32, 2, 1)]
    for ratio, h, tx, nch, mult in cases:
        x = rng.standard_normal((nch, n)).astype(np.float32)
        if np.dtype(tx).kind == "c":
            x = (x + 1j * rng.standard_normal((nch, n)).astype(np.float32))
        x = x.astype(tx)
        ts = pkg.TimeShardedFilter(h, ratio, n, device=dev_index, multiple=mult)
        assert (ts.rank, ts.world_size) == (rank, world)
        # two consecutive blocks of the stream: the second enters with the state and the halo the first left (the last
        # rank's tail feeds rank 0); over RCCL the halo stays in device memory (mrhip_set_history_device)
        fos = {c: (O.FIRFilter(h, ratio, 32, tx=tx) if isinstance(ratio, float) else O.FIRFilter(h, ratio, tx=tx)) for c in (0, nch - 1)}
        xl = torch.empty(ts.local_slice(x).shape, dtype=torch.from_numpy(x[:1, :1]).dtype, device=dev)
        for blk in range(2):
            xb = x if blk == 0 else (np.roll(x, 101, axis=1) * np.float32(0.75)).astype(tx)
            # ONE device buffer refilled in place block after block, as a streaming caller does (the last
            # rank's kept tail was a view of it)
            xl.copy_(torch.from_numpy(np.ascontiguousarray(ts.local_slice(xb))))
            y_local = ts.filt(xl)
            full = ts.gather(y_local if backend == "nccl" else y_local.cpu(), dst=0)
            if rank == 0:
                for c in sorted(fos):
                    ref = np.concatenate([fos[c].filt(xb[c, a:a + m]) for a, m in ts.slices])
                    got = full[c].cpu().numpy()
                    assert got.shape == ref.shape and got.dtype == ref.dtype, (ratio, got.shape, ref.shape, got.dtype, ref.dtype)
                    assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)), f"time-sharded {ratio} block {blk} channel {c} differs from the chunk loop"
            else:
                assert full is None
        ts.filter.close()
    formed = torch.ones(1, dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(formed)
    if rank == 0:
        print(f"MULTIGPU_OK ranks={int(formed.item())} backend={backend} mode=time cases={len(cases)}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
