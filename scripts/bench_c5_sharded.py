#!/usr/bin/env python3
"""BASELINE.json config 5: 4096-channel ComplexF32 147//160 resample sharded by channel over the GPUs of one node,
with the final gather of the outputs (RCCL over xGMI) timed separately.

    python scripts/bench_c5_sharded.py                       # 1 GPU: all 4096 channels
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        scripts/bench_c5_sharded.py [--gather root|all|none]

Total work is fixed (4096 channels x --samples, "strong" scaling): rank r filters channels shard_channels(4096, N, r).
The compute phase has no collective; `--gather root` moves every shard to rank 0 (bound by one GPU's 7 x ~153 GB/s
xGMI ingress), `--gather all` is a ring all-gather.  One JSON line from rank 0.  MRHIP_BENCH_BACKEND=gloo is the
plumbing check on a box with fewer GPUs than ranks (ranks share devices; never a measurement)."""
import argparse
import json
import os
import sys
import time
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gather", choices=["root", "all", "none"], default="root")
    args = ap.parse_args()

    import numpy as np
    import torch
    import __graft_entry__ as ge
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("MRHIP_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    pkg = ge.load_package()
    L, M = 147, 160
    h = pkg.firdes(24 * L, 0.5 / L, beta=7.8562).astype(np.float32)
    sh = pkg.ChannelShardedFilter(h, Fraction(L, M), args.channels, rank=rank, world_size=world, device=dev_index)
    n = args.samples
    x = torch.view_as_complex(torch.rand((sh.count, n, 2), dtype=torch.float32, device=dev,
                                         generator=torch.Generator(device=dev).manual_seed(0xC5 + rank)))
    n_out = (n * L + M - 1) // M
    y = torch.empty((sh.count, n_out), dtype=torch.complex64, device=dev)
    sh.filter.bind(np.complex64, sh.count)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def compute():
        sh.filter.reset()
        assert sh.filter.filt_into(y, x) == n_out

    def gather():
        if args.gather == "none" or dist is None:
            return None
        if backend != "nccl":                       # gloo plumbing check: collectives on host tensors
            yl = y.cpu()
            return sh.gather(yl) if args.gather == "root" else sh.all_gather(yl)
        return sh.gather(y) if args.gather == "root" else sh.all_gather(y)

    for _ in range(args.warmup):
        compute(); g = gather(); del g
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        compute()
    barrier()
    t_compute = (time.perf_counter() - t0) / args.steps
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g = gather(); del g
    barrier()
    t_gather = (time.perf_counter() - t0) / args.steps
    if dist is not None:
        t = torch.tensor([t_compute, t_gather], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_compute, t_gather = float(t[0]), float(t[1])
    if rank == 0:
        total_in = float(args.channels) * n
        out_bytes = float(args.channels) * n_out * 8
        print(json.dumps({
            "config": f"C5 FIRRational 147//160 ComplexF32, {args.channels} channels x {n} samples sharded by channel over {world} GPU(s)",
            "n_gpus": world, "scaling": "strong", "channels_per_gpu": sh.count, "kernel": sh.filter.last_kernel_name(),
            "compute_ms": round(t_compute * 1e3, 3), "Msamples_per_s_in": round(total_in / t_compute / 1e6, 1),
            "algorithmic_GBps_all_gpus": round(total_in * 15.35 / t_compute / 1e9, 1),
            "gather": args.gather if world > 1 else "none (1 GPU)", "gather_ms": round(t_gather * 1e3, 3),
            "gather_GBps": round(out_bytes * (world - 1) / world / t_gather / 1e9, 1) if world > 1 and t_gather > 0 and args.gather != "none" else None,
            "backend": backend}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
