#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X polyphase resampling engine.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json metric "Msamples/s in (Float32, 147//160, 24*147 taps)"): FIRRational
147//160, 3528 windowed-sinc x Kaiser(7.8562) taps (README.md:172-179 recipe), Float32 taps and
samples, 64 independent channels x 1e8 samples per channel per GPU -- the shape BASELINE.json's north star
quotes the roofline target on.  One step = one pass over the whole batch = ONE filt! call on a stateful
FIRFilter (reset before every pass).  The same batch STREAMED in 1e6-sample chunks (configs[1]'s chunking
applied to the 64-channel batch: 100 filt! calls per pass, state and history carried on the device; bit-identical
outputs) is measured right after the timed region and reported in the extra object `streamed_1e6_chunks`
(N = 1 only); `--chunk 1000000` makes it the timed workload instead.
Inputs are synthetic uniform [0,1) samples generated on the device before the timed region and
stay resident in HBM; outputs are written to a resident HBM buffer.

For N > 1 (launched by torch.distributed.run, one rank per GPU) every rank filters its own 64-channel
shard -- channels are independent, so there is no data-path collective -- and the
reported value is all ranks' input samples / max-over-ranks time ("scaling": "weak").

One JSON line is printed by rank 0.  `roofline.achieved` = algorithmic bytes per launch
(7.675 B per input sample per channel = 4 B read + 0.91875 * 4 B written) x samples per
launch / average launch duration of the dominant kernel, measured with HIP events recorded on the
launch stream around the compute-kernel launches of the timed region (mrhip_set_timing / mrhip_timing_read; every
launch when a pass is one call, every 4th in the chunked stream: the event records themselves cost a few
microseconds of stream time per launch, so bracketing every 120 us launch would slow the very throughput being measured).  `cpu_baseline` = the CPU oracle (a C port of the reference algorithm; the
reference itself is Julia-0.3 source and cannot run) on one core over a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from fractions import Fraction

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
L, M, TAPS_PER_PHI = 147, 160, 24
BYTES_PER_INPUT_SAMPLE = 4.0 + (L / M) * 4.0   # 7.675


def cpu_baseline(h, seconds_budget=20.0):
    """Time the oracle (port of the reference's filt, one thread) on a bounded sample of the workload."""
    import numpy as np
    from oracle import oracle as O
    n = 300_000_000                            # ~2.4 s per run on one core: ~10 s of CPU work in all
    x = np.random.default_rng(0).random(n, dtype=np.float32)
    f = O.FIRFilter(h, Fraction(L, M), tx=np.float32)
    f.filt(x[:1_000_000])                     # warm-up
    times = []
    t_all = time.perf_counter()
    while len(times) < 3 and (time.perf_counter() - t_all) < seconds_budget:
        f = O.FIRFilter(h, Fraction(L, M), tx=np.float32)
        t0 = time.perf_counter()
        f.filt(x)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(n / med / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"1 channel x {n} Float32 samples, 147//160, 3528 taps, median of {len(times)} runs of "
                      "oracle/multirate_oracle.c (gcc -O3, strict order, no FMA); reference is Julia 0.3 and "
                      "cannot run; README.md:172-193 quotes 17.56 Msamples/s on unnamed 2014 hardware"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)   # the clocks take a few 10 ms launches to settle (kernel trace: 12.7, 11.1, 10.5, 10.3, 10.2 ms)
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--samples", type=int, default=100_000_000, help="input samples per channel per step")
    ap.add_argument("--chunk", type=int, default=0, help="samples per channel per filt! call (0 = the whole batch in one call)")
    ap.add_argument("--numerics", choices=["strict", "fused"], default="strict")
    ap.add_argument("--time-every", type=int, default=0, help="bracket every n-th kernel launch of the timed region with HIP events "
                    "(0 = every launch when a pass is one call, every 4th when it is chunked: the brackets cost stream time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-streamed", action="store_true", help="skip the extra chunked passes reported as `streamed_1e6_chunks`")
    ap.add_argument("--no-check", action="store_true", help="skip the post-run oracle spot check (timing experiments)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import __graft_entry__ as ge

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    # one rank per GPU.  MRHIP_BENCH_BACKEND=gloo (plumbing check on a box with fewer GPUs than ranks: ranks then
    # share devices round-robin and the one timing reduction goes through the host) is never used by the driver.
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
    h = pkg.firdes(TAPS_PER_PHI * L, 0.5 / L, beta=7.8562).astype(np.float32)
    nch, n = args.channels, args.samples
    chunk = args.chunk if 0 < args.chunk < n else n
    time_every = args.time_every if args.time_every > 0 else (1 if chunk == n else 4)
    n_out_total = (n * L + M - 1) // M

    gen = torch.Generator(device=dev).manual_seed(0x4D520000 + rank)
    x = torch.empty((nch, n), dtype=torch.float32, device=dev)
    for c in range(nch):                       # per-row fill keeps the RNG scratch small
        x[c].uniform_(0.0, 1.0, generator=gen)
    y = torch.empty((nch, n_out_total), dtype=torch.float32, device=dev)

    filt = pkg.FIRFilter(h, Fraction(L, M), device=dev_index,
                         numerics=pkg.NUMERICS_FUSED if args.numerics == "fused" else pkg.NUMERICS_STRICT)
    filt.bind(np.float32, nch)

    def one_step(chunk=chunk):
        """one pass over the batch: n samples per channel through the (reset) filter in pieces of `chunk`"""
        filt.reset()
        k = 0
        for a in range(0, n, chunk):
            b = min(a + chunk, n)
            cnt = filt.next_output_count(b - a)
            filt.filt_into(y[:, k:k + cnt], x[:, a:b])
            k += cnt
        return k

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        produced = one_step()
    barrier()
    filt.set_timing(time_every)   # HIP events around every n-th launch (they cost stream time themselves)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        produced = one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    n_launch, kern_ms = filt.timing_read()
    filt.set_timing(False)
    assert produced == n_out_total, (produced, n_out_total)

    # The same batch streamed in 1e6-sample chunks (100 filt! calls per pass), reported next to the timed workload,
    # never as `value`.  Outputs are bit-identical to the single call (chunked == unchunked is a tested invariant).
    streamed = None
    if world == 1 and chunk == n and n > 1_000_000 and not args.no_streamed:
        sc = 1_000_000
        one_step(sc); torch.cuda.synchronize(dev)
        filt.set_timing(4)
        ts = time.perf_counter()
        for _ in range(args.steps):
            one_step(sc)
        torch.cuda.synchronize(dev)
        el1 = time.perf_counter() - ts
        nl1, ms1 = filt.timing_read()
        filt.set_timing(False)
        streamed = {"chunk": sc, "launches_per_step": (n + sc - 1) // sc,
                    "Msamples_per_s": round(float(nch) * n * args.steps / el1 / 1e6, 3),
                    "avg_launch_ms": round(ms1 / max(nl1, 1), 5), "launches_timed": nl1,
                    "achieved_GBps": round(nch * sc * BYTES_PER_INPUT_SAMPLE / (ms1 / max(nl1, 1) / 1e3) / 1e9, 2)}
        streamed["frac"] = round(streamed["achieved_GBps"] / HBM_PEAK_GBPS, 4)

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # light sanity check of the timed output against the oracle (checker only, after the timed region)
    if rank == 0 and not args.no_check:
        from oracle import oracle as O
        O.set_fused(args.numerics == "fused")     # the oracle mirrors the library's opt-in fused mode bit for bit
        fo = O.FIRFilter(h, Fraction(L, M), tx=np.float32)
        yo = fo.filt(x[nch - 1, :200_000].cpu().numpy())
        O.set_fused(False)
        got = y[nch - 1, :len(yo)].cpu().numpy()
        assert np.array_equal(got.view(np.uint32), yo.view(np.uint32)), "bench output differs from oracle"

    total_in = float(nch) * n * args.steps * world
    ms_per_step = elapsed / args.steps * 1e3
    value = total_in / elapsed / 1e6
    avg_launch_s = (kern_ms / 1e3) / max(n_launch, 1)
    bytes_per_launch = nch * min(chunk, n) * BYTES_PER_INPUT_SAMPLE
    achieved = bytes_per_launch / avg_launch_s / 1e9 if n_launch else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):      # PMC-measured HBM bytes of a launch of THIS size (profiles/: separate --pmc passes)
        try:
            for e in json.load(open(tpath)).get("entries", []):
                if abs(e["algorithmic_bytes_per_launch"] - bytes_per_launch) <= 1e-3 * bytes_per_launch:
                    traffic = e["hbm_bytes_per_launch"]
        except Exception:
            traffic = None

    if rank == 0:
        line = {
            "metric": "Msamples/s in (Float32, 147//160, 24*147 taps) + achieved HBM GB/s vs roofline",
            "value": round(value, 3), "unit": "Msamples/s (input samples, all channels, all GPUs)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"FIRRational 147//160, 3528 taps, Float32, {nch} channels x {n} samples per GPU, "
                                   + ("one filt! call per pass" if chunk == n else f"streamed in {chunk}-sample chunks through one stateful FIRFilter")
                                   + " (inputs and outputs resident in HBM)",
                       "channels_per_gpu": nch, "samples_per_channel": n, "chunk": chunk,
                       "numerics": args.numerics, "parallelism": f"channel-shard x{world}, no collective"},
            "output_msamples_s": round(value * L / M, 3),
            "kernel": filt.last_kernel_name(),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "avg_launch_ms": round(avg_launch_s * 1e3, 5), "launches_timed": n_launch,
                         "whole_step_GBps": round(nch * n * BYTES_PER_INPUT_SAMPLE / (ms_per_step / 1e3) / 1e9, 2)},
        }
        if streamed is not None:
            line["streamed_1e6_chunks"] = streamed
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(h)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
