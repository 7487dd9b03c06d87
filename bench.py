#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X polyphase resampling engine.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config headline|c5]

Workload `headline` (BASELINE.json metric "Msamples/s in (Float32, 147//160, 24*147 taps)"): FIRRational
147//160, 3528 windowed-sinc x Kaiser(7.8562) taps (README.md:172-179 recipe), Float32 taps and
samples, 64 independent channels x 1e8 samples per channel per GPU -- the shape BASELINE.json's north star
quotes the roofline target on.  One step = one pass over the whole batch = ONE filt! call on a stateful
FIRFilter (reset before every pass).  The same batch STREAMED in 1e6-sample chunks (configs[1]'s chunking
applied to the 64-channel batch: 100 filt! calls per pass, state and history carried on the device; bit-identical
outputs) is measured right after the timed region and reported in the extra object `streamed_1e6_chunks`
(N = 1 only); `--chunk 1000000` makes it the timed workload instead.
Inputs are synthetic uniform [0,1) samples generated on the device before the timed region and
stay resident in HBM; outputs are written to a resident HBM buffer.

Workload `c5` (BASELINE.json configs[4]): 4096 ComplexF32 channels x 1e6 samples, 147//160, sharded by channel over the
N GPUs ("scaling": "strong": the total is fixed, rank r filters shard_channels(4096, N, r)).  `value` is the
compute-only rate (no collective on the data path); the final gather of the outputs over RCCL/xGMI is timed
separately, to rank 0 (`gather.root`) and as an all-gather (`gather.all`), and reported next to it.

Multi-GPU: one process per GPU.  Under torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE in the environment)
this process IS one rank.  Started directly as `python bench.py --gpus N` with N > 1 it is only a launcher: it
makes NO GPU call, starts N fresh child processes of this script (one per GPU, rendezvous on 127.0.0.1) and
exits with their status -- a process that has touched the GPU is never re-executed.  For `headline` every rank
filters its own 64-channel shard -- channels are independent, so there is no data-path collective --
and the reported value is all ranks' input samples / max-over-ranks time ("scaling": "weak").  `n_gpus` is the
world size the process group actually formed (asserted equal to the all-reduced rank count).

Rank 0 prints ONE compact JSON line (< 4 KB: the contract keys, `roofline` with the BASELINE rows as
name -> {kernel_ms, frac, frac_wall}, `cpu_baseline`) as the only line on stdout; the long record (every config row, the
FUSED rows, the parity object) goes to --full-out (default gpurun_out/bench_full.json).  `roofline.achieved` = algorithmic bytes per launch
(7.675 B per input sample per channel = 4 B read + 0.91875 * 4 B written; 15.35 B for ComplexF32) x
samples per launch / average launch duration of the dominant kernel, measured with HIP events recorded on the
launch stream around the compute-kernel launches of the timed region (mrhip_set_timing / mrhip_timing_read; every
launch when a pass is one call; in the chunked stream one bracket around every 10 consecutive launches, the gaps
between them included: an event pair around a single 110 us launch reads ~5 us more than the kernel's own
timestamps in a rocprofv3 trace and costs stream time itself).  `cpu_baseline` = the CPU oracle (a C port of the reference algorithm; the reference itself is
Julia-0.3 source and cannot run) on ONE core (the reference is single-threaded), median of 5 runs over a bounded
sample; `cpu_baseline_all_cores` = the same port with one channel per logical core.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from fractions import Fraction

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
L, M, TAPS_PER_PHI = 147, 160, 24
BYTES_PER_INPUT_SAMPLE = 4.0 + (L / M) * 4.0   # 7.675
BYTES_PER_INPUT_SAMPLE_C64 = 2 * BYTES_PER_INPUT_SAMPLE   # 15.35
METRIC = "Msamples/s in (Float32, 147//160, 24*147 taps) + achieved HBM GB/s vs roofline"


# ---------------------------------------------------------------------------------------------
# CPU baseline (checker-side code: the only place outside tests/ and smoke() that touches oracle/)
# ---------------------------------------------------------------------------------------------
def host_description():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    logical = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = logical
    return {"cpu_model": model, "logical_cores": logical, "usable_cores": usable}


def cpu_baseline(h, seconds_budget=25.0):
    """Time the oracle (port of the reference's filt) on a bounded sample of the workload: one thread (the
    like-for-like figure: the reference is single-threaded), then one channel per usable core."""
    import threading

    import numpy as np
    from oracle import oracle as O
    host = host_description()
    n = 300_000_000                            # ~2.4 s per run on one core: ~12 s of CPU work for 5 runs
    x = np.random.default_rng(0).random(n, dtype=np.float32)
    O.FIRFilter(h, Fraction(L, M), tx=np.float32).filt(x[:1_000_000])   # warm-up
    times = []
    t_all = time.perf_counter()
    while len(times) < 5 and (time.perf_counter() - t_all) < seconds_budget:
        f = O.FIRFilter(h, Fraction(L, M), tx=np.float32)
        t0 = time.perf_counter()
        f.filt(x)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    one = {"value": round(n / med / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
           "cpu_model": host["cpu_model"], "host_logical_cores": host["logical_cores"], "runs": len(times),
           "sample": f"1 ch x {n:.3g} f32 samples, C1 filter; median of {len(times)} runs of oracle/multirate_oracle.c, gcc -O3, 1 core",
           "note": "strict order, no FMA; the reference is Julia 0.3 and cannot run; README.md:172-193 quotes 17.56 Msamples/s (Float64 taps) "
                   "on unnamed 2014 hardware"}

    # all cores: one channel per usable core, one thread each (ctypes releases the GIL inside the C call)
    cores = max(1, host["usable_cores"])
    n2 = 4_000_000 if cores > 32 else 20_000_000      # keeps five oversubscription-free runs inside a few seconds
    xs = x[:n2]
    mtimes = []
    t_all = time.perf_counter()
    while len(mtimes) < 5 and (time.perf_counter() - t_all) < 10.0:
        filters = [O.FIRFilter(h, Fraction(L, M), tx=np.float32) for _ in range(cores)]
        threads = [threading.Thread(target=f.filt, args=(xs,)) for f in filters]
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        mtimes.append(time.perf_counter() - t0)
    mtimes.sort()
    mmed = mtimes[len(mtimes) // 2]
    allc = {"value": round(cores * n2 / mmed / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "cpu_model": host["cpu_model"], "host_logical_cores": host["logical_cores"], "runs": len(mtimes),
            "sample": f"{cores} channels x {n2} Float32 samples, one channel per core (one thread each), "
                      f"median of {len(mtimes)} runs"}
    return one, allc


def cpu_baseline_simd(h, strict_value, seconds_budget=8.0):
    """The same C port built the way the reference's own code generation treats its dot loops -- they are @simd
    (src/support.jl:9,23,26,37,47,50): vectorised, the reduction reassociated -- for THIS host's CPU (`make -C oracle simd`:
    -O3 -march=native -fassociative-math -fno-signed-zeros), one thread.  Its bits depend on the vector width: a baseline, never
    the checker; its output is compared with the strict port's to a tolerance only."""
    import ctypes as C

    import numpy as np
    from oracle import oracle as O
    try:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "simd"])
        lib = C.CDLL(os.path.join(ROOT, "oracle", "libmultirate_oracle_simd.so"))
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}
    vp, cl, ci = C.c_void_p, C.c_long, C.c_int
    lib.mro_create_rational.restype = vp
    lib.mro_create_rational.argtypes = [vp, cl, ci, cl, cl, ci]
    lib.mro_filt.restype = cl
    lib.mro_filt.argtypes = [vp, vp, cl, vp, cl]
    lib.mro_destroy.argtypes = [vp]
    n = 100_000_000
    x = np.random.default_rng(0).random(n, dtype=np.float32)
    y = np.empty(n * L // M + 16, dtype=np.float32)
    times, cnt = [], 0
    t_all = time.perf_counter()
    while len(times) < 5 and (time.perf_counter() - t_all) < seconds_budget:
        f = lib.mro_create_rational(h.ctypes.data, len(h), 0, L, M, 0)
        t0 = time.perf_counter()
        cnt = lib.mro_filt(f, x.ctypes.data, n, y.ctypes.data, len(y))
        times.append(time.perf_counter() - t0)
        lib.mro_destroy(f)
    times.sort()
    med = times[len(times) // 2]
    ref = O.FIRFilter(h, Fraction(L, M), tx=np.float32).filt(x[:200_000])
    dev = float(np.abs(y[:len(ref)] - ref).max())
    host = host_description()
    out = {"value": round(n / med / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port", "cpu_model": host["cpu_model"], "runs": len(times),
           "flags": "gcc -O3 -march=native -ffp-contract=off -fassociative-math -fno-signed-zeros -fno-trapping-math",
           "max_abs_dev_from_strict_port": dev, "outputs": int(cnt),
           "sample": f"1 channel x {n} Float32 samples, 147//160, 3528 taps, median of {len(times)} runs; the reference's dot loops are @simd "
                     "(support.jl:9), i.e. vectorised with a reassociated reduction: this build gives the compiler the same licence"}
    if strict_value:
        out["speedup_over_strict_port"] = round(out["value"] / strict_value, 2)
    return out


# What "parity" rests on.  GPU == oracle is bit-exact; oracle == reference is ALGORITHMIC: the reference is
# Julia-0.3 source that nothing in this image can run, and its @simd dot loops make its own bits compiler dependent.
PARITY_PIN = {
    "kind": "algorithmic",
    "gpu_vs_oracle": "bit-exact (every kind, Float32 / Float64 / ComplexF32 / ComplexF64 and the README's mixed case, chunked == unchunked)",
    "oracle_vs_reference": "restatement of src/support.jl:5-80 and src/Filters.jl:15-846 loop for loop, plus a second independent restatement "
                           "(tests/strict_restatement.py) bit-equal on 240 cases; no reference executable exists to compare bits with",
    "reference_held_data": ["taps2pfb([1:9], 4) doc example (src/Filters.jl:276-282)",
                            "README.md:58-141: 3//17 resampling of 1.0:100 with h = [ones(3); zeros(6)] (Float64), whole and in chunks, counts 1/4/13",
                            "nextphase table (test/runtests.jl:423-438)",
                            "FIRFarrow notebook run: 126 outputs (a length only)",
                            "all four in tests/golden/reference_known_answers.json, checked on oracle and GPU"],
    "unpinned_by_the_reference": ["every Float32 / complex / FIRArbitrary VALUE", "FIRFarrow's polynomial fit", "firdes tap values (DSP.jl un-vendored)"],
}


# ---------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` started directly (no torch.distributed.run around it)
# ---------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n):
    """Start n child processes of this script, one per GPU, and return the worst exit status.  This process
    makes no GPU call: torch.cuda.device_count() only counts (it does not create a context on this image), and
    nothing that has initialised the GPU is ever exec'ed -- the ranks are ordinary fresh children."""
    backend = os.environ.get("MRHIP_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        import torch
        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py: --gpus {n} requested but only {have} GPU(s) are visible", file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        e = dict(env)
        e.update(RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e))
    worst = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                rc = p.poll()
                if rc is None:
                    continue
                pending.remove(p)
                if rc != 0:
                    worst = worst or rc
                    for q in pending:        # one rank failed: the others would wait at the rendezvous forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return worst


# ---------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------
class Rank:
    def __init__(self, args):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
        # one rank per GPU.  MRHIP_BENCH_BACKEND=gloo (plumbing check on a box with fewer GPUs than ranks: ranks then
        # share devices round-robin and the collectives go through the host) is never used by the driver.
        self.backend = os.environ.get("MRHIP_BENCH_BACKEND", "nccl")
        self.dev_index = self.local_rank if self.backend == "nccl" else self.local_rank % torch.cuda.device_count()
        torch.cuda.set_device(self.dev_index)
        self.dev = torch.device("cuda", self.dev_index)
        self.dist = None
        self.formed = 1
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
            self.dist = dist
            # the number of ranks the communicator really has: every rank contributes 1
            one = torch.ones(1, dtype=torch.int64, device=self.coll_device())
            dist.all_reduce(one)
            self.formed = int(one.item())
            assert self.formed == dist.get_world_size() == self.world, (self.formed, self.world)

    def coll_device(self):
        return self.dev if self.backend == "nccl" else "cpu"

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, values):
        if self.dist is None:
            return list(values)
        t = self.torch.tensor(list(values), dtype=self.torch.float64, device=self.coll_device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(v) for v in t.tolist()]

    def finish(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def traffic_for(bytes_per_launch):
    """(PMC-measured HBM bytes of a launch of THIS size, where that number comes from), or (None, None).  The counters
    need their own rocprofv3 --pmc passes (separate from any trace, MI355X_MICROARCH.md), so this is a RECORDED value
    from profiles/traffic_latest.json -- measured on the same kernel and launch size by scripts/profile_round.sh, not in
    this run; `traffic_source` in the JSON line says so."""
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            doc = json.load(open(tpath))
            for e in doc.get("entries", []):
                if abs(e["algorithmic_bytes_per_launch"] - bytes_per_launch) <= 1e-3 * bytes_per_launch:
                    src = {"file": "profiles/traffic_latest.json", "from": e.get("source"), "measured": e.get("measured", doc.get("measured")),
                           "how": "recorded, not measured in this run: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the same "
                                  "kernel and launch size (FETCH_SIZE x2 on gfx950)", "kernel": doc.get("kernel")}
                    return e["hbm_bytes_per_launch"], src
        except Exception:
            return None, None
    return None, None


_BC = None


def bench_configs_module():
    global _BC
    if _BC is None:
        import importlib.util
        spec = importlib.util.spec_from_file_location("mrhip_bench_configs", os.path.join(ROOT, "scripts", "bench_configs.py"))
        _BC = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_BC)
    return _BC


def config_rows(fused=False):
    """Every other BASELINE config on this GPU (and the reference's own FIRArbitrary / FIRFarrow benchmark shape), a few
    passes each, measured by scripts/bench_configs.py's harness: kernel time from HIP events, wall time with the host.
    fused: the BASELINE rows once more under the opt-in FUSED numerics (one fma per tap)."""
    bc = bench_configs_module()
    out = []
    for r in bc.run_rows(bc.FUSED_ROWS if fused else bc.BENCH_ROWS, fused=fused):
        out.append({"name": r["config"], "kernel": r["kernel"], "launches_per_pass": r["launches_per_pass"],
                    "kernel_ms": r["kernel_ms_per_pass"], "wall_ms": r["wall_ms_per_pass_incl_host"],
                    "Msamples_per_s_in": r["Msamples_per_s_in"], "Msamples_per_s_in_wall": r["Msamples_per_s_in_wall"],
                    "achieved_GBps": r["algorithmic_GBps"], "frac": r["frac_of_8TBps"], "frac_wall": r["frac_of_8TBps_wall"],
                    "arith": r["arith"], "TFLOPs": r["TFLOPs"], "strict_valu_frac": r["frac_of_strict_valu"]})
        if "wall_ms_per_call_continuing_stream" in r:      # FIRArbitrary / FIRFarrow: wall_ms IS the continuing stream's (no schedule memo); the memo figure beside it
            out[-1]["wall_ms_continuing_stream"] = r["wall_ms_per_call_continuing_stream"]
            out[-1]["wall_ms_with_schedule_memo"] = r.get("wall_ms_with_schedule_memo")
        for k in ("us_per_chunk", "chunks_per_pass", "note"):
            if k in r and r["config"].split()[0] in ("C2r", "C2rp", "C2rq", "C2rd", "C2s", "C1"):
                out[-1][k] = r[k]
    return out


def run_headline(args, R):
    import numpy as np
    import __graft_entry__ as ge
    torch, dev, rank, world = R.torch, R.dev, R.rank, R.world
    pkg = ge.load_package()
    h = pkg.firdes(TAPS_PER_PHI * L, 0.5 / L, beta=7.8562).astype(np.float32)
    nch, n = args.channels or 64, args.samples or 100_000_000
    chunk = args.chunk if 0 < args.chunk < n else n
    time_every = args.time_every if args.time_every != 0 else (1 if chunk == n else -10)
    n_out_total = (n * L + M - 1) // M

    gen = torch.Generator(device=dev).manual_seed(0x4D520000 + rank)
    x = torch.empty((nch, n), dtype=torch.float32, device=dev)
    for c in range(nch):                       # per-row fill keeps the RNG scratch small
        x[c].uniform_(0.0, 1.0, generator=gen)
    y = torch.empty((nch, n_out_total), dtype=torch.float32, device=dev)

    filt = pkg.FIRFilter(h, Fraction(L, M), device=R.dev_index,
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

    for _ in range(args.warmup):
        produced = one_step()
    R.barrier()
    filt.set_timing(time_every)   # HIP events around every n-th launch (they cost stream time themselves)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        produced = one_step()
    R.barrier()
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
        filt.set_timing(-10)         # one HIP-event bracket around every 10 consecutive launches (gaps included)
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
        streamed["whole_step_GBps"] = round(nch * n * BYTES_PER_INPUT_SAMPLE * args.steps / el1 / 1e9, 2)

    elapsed = R.max_over_ranks([elapsed])[0]
    last_kernel = filt.last_kernel_name()

    # light sanity check of the timed output against the oracle (checker only, after the timed region)
    if rank == 0 and not args.no_check:
        from oracle import oracle as O
        O.set_fused(args.numerics == "fused")     # the oracle mirrors the library's opt-in fused mode bit for bit
        fo = O.FIRFilter(h, Fraction(L, M), tx=np.float32)
        yo = fo.filt(x[nch - 1, :200_000].cpu().numpy())
        O.set_fused(False)
        got = y[nch - 1, :len(yo)].cpu().numpy()
        assert np.array_equal(got.view(np.uint32), yo.view(np.uint32)), "bench output differs from oracle"

    # the opt-in FUSED numerics (one fma per tap, same order; the oracle has the same switch) on the SAME workload, after the
    # timed region: reported under `fused`, never as `value`
    fused_headline = None
    if world == 1 and chunk == n and args.numerics == "strict" and not args.no_configs:
        ff = pkg.FIRFilter(h, Fraction(L, M), device=R.dev_index, numerics=pkg.NUMERICS_FUSED)
        ff.bind(np.float32, nch)
        for _ in range(2):
            ff.reset(); ff.filt_into(y, x)
        ff.set_timing(1)
        torch.cuda.synchronize(dev)
        tf = time.perf_counter()
        for _ in range(args.steps):
            ff.reset(); ff.filt_into(y, x)
        torch.cuda.synchronize(dev)
        elf = time.perf_counter() - tf
        nlf, msf = ff.timing_read()
        ff.set_timing(False)
        gb = nch * n * BYTES_PER_INPUT_SAMPLE / 1e9
        fused_headline = {"kernel": ff.last_kernel_name(), "kernel_ms": round(msf / max(nlf, 1), 4), "wall_ms": round(elf / args.steps * 1e3, 4),
                          "achieved_GBps": round(gb / (msf / max(nlf, 1) / 1e3), 2), "frac": round(gb / (msf / max(nlf, 1) / 1e3) / HBM_PEAK_GBPS, 4)}
        ff.close()

    configs = fused_rows = None
    filt.close()
    del x, y
    torch.cuda.empty_cache()
    if world == 1 and chunk == n and not args.no_configs:
        try:
            configs = config_rows()
        except Exception as e:       # the headline line is the contract: a failing side measurement must not take it down
            configs = [{"name": "configs failed", "error": f"{type(e).__name__}: {e}"}]
        if args.numerics == "strict":
            try:
                fused_rows = config_rows(fused=True)
            except Exception as e:
                fused_rows = [{"name": "fused rows failed", "error": f"{type(e).__name__}: {e}"}]

    total_in = float(nch) * n * args.steps * world
    ms_per_step = elapsed / args.steps * 1e3
    value = total_in / elapsed / 1e6
    avg_launch_s = (kern_ms / 1e3) / max(n_launch, 1)
    bytes_per_launch = nch * min(chunk, n) * BYTES_PER_INPUT_SAMPLE
    achieved = bytes_per_launch / avg_launch_s / 1e9 if n_launch else 0.0

    line = None
    if rank == 0:
        traffic, traffic_source = traffic_for(bytes_per_launch)
        line = {
            "metric": METRIC,
            "value": round(value, 3), "unit": "Msamples/s (input samples, all channels, all GPUs)",
            "n_gpus": R.formed, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"FIRRational 147//160 3528 taps f32, {nch} ch x {n:.3g} samples/GPU, "
                                   + ("one filt! per pass" if chunk == n else f"{chunk}-sample chunks, one stateful FIRFilter") + ", HBM-resident",
                       "channels_per_gpu": nch, "samples_per_channel": n, "chunk": chunk,
                       "numerics": args.numerics, "parallelism": f"channel-shard x{world}, no collective",
                       "backend": R.backend if world > 1 else None},
            "output_msamples_s": round(value * L / M, 3),
            "kernel": last_kernel,
            "parity_pin": PARITY_PIN["kind"],
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "avg_launch_ms": round(avg_launch_s * 1e3, 5), "launches_timed": n_launch,
                         "whole_step_GBps": round(nch * n * BYTES_PER_INPUT_SAMPLE / (ms_per_step / 1e3) / 1e9, 2)},
            "parity": PARITY_PIN,
        }
        if fused_headline is not None or fused_rows is not None:
            fz = {"note": "opt-in MRHIP_NUMERICS_FUSED: one fma per tap in the reference's order; differs from STRICT by <= 1 rounding per tap; "
                          "bit-equal to the oracle's fused switch; never `value`"}
            if fused_headline is not None:
                fz["headline"] = fused_headline
            for r in fused_rows or []:
                if "error" in r:
                    fz["error"] = r["error"]
                else:
                    fz[r["name"].split()[0]] = {"kernel": r["kernel"], "kernel_ms": r["kernel_ms"], "wall_ms": r["wall_ms"], "frac": r["frac"]}
            line["fused"] = fz
        if streamed is not None:
            line["streamed_1e6_chunks"] = streamed
        if configs is not None:
            line["configs"] = configs
            # the BASELINE rows once more, compact and inside `roofline` (a record that keeps only the parsed top-level keys,
            # or only the end of the line, still carries them): name -> kernel ms, wall ms, fraction of the HBM roof
            base = {}
            for r in configs:
                nm = r.get("name", "")
                if nm.split()[0] in BASELINE_ROW_TAGS:
                    base[nm.split()[0]] = {"kernel": r["kernel"], "kernel_ms": r["kernel_ms"], "wall_ms": r["wall_ms"], "frac": r["frac"],
                                           "frac_wall": r["frac_wall"]}
                    if "wall_ms_continuing_stream" in r:
                        base[nm.split()[0]]["wall_ms_continuing_stream"] = r["wall_ms_continuing_stream"]
            line["roofline"]["baseline_configs"] = base
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], line["cpu_baseline_all_cores"] = cpu_baseline(h)
            line["cpu_baseline_simd"] = cpu_baseline_simd(h, line["cpu_baseline"]["value"])
    return line


def run_c5(args, R):
    """BASELINE.json configs[4]: 4096 ComplexF32 channels, 147//160, sharded by channel over the ranks (strong scaling)."""
    import numpy as np
    import __graft_entry__ as ge
    torch, dev, rank, world = R.torch, R.dev, R.rank, R.world
    pkg = ge.load_package()
    h = pkg.firdes(TAPS_PER_PHI * L, 0.5 / L, beta=7.8562).astype(np.float32)
    nch_total, n = args.channels or 4096, args.samples or 1_000_000
    n_out = (n * L + M - 1) // M
    sh = pkg.ChannelShardedFilter(h, Fraction(L, M), nch_total, rank=rank, world_size=world, device=R.dev_index)
    x = torch.view_as_complex(torch.rand((sh.count, n, 2), dtype=torch.float32, device=dev,
                                         generator=torch.Generator(device=dev).manual_seed(0xC5000000 + sh.start)))
    y = torch.empty((sh.count, n_out), dtype=torch.complex64, device=dev)
    if sh.filter is not None:
        sh.filter.bind(np.complex64, sh.count)

    def one_step():
        if sh.filter is None:
            return
        sh.filter.reset()
        got = sh.filter.filt_into(y, x)
        assert got == n_out, (got, n_out)

    for _ in range(args.warmup):
        one_step()
    R.barrier()
    if sh.filter is not None:
        sh.filter.set_timing(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    R.barrier()
    t_compute = time.perf_counter() - t0
    n_launch, kern_ms = sh.filter.timing_read() if sh.filter is not None else (0, 0.0)
    if sh.filter is not None:
        sh.filter.set_timing(False)

    # the final gather, timed on its own: to rank 0, and as an all-gather (every rank ends with all 4096 channels)
    gather = {}
    out_bytes = float(nch_total) * n_out * 8
    if world > 1 and not args.no_gather:
        yc = y if R.backend == "nccl" else y.cpu()          # gloo plumbing check: collectives on host tensors
        for kind, fn in (("root", sh.gather), ("all", sh.all_gather)):
            g = fn(yc, n_out=n_out); del g                   # warm-up (communicator set-up, allocator)
            R.barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                g = fn(yc, n_out=n_out); del g
            R.barrier()
            tg = R.max_over_ranks([(time.perf_counter() - t1) / args.steps])[0]
            moved = out_bytes * (world - 1) / world          # bytes that cross a link: into the root / into each rank
            gather[kind] = {"ms": round(tg * 1e3, 3), "GBps_into_one_gpu": round(moved / tg / 1e9, 1),
                            "compute_plus_gather_Msamples_per_s": None}
    t_compute = R.max_over_ranks([t_compute])[0]

    # parity spot check against the oracle (checker only, after the timed region): rank 0's first and last channel
    if rank == 0 and not args.no_check and sh.filter is not None:
        from oracle import oracle as O
        for c in sorted({0, sh.count - 1}):
            yo = O.FIRFilter(h, Fraction(L, M), tx=np.complex64).filt(x[c, :100_000].cpu().numpy())
            got = y[c, :len(yo)].cpu().numpy()
            assert np.array_equal(got.view(np.uint32), yo.view(np.uint32)), "bench output differs from oracle"

    if rank == 0:
        per_step = t_compute / args.steps
        total_in = float(nch_total) * n
        value = total_in / per_step / 1e6
        for kind in gather:
            gather[kind]["compute_plus_gather_Msamples_per_s"] = round(total_in / (per_step + gather[kind]["ms"] / 1e3) / 1e6, 3)
        avg_launch_s = (kern_ms / 1e3) / max(n_launch, 1)
        bytes_per_launch = sh.count * n * BYTES_PER_INPUT_SAMPLE_C64
        achieved = bytes_per_launch / avg_launch_s / 1e9 if n_launch else 0.0
        line = {
            "metric": "Msamples/s in (ComplexF32, 147//160, 24*147 taps), 4096 channels sharded by channel + achieved HBM GB/s vs roofline",
            "value": round(value, 3), "unit": "Msamples/s (input samples, all channels, all GPUs; compute only)",
            "n_gpus": R.formed, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(per_step * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C5: FIRRational 147//160 3528 taps c64 x f32 taps, {nch_total} ch x {n:.3g} samples in all, "
                                   f"channel-sharded over {world} GPU(s), HBM-resident",
                       "channels_total": nch_total, "channels_per_gpu": sh.count, "samples_per_channel": n,
                       "parallelism": f"channel-shard x{world}, no data-path collective; final gather timed separately",
                       "backend": R.backend if world > 1 else None},
            "output_msamples_s": round(value * L / M, 3),
            "kernel": sh.filter.last_kernel_name() if sh.filter is not None else None,
            "gather": gather or None,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic_for(bytes_per_launch)[0],
                         "traffic_source": traffic_for(bytes_per_launch)[1],
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "avg_launch_ms": round(avg_launch_s * 1e3, 5), "launches_timed": n_launch,
                         "note": "rank 0's kernel; per-GPU fraction (every rank runs the same shard size +-1 channel)"},
        }
        return line
    return None



# ---------------------------------------------------------------------------------------------
# what is printed: ONE compact JSON line (< 4 KB) on stdout; the long record goes to a file
# ---------------------------------------------------------------------------------------------
COMPACT_MAX = 4000
BASELINE_ROW_TAGS = ("C1", "C2", "C2s", "C2r", "C2rp", "C2rq", "C3a", "C3b", "C4", "C4f", "C5", "C2rd")


def _short(s, n=110):
    s = str(s)
    return s if len(s) <= n else s[:n - 1] + "~"


def compact_line(full, full_path=None):
    """The contract line the driver parses (bench contract keys + `roofline` + `cpu_baseline`), built from the full record:
    short strings, the BASELINE rows as name -> {kernel_ms, frac, frac_wall} inside `roofline`, nothing else.  Always
    shorter than COMPACT_MAX bytes: optional keys are dropped, last first, until it fits (round 5's 25.8 KB line left the
    driver's record unparsed)."""
    rf = full.get("roofline", {})
    cfg = full.get("config", {})
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    out["unit"] = _short(out["unit"], 60)
    out["config"] = {k: (_short(v) if isinstance(v, str) else v) for k, v in cfg.items() if v is not None}
    out["kernel"] = _short(full.get("kernel"), 90)
    out["parity_pin"] = full.get("parity_pin")
    out["roofline"] = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                                               "avg_launch_ms", "launches_timed") if k in rf}
    if rf.get("traffic_source"):
        out["roofline"]["traffic_source"] = "recorded: " + _short(rf["traffic_source"].get("file"), 50)
    if rf.get("note"):
        out["roofline"]["note"] = _short(rf["note"], 100)
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: (_short(v, 100) if isinstance(v, str) else v) for k, v in cb.items()
                               if k in ("value", "unit", "cores", "kind", "sample", "cpu_model", "runs")}
    optional = []        # (key path, value), most dispensable LAST
    base = rf.get("baseline_configs")
    if base:             # [kernel_ms, frac of 8 TB/s (kernel), frac (wall)] per BASELINE row, both from the same passes
        optional.append((("roofline", "baseline_configs"),
                         {k: {"kernel_ms": v.get("kernel_ms"), "frac": v.get("frac"), "frac_wall": v.get("frac_wall")} for k, v in base.items()}))
    if full.get("c5"):
        c5 = full["c5"]
        optional.append((("c5",), {"value": c5.get("value"), "unit": _short(c5.get("unit"), 40), "scaling": c5.get("scaling"), "ms_per_step": c5.get("ms_per_step"),
                                   "n_gpus": c5.get("n_gpus"), "channels_total": c5.get("channels_total"), "channels_per_gpu": c5.get("channels_per_gpu"),
                                   "samples_per_channel": c5.get("samples_per_channel"), "kernel": _short(c5.get("kernel"), 60),
                                   "roofline": {k: (c5.get("roofline") or {}).get(k) for k in ("achieved", "frac", "avg_launch_ms")},
                                   "gather": {k: {"ms": v.get("ms"), "GBps_into_one_gpu": v.get("GBps_into_one_gpu")} for k, v in (c5.get("gather") or {}).items()}}))
    if full.get("gather"):
        optional.append((("gather",), {k: {"ms": v.get("ms"), "GBps_into_one_gpu": v.get("GBps_into_one_gpu")} for k, v in full["gather"].items()}))
    if full.get("cpu_baseline_simd") and "value" in full["cpu_baseline_simd"]:
        optional.append((("cpu_baseline_simd",), {"value": full["cpu_baseline_simd"]["value"], "cores": 1}))
    if full.get("cpu_baseline_all_cores"):
        optional.append((("cpu_baseline_all_cores",), {"value": full["cpu_baseline_all_cores"]["value"], "cores": full["cpu_baseline_all_cores"]["cores"]}))
    if full.get("streamed_1e6_chunks"):
        st = full["streamed_1e6_chunks"]
        optional.append((("streamed_1e6_chunks",), {"Msamples_per_s": st.get("Msamples_per_s"), "frac": st.get("frac")}))
    if full.get("fused"):
        optional.append((("fused",), {k: v.get("frac") for k, v in full["fused"].items() if isinstance(v, dict)}))
    if full.get("output_msamples_s") is not None:
        optional.append((("output_msamples_s",), full["output_msamples_s"]))
    if full_path:
        optional.append((("full_record",), full_path))

    def put(d, path, v):
        for k in path[:-1]:
            d = d.setdefault(k, {})
        d[path[-1]] = v

    def drop(d, path):
        for k in path[:-1]:
            d = d[k]
        d.pop(path[-1], None)

    for path, v in optional:
        put(out, path, v)
    text = json.dumps(out, separators=(",", ":"))
    while len(text) >= COMPACT_MAX and optional:
        path, _ = optional.pop()
        drop(out, path)
        text = json.dumps(out, separators=(",", ":"))
    assert len(text) < COMPACT_MAX, len(text)
    return text


def emit(full, args):
    """Write the long record to a file (and to stderr with --full-stderr), print the compact line as the ONLY line on stdout."""
    path = args.full_out
    rel = None
    if path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            with open(path, "w") as fh:
                json.dump(full, fh)
                fh.write("\n")
            rel = os.path.relpath(os.path.abspath(path), ROOT)
        except OSError as e:
            print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
    if args.full_stderr:
        print(json.dumps(full), file=sys.stderr, flush=True)
    print(compact_line(full, rel), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)   # the clocks take a few 10 ms launches to settle (kernel trace: 12.7, 11.1, 10.5, 10.3, 10.2 ms)
    ap.add_argument("--config", choices=["headline", "c5"], default="headline")
    ap.add_argument("--channels", type=int, default=0, help="channels per GPU (headline, default 64) / in all (c5, default 4096)")
    ap.add_argument("--samples", type=int, default=0, help="input samples per channel per step (default 1e8 headline, 1e6 c5)")
    ap.add_argument("--chunk", type=int, default=0, help="samples per channel per filt! call (0 = the whole batch in one call)")
    ap.add_argument("--numerics", choices=["strict", "fused"], default="strict")
    ap.add_argument("--time-every", type=int, default=0, help="n > 0: bracket every n-th kernel launch of the timed region with HIP events; "
                    "-n: one bracket around every n consecutive launches (0 = every launch when a pass is one call, groups of 10 when it is chunked)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-streamed", action="store_true", help="skip the extra chunked passes reported as `streamed_1e6_chunks`")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configs reported as `configs` (N = 1 only)")
    ap.add_argument("--no-check", action="store_true", help="skip the post-run oracle spot check (timing experiments)")
    ap.add_argument("--no-gather", action="store_true", help="c5: skip the gather timings")
    ap.add_argument("--full-out", default=os.path.join(ROOT, "gpurun_out", "bench_full.json"),
                    help="file that receives the long record (every config row, fused rows, parity object); '' = none")
    ap.add_argument("--full-stderr", action="store_true", help="also print the long record on stderr")
    ap.add_argument("--no-c5", action="store_true", help="multi-GPU headline: skip the config-5 pass appended as `c5`")
    ap.add_argument("--c5-channels", type=int, default=0, help="multi-GPU headline: channels in all of the appended config-5 pass (default 4096)")
    ap.add_argument("--c5-samples", type=int, default=0, help="... and its samples per channel (default 1e6)")
    args = ap.parse_args()

    if "RANK" not in os.environ and args.gpus > 1:
        # started directly, not by torch.distributed.run: become the launcher (no GPU call in this process)
        sys.exit(launch_ranks(args.gpus))

    R = Rank(args)
    if R.world != args.gpus and R.world > 1:
        args.gpus = R.world
    if args.config == "c5":
        line = run_c5(args, R)
    else:
        line = run_headline(args, R)
        if R.world > 1 and not args.no_c5:
            # BASELINE.json configs[4] in the SAME line of a multi-GPU run (the driver's scaling runs pass no --config): 4096
            # ComplexF32 channels sharded by channel over the ranks (strong scaling), compute-only rate, the final RCCL gather
            # timed apart -- after the headline's timed region, on every rank
            c5_args = argparse.Namespace(**vars(args))
            c5_args.channels, c5_args.samples = args.c5_channels, args.c5_samples
            c5 = run_c5(c5_args, R)
            if line is not None and c5 is not None:
                line["c5"] = {"metric": c5["metric"], "value": c5["value"], "unit": c5["unit"], "scaling": c5["scaling"], "ms_per_step": c5["ms_per_step"],
                              "n_gpus": c5["n_gpus"], "channels_total": c5["config"]["channels_total"], "channels_per_gpu": c5["config"]["channels_per_gpu"],
                              "samples_per_channel": c5["config"]["samples_per_channel"], "kernel": c5["kernel"], "gather": c5["gather"],
                              "roofline": {k: c5["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches_timed", "note")},
                              "workload": c5["config"]["workload"]}
    if line is not None:
        emit(line, args)
    R.finish()      # only on success: a rank that raised must not wait for the others at a barrier (the launcher ends them)


if __name__ == "__main__":
    main()
