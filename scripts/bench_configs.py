#!/usr/bin/env python3
"""Secondary measurements: BASELINE.json configs 1-5, the reference's own Arb-vs-Farrow benchmark shape and the non-headline
shapes on one MI355X.  Kernel-only figures come from the library's HIP-event log; wall figures include the host (Python +
library + whatever part of the FIRArbitrary phase schedule runs there).  Every row is priced against BOTH roofs: HBM
(8 TB/s) with the algorithmic bytes of SURVEY.md 8d, and the vector ALU with the reference's 2 flops per tap --
`frac_of_strict_valu` against the rate of separately rounded multiply + add (the default STRICT numerics cannot fuse: half
of the FMA peak, 78.6 TF f32 / 39.3 TF f64) and `frac_of_fma_valu` against the FMA peak (157.3 TF f32 / 78.6 TF f64,
MI355X_MICROARCH.md).  Not the headline bench (that is bench.py, which imports `run_rows` to put the BASELINE rows into
its JSON line); used for DESIGN.md's per-kernel table.

    python scripts/bench_configs.py [names...] [--numerics fused]
"""
import json
import math
import os
import sys
import time
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge

pkg = ge.load_package()
dev = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
FUSED = __name__ == "__main__" and "--numerics" in sys.argv and sys.argv[sys.argv.index("--numerics") + 1] == "fused"
EMIT = print                                   # run_rows() points it at a collector
HBM_GBPS = 8000.0
SETTLE_MS = float(os.environ.get("BENCH_SETTLE_MS", "60"))     # sustained load in front of the timed passes of a row
FMA_TF = {False: 157.3, True: 78.6}          # vector FMA peak, f32 / f64


def rand(shape, dtype):
    if dtype.is_complex:
        return torch.view_as_complex(torch.rand(shape + (2,), device=dev, dtype=torch.float32 if dtype == torch.complex64 else torch.float64))
    return torch.rand(shape, device=dev, dtype=dtype)


def run(name, h, ratio, nphi, nch, n, dtype, bytes_per_in, flops_per_in, reps=5, chunk=None, polyorder=None, per_call=False, note=None):
    """bytes_per_in / flops_per_in: algorithmic bytes and flops (2 per tap per real component) per input sample per channel"""
    x = rand((nch, n), dtype)
    f = pkg.FIRFilter(h, ratio, nphi, polyorder, device=dev.index or 0, numerics=pkg.NUMERICS_FUSED if FUSED else pkg.NUMERICS_STRICT)
    chunk = chunk or n
    f.filt(x[:, :chunk])                       # warm-up + bind
    out_dtype = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
                 np.dtype(np.complex64): torch.complex64, np.dtype(np.complex128): torch.complex128}[np.dtype(f.output_dtype)]
    r_f64 = out_dtype in (torch.float64, torch.complex128)
    f.reset()
    ychunked = None
    if chunk != n and not per_call:            # streaming through the library's chunk loop (resident signal)
        ychunked = torch.empty((nch, f.outputlength(n) + 8), dtype=out_dtype, device=dev)
    if per_call:
        os.environ["MRHIP_CHUNKED_PER_CALL"] = "1"
    # filt!(buffer, self, x) into a buffer allocated once (the reference's filt allocates per call, Filters.jl:744-751: an
    # allocator's time is not the engine's); FIRArbitrary / FIRFarrow size it from the outputlength bound like that code does
    ybuf = None
    if ychunked is None:
        ybuf = torch.empty((nch, max(f.outputlength_bound(chunk), 1)), dtype=out_dtype, device=dev)
    # settle the clocks: the chip takes ~30 ms of sustained load to reach the clock it then holds (config 4, the same call 40 times:
    # 4.6, 4.3, 4.1, 4.0 ... 3.90 ms from the eighth call on; with 50 ms of idle between calls every one is 4.7 --
    # scripts/exp_c4_drift.py): at least 2 passes, then on until SETTLE_MS of work have run or 20 passes
    t_settle = time.perf_counter()
    for i_settle in range(20):
        f.reset()
        if ychunked is not None:
            f.filt_into_chunked(ychunked, x, chunk)
        else:
            for a in range(0, n, chunk):
                f.filt_into(ybuf, x[:, a:a + chunk])
        torch.cuda.synchronize()
        el_ms = (time.perf_counter() - t_settle) * 1e3
        # (a row whose whole pass is a burst of well under a millisecond -- configs 1 and 2: 11 and 160 us -- is measured as a burst: repeated back
        #  to back without a pause such a launch slows after ~2 ms of them, 0.16 -> 0.22 ms for config 2 -- the chip's fast power limiter, not the
        #  clock ramp this loop is for; scripts/exp_c2_drift.py)
        if i_settle >= 1 and (el_ms >= SETTLE_MS or el_ms / (i_settle + 1) < 1.0):
            break
    settle_passes, settle_ms = i_settle + 1, (time.perf_counter() - t_settle) * 1e3
    f.set_timing(True)
    torch.cuda.synchronize()
    t_wall = time.perf_counter()
    for _ in range(reps):
        f.reset()
        if ychunked is not None:
            f.filt_into_chunked(ychunked, x, chunk)
            continue
        for a in range(0, n, chunk):
            f.filt_into(ybuf, x[:, a:a + chunk])
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t_wall) * 1e3 / reps
    nl, ms = f.timing_read()
    per_pass_ms = ms / reps
    gbps = nch * n * bytes_per_in / (per_pass_ms * 1e-3) / 1e9
    tflops = nch * n * flops_per_in / (per_pass_ms * 1e-3) / 1e12
    cont_ms = None
    if isinstance(ratio, float) and chunk == n:
        # FIRArbitrary / FIRFarrow: the passes above reset the filter and repeat the block, so from the second pass on the phase
        # schedule is the memo of the identical earlier call.  A stream that goes on pays for its schedule every call (evaluated
        # on the device beside the previous call's filter kernel): the same calls WITHOUT the reset.
        memo_kernel_ms = per_pass_ms
        f.set_timing(False)
        f.reset()
        f.filt_into(ybuf, x)
        for _ in range(2):                     # (the clocks are settled; the schedule's own pipeline -- one call ahead -- is not yet)
            f.filt_into(ybuf, x)
        f.set_timing(True)
        torch.cuda.synchronize()
        t_c = time.perf_counter()
        for _ in range(max(reps, 3)):
            f.filt_into(ybuf, x)
        torch.cuda.synchronize()
        cont_ms = (time.perf_counter() - t_c) * 1e3 / max(reps, 3)
        # kernel time and wall time of a row come from the SAME passes (a kernel cannot outlast its wall): the continuing stream's
        nl_c, ms_c = f.timing_read()
        if nl_c:
            per_pass_ms = ms_c / max(reps, 3)
            nl = nl_c * reps // max(reps, 3)
            gbps = nch * n * bytes_per_in / (per_pass_ms * 1e-3) / 1e9
            tflops = nch * n * flops_per_in / (per_pass_ms * 1e-3) / 1e12
    out = {"config": name, "kernel": f.last_kernel_name(), "numerics": "fused" if FUSED else "strict", "channels": nch, "samples_per_channel": n,
           "untimed_passes": settle_passes, "untimed_ms": round(settle_ms, 1),
           "kernel_ms_per_pass": round(per_pass_ms, 4), "wall_ms_per_pass_incl_host": round(wall_ms, 3), "launches_per_pass": nl // reps,
           "Msamples_per_s_in": round(nch * n / (per_pass_ms * 1e-3) / 1e6, 1),
           "Msamples_per_s_in_wall": round(nch * n / (wall_ms * 1e-3) / 1e6, 1),
           "algorithmic_GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / HBM_GBPS, 4),
           "frac_of_8TBps_wall": round(nch * n * bytes_per_in / (wall_ms * 1e-3) / 1e9 / HBM_GBPS, 4),
           "flops_per_input_sample": round(flops_per_in, 2), "TFLOPs": round(tflops, 2),
           "arith": "f64" if r_f64 else "f32",
           "frac_of_strict_valu": round(tflops / (FMA_TF[r_f64] / 2), 4), "frac_of_fma_valu": round(tflops / FMA_TF[r_f64], 4)}
    if cont_ms is not None:
        # FIRArbitrary / FIRFarrow: the figure a STREAM pays is the continuing one (every call evaluates its phase schedule); the passes above
        # reset and repeat one block, so from the second on they reuse the schedule of the identical earlier call (the memo): a side key
        out["wall_ms_with_schedule_memo"] = out["wall_ms_per_pass_incl_host"]
        out["kernel_ms_with_schedule_memo"] = round(memo_kernel_ms, 4)
        out["wall_ms_per_pass_incl_host"] = round(cont_ms, 3)
        out["wall_ms_per_call_continuing_stream"] = round(cont_ms, 3)
        out["Msamples_per_s_in_wall"] = round(nch * n / (cont_ms * 1e-3) / 1e6, 1)
        out["frac_of_8TBps_wall"] = round(nch * n * bytes_per_in / (cont_ms * 1e-3) / 1e9 / HBM_GBPS, 4)
    if note:
        out["note"] = note
    EMIT(json.dumps(out))
    if per_call:
        os.environ.pop("MRHIP_CHUNKED_PER_CALL", None)
    f.close()
    del x
    torch.cuda.empty_cache()


DEFAULT_ROWS = ["c1", "c2", "c2s", "c3a", "c3b", "c4", "c4f", "c5", "x160", "xf64", "xmix", "xstd", "x32"]
# what bench.py reports next to the headline: every BASELINE config on one GPU, the README's mixed precision, the reference's
# own FIRArbitrary / FIRFarrow benchmark shape
# (the BASELINE rows LAST: the driver's record keeps the END of the line)
BENCH_ROWS = ["xarb", "af", "xdec", "xlarge", "ms", "xmix64", "c1", "c2", "c2s", "c2r", "c3a", "c3b", "c4", "c4f", "c5"]
# the opt-in FUSED numerics (one fma per tap, same order) for the BASELINE rows: bench.py reports them under `fused`
FUSED_ROWS = ["c3a", "c3b", "c4", "c5"]


def rows(which):
    h147 = pkg.firdes(24 * 147, 0.5 / 147, beta=7.8562).astype(np.float32)
    h128 = pkg.firdes(128, 0.5 / 4, beta=7.8562).astype(np.float32)
    harb = pkg.firdes(32 * 32, 0.45 / 32, beta=7.8562) * 32
    R147 = 147 / 160

    def _c1():
        # (200 passes: with the default 5 the row's wall time was the device synchronisation that ends the timed loop, ~60 us, shared by five calls --
        #  24-25 us "per call"; a loop of reset() + filt!() runs at 13 us per call, filt!() alone at 9.5: scripts/attic/exp_c1_wall.py)
        run("C1 rational 147//160 f32 1ch x 1e6 (one call)", h147, Fraction(147, 160), 32, 1, 1_000_000, torch.float32, 7.675, 48 * R147, reps=200)
    def _c2():
        run("C2 rational 147//160 f32 1ch x 1e8 in 1e6 chunks (resident signal: mrhip_filt_device_chunked)", h147, Fraction(147, 160), 32, 1, 100_000_000, torch.float32, 7.675, 48 * R147, reps=3, chunk=1_000_000)
    def _c2s():
        run("C2s the same, one launch per arriving 1e6-sample chunk (MRHIP_CHUNKED_PER_CALL=1)", h147, Fraction(147, 160), 32, 1, 100_000_000, torch.float32, 7.675, 48 * R147, reps=2, chunk=1_000_000, per_call=True)
    def _c2r():
        # BASELINE config 2 as stated -- ONE channel, 1e8 samples ARRIVING in 1e6-sample chunks, state carried from chunk to chunk -- through
        # the ring of arriving chunks (mrhip_ring_*): one descriptor per chunk into ONE resident kernel instead of one launch per chunk.
        # No launch to bracket with events: the times are the host's, from the first push to the last chunk's completion flag.
        n, chunk, reps = 100_000_000, 1_000_000, 3
        x = rand((1, n), torch.float32)
        f = pkg.FIRFilter(h147, Fraction(147, 160), device=dev.index or 0).bind(np.float32, 1)
        y = torch.empty((1, f.outputlength(n) + 8), dtype=torch.float32, device=dev)
        gb = n * 7.675 / 1e9
        for tag, label, how in (("C2r", "pushed by the library's chunk loop (mrhip_ring_push_chunks)", "lib"), ("C2rp", "pushed one by one from Python (mrhip_ring_push, tensor checks per push)", "py"),
                                ("C2rq", "one mrhip_ring_push per chunk on precomputed addresses (what a loop in another language pays, plus ctypes)", "raw")):
            t_push, t_all = [], []
            for rep in range(reps + 1):
                f.reset()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ring = f.open_ring()
                t1 = time.perf_counter()
                if how == "lib":
                    total, _ = ring.push_chunks(y, x, chunk)
                elif how == "raw":
                    xp, yp, k, cap = x.data_ptr(), y.data_ptr(), 0, y.shape[-1]
                    for a in range(0, n, chunk):
                        cnt, _ = ring.push_raw(xp + 4 * a, chunk, chunk, yp + 4 * k, cap - k, cap - k)
                        k += cnt
                else:
                    k = 0
                    for a in range(0, n, chunk):
                        cnt, _ = ring.push(y[:, k:], x[:, a:a + chunk])
                        k += cnt
                ring.drain()
                t2 = time.perf_counter()
                resident = ring.info()["resident"]
                ring.close()
                t3 = time.perf_counter()
                if rep:                                    # (the first pass settles the clocks)
                    t_push.append(t2 - t1); t_all.append(t3 - t0)
            ms, ms_all = 1e3 * sorted(t_push)[len(t_push) // 2], 1e3 * sorted(t_all)[len(t_all) // 2]
            EMIT(json.dumps({"config": f"{tag} the same through the ring of arriving chunks: 100 descriptors into one resident kernel, {label}", "kernel": "rational_opair_kernel (resident)" if resident else f.last_kernel_name(),
                             "numerics": "strict", "channels": 1, "samples_per_channel": n, "kernel_ms_per_pass": round(ms, 4),
                             "wall_ms_per_pass_incl_host": round(ms_all, 3), "launches_per_pass": 1, "chunks_per_pass": n // chunk,
                             "us_per_chunk": round(1e3 * ms / (n // chunk), 3),
                             "Msamples_per_s_in": round(n / (ms * 1e-3) / 1e6, 1), "Msamples_per_s_in_wall": round(n / (ms_all * 1e-3) / 1e6, 1),
                             "algorithmic_GBps": round(gb / (ms * 1e-3), 1), "frac_of_8TBps": round(gb / (ms * 1e-3) / HBM_GBPS, 4),
                             "frac_of_8TBps_wall": round(gb / (ms_all * 1e-3) / HBM_GBPS, 4), "flops_per_input_sample": round(48 * R147, 2),
                             "TFLOPs": round(n * 48 * R147 / (ms * 1e-3) / 1e12, 2), "arith": "f32",
                             "frac_of_strict_valu": round(n * 48 * R147 / (ms * 1e-3) / 1e12 / (FMA_TF[False] / 2), 4), "frac_of_fma_valu": round(n * 48 * R147 / (ms * 1e-3) / 1e12 / FMA_TF[False], 4),
                             "note": "kernel_ms = first push -> last chunk's completion flag (host clock; ring already open); wall_ms adds opening the ring (one launch) and closing it"}))
        f.close()
        del x, y
        torch.cuda.empty_cache()
    def _c3a():
        run("C3a interpolator 4//1 128 taps c64 256ch x 1e6", h128, Fraction(4, 1), 32, 256, 1_000_000, torch.complex64, 40.0, 2 * 2 * 32 * 4)
    def _c3b():
        run("C3b decimator 1//4 128 taps c64 256ch x 1e6", h128, Fraction(1, 4), 32, 256, 1_000_000, torch.complex64, 10.0, 2 * 2 * 128 / 4)
    def _c4():
        run("C4 arbitrary pi/3 32x32 taps f64 64ch x 1e7", harb, float(math.pi / 3), 32, 64, 10_000_000, torch.float64, 8 + 8 * math.pi / 3, (2 * 64 + 2) * math.pi / 3, reps=2)
    def _c4f():
        run("C4f farrow pi/3 32x32 taps polyorder 4 f64 64ch x 1e7", harb, float(math.pi / 3), 32, 64, 10_000_000, torch.float64, 8 + 8 * math.pi / 3, (2 * 32 + 2 * 4 * 32 / 64) * math.pi / 3, reps=2, polyorder=4,
            note="flops include the Float64 Horner evaluation of the 32 taps of every output index, done once for all 64 channels")
    def _c5():
        run("C5 rational 147//160 c64 512ch x 1e6 (one GPU's shard of 4096)", h147, Fraction(147, 160), 32, 512, 1_000_000, torch.complex64, 15.35, 2 * 48 * R147)

    def _xdec():
        # decimations the run-time-M streaming kernel serves (fir_stream_rt_kernel): bytes per input sample sizeof(Tx) + sizeof(Tb)/M
        for M, dt, nm, es in ((32, torch.float32, "f32", 4), (36, torch.float32, "f32", 4), (100, torch.float32, "f32", 4), (24, torch.complex64, "c64", 8)):
            hd = pkg.firdes(128, 0.5 / M, beta=7.8562).astype(np.float32)
            nc = 2 if dt == torch.complex64 else 1
            run(f"X decimator 1//{M} 128 taps {nm} 64ch x 4e6", hd, Fraction(1, M), 32, 64, 4_000_000, dt, es + es / M, nc * 2 * 128 / M)

    def _xlarge():
        # L > 512: ordinary clock-trim ratios (1-14 % of the roofline on poly_tiled / poly_generic): the output-pair kernel in period blocks
        for L_, M_ in ((625, 512), (1000, 999), (640, 441), (4096, 4095)):
            hl = pkg.firdes(24 * L_, 0.5 / max(L_, M_), beta=7.8562).astype(np.float32)
            run(f"X rational {L_}//{M_} 24 taps per phase f32 64ch x 1e6", hl, Fraction(L_, M_), 32, 64, 1_000_000, torch.float32, 4 + 4 * L_ / M_, 48 * L_ / M_)

    # shapes outside BASELINE.json (where the non-headline kernels stand)
    def _x160():
        h160 = pkg.firdes(24 * 160, 0.5 / 160, beta=7.8562).astype(np.float32)
        run("X rational 160//147 (44.1k->48k) f32 64ch x 1e6", h160, Fraction(160, 147), 32, 64, 1_000_000, torch.float32, 4 + 4 * 160 / 147, 48 * 160 / 147)
    def _xf64():
        run("X rational 147//160 f64 64ch x 1e6", h147.astype(np.float64), Fraction(147, 160), 32, 64, 1_000_000, torch.float64, 2 * 7.675, 48 * R147)
    def _xmix():
        # the reference's own published benchmark (README.md:172-193): firdes returns Float64 taps, x = rand(Float32, 1_000_000)
        # -> Float64 output (Filters.jl:581); 0.0569 s = 17.56 Msamples/s on unnamed 2014 hardware, one channel
        run("X README mixed precision: 147//160 Float64 taps x Float32 samples -> Float64, 64ch x 1e6", h147.astype(np.float64), Fraction(147, 160), 32, 64, 1_000_000,
            torch.float32, 4 + 8 * R147, 48 * R147, note="reference README.md:172-193: 17.56 Msamples/s in (0.0569 s for 1e6 samples, 1 channel, Julia 0.3, unnamed 2014 CPU)")
        run("X README mixed precision, the README's own size: 1ch x 1e6", h147.astype(np.float64), Fraction(147, 160), 32, 1, 1_000_000,
            torch.float32, 4 + 8 * R147, 48 * R147, note="reference README.md:172-193: 0.0569 s per call = 17.56 Msamples/s")
    def _xstd():
        run("X standard 1//1 128 taps f32 64ch x 4e6", h128, Fraction(1, 1), 32, 64, 4_000_000, torch.float32, 8.0, 2 * 128)
    def _x32():
        h32 = pkg.firdes(24 * 3, 0.5 / 3, beta=7.8562).astype(np.float32)
        run("X rational 3//2 f32 64ch x 1e6", h32, Fraction(3, 2), 32, 64, 1_000_000, torch.float32, 4 + 6, 48 * 1.5)
        run("X rational 2//3 f32 64ch x 1e6", h32, Fraction(2, 3), 32, 64, 1_000_000, torch.float32, 4 + 8 / 3, 72 * 2 / 3)
    def _xc32():   # the headline's launch size (491 MB) on the ComplexF32 kernel: 32 complex channels x 1e6 per launch
        run("X rational 147//160 c64 32ch x 2e7 in 1e6 chunks, per call", h147, Fraction(147, 160), 32, 32, 20_000_000, torch.complex64, 15.35, 2 * 48 * R147, reps=3, chunk=1_000_000, per_call=True)
        run("X rational 147//160 f32 64ch x 2e7 in 1e6 chunks, per call", h147, Fraction(147, 160), 32, 64, 20_000_000, torch.float32, 7.675, 48 * R147, reps=3, chunk=1_000_000, per_call=True)

    def _xarb():   # config 4's shape in the other sample / tap types (which FIRArbitrary / FIRFarrow kernel serves them, and how fast)
        for th, dt, sb, nm in ((np.float32, torch.float32, 4, "f32 taps x f32"), (np.float32, torch.complex64, 8, "f32 taps x c64"),
                               (np.float64, torch.float32, 4, "f64 taps x f32"), (np.float64, torch.complex64, 8, "f64 taps x c64"),
                               (np.float64, torch.complex128, 16, "f64 taps x c128")):
            nc = 2 if dt.is_complex else 1
            ob = (8 if th == np.float64 else 4) * nc
            for kind, po in (("arbitrary", None), ("farrow", 4)):
                fl = ((2 * 64 + 2) if po is None else (2 * 32 + 2 * 4 * 32 / 64)) * nc * math.pi / 3
                run(f"X {kind} pi/3 32x32 {nm} 64ch x 4e6", harb.astype(th), float(math.pi / 3), 32, 64, 4_000_000, dt, sb + ob * math.pi / 3, fl, reps=2, polyorder=po)

    def _xmix64():
        run("X README mixed precision: 147//160 Float64 taps x Float32 samples -> Float64, 64ch x 1e6", h147.astype(np.float64), Fraction(147, 160), 32, 64, 1_000_000,
            torch.float32, 4 + 8 * R147, 48 * R147, note="reference README.md:172-193: 17.56 Msamples/s in (0.0569 s for 1e6 samples, 1 channel, Julia 0.3, unnamed 2014 CPU)")
    def _af():
        # examples/Arb-Farrow Speed Comparison.jl:38-54: N𝜙 = 32, 10 taps per phase, polyorder 4, x = rand(Tx, 10_000_000), ONE channel,
        # rates 1.0 and 1/2.123456789, Tx in (Float32, Float64, Complex64, Complex128); h = firdes(...) .* N𝜙 is Float64 there (the
        # script's `Th = Float32` is never applied); the reference prints samples/s and records no result
        haf = pkg.firdes(320, 0.45 / 32, beta=7.8562) * 32
        for rate in (1.0, 1 / 2.123456789):
            for dt, sb in ((torch.float32, 4), (torch.float64, 8), (torch.complex64, 8), (torch.complex128, 16)):
                nc = 2 if dt.is_complex else 1
                ob = 8 * nc                                   # Float64 taps: the output is Float64 / ComplexF64
                for kind, po in (("FIRArbitrary", None), ("FIRFarrow", 4)):
                    fl = (2 * 20 + 2 if po is None else 2 * 10) * nc * rate
                    run(f"AF {kind} rate {rate:.9g} {str(dt).replace('torch.', '')} 1ch x 1e7 (Arb-Farrow Speed Comparison.jl shape)", haf, float(rate), 32, 1,
                        10_000_000, dt, sb + ob * rate, fl, reps=2, polyorder=po)

    def _ms():
        # the north star's "one-channel-per-stream": 64 INDEPENDENT single-channel FIRFilters (README.md:87-141: one object per
        # signal), chunks of about 1e6 samples of UNEQUAL lengths arriving round after round, one launch per round
        # (mrhip_filt_device_multi) -- against one launch per stream and round (the plain loop)
        ns, rounds = 64, 5
        rng = np.random.default_rng(1)
        lens = [int(v) for v in rng.integers(900_000, 1_100_000, size=ns)]
        fs = [pkg.FIRFilter(h147, Fraction(147, 160), device=dev.index or 0, numerics=pkg.NUMERICS_FUSED if FUSED else pkg.NUMERICS_STRICT).bind(np.float32, 1) for _ in range(ns)]
        xs = [torch.rand((1, n_), device=dev, dtype=torch.float32) for n_ in lens]
        for i, f in enumerate(fs):                       # every stream somewhere else in its phase cycle
            f.filt(xs[i][:, :1000 + 13 * i])
        ys = [torch.empty((1, f.outputlength_bound(n_)), device=dev, dtype=torch.float32) for f, n_ in zip(fs, lens)]
        total = float(sum(lens))
        for label, multi in (("one launch per round (mrhip_filt_device_multi)", True), ("one launch per stream and round (plain loop)", False)):
            ms = pkg.MultiStream(fs, ys, xs)
            def go():
                if multi:
                    ms.run()
                else:
                    for f, y, x in zip(fs, ys, xs):
                        f.filt_into(y, x)
            go(); go()
            for f in fs:
                f.set_timing(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(rounds):
                go()
            torch.cuda.synchronize()
            wall_ms = (time.perf_counter() - t0) * 1e3 / rounds
            ms_k, nl = 0.0, 0
            for f in fs:
                a, b = f.timing_read()
                nl += a; ms_k += b
                f.set_timing(False)
            per = ms_k / rounds
            gb = total * 7.675 / 1e9
            EMIT(json.dumps({"config": f"MS 64 independent 147//160 f32 streams x ~1e6-sample chunks of unequal lengths, {label}", "kernel": fs[0].last_kernel_name(),
                             "numerics": "fused" if FUSED else "strict", "channels": ns, "samples_per_channel": int(total / ns),
                             "kernel_ms_per_pass": round(per, 4), "wall_ms_per_pass_incl_host": round(wall_ms, 3), "launches_per_pass": nl // rounds,
                             "Msamples_per_s_in": round(total / (per * 1e-3) / 1e6, 1), "Msamples_per_s_in_wall": round(total / (wall_ms * 1e-3) / 1e6, 1),
                             "algorithmic_GBps": round(gb / (per * 1e-3), 1), "frac_of_8TBps": round(gb / (per * 1e-3) / HBM_GBPS, 4),
                             "frac_of_8TBps_wall": round(gb / (wall_ms * 1e-3) / HBM_GBPS, 4), "flops_per_input_sample": round(48 * R147, 2),
                             "TFLOPs": round(total * 48 * R147 / (per * 1e-3) / 1e12, 2), "arith": "f32",
                             "frac_of_strict_valu": round(total * 48 * R147 / (per * 1e-3) / 1e12 / (FMA_TF[False] / 2), 4),
                             "frac_of_fma_valu": round(total * 48 * R147 / (per * 1e-3) / 1e12 / FMA_TF[False], 4)}))
        for f in fs:
            f.close()

    table = {"ms": _ms, "c1": _c1, "c2": _c2, "c2s": _c2s, "c2r": _c2r, "c3a": _c3a, "c3b": _c3b, "c4": _c4, "c4f": _c4f, "c5": _c5, "x160": _x160, "xf64": _xf64, "xmix": _xmix, "xstd": _xstd, "x32": _x32, "xc32": _xc32, "xarb": _xarb, "xmix64": _xmix64, "af": _af, "xdec": _xdec, "xlarge": _xlarge}
    for name in which:                # in the order asked for (bench.py wants the BASELINE rows last)
        table[name]()


def run_rows(which, reps_note=None, fused=False):
    """The rows named in `which` as a list of dicts (bench.py); fused: under the opt-in FUSED numerics."""
    global EMIT, FUSED
    got = []
    EMIT = lambda line: got.append(json.loads(line))
    was = FUSED
    FUSED = bool(fused)
    try:
        rows(which)
    finally:
        EMIT = print
        FUSED = was
    return got


if __name__ == "__main__":
    rows([a for a in sys.argv[1:] if not a.startswith("--") and a not in ("strict", "fused")] or DEFAULT_ROWS)
