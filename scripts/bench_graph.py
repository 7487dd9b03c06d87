"""What a call costs the HOST in a stream of small chunks: the synchronous call (filt_into: the count comes back), the
asynchronous call (filt_into_async: planned on the device, nothing comes back) and the same calls captured ONCE into a HIP
graph and replayed.  Every row filters the same resident signal chunk after chunk; wall = host clock round the whole stream
with one synchronize at the end, per call.  The kernel time beside it is the filter launches alone (HIP events, from the
synchronous pass).
Usage: python scripts/bench_graph.py            (prints one JSON line per row)"""
import json
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
dev = torch.device("cuda", 0)
TD = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64, np.dtype(np.complex64): torch.complex64, np.dtype(np.complex128): torch.complex128}


def stream_row(name, make, nch, chunk, ncalls, dtype, reps=5):
    """make() -> a filter or a cascade; ncalls chunks of `chunk` samples per pass"""
    x = torch.rand((nch, chunk * ncalls), device=dev, dtype=dtype)
    f = make()
    is_cas = isinstance(f, pkg.FilterCascade)
    stages = f.stages if is_cas else (f,)
    if is_cas:
        f._ensure(np.dtype(np.float32 if dtype == torch.float32 else np.float64), nch)
    else:
        f.bind(np.float32 if dtype == torch.float32 else (np.float64 if dtype == torch.float64 else np.complex64), nch)
    bound = f.outputlength_bound(chunk)
    ys = torch.empty((ncalls, nch, bound), dtype=TD[np.dtype(f.output_dtype)], device=dev)
    cnt = torch.zeros(ncalls, dtype=torch.int64, device=dev)

    def wall(fn):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3 / ncalls)
        return best

    def sync_pass():
        for i in range(ncalls):
            f.filt_into(ys[i], x[:, i * chunk:(i + 1) * chunk])

    def async_pass():
        for i in range(ncalls):
            f.filt_into_async(ys[i], x[:, i * chunk:(i + 1) * chunk], cnt[i:i + 1])

    w_sync = wall(sync_pass)
    kernel_ms = None
    if not is_cas:
        f.set_timing(True)
        sync_pass(); torch.cuda.synchronize()
        nl, ms = f.timing_read()
        f.set_timing(False)
        kernel_ms = ms / ncalls
    w_async = wall(async_pass)
    for s_ in stages:
        s_.sync_state()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=st):
        async_pass()
    w_graph = wall(g.replay)
    for s_ in stages:
        s_.sync_state()
    out = {"row": name, "kernel": stages[-1].last_kernel_name(), "channels": nch, "chunk": chunk, "calls_per_pass": ncalls,
           "kernel_ms_per_call": None if kernel_ms is None else round(kernel_ms, 4),
           "wall_ms_per_call_sync": round(w_sync, 4), "wall_ms_per_call_async": round(w_async, 4), "wall_ms_per_call_graph_replay": round(w_graph, 4)}
    print(json.dumps(out), flush=True)
    f.close()
    del x, ys
    torch.cuda.empty_cache()


def main():
    global stream_row
    only = [a for a in sys.argv[1:] if not a.startswith("-") and not a.startswith("chunk=")]
    chunks = [int(a[6:]) for a in sys.argv[1:] if a.startswith("chunk=")]
    if only or chunks:             # rows whose name contains any of the words given (and whose chunk size is one of chunk=<n>)
        all_rows = stream_row
        stream_row = lambda name, make, nch, chunk, *a, **k: all_rows(name, make, nch, chunk, *a, **k) if (not only or any(w in name for w in only)) and (not chunks or chunk in chunks) else None
    h147 = pkg.firdes(24 * 147, 0.5 / 147, beta=7.8562).astype(np.float32)
    h128 = pkg.firdes(128, 0.5 / 4, beta=7.8562).astype(np.float32)
    haf = pkg.firdes(320, 0.45 / 32, beta=7.8562) * 32
    F = pkg.FIRFilter
    # chunk sizes that are NOT multiples of the decimation: state and count change from call to call
    stream_row("rational 147//160 f32, 1 channel", lambda: F(h147, Fraction(147, 160)), 1, 99_991, 50, torch.float32)
    stream_row("rational 147//160 f32, 1 channel", lambda: F(h147, Fraction(147, 160)), 1, 999_983, 20, torch.float32)
    stream_row("rational 147//160 f32, 64 channels", lambda: F(h147, Fraction(147, 160)), 64, 99_991, 50, torch.float32)
    stream_row("decimator 1//4 128 taps f32, 1 channel", lambda: F(h128, Fraction(1, 4)), 1, 99_991, 50, torch.float32)
    stream_row("FIRArbitrary rate 1/2.123456789, 10 taps per phase, f32, 1 channel", lambda: F(haf, 1 / 2.123456789, 32), 1, 99_991, 50, torch.float32)
    stream_row("FIRFarrow rate 1/2.123456789, 10 taps per phase, polyorder 4, f32, 1 channel", lambda: F(haf, 1 / 2.123456789, 32, 4), 1, 99_991, 50, torch.float32)
    stream_row("FIRArbitrary rate 1/2.123456789, 10 taps per phase, f32, 1 channel", lambda: F(haf, 1 / 2.123456789, 32), 1, 19_997, 50, torch.float32)
    stream_row("FIRArbitrary rate 1/2.123456789, 10 taps per phase, f32, 1 channel", lambda: F(haf, 1 / 2.123456789, 32), 1, 49_999, 50, torch.float32)
    stream_row("FIRArbitrary rate 1/2.123456789, 10 taps per phase, f32, 1 channel", lambda: F(haf, 1 / 2.123456789, 32), 1, 299_993, 30, torch.float32)
    stream_row("FIRArbitrary rate pi/3, 32 taps per phase, f64, 64 channels", lambda: F(pkg.firdes(1024, 0.45 / 32, beta=7.8562) * 32, float(np.pi / 3), 32), 64, 99_991, 20, torch.float64)
    stream_row("cascade: decimator 1//4, then 147//160, f32, 1 channel", lambda: pkg.FilterCascade(F(h128, Fraction(1, 4)), F(h147, Fraction(147, 160))), 1, 99_991, 50, torch.float32)
    stream_row("cascade: decimator 1//4, then 147//160, f32, 16 channels", lambda: pkg.FilterCascade(F(h128, Fraction(1, 4)), F(h147, Fraction(147, 160))), 16, 99_991, 50, torch.float32)


if __name__ == "__main__":
    main()
