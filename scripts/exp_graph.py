"""HIP-graph capture of a fixed-chunk streaming loop: when chunk % M == 0 the rational state
(phiIdx, inputDeficit) is the same at every call, so a graph that captured an even number of filt! calls
(history ping-pong) can be replayed over static buffers.  Compares with the plain loop and times both."""
import os, sys, time
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
L, M = 147, 160
h = pkg.firdes(24 * L, 0.5 / L, beta=7.8562).astype(np.float32)
nch = int(os.environ.get("NCH", "1"))
chunk, ncalls = 1_000_000, 100
x = torch.rand((nch, chunk * ncalls), dtype=torch.float32, device="cuda")
nout = chunk * L // M
y_ref = torch.empty((nch, nout * ncalls), dtype=torch.float32, device="cuda")
y_g = torch.zeros_like(y_ref)
f = pkg.FIRFilter(h, Fraction(L, M)); f.bind(np.float32, nch)

def loop(y):
    for i in range(ncalls):
        f.filt_into(y[:, i * nout:(i + 1) * nout], x[:, i * chunk:(i + 1) * chunk])

f.reset(); loop(y_ref); torch.cuda.synchronize()
for rep in range(2):
    f.reset(); torch.cuda.synchronize(); t0 = time.perf_counter(); loop(y_ref); torch.cuda.synchronize()
    print(f"plain loop: {(time.perf_counter() - t0) / ncalls * 1e6:.2f} us per chunk", flush=True)
f.reset(); torch.cuda.synchronize(); t0 = time.perf_counter(); n = f.filt_into_chunked(y_ref, x, chunk); torch.cuda.synchronize()
print(f"library chunk loop: {(time.perf_counter() - t0) / ncalls * 1e6:.2f} us per chunk", flush=True)

# graph: capture all calls of the pass (even count) on a side stream
f.reset()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
try:
    with torch.cuda.graph(g, stream=s):
        loop(y_g)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    print("graph replay equals plain loop:", torch.equal(y_g.view(torch.int32), y_ref.view(torch.int32)), flush=True)
    y_g.zero_()
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
        print(f"graph replay: {(time.perf_counter() - t0) / ncalls * 1e6:.2f} us per chunk", flush=True)
    print("graph replay (2nd) equals plain loop:", torch.equal(y_g.view(torch.int32), y_ref.view(torch.int32)), flush=True)
except Exception as e:
    print("graph capture failed:", repr(e)[:500], flush=True)
