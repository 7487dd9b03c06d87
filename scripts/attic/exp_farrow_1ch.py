"""Few-channel FIRFarrow (the reference's Arb-Farrow Speed Comparison.jl shape): kernel time under the library's switches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MRHIP_ENV_DYNAMIC"] = "1"
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
haf = pkg.firdes(320, 0.45 / 32, beta=7.8562) * 32
x = torch.rand((1, 10_000_000), device="cuda", dtype=torch.float32)
def run(tag, kind_po, env, nch=1, dt=torch.float32, rate=1.0):
    for k, v in env.items(): os.environ[k] = v
    xx = x if nch == 1 else torch.rand((nch, 10_000_000 // nch), device="cuda", dtype=torch.float32)
    f = pkg.FIRFilter(haf, rate, 32, kind_po)
    y = torch.empty((nch, f.bind(np.float32, nch).outputlength_bound(xx.shape[1])), dtype=torch.float64, device="cuda")
    for _ in range(3):
        f.reset(); f.filt_into(y, xx)
    f.set_timing(True)
    for _ in range(5):
        f.reset(); f.filt_into(y, xx)
    torch.cuda.synchronize()
    nl, ms = f.timing_read()
    print(f"{tag:40s} {f.last_kernel_name():22s} {ms / 5:.4f} ms", flush=True)
    f.close()
    for k in env: os.environ.pop(k, None)
os.environ["MRHIP_DEBUG"] = "1"
run("farrow 1ch", 4, {})
run("arb 1ch", None, {})
os.environ.pop("MRHIP_DEBUG")
run("farrow 1ch no DMA", 4, {"MRHIP_PIPE_DMA": "0"})
run("farrow 1ch tiled", 4, {"MRHIP_FARROW_PIPE": "0"})
run("farrow 1ch generic", 4, {"MRHIP_FARROW_PIPE": "0", "MRHIP_FARROW_TILED": "0"})
run("farrow 2ch", 4, {}, nch=2)
run("farrow 4ch", 4, {}, nch=4)
run("farrow 8ch", 4, {}, nch=8)
run("arb 4ch", None, {}, nch=4)
run("farrow 1ch polyorder 1", 1, {})
run("farrow 1ch polyorder 8", 8, {})
