"""FIRFarrow: farrow_wave_kernel (one lane per output, windows from global memory) against farrow_pipe_kernel (LDS tiles) by channel
count, total size fixed (1e7 samples); Float32 and Float64 samples, Float64 taps, 10 taps per phase (the reference's example) and 32."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MRHIP_ENV_DYNAMIC"] = "1"
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
for tpp in (10, 32):
    h = pkg.firdes(32 * tpp, 0.45 / 32, beta=7.8562) * 32
    for dt in (torch.float32, torch.float64):
        for nch in (1, 2, 3, 4, 6, 8, 16, 64):
            n = 10_000_000 // nch if nch < 64 else 1_000_000
            x = torch.rand((nch, n), device="cuda", dtype=dt)
            out = []
            for label, env in (("wave", {"MRHIP_FARROW_WAVE_MAXCH": "64"}), ("pipe", {"MRHIP_FARROW_WAVE_MAXCH": "0"})):
                os.environ.update(env)
                f = pkg.FIRFilter(h, 1.0, 32, 4)
                y = torch.empty((nch, f.bind(np.float32 if dt == torch.float32 else np.float64, nch).outputlength_bound(n)), dtype=torch.float64, device="cuda")
                f.filt_into(y, x); torch.cuda.synchronize()
                f.set_timing(True)
                for _ in range(5):
                    f.reset(); f.filt_into(y, x)
                torch.cuda.synchronize()
                nl, ms = f.timing_read()
                out.append(f"{label}: {ms / 5:7.4f} ms {f.last_kernel_name()[:18]}")
                f.close()
            print(f"tapsPerPhi={tpp:2d} {str(dt)[6:]:8s} nch={nch:2d} n={n:8d}  " + " | ".join(out), flush=True)
