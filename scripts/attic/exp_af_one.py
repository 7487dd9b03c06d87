import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
haf = pkg.firdes(320, 0.45 / 32, beta=7.8562) * 32
f = pkg.FIRFilter(haf, 0.470930233, 32).bind(np.float32, 1)
x = torch.rand((1, 10_000_000), device="cuda")
y = torch.empty((1, f.outputlength_bound(x.shape[1])), device="cuda", dtype=torch.float64)
for _ in range(4):
    f.filt_into(y, x)
torch.cuda.synchronize()
