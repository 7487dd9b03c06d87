import os, sys
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as O
pkg = ge.load_package()
os.environ.setdefault("MRHIP_RING_IDLE_MS", "500")
ratio, T, th, tx, nch = Fraction(5, 3), 24, np.float64, np.float64, int(os.environ.get("NCH", "5"))
sizes = eval(os.environ.get("SIZES", "[15567, 23797, 17, 37742, 33328, 37889, 26609, 13129, 20436, 37278, 40601, 47597, 5894, 51521, 15, 1554, 31927, 30770, 58479, 53535, 27]"))
rng = np.random.default_rng(3)
L = ratio.numerator
h = (pkg.firdes(T * L, 0.45 / max(L, ratio.denominator), beta=7.0) * L).astype(th)
n = sum(sizes)
x = rng.standard_normal((nch, n)).astype(tx)
f = pkg.FIRFilter(h, ratio, device=0).bind(tx, nch)
xd = torch.from_numpy(x).cuda()
cuts = np.concatenate([[0], np.cumsum(sizes)])
ys = torch.zeros((len(sizes), nch, max(f.outputlength_bound(s) for s in sizes)), dtype=torch.float64, device="cuda")
torch.cuda.current_stream().synchronize()
try:
    with f.open_ring() as ring:
        print("resident", ring.info()["resident"], "grab", ring.info())
        for i in range(len(sizes)):
            ring.push(ys[i], xd[:, cuts[i]:cuts[i + 1]])
        ring.drain()
    print("OK")
except Exception as e:
    print("FAILED", str(e)[:120])
