"""repro: rate 2.5 N𝜙 10 mixed synchronous / asynchronous calls (stress seed 111)"""
import itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
rng = np.random.default_rng(0)
rate, nphi, hl, nch = 2.5, 10, 36, 3
sizes = [17696, 104308, 6459, 21814]
h = rng.standard_normal(hl).astype(np.float32)
x = torch.from_numpy(rng.standard_normal((nch, sum(sizes))).astype(np.float32)).cuda()
os.environ["MRHIP_SCHED_DEVICE"] = "0"
g = pkg.FIRFilter(h, rate, nphi)
ref, pos = [], 0
for s in sizes:
    ref.append(g.filt(x[:, pos:pos + s])); pos += s
os.environ.pop("MRHIP_SCHED_DEVICE")
bad = 0
for pat in itertools.product([0, 1], repeat=4):
    f = pkg.FIRFilter(h, rate, nphi).bind(np.float32, nch)
    cnt = torch.full((4,), -1, dtype=torch.int64, device="cuda")
    outs, pos = [], 0
    try:
        for i, s in enumerate(sizes):
            if pat[i]:
                yb = torch.empty((nch, f.outputlength_bound(s)), dtype=torch.float32, device="cuda")
                f.filt_into_async(yb, x[:, pos:pos + s], cnt[i:i + 1])
                outs.append(yb)
            else:
                outs.append(f.filt(x[:, pos:pos + s]))
            pos += s
        f.sync_state()
        c = cnt.cpu().tolist()
        ok = all(torch.equal((o[:, :c[i]] if pat[i] else o), ref[i]) for i, o in enumerate(outs))
        print(pat, "ok" if ok else "WRONG", f.schedule_info(), flush=True)
        bad += not ok
    except Exception as e:
        print(pat, "EXC", e, f.schedule_info(), flush=True)
        bad += 1
    f.close()
print("bad", bad)
