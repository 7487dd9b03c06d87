R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
{
/opt/rocm/lib/llvm/bin/clang++ -O3 -fPIC -std=c++17 scripts/ubench/arb_recurrence_core.cpp -o /tmp/arb_core; for r in 1.0471975511965976 1.3 2.0 0.7; do /tmp/arb_core 32 $r | tail -1; done; /tmp/arb_core 10 | tail -1
MRHIP_DEBUG=2 python - <<'PY'
import sys, os, time, math
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
harb = pkg.firdes(32 * 32, 0.45 / 32, beta=7.8562) * 32
for nch in (64,):
    x = torch.rand((nch, 10_000_000), dtype=torch.float64, device="cuda")
    f = pkg.FIRFilter(harb, math.pi / 3, 32)
    f.filt(x[:, :100000]); f.reset(); torch.cuda.synchronize()
    y = None
    for rep in range(3):
        f.reset(); del y; t0 = time.perf_counter(); y = f.filt(x); torch.cuda.synchronize()
        print(f"nch={nch}: filt wall {1e3 * (time.perf_counter() - t0):.2f} ms, outputs {y.shape[-1]}", flush=True)
    f.close(); del x, y
PY
} > gpurun_out/exp_arb_host.log 2>&1
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 >> gpurun_out/exp_arb_host.log
python scripts/bench_configs.py c4 c4f > gpurun_out/configs16.jsonl 2> gpurun_out/configs16.err
