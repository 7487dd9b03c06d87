import os, sys, math, time, subprocess
# each configuration in a fresh process (the knobs are read once per process)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ".")
    import numpy as np, torch
    import __graft_entry__ as ge
    pkg = ge.load_package()
    th, dt = (np.float32, torch.float32) if sys.argv[2] == "f32" else (np.float64, torch.float64)
    nch, n = 256, 2_000_000
    h = (pkg.firdes(32 * 32, 0.45 / 32, beta=7.8562) * 32).astype(th)
    x = torch.rand((nch, n), device="cuda", dtype=dt)
    f = pkg.FIRFilter(h, float(math.pi / 3), 32)
    y = f.filt(x); f.set_timing(True)
    for _ in range(3): f.reset(); y = f.filt(x)
    torch.cuda.synchronize(); nl, ms = f.timing_read(); per = ms / 3
    b = nch * n * (x.element_size() + y.element_size() * math.pi / 3)
    print(f"{sys.argv[2]} CPL={os.environ.get('MRHIP_ARB_CPL','-')} TILE={os.environ.get('MRHIP_ARB_TILE','-')}: kernel {per:.3f} ms {b / (per * 1e-3) / 8e12 * 100:.1f} % HBM", flush=True)
else:
    for dtn in ("f32", "f64"):
        for cpl in ("", "2", "4", "8"):
            for tile in ("", "512", "1024"):
                env = dict(os.environ)
                if cpl: env["MRHIP_ARB_CPL"] = cpl
                if tile: env["MRHIP_ARB_TILE"] = tile
                subprocess.run([sys.executable, sys.argv[0], "child", dtn], env=env, stderr=subprocess.DEVNULL)
