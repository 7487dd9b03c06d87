"""CPU sanitizer leg (SURVEY.md section 5 "race detection / sanitizers"; GPU AddressSanitizer is not available on this pool): the
library's host logic -- csrc/host_logic.cpp and design.cpp, linked with the test-only C face csrc/host_abi_asan.cpp -- and the
oracle, both built by gcc with -fsanitize=address,undefined (`make -C multirate.jl_amd/csrc asan`, `make -C oracle asan`), driven
from a child process that preloads libasan.  The child compares the two while it is at it: the closed form of the rational
state recurrence against the oracle's loop (src/Filters.jl:558-571), the FIRArbitrary phase schedule against the oracle's in BOTH
mod() forms (:663-673), taps2pfb, nextphase, polyfit, firdes.  Any sanitizer report fails the test."""
import os
import shutil
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent(r'''
    import ctypes as C, os, sys
    import numpy as np
    from fractions import Fraction
    ROOT = sys.argv[1]
    H = C.CDLL(os.path.join(ROOT, "multirate.jl_amd", "libmultirate_host_asan.so"))
    O = C.CDLL(os.path.join(ROOT, "oracle", "libmultirate_oracle_asan.so"))
    vp, i64, ci, cd, cl = C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_long
    H.mrhip_taps2pfb.restype = i64; H.mrhip_taps2pfb.argtypes = [vp, i64, ci, i64, vp]
    H.mrhip_nextphase.restype = i64; H.mrhip_nextphase.argtypes = [i64, i64, i64]
    H.mrhip_outputlength_ratio.restype = i64; H.mrhip_outputlength_ratio.argtypes = [i64] * 4
    H.mrhip_inputlength_ratio.restype = i64; H.mrhip_inputlength_ratio.argtypes = [i64] * 4
    H.mrhip_polyfit.restype = ci; H.mrhip_polyfit.argtypes = [vp, i64, i64, vp]
    H.mrhip_host_plan_rational.restype = ci; H.mrhip_host_plan_rational.argtypes = [ci, i64, i64, i64, i64, i64, vp]
    H.mrhip_host_arbitrary_schedule.restype = i64; H.mrhip_host_arbitrary_schedule.argtypes = [vp, cd, i64, i64, ci, vp, vp, i64]
    H.mrhip_firdes.restype = i64; H.mrhip_firdes.argtypes = [i64, vp, ci, ci, cd, cd, vp, vp]
    H.mrhip_kaiserlength.restype = ci; H.mrhip_kaiserlength.argtypes = [cd, cd, cd, vp, vp]
    O.mro_create_rational.restype = vp; O.mro_create_rational.argtypes = [vp, cl, ci, cl, cl, ci]
    O.mro_create_arbitrary.restype = vp; O.mro_create_arbitrary.argtypes = [vp, cl, ci, cd, cl, ci]
    O.mro_filt.restype = cl; O.mro_filt.argtypes = [vp, vp, cl, vp, cl]
    O.mro_filt_sched.restype = cl; O.mro_filt_sched.argtypes = [vp, vp, cl, vp, cl, vp]
    O.mro_destroy.argtypes = [vp]
    O.mro_taps2pfb.restype = cl; O.mro_taps2pfb.argtypes = [vp, cl, ci, cl, vp]
    O.mro_nextphase.restype = cl; O.mro_nextphase.argtypes = [cl, cl, cl]
    O.mro_set_mod_form.argtypes = [ci]
    class St(C.Structure):
        _fields_ = [("kind", ci), ("phiIdx", cl), ("inputDeficit", cl), ("phiAccumulator", cd), ("alpha", cd), ("delta", cd),
                    ("xIdx", cl), ("tapsPerPhi", cl), ("Nphi", cl), ("historyLen", cl), ("L", cl), ("M", cl), ("hLen", cl)]
    class Sched(C.Structure):
        _fields_ = [("xIdx", cl), ("phiIdx", cl), ("alpha", cd)]
    O.mro_get_state.argtypes = [vp, C.POINTER(St)]
    rng = np.random.default_rng(3)
    checks = 0
    # taps2pfb / nextphase: library == oracle (src/Filters.jl:284-298, 433-439)
    for hl, nphi in ((9, 4), (1, 1), (3528, 147), (100, 7), (33, 64)):
        h = rng.standard_normal(hl)
        t = H.mrhip_taps2pfb(h.ctypes.data, hl, 1, nphi, None)
        a = np.empty(t * nphi); b = np.empty(t * nphi)
        H.mrhip_taps2pfb(h.ctypes.data, hl, 1, nphi, a.ctypes.data)
        assert O.mro_taps2pfb(h.ctypes.data, hl, 1, nphi, b.ctypes.data) == t and np.array_equal(a, b)
        checks += 1
    for L in range(1, 9):
        for M in range(1, 9):
            for p in range(1, L + 1):
                assert H.mrhip_nextphase(p, L, M) == O.mro_nextphase(p, L, M)
    # the closed form of the rational state recurrence against the oracle's loop, chunk after chunk (every kind of the family)
    for (L, M) in ((147, 160), (3, 17), (1, 4), (4, 1), (1, 1), (7, 2), (160, 147)):
        g = np.gcd(L, M)
        kind = 0 if L == M else 1 if L // g == 1 else 2 if M // g == 1 else 3
        h = rng.standard_normal(max(L, M) * 3 + 1)
        f = O.mro_create_rational(h.ctypes.data, len(h), 1, L, M, 1)
        phi, d = 1, 1
        for n in (1, 2, 5, 40, 1, 333, 0, 7, 1000):
            x = rng.standard_normal(max(n, 1))
            y = np.empty(n * L // M + L + 8)
            cnt = O.mro_filt(f, x.ctypes.data, n, y.ctypes.data, len(y))
            out = (i64 * 5)()
            H.mrhip_host_plan_rational(kind, L // g, M // g, phi, d, n, out)
            st = St(); O.mro_get_state(f, C.byref(st))
            if n > 0:
                assert out[0] == cnt, (L, M, n, out[0], cnt)
                if kind in (1, 3):
                    assert (out[3] if kind == 3 else 1, out[4]) == (st.phiIdx if kind == 3 else 1, st.inputDeficit), (L, M, n, list(out), st.phiIdx, st.inputDeficit)
                phi, d = out[3], out[4]
            checks += 1
        O.mro_destroy(f)
    # the FIRArbitrary phase schedule by the library's host loop (one piece and ragged resumed pieces) == the oracle's, both mod() forms
    for (nphi, rate) in ((32, np.pi / 3), (10, np.pi / 3), (7, 1.7), (12, 0.37), (32, 0.013), (5, 23.7)):
        for form in (0, 1):
            O.mro_set_mod_form(form)
            h = rng.standard_normal(nphi * 4)
            f = O.mro_create_arbitrary(h.ctypes.data, len(h), 1, float(rate), nphi, 1)
            st_h = (cd * 4)(1.0, 1.0, 0.0, 0.0)
            for n in (3000, 1, 2, 7777, 5):
                x = rng.standard_normal(n)
                cap = int(n * rate) + 16
                y = np.empty(cap); sc = (Sched * cap)()
                cnt = O.mro_filt_sched(f, x.ctypes.data, n, y.ctypes.data, cap, C.cast(sc, vp))
                nidx = np.zeros(cap, dtype=np.int32); acc = np.zeros(cap)
                got = H.mrhip_host_arbitrary_schedule(st_h, nphi / float(rate), nphi, n, form, nidx.ctypes.data, acc.ctypes.data, cap)
                assert got == cnt, (nphi, rate, form, n, got, cnt)
                for k in range(cnt):
                    assert nidx[k] == sc[k].xIdx and int(np.floor(acc[k])) == sc[k].phiIdx and acc[k] - np.floor(acc[k]) == sc[k].alpha, (nphi, rate, form, k)
                st = St(); O.mro_get_state(f, C.byref(st))
                assert (st_h[0], int(st_h[1])) == (st.phiAccumulator, st.inputDeficit), (nphi, rate, form, n)
                checks += 1
            O.mro_destroy(f)
        O.mro_set_mod_form(0)
    # polyfit: exact on a cubic; rank deficiency reported (support.jl:85-88)
    xs = np.arange(1, 33, dtype=np.float64); yv = 2.0 - xs + 0.5 * xs ** 2 - 0.01 * xs ** 3
    coef = np.empty(4)
    assert H.mrhip_polyfit(yv.ctypes.data, 32, 3, coef.ctypes.data) == 0 and np.allclose(coef, [2.0, -1.0, 0.5, -0.01], atol=1e-9)
    assert H.mrhip_polyfit(yv.ctypes.data, 2, 3, coef.ctypes.data) != 0
    # firdes / kaiserlength (src/FIRDesign.jl:18-95): every response, odd and even lengths
    for resp, cut in ((0, [0.1]), (2, [0.2]), (1, [0.1, 0.3]), (3, [0.1, 0.3])):
        for nt in (1, 2, 31, 64):
            c = np.array(cut)
            n = H.mrhip_firdes(nt, c.ctypes.data, len(c), resp, 1.0, 6.75, None, None)
            out = np.empty(max(n, 1))
            assert n >= nt and H.mrhip_firdes(nt, c.ctypes.data, len(c), resp, 1.0, 6.75, None, out.ctypes.data) == n and np.all(np.isfinite(out[:n]))
            checks += 1
    nn, bb = i64(0), cd(0.0)
    assert H.mrhip_kaiserlength(0.05, 60.0, 1.0, C.byref(nn), C.byref(bb)) == 0 and nn.value > 0
    print("SANITIZED_OK", checks)
''')


def test_host_logic_and_oracle_under_asan_and_ubsan(tmp_path):
    if not shutil.which("gcc") or not shutil.which("g++"):
        pytest.skip("no gcc")
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not libasan or not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan for this gcc")
    if not os.path.isdir("/opt/rocm/include/hip"):
        pytest.skip("no HIP headers (host_logic.cpp includes the shared internal header)")
    subprocess.run(["make", "-C", os.path.join(ROOT, "multirate.jl_amd", "csrc"), "-s", "asan"], check=True, capture_output=True, timeout=600)
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"], check=True, capture_output=True, timeout=600)
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    p = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=900)
    report = p.stdout[-3000:] + p.stderr[-6000:]
    assert p.returncode == 0 and "SANITIZED_OK" in p.stdout, report
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error:" not in p.stderr, report
