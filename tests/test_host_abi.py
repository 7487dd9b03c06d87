"""CPU tests: the C-ABI library loads and exports every symbol include/multirate_hip.h declares,
host-only helpers agree with the oracle, and the product path fails loudly without a GPU."""
import ctypes as C
import math
import os
import re
from fractions import Fraction

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "multirate_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mrhip_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_header_symbol(pkg):
    lib = C.CDLL(pkg.library_path())
    syms = header_symbols()
    assert len(syms) >= 28
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in multirate_hip.h but not exported"
    # and the Python binding table covers the header exactly
    from multirate_jl_amd import host
    assert sorted(n for n, _, _ in host.ABI) == syms


def test_abi_version_and_error_string(pkg):
    lib = pkg.load_library()
    assert lib.mrhip_abi_version() == 1
    assert isinstance(lib.mrhip_last_error(), bytes)


def test_taps2pfb_matches_reference_example_and_oracle(pkg, O, known_answers):
    ka = known_answers["taps2pfb"]
    assert np.array_equal(pkg.taps2pfb(np.array(ka["h"], dtype=np.float64), ka["Nphi"]),
                          np.array(ka["pfb_rows"], dtype=np.float64))
    rng = np.random.default_rng(0)
    for _ in range(50):
        n, L = int(rng.integers(1, 200)), int(rng.integers(1, 40))
        for dt in (np.float32, np.float64):
            h = rng.random(n).astype(dt)
            assert np.array_equal(pkg.taps2pfb(h, L), O.taps2pfb(h, L))


def test_nextphase_and_lengths_match_oracle(pkg, O):
    lib, olib = pkg.load_library(), O.lib()
    for L in range(1, 12):
        for M in range(1, 12):
            if math.gcd(L, M) != 1:
                continue
            for p in range(1, L + 1):
                assert pkg.nextphase(p, Fraction(L, M)) == O.nextphase(p, L, M)
                for n in (0, 1, 2, 7, 100, 12345):
                    assert lib.mrhip_outputlength_ratio(n, L, M, p) == olib.mro_outputlength_ratio(n, L, M, p)
                    assert lib.mrhip_inputlength_ratio(n, L, M, p) == olib.mro_inputlength_ratio(n, L, M, p)
    for th in range(2):
        for tx in range(4):
            assert lib.mrhip_output_dtype(th, tx) == olib.mro_output_dtype(th, tx)


def test_kernel_selection_matches_reference_constructor(pkg):
    h = np.ones(30)
    K = {(1, 1): "FIRStandard", (3, 3): "FIRStandard", (1, 4): "FIRDecimator", (2, 8): "FIRDecimator",
         (4, 1): "FIRInterpolator", (147, 160): "FIRRational", (6, 4): "FIRRational"}
    for (a, b), name in K.items():
        f = pkg.FIRFilter(h, Fraction(a, b))            # src/Filters.jl:163-175
        assert f.kernel_name == name
        assert f.historyLen == (29 if name in ("FIRStandard", "FIRDecimator") else f.tapsPerPhi - 1)
    fa = pkg.FIRFilter(h, 0.77, 8)
    assert fa.kernel_name == "FIRArbitrary" and fa.tapsPerPhi == 4 and fa.historyLen == 3
    with pytest.raises(pkg.MultirateHIPError):
        pkg.FIRFilter(h, -1.0)                          # "rate must be greater than 0", Filters.jl:184
    with pytest.raises(pkg.MultirateHIPError):
        pkg.FIRFilter(np.ones(4, dtype=np.complex64))   # complex taps unsupported


def test_firdes_and_kaiserlength(pkg):
    n, beta = pkg.kaiserlength(0.05, samplerate=32)     # test/runtests.jl:336
    assert n == int(math.ceil((60 - 7.95) / (2 * math.pi * 2.285 * 0.05 / 32)))
    assert abs(beta - 0.1102 * (60 - 8.7)) < 1e-12
    h = pkg.firdes(3528, 0.5 / 147, beta=7.8562)
    assert h.shape == (3528,) and np.allclose(h, h[::-1]) and abs(h.sum() - 1) < 1e-3


def test_firdes_all_responses_match_windowed_sinc(pkg):
    """src/FIRDesign.jl:47-95: the four FIRResponse prototypes times a Kaiser window, against scipy's
    independent windowed-sinc design (firwin, unscaled): equal to rounding.  F is in cycles/sample
    (the reference's convention), firwin's cutoff is relative to Nyquist."""
    import scipy.signal as sg
    from multirate_jl_amd import design as D
    beta = 6.3
    assert np.abs(pkg.firdes(101, 0.2, beta=beta) - sg.firwin(101, 0.4, window=("kaiser", beta), scale=False)).max() < 1e-15
    hb = D.firdes(101, [0.3, 0.1], response=D.BANDPASS, beta=beta)      # 2(F1 sinc - F2 sinc): pass band F2..F1
    assert np.abs(hb - sg.firwin(101, [0.2, 0.6], window=("kaiser", beta), pass_zero=False, scale=False)).max() < 1e-15
    # BANDSTOP as the reference writes it, 2(F2 sinc - F1 sinc) (FIRDesign.jl:59), has no unit impulse: it is the
    # BANDPASS formula with the pair swapped -- reproduced as written (a reference quirk, not "fixed" here)
    hs = D.firdes(101, [0.1, 0.3], response=D.BANDSTOP, beta=beta)
    assert np.abs(hs - sg.firwin(101, [0.2, 0.6], window=("kaiser", beta), pass_zero=False, scale=False)).max() < 1e-15
    hh = D.firdes(100, 0.2, response=D.HIGHPASS, beta=beta)             # even numtaps -> one more tap (:55)
    assert len(hh) == 101
    assert np.abs(hh - sg.firwin(101, 0.4, window=("kaiser", beta), pass_zero=False, scale=False)).max() < 1e-15
    # firdes(cutoff, transitionwidth, attenuation): length and beta from kaiserlength (:90-95)
    n, b = pkg.kaiserlength(0.05, 80.0)
    h2 = D.firdes(0.1, 0.05, 80.0)
    assert len(h2) == n and np.array_equal(h2, D.firdes(n, 0.1, beta=b))
    assert D.firdes(0.1, 0.05, samplerate=2.0).shape == (pkg.kaiserlength(0.05, 60.0, samplerate=2.0)[0],)
    with pytest.raises(ValueError):
        D.firprototype(11, 0.1, response=7)
    # custom window function (FIRDesign.jl:85)
    assert np.array_equal(D.firdes(21, 0.1, np.hanning), D.firprototype(21, 0.1) * np.hanning(21))


def test_design_entry_points_match_numpy_restatement(pkg):
    """mrhip_kaiser / mrhip_firprototype / mrhip_firdes / mrhip_kaiserlength (csrc/design.cpp, the entry points the
    Julia shim binds) against an independent numpy restatement of src/FIRDesign.jl:18-95 (np.sinc, np.kaiser)."""
    from multirate_jl_amd import design as D
    rng = np.random.default_rng(11)

    def proto(numtaps, F, response):
        M = numtaps - 1
        if response == D.HIGHPASS and M % 2:
            M += 1
        n = np.arange(M + 1, dtype=np.float64) - M / 2.0
        if response == D.LOWPASS:
            return 2.0 * F * np.sinc(2.0 * F * n)
        if response == D.BANDPASS:
            return 2.0 * (F[0] * np.sinc(2.0 * F[0] * n) - F[1] * np.sinc(2.0 * F[1] * n))
        if response == D.HIGHPASS:
            return np.sinc(n) - 2.0 * F * np.sinc(2.0 * F * n)
        return 2.0 * (F[1] * np.sinc(2.0 * F[1] * n) - F[0] * np.sinc(2.0 * F[0] * n))

    for n in (1, 2, 3, 16, 129, 3528):
        for beta in (0.0, 2.5, 7.8562, 14.0):
            assert np.allclose(D.kaiser(n, beta), np.kaiser(n, beta), rtol=1e-13, atol=1e-16)
    for _ in range(40):
        numtaps = int(rng.integers(2, 400))
        f1, f2 = sorted(rng.uniform(0.01, 0.49, 2))
        sr = float(rng.choice([1.0, 2.0, 48000.0]))
        beta = float(rng.uniform(0, 12))
        for resp, F in ((D.LOWPASS, f1), (D.HIGHPASS, f1), (D.BANDPASS, [f2, f1]), (D.BANDSTOP, [f1, f2])):
            want = proto(numtaps, F, resp)
            got = D.firprototype(numtaps, F, resp)
            assert got.shape == want.shape and np.allclose(got, want, rtol=0, atol=2e-16 * max(1.0, np.abs(want).max()) * 8)
            Fs = [v * sr for v in F] if isinstance(F, list) else F * sr
            h = D.firdes(numtaps, Fs, response=resp, samplerate=sr, beta=beta)
            assert np.allclose(h, want * np.kaiser(len(want), beta), rtol=0, atol=1e-14)
    for tw, att, sr in ((0.05, 60.0, 1.0), (0.05, 80.0, 32.0), (0.01, 30.0, 1.0), (0.2, 10.0, 2.0), (100.0, 50.0, 48000.0)):
        n, b = D.kaiserlength(tw, att, sr)
        assert n == int(math.ceil((att - 7.95) / (2 * math.pi * 2.285 * tw / sr)))
        want_b = 0.1102 * (att - 8.7) if att > 50 else (0.5842 * (att - 21) ** 0.4 + 0.07886 * (att - 21) if att >= 21 else 0.0)
        assert abs(b - want_b) <= 1e-15 * max(1.0, want_b)
    lib = pkg.load_library()
    assert lib.mrhip_firprototype(10, None, 1, 0, None) == -1 and b"numtaps" in lib.mrhip_last_error()


def test_mrhip_state_layout_matches_julia_shim(pkg, tmp_path):
    """No Julia runs here, so the struct the shim reads through mrhip_get_state (MRHIPState in MultirateHIP.jl, an
    isbits struct laid out like C) is pinned by a C program: it prints sizeof/offsetof of mrhip_state from the real
    header; the shim's field list, laid out by the C rules for its declared Julia types, must give the same numbers.
    The same program links the library and calls a few host-only entry points the way a C host would."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    fields = ["kind", "tap_dtype", "sample_dtype", "output_dtype", "nchannels", "hLen", "interpolation", "decimation", "Nphi",
              "tapsPerPhi", "historyLen", "phiIdx", "inputDeficit", "xIdx", "rate", "phiAccumulator", "alpha", "delta"]
    src = tmp_path / "layout.c"
    src.write_text(
        "#include <stdio.h>\n#include <stddef.h>\n#include \"multirate_hip.h\"\n"
        "int main(void) {\n  printf(\"sizeof %zu\\n\", sizeof(mrhip_state));\n"
        + "".join(f'  printf("{f} %zu %zu\\n", offsetof(mrhip_state, {f}), sizeof(((mrhip_state *)0)->{f}));\n' for f in fields)
        + "  double w[5]; if (mrhip_kaiser(5, 3.0, w) != 0) return 2;\n"
          "  int64_t n = 0; double b = 0; if (mrhip_kaiserlength(0.05, 60.0, 1.0, &n, &b) != 0) return 3;\n"
          "  printf(\"abi %d kaiser_mid %.17g numtaps %lld nextphase %lld\\n\", mrhip_abi_version(), w[2], (long long)n,\n"
          "         (long long)mrhip_nextphase(1, 147, 160));\n  return 0;\n}\n")
    exe = tmp_path / "layout"
    libdir = os.path.dirname(pkg.library_path())
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lmultirate_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe)], text=True).split("\n")
    c_layout = {ln.split()[0]: tuple(int(v) for v in ln.split()[1:]) for ln in out if ln and ln.split()[0] in fields}
    c_size = int(out[0].split()[1])
    assert "abi 1 kaiser_mid 1 numtaps" in out[-2] and out[-2].endswith("nextphase 14")
    # the shim's struct: parse `struct MRHIPState ... end` and lay it out with the C rules (natural alignment)
    jl = open(os.path.join(ROOT, "multirate.jl_amd", "julia", "MultirateHIP.jl")).read()
    body = re.search(r"struct MRHIPState[^\n]*\n(.*?)\nend", jl, flags=re.S).group(1)
    jl_fields = re.findall(r"(\w+)::(Int32|Int64|Float64)", body)
    size_of = {"Int32": 4, "Int64": 8, "Float64": 8}
    off, jl_layout = 0, {}
    for name, ty in jl_fields:
        sz = size_of[ty]
        off = (off + sz - 1) // sz * sz
        jl_layout[name] = (off, sz)
        off += sz
    jl_size = (off + 7) // 8 * 8
    assert [n for n, _ in jl_fields] == fields, "field order of MRHIPState differs from mrhip_state"
    assert jl_layout == c_layout and jl_size == c_size == 128
    # ... and the ctypes mirror the tests drive
    from multirate_jl_amd import host
    assert C.sizeof(host._State) == c_size
    for f in fields:
        assert (getattr(host._State, f).offset, getattr(host._State, f).size) == c_layout[f]


def test_julia_shim_ccalls_match_header():
    """The Julia shim cannot be executed here: check statically that every `ccall((:mrhip_x, libmr), ret, (argtypes...), ...)`
    names a symbol the header declares, with the header's number of parameters."""
    hdr = open(os.path.join(ROOT, "include", "multirate_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    arity = {}
    for m in re.finditer(r"\b(mrhip_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        params = m.group(2).strip()
        arity[m.group(1)] = 0 if params in ("", "void") else params.count(",") + 1
    jl = open(os.path.join(ROOT, "multirate.jl_amd", "julia", "MultirateHIP.jl")).read()
    calls = re.findall(r"ccall\(\(:(mrhip_[a-z0-9_]+), libmr\),\s*\w+,\s*(\w+|\((?:[^()]|\([^()]*\))*\))", jl)
    assert len(calls) >= 25
    seen = set()
    for name, argt in calls:
        assert name in arity, f"{name} is not declared in multirate_hip.h"
        if argt.startswith("("):
            inner = argt[1:-1].strip().rstrip(",")
            n = 0 if not inner else len(re.sub(r"\{[^{}]*\}", "", re.sub(r"\{[^{}]*\}", "", inner)).split(","))
            assert n == arity[name], f"{name}: shim passes {n} arguments, header declares {arity[name]}"
        seen.add(name)
    # ... and with the header's parameter and return TYPES (Cint <-> int / enum, Int64 <-> int64_t, Cdouble <-> double,
    # Ptr{...} / Cstring <-> pointer): a swapped Cint / Int64 would pass the arity check and corrupt the call
    def c_class(t):
        t = re.sub(r"\bconst\b", "", t).strip()
        if "*" in t:
            return "ptr"
        base = t.split()[0] if t.split() else t
        return {"int": "i32", "int32_t": "i32", "int64_t": "i64", "double": "f64", "void": "void", "mrhip_status": "i32",
                "mrhip_dtype": "i32", "mrhip_kind": "i32", "mrhip_numerics": "i32", "size_t": "u64", "uint64_t": "u64"}[base]

    def jl_class(t):
        t = t.strip()
        if t.startswith("Ptr") or t.startswith("Ref") or t == "Cstring":
            return "ptr"
        return {"Cint": "i32", "Int32": "i32", "Int64": "i64", "Cdouble": "f64", "Float64": "f64", "Cvoid": "void", "Csize_t": "u64", "UInt64": "u64"}[t]

    proto = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ ]*?[ \*]+)(mrhip_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        ret, name, params = m.group(1), m.group(2), m.group(3).strip()
        ptypes = []
        if params not in ("", "void"):
            for prm in params.split(","):
                prm = prm.strip()
                ptypes.append(c_class(prm if "*" in prm else " ".join(prm.split()[:-1])))
        proto[name] = (c_class(ret), ptypes)
    checked = 0
    for m in re.finditer(r"ccall\(\(:(mrhip_[a-z0-9_]+), libmr\),\s*(\w+),\s*(\w+|\((?:[^()]|\([^()]*\))*\))", jl):
        name, ret, argt = m.group(1), m.group(2), m.group(3)
        if not argt.startswith("("):          # a named tuple of argument types (`args`): resolved below
            defs = [t for t in re.finditer(r"\b" + argt + r"\s*=\s*(\((?:[^()]|\([^()]*\))*\))", jl) if t.start() < m.start()]
            assert defs, f"{name}: cannot resolve the argument-type tuple `{argt}`"
            argt = defs[-1].group(1)                    # the nearest definition above the call
        inner = argt[1:-1].strip().rstrip(",")
        flat = re.sub(r"\{[^{}]*\}", "", re.sub(r"\{[^{}]*\}", "", inner))
        jt = [jl_class(a) for a in flat.split(",")] if flat else []
        want_ret, want = proto[name]
        assert jl_class(ret) == want_ret, f"{name}: shim return {ret}, header {want_ret}"
        assert jt == want, f"{name}: shim argument types {jt}, header {want}"
        checked += 1
    assert checked >= 30
    # the shim binds the whole design / cascade / hot-path surface
    for must in ("mrhip_firdes", "mrhip_firdes_kaiser", "mrhip_firprototype", "mrhip_kaiserlength", "mrhip_kaiser",
                 "mrhip_arbitrary_tapsforphase", "mrhip_farrow_tapsforphase", "mrhip_cascade_create", "mrhip_cascade_filt_device",
                 "mrhip_filt_host", "mrhip_filt_device", "mrhip_filt_device_chunked", "mrhip_create_rational",
                 "mrhip_create_arbitrary", "mrhip_create_farrow", "mrhip_get_state", "mrhip_set_state", "mrhip_reset", "mrhip_schedule_info"):
        assert must in seen, must
    # the reference's exported names (src/Multirate.jl:26-41) and the kernel-field access of its own example (examples/FIRFarrow.jl:23-30)
    for name in ("tapsforphase!", "tapsforphase", "setphase", "reset", "outputlength", "inputlength", "taps2pfb", "filt!", "filt", "firdes", "kaiserlength"):
        assert re.search(r"export[^#]*\b" + re.escape(name), jl, flags=re.S), name
    assert "Base.getproperty(f::FIRFilter, name::Symbol)" in jl and "name === :kernel" in jl
    for field in (":inputDeficit", '"𝜙Idx"', '"α"', '"𝜙Accumulator"'):
        assert field in jl


def test_product_path_fails_loudly_without_gpu(pkg):
    """No CPU fallback: on a box without a gfx950 device constructing the device object raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    f = pkg.FIRFilter(np.ones(8, dtype=np.float32), Fraction(2, 3))
    with pytest.raises(pkg.MultirateHIPError) as ei:
        f.filt(np.ones(16, dtype=np.float32))
    assert ei.value.code == 4                           # MRHIP_ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value) or "gfx950" in str(ei.value)


def test_package_never_imports_oracle(pkg):
    """The shipped path must not route through oracle/ (nor reference it) in any form."""
    pdir = os.path.dirname(pkg.library_path())
    for dp, _, files in os.walk(pdir):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h", ".jl")):
                txt = open(os.path.join(dp, fn), errors="replace").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libmultirate_oracle" not in txt, fn
                assert "mro_" not in txt, fn


def test_julia_shim_defines_every_name_the_reference_exports():
    """src/Multirate.jl:15-41 is the reference's export list for FIRDesign.jl and Filters.jl (the names as data; `HIGPASS` there is a
    typo for the HIGHPASS that src/FIRDesign.jl:7 defines).  The shim cannot be executed here: check statically that each name is
    exported AND defined in MultirateHIP.jl, and that the kernel types are parametric in the tap type like the reference's
    (`type FIRRational{T} <: FIRKernel`, src/Filters.jl:15,28,45,62,91,123), so that FIRFilter{FIRRational{Float32}} names a type."""
    exported_by_reference = ["firdes", "kaiserlength", "FIRResponse", "LOWPASS", "HIGHPASS", "BANDPASS", "BANDSTOP",          # Multirate.jl:15-22
                             "FIRFilter", "FIRInterpolator", "FIRArbitrary", "FIRDecimator", "FIRFarrow", "FIRRational", "FIRStandard",
                             "filt!", "filt", "setphase", "tapsforphase!", "tapsforphase", "taps2pfb", "reset", "outputlength", "inputlength"]   # :26-41
    jl = open(os.path.join(ROOT, "multirate.jl_amd", "julia", "MultirateHIP.jl")).read()
    exports = set(re.findall(r"[\w!]+", " ".join(re.findall(r"^export ((?:[^\n]*,\n)*[^\n]*)", jl, flags=re.M))))
    missing = []
    for name in exported_by_reference:
        q = re.escape(name)
        defined = re.search(r"^(?:function |mutable struct |struct |@enum |const )?(?:Base\.)?" + q + r"(?:\{[^}]*\})?\s*(?:\(|<:|=|$)", jl, flags=re.M) or \
            re.search(r"@enum [^\n]*\b" + q + r"\b", jl)
        if name not in exports or not defined:
            missing.append(name)
    assert not missing, f"exported by the reference but not exported / defined in the shim: {missing}"
    for kind in ("FIRStandard", "FIRDecimator", "FIRInterpolator", "FIRRational", "FIRArbitrary", "FIRFarrow"):
        assert re.search(r"^struct " + kind + r"\{T\} <: FIRKernel end", jl, flags=re.M), f"{kind} is not parametric in the tap type"
    # the constructors instantiate the parametric kernel type with the tap type (FIRFilter{FIRRational{Float32}} for Float32 taps)
    assert "FIRFilter{kindof(r){Th}}" in jl and "FIRFilter{FIRArbitrary{Th}}" in jl and "FIRFilter{FIRFarrow{Th}}" in jl
    # no method is left dispatching on the bare (now UnionAll) kernel name as an exact type parameter
    assert not re.search(r"FIRFilter\{FIR(?:Standard|Decimator|Interpolator|Rational|Arbitrary|Farrow)\}", jl)
