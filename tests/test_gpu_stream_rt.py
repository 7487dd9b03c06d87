"""fir_stream_rt_kernel (kernels_fir_stream_rt.hip): FIRStandard / FIRDecimator (src/Filters.jl:450-473, :598-631) with the
decimation as a run-time value.  Bit-exact against the oracle, the universal kernel and -- where both exist -- the per-M
instantiation of fir_stream_kernel, in both numerics modes; chunked, so the start-from-zero seam of support.jl:46 and the
carried history take part."""
from fractions import Fraction

import numpy as np
import pytest

from conftest import assert_bit_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


def _rand(rng, shape, tx):
    if np.issubdtype(tx, np.complexfloating):
        return (rng.random(shape) + 1j * rng.random(shape)).astype(tx)
    return rng.random(shape).astype(tx)


def _run_chunks(f, x, sizes):
    outs, pos = [], 0
    for s in sizes:
        outs.append(f.filt(x[..., pos:pos + s]))
        pos += s
    return outs


def _rt_plans(M, T, es, single=None):
    """kernels_fir_stream.hip plan_fir_stream, the run-time-M branch: one or two outputs per lane, and does a step fit two LDS
    stages (pairs of 8- and 16-byte samples: with room for two workgroups per CU)?  `single`: None = the plan's own choice."""
    bs = 8 if es == 16 else 16
    if single is None:
        single = (M * es) % 8 == 0 and not (M % bs == 0 and 128 <= M * es <= 192)
    if single and (M * es) % 8:
        single = False
    if es == 16 and M > 32:
        return False
    S = (1 if single else 2) * M * es
    cd = S // 16 if S % 16 == 0 else 0
    pad = cd if cd >= 2 and cd % 2 == 0 else 0
    opw = 64 if single else 128
    ncw = 3
    while ncw > 1 and opw * ncw * M * es > 24 * 1024:
        ncw -= 1
    nchunks = ((opw * ncw * M + T + 16) * es + 15) // 16
    if pad:
        nchunks = (nchunks + pad - 1) // pad * (pad + 1)
    nslots = (nchunks + 63) // 64
    if nslots > 60 or nslots * 2048 > 150 * 1024 or M + 16 > 256:
        return False
    return not (not single and es >= 8 and nslots * 2048 + 64 > 78 * 1024)


TYPES = ((np.float32, np.float32), (np.float32, np.complex64), (np.float64, np.float64), (np.float64, np.float32),
         (np.float64, np.complex64), (np.float64, np.complex128))


def _case(pkg, O, torch, monkeypatch, rng, M, T, th, tx, numerics, want_rt=True):
    nch = int(rng.integers(1, 5))
    n = 30_011 + 64 * M
    h = rng.standard_normal(T).astype(th)
    if T > 3:
        h[rng.integers(0, T, 2)] = 0.0
    x = _rand(rng, (nch, n), tx) - 0.5
    xr = x.view(np.float64 if tx in (np.float64, np.complex128) else np.float32)
    xr[:, 300:300 + 2 * T] = -0.0                      # all-(-0) windows: the zero-start quirk shows as a sign
    xr[0, 5000] = np.inf; xr[0, 5100] = -np.inf; xr[nch - 1, 9000:9003] = np.nan
    xd = torch.from_numpy(x).cuda()
    sizes = [10_007, 1, 13, max(T // 2, 1), 9_990]
    sizes.append(n - sum(sizes))
    ys = {}
    es = np.dtype(tx).itemsize
    for mode in ("rt", "pair", "single", "generic"):
        monkeypatch.setenv("MRHIP_STREAM_RT", "2")
        monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False); monkeypatch.delenv("MRHIP_STREAM_RT_SINGLE", raising=False)
        if mode == "generic":
            monkeypatch.setenv("MRHIP_FORCE_GENERIC", "1")
        elif mode != "rt":
            monkeypatch.setenv("MRHIP_STREAM_RT_SINGLE", "1" if mode == "single" else "0")
        f = pkg.FIRFilter(h, Fraction(1, M), numerics=numerics)
        y = torch.cat(_run_chunks(f, xd, sizes), dim=-1).cpu().numpy()
        ys[mode] = (y, f.history.copy(), f.last_kernel_name(), (f.state.phiIdx, f.state.inputDeficit))
        f.close()
    monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False); monkeypatch.delenv("MRHIP_STREAM_RT_SINGLE", raising=False)
    what = f"M={M} T={T} {th.__name__} x {tx.__name__} numerics={numerics}"
    if want_rt:
        assert ys["rt"][2] == "fir_stream_rt_kernel", (what, ys["rt"][2])
    for mode, single in (("pair", False), ("single", True)):       # the two lane maps, forced
        if _rt_plans(M, T, es, single) and not (single and (M * es) % 8):
            assert ys[mode][2] == "fir_stream_rt_kernel", (what, mode, ys[mode][2])
        assert_bit_equal(ys[mode][0], ys["generic"][0], f"{mode} vs generic {what}")
        assert_bit_equal(ys[mode][1], ys["generic"][1], f"history {mode} {what}")
        assert ys[mode][3] == ys["generic"][3]
    assert ys["generic"][2] == "poly_generic_kernel"
    assert_bit_equal(ys["rt"][0], ys["generic"][0], f"rt vs generic {what}")
    assert_bit_equal(ys["rt"][1], ys["generic"][1], f"history {what}")
    assert ys["rt"][3] == ys["generic"][3]
    O.set_fused(numerics == pkg.NUMERICS_FUSED)
    try:
        fo = O.FIRFilter(h, Fraction(1, M), tx=tx)
        yo = np.concatenate([fo.filt(p) for p in np.split(x[nch - 1], np.cumsum(sizes)[:-1])])
    finally:
        O.set_fused(False)
    ft = np.float64 if yo.dtype in (np.float64, np.complex128) else np.float32
    got, want = ys["rt"][0][nch - 1].view(ft), yo.view(ft)
    assert np.array_equal(np.isnan(got), np.isnan(want)), what
    ok = ~np.isnan(want)
    assert_bit_equal(got[ok], want[ok], f"rt vs oracle {what}")
    return ys["rt"][2]


def test_runtime_decimation_kernel_matrix(pkg, O, torch_cuda, monkeypatch):
    """Every block class of the kernel (B / A / D / the sample-by-sample one) under both lane maps (two outputs per lane, one output
    per lane -- each forced, and the plan's own choice): decimations below, at and above the block size, windows shorter than the
    decimation, tap counts that are and are not whole blocks; every sample / tap type pairing."""
    rng = np.random.default_rng(4101)
    n_rt = 0
    for M in (1, 2, 3, 5, 8, 10, 15, 16, 17, 31, 32, 33, 36, 38, 47, 48, 55):
        for T in (2, 3, 15, 16, 17, 24, 48, 127, 128, 257):
            if M not in (1, 3, 16, 17, 36) and T in (3, 15, 17, 127, 257):
                continue
            for th, tx in TYPES:
                if th == np.float64 and M not in (1, 3, 8, 17, 36, 55) and T != 48:
                    continue
                if tx == np.complex128 and T not in (16, 48, 128):
                    continue
                for numerics in (pkg.NUMERICS_STRICT, pkg.NUMERICS_FUSED):
                    if numerics == pkg.NUMERICS_FUSED and T not in (17, 48, 128):
                        continue
                    fits = _rt_plans(M, T, np.dtype(tx).itemsize)
                    k = _case(pkg, O, torch_cuda, monkeypatch, rng, M, T, th, tx, numerics, want_rt=fits)
                    n_rt += k == "fir_stream_rt_kernel"
    assert n_rt > 300


def test_runtime_decimation_kernel_large_decimations(pkg, O, torch_cuda, monkeypatch):
    """Decimations no instantiation ever covered (65 ... 109 for real Float32, up to 55 for ComplexF64) and, beyond what a stage
    holds, the fallback -- same results either way."""
    rng = np.random.default_rng(4102)
    seen = set()
    for M, th, tx in ((65, np.float32, np.float32), (77, np.float32, np.float32), (100, np.float32, np.float32), (109, np.float32, np.float32),
                      (37, np.float32, np.complex64), (51, np.float32, np.complex64), (35, np.float64, np.float64), (54, np.float64, np.float64),
                      (18, np.float64, np.complex128), (27, np.float64, np.complex128), (200, np.float32, np.float32), (64, np.float64, np.complex128)):
        for T in (24, 128, 2 * M + 5):
            for numerics in (pkg.NUMERICS_STRICT, pkg.NUMERICS_FUSED):
                seen.add(_case(pkg, O, torch_cuda, monkeypatch, rng, M, T, th, tx, numerics, want_rt=_rt_plans(M, T, np.dtype(tx).itemsize)))
        monkeypatch.setenv("MRHIP_STREAM_RT_ONE_WG", "1")          # ... and with one workgroup per CU allowed for wide samples
        _case(pkg, O, torch_cuda, monkeypatch, rng, M, 128, th, tx, pkg.NUMERICS_STRICT, want_rt=False)
        monkeypatch.delenv("MRHIP_STREAM_RT_ONE_WG")
    assert "fir_stream_rt_kernel" in seen and len(seen) >= 2, seen


def test_runtime_kernel_equals_the_per_decimation_instantiations(pkg, torch_cuda, monkeypatch):
    """MRHIP_STREAM_RT=2 (always the run-time kernel) against MRHIP_STREAM_RT=0 (only the instantiations): same bits, one
    long multi-channel call each (many tiles, dynamic grabs)."""
    torch = torch_cuda
    rng = np.random.default_rng(4103)
    for M, T, tx in ((1, 128, np.float32), (2, 64, np.complex64), (10, 128, np.float32), (15, 128, np.complex64), (13, 96, np.float32), (7, 200, np.float64)):
        th = np.float64 if tx == np.float64 else np.float32
        h = rng.standard_normal(T).astype(th)
        x = torch.from_numpy(_rand(rng, (9, 600_000), tx) - 0.5).cuda()
        ys = {}
        for mode in ("0", "2"):
            monkeypatch.setenv("MRHIP_STREAM_RT", mode)
            f = pkg.FIRFilter(h, Fraction(1, M))
            ys[mode] = (f.filt(x).cpu().numpy(), f.last_kernel_name())
            f.close()
        assert ys["0"][1] == "fir_stream_kernel" and ys["2"][1] == "fir_stream_rt_kernel", (M, T, tx, ys["0"][1], ys["2"][1])
        assert_bit_equal(ys["0"][0], ys["2"][0], f"M={M} T={T} {tx}")
