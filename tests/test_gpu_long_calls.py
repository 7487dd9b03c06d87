"""GPU tests: a filt! call longer than one launch can index (the reference takes a Vector of any length, src/Filters.jl:536-575)
is cut into launch-sized pieces inside mrhip_filt_device / mrhip_filt_host and must equal the unsplit call bit for bit --
including the one place where a piece differs from a call: the Vector seam's start-from-zero (support.jl:46), visible
as the sign of an all-(-0.0) sum."""
import math
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


CASES = [("standard", Fraction(1, 1), None), ("decimator", Fraction(1, 5), None), ("interpolator", Fraction(3, 1), None),
         ("rational", Fraction(7, 9), None), ("arbitrary", 1.2345, None), ("farrow", 0.777, 3)]


@pytest.mark.parametrize("name,ratio,polyorder", CASES)
def test_split_device_call_equals_one_reference_call(pkg, O, torch_cuda, monkeypatch, name, ratio, polyorder):
    torch = torch_cuda
    monkeypatch.setenv("MRHIP_LAUNCH_MAX", "10007")       # a 60 011-sample call becomes six launches
    rng = np.random.default_rng(17)
    h = rng.standard_normal(96).astype(np.float32)
    n = 60_011
    x = rng.standard_normal((2, n)).astype(np.float32)
    x[:, :7000] = -0.0                                     # all-(-0) sums across the first piece seams: the sign must survive
    x[1, 20_000:20_300] = -0.0
    kw = {} if polyorder is None else {"polyorder": polyorder}
    f = pkg.FIRFilter(h, ratio, 32, polyorder) if polyorder is not None else pkg.FIRFilter(h, ratio, 32)
    if polyorder is not None:
        f.bind(np.float32, 2)
        kw["pnfb"] = f.pnfb()
    y = f.filt(torch.from_numpy(x).cuda()).cpu().numpy()
    for c in range(2):
        fo = O.FIRFilter(h, ratio, 32, tx=np.float32, **kw)
        yo = fo.filt(x[c])                                 # ONE reference call
        assert_bit_equal(y[c], yo, f"{name} channel {c}")
    st, so = f.state, fo.state
    assert (st.phiIdx, st.inputDeficit, st.phiAccumulator) == (so.phiIdx, so.inputDeficit, so.phiAccumulator)
    assert_bit_equal(f.history[1], fo.history, "history")
    f.close()


@pytest.mark.parametrize("name,ratio", [("standard", Fraction(1, 1)), ("decimator", Fraction(1, 4)), ("rational", Fraction(3, 2))])
def test_host_path_pieces_keep_the_sign_of_zero(pkg, O, torch_cuda, monkeypatch, name, ratio):
    """mrhip_filt_host cuts a call into staging pieces (MRHIP_HOST_PIECE_KB): the pieces must stay invisible."""
    monkeypatch.setenv("MRHIP_HOST_PIECE_KB", "16")        # 4096 Float32 samples per piece
    rng = np.random.default_rng(18)
    h = rng.standard_normal(64).astype(np.float32)
    x = np.full(30_000, -0.0, dtype=np.float32)
    x[12_345:] = rng.standard_normal(30_000 - 12_345).astype(np.float32)
    f = pkg.FIRFilter(h, ratio)
    y = f.filt(x)                                          # numpy input: the pipelined host path
    yo = O.FIRFilter(h, ratio, tx=np.float32).filt(x)
    assert_bit_equal(y, yo, name)
    f.close()


def test_call_longer_than_2_31_samples(pkg, O, torch_cuda):
    """1 channel x 2.3e9 Float32 samples through FIRRational 147//160 in ONE mrhip_filt_device call (9.2 GB in, 8.5 GB out):
    the count, the end state and windows of the output against the oracle (a window that starts at a multiple of 160
    inputs starts at phase 1 with deficit 1 again, so the oracle can be started there on the preceding samples)."""
    torch = torch_cuda
    free, _ = torch.cuda.mem_get_info()
    n = 2_300_000_000
    if free < (n * 4 * 2 + (2 << 30)):
        pytest.skip("not enough free device memory for a 2.3e9-sample signal")
    h = pkg.firdes(24 * 147, 0.5 / 147, beta=7.8562).astype(np.float32)
    g = torch.Generator(device="cuda").manual_seed(23)
    x = torch.empty(n, dtype=torch.float32, device="cuda")
    for a in range(0, n, 1 << 28):
        x[a:a + (1 << 28)].uniform_(-1.0, 1.0, generator=g)
    f = pkg.FIRFilter(h, Fraction(147, 160))
    y = f.filt(x)
    assert y.shape[0] == -(-n * 147 // 160)
    st = f.state
    u_end = y.shape[0] * 160
    assert (st.phiIdx, st.inputDeficit) == (u_end % 147 + 1, 1 + u_end // 147 - n)
    for a in (0, 160 * 6_710_886 - 160 * 5, 160 * 13_421_773, n - n % 160 - 160 * 2000):   # start, a launch seam, beyond 2^31, the end
        lead = 160 * 2 if a else 0                         # two periods of lead-in rebuild the 23-sample history
        xo = x[a - lead:a + 40_000].cpu().numpy()
        yo = O.FIRFilter(h, Fraction(147, 160), tx=np.float32).filt(xo)
        k0 = (a - lead) // 160 * 147
        skip = lead // 160 * 147
        got = y[k0 + skip:k0 + len(yo)].cpu().numpy()
        assert_bit_equal(got, yo[skip:], f"window at {a}")
    f.close()
    del x, y
    torch.cuda.empty_cache()


@pytest.mark.parametrize("nch", [1, 5, 12])
def test_farrow_pieces_of_a_split_call_do_not_restart_the_seam(pkg, O, torch_cuda, monkeypatch, nch):
    """FIRFarrow's dot is the Vector seam variant (support.jl:46: starts from zero) for the first tapsPerPhi-1
    inputs of a CALL; the pieces mrhip_filt_device cuts a long call into continue that call, so their first outputs must not
    start from zero again -- visible only as the sign of an all-(-0.0) sum: positive taps (a constant polynomial bank), -0.0
    samples across every piece seam.  One and five channels: farrow_wave_kernel; twelve: farrow_pipe_kernel."""
    torch = torch_cuda
    monkeypatch.setenv("MRHIP_LAUNCH_MAX", "4099")
    Nphi, T = 8, 6
    h = np.ones(Nphi * T, dtype=np.float64)
    n = 20_011
    x = np.full((nch, n), -0.0, dtype=np.float64)
    x[:, 15_000:] = np.random.default_rng(4).standard_normal((nch, n - 15_000))
    f = pkg.FIRFilter(h, 0.9, Nphi, 2).bind(np.float64, nch)
    fo = O.FIRFilter(h, 0.9, Nphi, tx=np.float64, polyorder=2, pnfb=f.pnfb())
    y = f.filt(torch.from_numpy(x).cuda()).cpu().numpy()
    yo = fo.filt(x[nch - 1])                                # ONE reference call
    assert_bit_equal(y[nch - 1], yo, "split FIRFarrow call")
    assert np.signbit(yo[T + 5:5000]).all() and not np.signbit(yo[:T - 1]).any()       # -0 past the call's own seam, +0 on it
    assert f.last_kernel_name() == ("farrow_wave_kernel" if nch <= 8 else "farrow_pipe_kernel")
    f.close()


def test_launch_pieces_keep_inputs_and_outputs_below_2_31(pkg, torch_cuda):
    """a FIRRational with L > M writes more than it reads: the per-launch step of a split call must bound the
    OUTPUTS too (5//2 with 2^30 inputs per launch would have indexed 2.7e9 outputs in 31 bits).  Plan-level: the bound of a
    one-launch call (mrhip_outputlength_bound of the largest x_len a single launch takes) stays below 2^31 for every kind."""
    h = np.ones(20, dtype=np.float32)
    for ratio in (Fraction(5, 2), Fraction(441, 160), Fraction(7, 1), Fraction(2, 5), Fraction(1, 1)):
        f = pkg.FIRFilter(h, ratio).bind(np.float32, 1)
        L, M = ratio.numerator, ratio.denominator
        step = (1 << 30) * M // L if L > M else 1 << 30
        assert f.outputlength_bound(step) < 2 ** 31, (ratio, f.outputlength_bound(step))
        f.close()
