"""GPU parity tests (need one MI355X): the HIP path, called through the C ABI, against the CPU oracle
and the committed golden vectors.  Bar: BIT-EXACT for every dtype (both sides evaluate the dot
product in the order the reference source states, separately rounded multiply/add), which is
inside the north star's "within 1 ULP" for Float32/Float64/ComplexF32."""
import math
import os
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


def test_extension_is_loaded_and_device_is_gfx950(pkg, torch_cuda):
    lib = pkg.load_library()
    assert lib.mrhip_device_count() >= 1
    maps = open("/proc/self/maps").read()
    assert "libmultirate_hip.so" in maps, "native library not loaded"


def test_reference_known_answers_on_gpu(pkg, known_answers, torch_cuda):
    ka = known_answers["readme_stream"]                  # README.md:58-141
    x = np.arange(ka["x_first"], ka["x_last"] + 1, dtype=np.float64)
    h = np.array(ka["h"], dtype=np.float64)
    f = pkg.FIRFilter(h, Fraction(*ka["ratio"]))
    ys = []
    for (a, b), want in zip(ka["chunks"], ka["y"]):
        y = f.filt(x[a - 1:b])
        assert y.tolist() == want
        ys.append(y)
    assert np.array_equal(f.taps(), np.array(ka["pfb_rows"], dtype=np.float64))
    st = f.state
    assert (st.phiIdx, st.inputDeficit) == (1, 3)
    assert np.sum(np.concatenate(ys) - pkg.filt(h, x, Fraction(*ka["ratio"]))) == 0.0


def test_notebook_farrow_output_length_on_gpu(pkg, O, known_answers, torch_cuda):
    """The reference's fourth held datum (doc/Polyphase Filtering Explained.ipynb, last cell): 40 samples through
    FIRFilter(h, float64(pi), 32, 4) give 126 outputs -- FIRFarrow and FIRArbitrary, on the GPU; values == oracle."""
    ka = known_answers["notebook_farrow"]
    rate = math.pi
    t = np.arange(ka["Nx"])
    x = np.cos(2 * np.pi * ka["xf1"] * t) + 0.5 * np.sin(2 * np.pi * ka["xf2"] * t * np.pi)
    hLen = ka["tapsPerPhi"] * ka["Nphi"]
    h = pkg.firdes(hLen, min(0.45 / ka["Nphi"], rate / ka["Nphi"]), beta=7.8562) * ka["Nphi"]
    ff = pkg.FIRFilter(h, rate, ka["Nphi"], ka["polyorder"])
    y = ff.filt(x)
    assert len(y) == ka["len_y"]
    assert_bit_equal(y, O.FIRFilter(h, rate, ka["Nphi"], tx=np.float64, polyorder=ka["polyorder"], pnfb=ff.pnfb()).filt(x), "notebook farrow")
    ya = pkg.FIRFilter(h, rate, ka["Nphi"]).filt(x)
    assert len(ya) == ka["len_y"]
    assert_bit_equal(ya, O.FIRFilter(h, rate, ka["Nphi"], tx=np.float64).filt(x), "notebook arbitrary")


def test_golden_vectors_host_path(pkg, golden, torch_cuda):
    """Every committed fixture through mrhip_filt_host: outputs, per-chunk counts, end state and
    history, all bit-exact."""
    meta, data = golden
    for m in meta:
        k = m["id"]
        h, x, sizes = data[k + "_h"], data[k + "_x"], data[k + "_sizes"]
        ratio = Fraction(m["L"], m["M"]) if m["kind"] == "rational" else float(m["rate"])
        f = pkg.FIRFilter(h, ratio, m.get("Nphi", 32))
        outs = _run_chunks(f, x, sizes)
        assert_bit_equal(np.concatenate(outs), data[k + "_y"], f"{k} {m}")
        assert [len(o) for o in outs] == data[k + "_counts"].tolist(), k
        st = f.state
        assert [st.phiIdx, st.inputDeficit] == data[k + "_state"].tolist(), k
        if m["kind"] == "arbitrary":
            assert st.phiAccumulator == float(data[k + "_acc"][0])
        assert_bit_equal(f.history, data[k + "_hist"], k + " history")
        f.close()


@pytest.mark.parametrize("seed", range(3))
def test_random_sweep_device_path_multichannel(pkg, O, torch_cuda, seed):
    """test_all of the reference (runtests.jl:389-421) re-run against the oracle: random L, M, hLen,
    dtypes; stateless, pivot split and ragged chunkings; 1..5 channels batched; torch device path."""
    torch = torch_cuda
    rng = np.random.default_rng(1000 + seed)
    for _ in range(40):
        L, M = int(rng.integers(1, 33)), int(rng.integers(1, 33))
        th = rng.choice([np.float32, np.float64])
        tx = rng.choice([np.float32, np.float64, np.complex64, np.complex128])
        nch = int(rng.integers(1, 6))
        h = rng.random(int(rng.integers(1, 129))).astype(th)
        n = int(rng.integers(200, 301))
        x = _rand(rng, (nch, n), tx)
        fr = Fraction(L, M)
        cuts = sorted(set(rng.integers(0, n + 1, size=int(rng.integers(0, 6))).tolist()))
        sizes = np.diff([0] + cuts + [n]).tolist()        # may contain zero-length chunks
        f = pkg.FIRFilter(h, fr)
        xd = torch.from_numpy(x).cuda()
        outs = [o.cpu().numpy() for o in _run_chunks(f, xd, sizes)]
        y = np.concatenate(outs, axis=1)
        for c in range(nch):
            fo = O.FIRFilter(h, fr, tx=tx)
            yo = np.concatenate([fo.filt(x[c, a:a + s]) for a, s in zip(np.cumsum([0] + sizes[:-1]), sizes)])
            assert_bit_equal(y[c], yo, f"L={L} M={M} th={th} tx={tx} ch={c} sizes={sizes}")
        so, st = fo.state, f.state
        assert (st.phiIdx, st.inputDeficit) == (so.phiIdx, so.inputDeficit)
        assert_bit_equal(f.history.reshape(nch, -1)[nch - 1], fo.history, "history")
        f.close()


def test_arbitrary_sweep(pkg, O, torch_cuda):
    torch = torch_cuda
    rng = np.random.default_rng(77)
    for trial in range(12):
        Nphi = int(rng.choice([8, 32, 10]))
        T = int(rng.integers(2, 40))
        th = rng.choice([np.float32, np.float64])
        tx = rng.choice([np.float32, np.float64, np.complex64, np.complex128])
        h = (pkg.firdes(T * Nphi - int(rng.integers(0, Nphi)), 0.45 / Nphi, beta=7.0) * Nphi).astype(th)
        rate = float(rng.choice([math.pi / 3, 0.1234, 1.0, 2.5, 31.7, 1 / 2.123456789]))
        nch = int(rng.integers(1, 4))
        n = 700
        x = _rand(rng, (nch, n), tx)
        sizes = [n] if trial % 3 == 0 else ([1] * 40 + [n - 40] if trial % 3 == 1 else [13] * (n // 13) + [n % 13])
        f = pkg.FIRFilter(h, rate, Nphi)
        outs = [o.cpu().numpy() for o in _run_chunks(f, torch.from_numpy(x).cuda(), sizes)]
        y = np.concatenate(outs, axis=1)
        for c in range(nch):
            fo = O.FIRFilter(h, rate, Nphi, tx=tx)
            yo = np.concatenate(_run_chunks(fo, x[c], sizes))
            assert_bit_equal(y[c], yo, f"arb rate={rate} Nphi={Nphi} T={T} {th} {tx}")
        assert f.state.phiAccumulator == fo.state.phiAccumulator
        assert f.state.inputDeficit == fo.state.inputDeficit
        f.close()


def test_config4_arbitrary_vs_naive_tolerance(pkg, torch_cuda):
    """BASELINE config 4: FIRArbitrary pi/3, 32 taps x 32 filters, Float64, 64 channels.  Stated
    tolerance vs NaiveResamplers' algorithm: |h[end]|*max|x| + 1e-12."""
    from oracle import naive as N
    torch = torch_cuda
    h = pkg.firdes(1024, 0.45 / 32, beta=7.8562) * 32
    rng = np.random.default_rng(9)
    x = rng.random((64, 3000))
    f = pkg.FIRFilter(h, float(math.pi / 3), 32)
    y = f.filt(torch.from_numpy(x).cuda()).cpu().numpy()
    for c in (0, 17, 63):
        ref = N.naive_arbitrary(h, x[c], float(math.pi / 3), 32)
        n = min(y.shape[1], len(ref))
        assert abs(y.shape[1] - len(ref)) <= 1
        assert np.max(np.abs(y[c, :n] - ref[:n])) <= abs(h[-1]) * 1.0 + 1e-12


def test_config3_integer_kernels_complex_batched(pkg, O, torch_cuda):
    """BASELINE config 3: interpolator 4//1 and decimator 1//4, 128 taps, ComplexF32, 256 channels."""
    torch = torch_cuda
    rng = np.random.default_rng(11)
    h = pkg.firdes(128, 0.5 / 4, beta=7.8562).astype(np.float32)
    x = _rand(rng, (256, 4096), np.complex64)
    xd = torch.from_numpy(x).cuda()
    for ratio in (Fraction(4, 1), Fraction(1, 4)):
        f = pkg.FIRFilter(h, ratio)
        y = torch.cat(_run_chunks(f, xd, [1000, 3, 3093]), dim=1).cpu().numpy()
        for c in (0, 1, 128, 255):
            fo = O.FIRFilter(h, ratio, tx=np.complex64)
            yo = np.concatenate(_run_chunks(fo, x[c], [1000, 3, 3093]))
            assert_bit_equal(y[c], yo, f"ratio {ratio} ch {c}")
        f.close()


def test_errors_and_short_inputs(pkg, torch_cuda):
    h = np.arange(1, 40, dtype=np.float64)
    f = pkg.FIRFilter(h, Fraction(1, 8))
    assert len(f.filt(np.ones(1))) == 1
    assert f.state.inputDeficit == 8
    assert len(f.filt(np.ones(3))) == 0                  # Filters.jl:638-643
    assert f.state.inputDeficit == 5
    assert len(f.filt(np.ones(0))) == 0
    assert len(f.filt(np.ones(5))) == 1
    # buffer too small: error, state untouched (Filters.jl:550)
    f2 = pkg.FIRFilter(h, Fraction(3, 2))
    f2.filt(np.ones(10))
    before = (f2.state.phiIdx, f2.state.inputDeficit)
    with pytest.raises(pkg.MultirateHIPError) as ei:
        pkg.filt_(np.empty(3), f2, np.ones(100))
    assert ei.value.code == 2 and "buffer is too small" in str(ei.value)
    assert (f2.state.phiIdx, f2.state.inputDeficit) == before
    # filt! return values: count for Rational, buffer for Standard (Filters.jl:574, :472)
    buf = np.empty(200)
    want = f2.next_output_count(100)
    got = pkg.filt_(buf, f2, np.ones(100))
    assert isinstance(got, int) and got == want == 150
    f3 = pkg.FIRFilter(h)
    b3 = np.empty(10)
    assert pkg.filt_(b3, f3, np.ones(10)) is b3
    # hLen == 1 works (reference throws, documented deviation)
    f4 = pkg.FIRFilter(np.ones(1, dtype=np.float32))
    assert f4.filt(np.array([2., 3.], dtype=np.float32)).tolist() == [2., 3.]
    # reset restores constructor state
    f2.reset()
    assert (f2.state.phiIdx, f2.state.inputDeficit) == (1, 1) and not f2.history.any()


def test_negative_zero_and_nonfinite_follow_reference(pkg, O, torch_cuda):
    """-0.0 products: the Vector seam variant starts from zero (support.jl:46) so the first hLen
    outputs of a call are +0.0 while steady-state outputs are -0.0; Inf/NaN propagate only through the
    window that contains them."""
    h = np.zeros(5, dtype=np.float32)
    x = -np.ones(12, dtype=np.float32)
    for ratio in (Fraction(1, 1), Fraction(1, 2), Fraction(2, 1), Fraction(2, 3)):
        assert_bit_equal(pkg.filt(h, x, ratio), O.filt(h, x, ratio), f"-0.0 ratio {ratio}")
    h = np.arange(1, 8, dtype=np.float64)
    x = np.ones(40)
    x[20] = np.inf
    x[30] = np.nan
    for ratio in (Fraction(1, 1), Fraction(1, 3), Fraction(3, 1), Fraction(3, 4)):
        assert_bit_equal(pkg.filt(h, x, ratio), O.filt(h, x, ratio), f"nonfinite ratio {ratio}")


def test_fused_numerics_close_to_strict(pkg, torch_cuda):
    rng = np.random.default_rng(4)
    h = pkg.firdes(24 * 147, 0.5 / 147, beta=7.8562).astype(np.float32)
    x = rng.random(20000).astype(np.float32)
    ys = pkg.filt(h, x, Fraction(147, 160))
    yf = pkg.filt(h, x, Fraction(147, 160), numerics=pkg.NUMERICS_FUSED)
    assert ys.shape == yf.shape
    assert np.max(np.abs(ys - yf)) <= 24 * np.finfo(np.float32).eps * 0.5


def test_headline_c2_streaming_1e8_vs_oracle(pkg, O, torch_cuda):
    """BASELINE config 2: 147//160, 3528 taps, Float32, 1e8 samples, one channel, streamed in 1e6-sample
    chunks AND in prime-sized chunks (999 983: 1e6 is a multiple of 160, which would hide carry
    bugs); compared with the oracle over the full 91 875 000 outputs."""
    torch = torch_cuda
    n = 100_000_000
    h = pkg.firdes(24 * 147, 0.5 / 147, beta=7.8562).astype(np.float32)
    g = torch.Generator(device="cuda").manual_seed(0x4D520000)
    xd = torch.rand(n, generator=g, device="cuda", dtype=torch.float32)
    x = xd.cpu().numpy()
    yo = O.FIRFilter(h, Fraction(147, 160), tx=np.float32).filt(x)
    assert len(yo) == 91_875_000
    for chunk in (1_000_000, 999_983):
        f = pkg.FIRFilter(h, Fraction(147, 160))
        outs = [f.filt(xd[a:a + chunk]) for a in range(0, n, chunk)]
        y = torch.cat(outs).cpu().numpy()
        assert_bit_equal(y, yo, f"chunk {chunk}")
        st = (f.state.phiIdx, f.state.inputDeficit)
        hist = f.history
        f.close()
        del outs, y
        # the library's own chunk loop (mrhip_filt_device_chunked: the resident signal in ONE launch): same outputs,
        # same end state, same history
        f = pkg.FIRFilter(h, Fraction(147, 160)).bind(np.float32, 1)
        yd = torch.empty(len(yo) + 8, device="cuda", dtype=torch.float32)
        assert f.filt_into_chunked(yd, xd, chunk) == len(yo)
        assert f.last_kernel_name() == "rational_opair_kernel"
        assert_bit_equal(yd[:len(yo)].cpu().numpy(), yo, f"chunked entry, chunk {chunk}")
        assert (f.state.phiIdx, f.state.inputDeficit) == st
        assert_bit_equal(f.history, hist, "history after the chunked entry")
        f.close()
        # config 2 AS STATED -- the chunks arrive one after the other -- through the ring of arriving chunks (mrhip_ring_*: one descriptor
        # per chunk into ONE resident kernel): the same 100 (101) chunks pushed one by one, then by the library's push loop; every one
        # of the 91 875 000 outputs, the end state and the history against the oracle's stream
        for how in ("push", "push_chunks"):
            f = pkg.FIRFilter(h, Fraction(147, 160)).bind(np.float32, 1)
            yd.zero_()
            torch.cuda.synchronize()
            ring = f.open_ring()
            assert ring.info()["resident"]
            if how == "push":
                k = 0
                for a in range(0, n, chunk):
                    cnt, _ = ring.push(yd[k:], xd[a:a + chunk])
                    k += cnt
            else:
                k, _ = ring.push_chunks(yd, xd, chunk)
            assert k == len(yo)
            ring.drain()
            ring.close()
            assert_bit_equal(yd[:len(yo)].cpu().numpy(), yo, f"ring ({how}), chunk {chunk}")
            assert (f.state.phiIdx, f.state.inputDeficit) == st
            assert_bit_equal(f.history, hist, f"history after the ring ({how})")
            # ... and the stream goes on in the filter behind the ring: one more call equals the oracle's next call
            f.close()
        del yd


def test_headline_64ch_properties(pkg, O, torch_cuda):
    """Headline shape (64 channels) at a size the box handles quickly, via size-independent
    properties: (1) whole == chunked (prime chunks) bit-for-bit through a checksum of all outputs;
    (2) channels are independent: channel c of the batch == the same signal run alone;
    (3) spot windows against the oracle restarted mid-stream from (state, history)."""
    torch = torch_cuda
    nch, n = 64, 4_000_000
    h = pkg.firdes(24 * 147, 0.5 / 147, beta=7.8562).astype(np.float32)
    g = torch.Generator(device="cuda").manual_seed(0x4D520001)
    xd = torch.rand((nch, n), generator=g, device="cuda", dtype=torch.float32)
    f = pkg.FIRFilter(h, Fraction(147, 160))
    y_whole = f.filt(xd)
    f2 = pkg.FIRFilter(h, Fraction(147, 160))
    chunk = 999_983
    y_chunk = torch.cat([f2.filt(xd[:, a:a + chunk].contiguous()) for a in range(0, n, chunk)], dim=1)
    assert y_whole.shape == y_chunk.shape == (nch, n * 147 // 160)
    assert torch.equal(y_whole.view(torch.int32), y_chunk.view(torch.int32))
    f1 = pkg.FIRFilter(h, Fraction(147, 160))
    y1 = f1.filt(xd[37].contiguous())
    assert torch.equal(y1.view(torch.int32), y_whole[37].view(torch.int32))
    # spot windows: restart the oracle at input offset a (multiple of 160 => phase 1, deficit 1)
    rng = np.random.default_rng(2)
    for _ in range(6):
        c = int(rng.integers(0, nch))
        a = int(rng.integers(1, n // 160 - 200)) * 160
        seg = xd[c, a - 23:a + 20000].cpu().numpy()
        fo = O.FIRFilter(h, Fraction(147, 160), tx=np.float32)
        fo.set_history(seg[:23])
        yo = fo.filt(seg[23:])
        k0 = a * 147 // 160
        assert_bit_equal(y_whole[c, k0:k0 + len(yo)].cpu().numpy(), yo, f"spot ch {c} offset {a}")


def test_tuned_and_generic_kernels_agree(pkg, torch_cuda, monkeypatch):
    """The phase-stationary kernel and the universal kernel are two schedules of the same arithmetic:
    outputs must be bit-identical (also checks the dispatcher really picks the tuned kernel)."""
    torch = torch_cuda
    rng = np.random.default_rng(21)
    cases = [(147, 160, 3528, np.float32, np.float32), (147, 160, 3500, np.float32, np.complex64),
             (4, 1, 128, np.float32, np.complex64), (3, 17, 50, np.float64, np.float64),
             (7, 5, 100, np.float64, np.float32), (160, 147, 1000, np.float64, np.complex128),
             (1, 3, 31, np.float32, np.float32), (1, 1, 17, np.float32, np.float64), (5, 2, 160, np.float32, np.float32),
             (9, 10, 200, np.float32, np.float32), (31, 32, 31 * 7, np.float32, np.float32), (5, 7, 33, np.float32, np.float32),
             (146, 147, 146 * 32, np.float32, np.float32), (147, 160, 147 * 24, np.float32, np.complex64),
             (9, 10, 200, np.float32, np.complex64), (5, 7, 33, np.float32, np.complex64), (1, 4, 128, np.float32, np.complex64), (1, 1, 300, np.float64, np.float64),
             (1, 7, 129, np.float64, np.complex128), (1, 32, 1, np.float32, np.float32), (1, 5, 64, np.float64, np.float32), (1, 1, 1, np.float32, np.complex64), (1, 3, 1, np.float64, np.float64),
             (1, 24, 100, np.float32, np.complex64), (1, 100, 333, np.float32, np.float32),
             (1, 4, 128, np.float32, np.float32), (1, 1, 64, np.float32, np.complex64), (1, 2, 200, np.float32, np.float32),
             (1, 8, 131, np.float32, np.complex64), (1, 1, 300, np.float32, np.float32), (1, 4, 509, np.float32, np.complex64),
             (5, 1, 160, np.float32, np.float32), (3, 1, 17, np.float32, np.complex64), (7, 1, 50, np.float32, np.float32),
             (2, 1, 64, np.float32, np.complex64), (33, 1, 33 * 32, np.float32, np.float32), (6, 1, 100, np.float64, np.float64),
             (160, 147, 24 * 160, np.float32, np.float32), (3, 2, 72, np.float32, np.complex64), (5, 9, 65, np.float32, np.float32)]
    tuned_seen = set()
    for (L, M, hl, th, tx) in cases:
        h = rng.standard_normal(hl).astype(th)
        x = _rand(rng, (3, 50_000), tx) - 0.5
        xd = torch.from_numpy(x).cuda()
        sizes = [20_000, 1, 29_999]
        monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False)
        f = pkg.FIRFilter(h, Fraction(L, M))
        y_t = torch.cat(_run_chunks(f, xd, sizes), dim=-1).cpu().numpy()
        assert f.last_kernel_name() in ("poly_phase_stationary_kernel", "rational_opair_kernel", "fir_stream_rt_kernel", "fir_stream_kernel"), (L, M, hl, f.last_kernel_name())
        tuned_seen.add(f.last_kernel_name())
        monkeypatch.setenv("MRHIP_FORCE_GENERIC", "1")
        g = pkg.FIRFilter(h, Fraction(L, M))
        y_g = torch.cat(_run_chunks(g, xd, sizes), dim=-1).cpu().numpy()
        assert g.last_kernel_name() == "poly_generic_kernel"
        monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False)
        assert_bit_equal(y_t, y_g, f"tuned vs generic L={L} M={M} hLen={hl} {th} {tx}")
        assert_bit_equal(f.history, g.history, "history")
    assert tuned_seen == {"poly_phase_stationary_kernel", "rational_opair_kernel", "fir_stream_rt_kernel", "fir_stream_kernel"}, tuned_seen


def test_output_pair_kernel_both_directions(pkg, O, torch_cuda, monkeypatch):
    """rational_opair_kernel (two adjacent OUTPUTS per lane, window offsets resolved by exact no-op slots): L > M
    ratios (160//147, 3//2, ... and beyond two: 441//160, 7//3, 511//2), M > L ratios (147//160, 5//9, ...) up to
    M/L < 6 (160//441, 3//17, 2//11: window distance SMIN = 2..5), every tapsPerPhi class (odd, even, 1, 32),
    Float32 / ComplexF32 / Float64 / ComplexF64 samples, Float32 and Float64 taps (incl. the README's Float64 taps x Float32
    samples and Float64 taps x ComplexF32 samples), STRICT and FUSED, multi-channel, chunked with 1-sample and prime pieces; inputs contain -0.0, +-Inf and
    NaN runs (a skipped slot must not turn into 0*Inf or flip the sign of an all-zero sum).  Bit-exact against the
    universal kernel on all channels and against the oracle on one."""
    torch = torch_cuda
    rng = np.random.default_rng(2025)
    cases = [(160, 147, 24 * 160), (160, 147, 24 * 160 - 77), (3, 2, 72), (3, 2, 3 * 32), (3, 2, 3),
             (7, 5, 100), (5, 3, 23), (9, 5, 9 * 17), (16, 9, 16 * 31 - 5), (32, 31, 32 * 8),
             (5, 9, 5 * 13), (4, 7, 4 * 32), (147, 160, 3528), (9, 10, 9 * 7), (31, 32, 31 * 2 - 1),
             (2, 3, 72), (3, 2, 3 * 33), (160, 147, 160 * 48 - 5), (5, 9, 5 * 47), (147, 160, 147 * 40), (7, 5, 7 * 41 - 3),   # 33..48 taps per phase: Float32 arithmetic only
             (441, 160, 441 * 24 - 3), (7, 3, 7 * 24), (5, 2, 5 * 9), (511, 2, 511 * 3), (9, 4, 9 * 32),                      # L >= 2M
             (160, 441, 160 * 24), (3, 17, 72), (2, 5, 2 * 31), (3, 10, 3 * 8), (2, 9, 64), (2, 11, 2 * 17 - 1), (3, 8, 3 * 32),   # M >= 2L: Float32 arithmetic, <= 32 taps per phase
             (147, 160, 147 * 56), (160, 147, 160 * 64 - 5), (2, 1, 2 * 50 - 1), (3, 2, 3 * 61)]                                 # 49..64 taps per phase: Float32 samples
    combos = [(np.float32, np.float32), (np.float32, np.complex64), (np.float64, np.float64), (np.float64, np.float32), (np.float64, np.complex64), (np.float64, np.complex128)]
    for (L, M, hl) in cases:
        for th, tx in combos:
            if th == np.float64 and (-(-hl // L) > 48 or (-(-hl // L) > 32 and tx in (np.complex64, np.complex128))):
                continue                                                # Float64 arithmetic: two columns of <= 48 taps (complex samples: 32)
            if M >= 2 * L and th == np.float64:
                continue                                                # window distances of 2..5 samples: Float32 arithmetic only
            if -(-hl // L) > 48 and th != np.float32:
                continue                                                # 49..64 taps per phase: Float32 arithmetic
            for numerics in (pkg.NUMERICS_STRICT, pkg.NUMERICS_FUSED):
                if numerics == pkg.NUMERICS_FUSED and (hl % 2 or th == np.float64 and L % 2):
                    continue                                            # (thin the matrix)
                nch = int(rng.integers(1, 5))
                h = rng.standard_normal(hl).astype(th)
                h[rng.integers(0, hl, 3)] = 0.0
                x = _rand(rng, (nch, 30_011), tx) - 0.5
                xr = x.view(np.float64 if tx in (np.float64, np.complex128) else np.float32)
                xr[:, 500:560] = -0.0                                   # an all-(-0) window
                xr[0, 2000] = np.inf; xr[0, 2100] = -np.inf; xr[nch - 1, 4000:4003] = np.nan
                xd = torch.from_numpy(x).cuda()
                sizes = [10_007, 1, 13, 19_990]
                f = pkg.FIRFilter(h, Fraction(L, M), numerics=numerics)
                y_t = torch.cat(_run_chunks(f, xd, sizes), dim=-1).cpu().numpy()
                kn = f.last_kernel_name()
                monkeypatch.setenv("MRHIP_FORCE_GENERIC", "1")
                g = pkg.FIRFilter(h, Fraction(L, M), numerics=numerics)
                y_g = torch.cat(_run_chunks(g, xd, sizes), dim=-1).cpu().numpy()
                monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False)
                # STRICT: every shape here is the output-pair kernel's; FUSED (opt-in) is instantiated for M/L < 2 and tapsPerPhi a
                # multiple of 4, the other FUSED shapes run on the phase-stationary / tiled kernels -- same results either way
                tpp = -(-hl // L)
                if numerics == pkg.NUMERICS_STRICT or (M < 2 * L and tpp % 4 == 0):
                    assert kn == "rational_opair_kernel", (L, M, hl, th, tx, numerics, kn)
                else:
                    assert kn in ("poly_phase_stationary_kernel", "poly_tiled_kernel"), (L, M, hl, th, tx, numerics, kn)
                assert_bit_equal(y_t, y_g, f"opair vs generic L={L} M={M} hLen={hl} {th} {tx} numerics={numerics}")
                assert_bit_equal(f.history, g.history, "history")
                assert (f.state.phiIdx, f.state.inputDeficit) == (g.state.phiIdx, g.state.inputDeficit)
                if numerics == pkg.NUMERICS_STRICT:
                    fo = O.FIRFilter(h, Fraction(L, M), tx=tx)
                    yo = np.concatenate([fo.filt(p) for p in np.split(x[nch - 1], np.cumsum(sizes)[:-1])])
                    # NaN payloads/signs are the host FPU's on the oracle side (x86's default NaN has the sign bit set,
                    # the GPU's does not): NaNs must sit in the same places, everything else is compared bit for bit
                    ft = np.float64 if yo.dtype in (np.float64, np.complex128) else np.float32
                    got, want = y_t[nch - 1].view(ft), yo.view(ft)
                    assert np.array_equal(np.isnan(got), np.isnan(want)), f"NaN positions L={L} M={M} hLen={hl} {th} {tx}"
                    ok = ~np.isnan(want)
                    assert_bit_equal(got[ok], want[ok], f"opair vs oracle L={L} M={M} hLen={hl} {th} {tx}")


def test_stream_kernel_standard_and_decimator(pkg, O, torch_cuda, monkeypatch):
    """fir_stream_kernel (FIRStandard / FIRDecimator, Float32 and Float64 arithmetic, loader-wave staging, padded LDS tile,
    scalar taps; M = 1..11, 13, 15: one instantiation per decimation) and fir_stream_rt_kernel (the decimation at run time: the dispatcher's
    choice from M = 16, forced with MRHIP_STREAM_RT=2 below it): M = 1..32, 40, 48, 50, 64, tap counts 2..512 (whole blocks and ragged),
    Float32 and ComplexF32, STRICT and FUSED, multi-channel, chunked with 1-sample pieces and pieces shorter than the history (the
    start-from-zero quirk of the Vector seam variant, support.jl:46, applies to the first hLen outputs of EVERY call); inputs contain
    -0.0, +-Inf, NaN.  Bit-exact against each other, the universal kernel and the oracle."""
    torch = torch_cuda
    rng = np.random.default_rng(77)
    for M in list(range(1, 17)) + [17, 19, 20, 22, 24, 25, 27, 30, 31, 32, 40, 48, 50, 64]:
        for T in (2, 3, 7, 16, 24, 32, 33, 48, 127, 128, 500, 512):
            if M > 16 and T not in (3, 24, 128, 512):
                continue
            if M not in (1, 2, 4, 8) and T in (7, 33, 127, 500):
                continue                                                # (thin the matrix for the later instantiations)
            for th, tx in ((np.float32, np.float32), (np.float32, np.complex64), (np.float64, np.float64), (np.float64, np.float32), (np.float64, np.complex64),
                           (np.float64, np.complex128), (np.float32, np.complex128)):
                if th == np.float64 and (T in (2, 7, 33, 127, 500) or (M not in (1, 2, 3, 4, 8, 16) and T != 48)):
                    continue                                            # (thin the matrix for Float64 arithmetic)
                if tx == np.complex128 and (T not in (48, 128) or (th == np.float32 and M not in (1, 5))):
                    continue
                if M * np.dtype(tx).itemsize > 256:
                    continue                                            # one step of 128 outputs would not fit a 60 KB stage: the tiled kernel's
                for numerics in (pkg.NUMERICS_STRICT, pkg.NUMERICS_FUSED):
                    if numerics == pkg.NUMERICS_FUSED and T not in (48, 128):
                        continue
                    nch = int(rng.integers(1, 6))
                    h = rng.standard_normal(T).astype(th)
                    h[rng.integers(0, T, 2)] = 0.0
                    x = _rand(rng, (nch, 40_009), tx) - 0.5
                    xr = x.view(np.float64 if tx in (np.float64, np.complex128) else np.float32)
                    xr[:, 300:300 + 2 * T] = -0.0                      # all-(-0) windows: the zero-start quirk shows as a sign
                    xr[0, 5000] = np.inf; xr[0, 5100] = -np.inf; xr[nch - 1, 9000:9003] = np.nan
                    xd = torch.from_numpy(x).cuda()
                    sizes = [10_007, 1, 13, T // 2, 19_990, 40_009 - 10_007 - 1 - 13 - T // 2 - 19_990]
                    ys = {}
                    for mode in ("stream", "rt", "generic"):
                        monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False); monkeypatch.delenv("MRHIP_STREAM_RT", raising=False)
                        if mode == "rt":
                            monkeypatch.setenv("MRHIP_STREAM_RT", "2")
                        if mode == "generic":
                            monkeypatch.setenv("MRHIP_FORCE_GENERIC", "1")
                        f = pkg.FIRFilter(h, Fraction(1, M), numerics=numerics)
                        y = torch.cat(_run_chunks(f, xd, sizes), dim=-1).cpu().numpy()
                        ys[mode] = (y, f.history.copy(), f.last_kernel_name(), (f.state.phiIdx, f.state.inputDeficit))
                    monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False); monkeypatch.delenv("MRHIP_STREAM_RT", raising=False)
                    assert ys["stream"][2] == ("fir_stream_kernel" if M <= 11 or M in (13, 15) else "fir_stream_rt_kernel") and ys["rt"][2] == "fir_stream_rt_kernel" and ys["generic"][2] == "poly_generic_kernel", (M, T, th, tx, ys["stream"][2], ys["rt"][2])
                    for other in ("rt", "generic"):
                        assert_bit_equal(ys["stream"][0], ys[other][0], f"stream vs {other} M={M} T={T} {th} {tx} numerics={numerics}")
                        assert_bit_equal(ys["stream"][1], ys[other][1], "history")
                        assert ys["stream"][3] == ys[other][3]
                    if numerics == pkg.NUMERICS_STRICT:
                        fo = O.FIRFilter(h, Fraction(1, M), tx=tx)
                        yo = np.concatenate([fo.filt(p) for p in np.split(x[nch - 1], np.cumsum(sizes)[:-1])])
                        ft = np.float64 if yo.dtype in (np.float64, np.complex128) else np.float32
                        got, want = ys["stream"][0][nch - 1].view(ft), yo.view(ft)
                        assert np.array_equal(np.isnan(got), np.isnan(want))
                        ok = ~np.isnan(want)
                        assert_bit_equal(got[ok], want[ok], f"stream vs oracle M={M} T={T} {tx}")


def test_stream_kernel_long_filters(pkg, O, torch_cuda, monkeypatch):
    """fir_stream_kernel beyond the direct kernel's 512 taps (scalar tap loads have no register budget): 1024 to 3001 taps,
    single-rate and decimating, Float32 / ComplexF32 / Float64; bit-exact against the universal kernel and the oracle."""
    torch = torch_cuda
    rng = np.random.default_rng(99)
    for (M, T, th, tx) in [(1, 1024, np.float32, np.float32), (4, 2049, np.float32, np.complex64), (3, 3001, np.float32, np.float32),
                           (1, 1500, np.float64, np.float64), (8, 1024, np.float64, np.float32), (5, 777, np.float32, np.complex64)]:
        nch = int(rng.integers(1, 4))
        h = rng.standard_normal(T).astype(th)
        x = _rand(rng, (nch, 30_011), tx) - 0.5
        xd = torch.from_numpy(x).cuda()
        sizes = [9_001, 1, T // 3, 30_011 - 9_001 - 1 - T // 3]
        monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False)
        f = pkg.FIRFilter(h, Fraction(1, M))
        y = torch.cat(_run_chunks(f, xd, sizes), dim=-1).cpu().numpy()
        assert f.last_kernel_name() == "fir_stream_kernel", (M, T, f.last_kernel_name())
        monkeypatch.setenv("MRHIP_STREAM_RT", "2")
        r = pkg.FIRFilter(h, Fraction(1, M))
        yr = torch.cat(_run_chunks(r, xd, sizes), dim=-1).cpu().numpy()
        assert r.last_kernel_name() == "fir_stream_rt_kernel", (M, T, r.last_kernel_name())
        monkeypatch.delenv("MRHIP_STREAM_RT", raising=False)
        assert_bit_equal(y, yr, f"per-M vs run-time-M kernel M={M} T={T} {th} {tx}")
        monkeypatch.setenv("MRHIP_FORCE_GENERIC", "1")
        g = pkg.FIRFilter(h, Fraction(1, M))
        yg = torch.cat(_run_chunks(g, xd, sizes), dim=-1).cpu().numpy()
        monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False)
        assert_bit_equal(y, yg, f"stream vs generic M={M} T={T} {th} {tx}")
        assert_bit_equal(f.history, g.history, "history")
        fo = O.FIRFilter(h, Fraction(1, M), tx=tx)
        yo = np.concatenate([fo.filt(p) for p in np.split(x[nch - 1], np.cumsum(sizes)[:-1])])
        assert_bit_equal(y[nch - 1], yo, f"stream vs oracle M={M} T={T} {th} {tx}")


def test_arbitrary_tuned_and_generic_agree(pkg, torch_cuda, monkeypatch):
    torch = torch_cuda
    rng = np.random.default_rng(31)
    for (Nphi, T, rate, th, tx) in [(32, 32, math.pi / 3, np.float64, np.float64), (32, 10, 0.37, np.float32, np.complex64),
                                    (8, 5, 2.7, np.float32, np.float32), (48, 17, 0.9991, np.float64, np.complex128),
                                    (32, 24, 1.0, np.float64, np.float32)]:
        h = (pkg.firdes(T * Nphi, 0.45 / Nphi, beta=7.0) * Nphi).astype(th)
        x = _rand(rng, (3, 40_000), tx) - 0.5
        xd = torch.from_numpy(x).cuda()
        sizes = [15_000, 2, 24_998]
        monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False)
        f = pkg.FIRFilter(h, float(rate), Nphi)
        y_t = torch.cat(_run_chunks(f, xd, sizes), dim=-1).cpu().numpy()
        pipe = np.dtype(tx).itemsize <= 8                          # Float32 and 8-byte samples; ComplexF64: arb_tiled_kernel
        assert f.last_kernel_name() == ("arb_pipe_kernel" if pipe else "arb_tiled_kernel")
        monkeypatch.setenv("MRHIP_FORCE_GENERIC", "1")
        g = pkg.FIRFilter(h, float(rate), Nphi)
        y_g = torch.cat(_run_chunks(g, xd, sizes), dim=-1).cpu().numpy()
        assert g.last_kernel_name() == "arb_generic_kernel"
        monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False)
        assert_bit_equal(y_t, y_g, f"arb tuned vs generic Nphi={Nphi} T={T} rate={rate} {th} {tx}")


def test_farrow_sweep_bit_exact_with_shared_polynomials(pkg, O, torch_cuda):
    """FIRFarrow (src/Filters.jl:123-147, 764-846): the polynomial bank is fitted once (the reference pins no
    bits of A \\ y) and handed to both sides; everything downstream -- Float64 Horner per tap, rounding to the
    tap type, the Vector unsafedot incl. the seam start-from-zero, the phase recurrence -- must then agree
    bit for bit, for every dtype, chunking and channel count."""
    torch = torch_cuda
    rng = np.random.default_rng(99)
    for trial in range(12):
        Nphi = int(rng.choice([8, 32, 12]))
        T = int(rng.integers(2, 36))
        order = int(rng.integers(0, 7))
        th = rng.choice([np.float32, np.float64])
        tx = rng.choice([np.float32, np.float64, np.complex64, np.complex128])
        h = (pkg.firdes(T * Nphi - int(rng.integers(0, Nphi)), 0.45 / Nphi, beta=7.0) * Nphi).astype(th)
        rate = float(rng.choice([math.pi / 3, 0.1234, 1.0, 2.5, 31.7, 1 / 2.123456789]))
        nch = int(rng.integers(1, 4))
        n = 600
        x = _rand(rng, (nch, n), tx)
        sizes = [n] if trial % 3 == 0 else ([1] * 40 + [n - 40] if trial % 3 == 1 else [13] * (n // 13) + [n % 13])
        pn = O.pfb2pnfb(O.taps2pfb(h, Nphi), order)
        f = pkg.FIRFilter(h, rate, Nphi, order, pnfb=pn)
        outs = [o.cpu().numpy() for o in _run_chunks(f, torch.from_numpy(x).cuda(), sizes)]
        y = np.concatenate(outs, axis=1)
        assert f.last_kernel_name() in ("farrow_kernel", "farrow_tiled_kernel", "farrow_pipe_kernel", "farrow_wave_kernel") and f.kernel_name == "FIRFarrow"
        assert np.array_equal(f.pnfb(), pn)
        for c in range(nch):
            fo = O.FIRFilter(h, rate, Nphi, tx=tx, polyorder=order, pnfb=pn)
            yo = np.concatenate(_run_chunks(fo, x[c], sizes))
            assert_bit_equal(y[c], yo, f"farrow rate={rate} Nphi={Nphi} T={T} order={order} {th} {tx}")
        assert f.state.phiAccumulator == fo.state.phiAccumulator
        assert f.state.inputDeficit == fo.state.inputDeficit
        # tapsforphase (Filters.jl:764-775) == the oracle's polyval, rounded to the tap type
        ph = float(rng.uniform(0, Nphi + 1))
        assert np.array_equal(f.tapsforphase(ph), np.array([O.polyval(r, ph) for r in pn]).astype(th))
        with pytest.raises(pkg.MultirateHIPError):
            f.tapsforphase(Nphi + 1.5)
        f.close()


def test_farrow_own_fit_and_setphase(pkg, O, torch_cuda):
    """The library's own least-squares fit (Householder QR, Float64) against numpy's: coefficients agree to
    rounding and the filtered output to ~1e-12 relative; setphase semantics (Filters.jl:210-235)."""
    rng = np.random.default_rng(4)
    h = pkg.firdes(32 * 32, 0.45 / 32, beta=7.8562) * 32
    x = rng.random(5000)
    f = pkg.FIRFilter(h, float(math.pi / 3), 32, 4)
    y = f.filt(x)
    pn = O.pfb2pnfb(O.taps2pfb(h, 32), 4)
    A = np.vander(np.arange(1.0, 33.0), 5, increasing=True)
    assert np.abs(f.pnfb() @ A.T - pn @ A.T).max() <= 1e-9 * np.abs(h).max()     # same fitted filter bank
    yo = O.FIRFilter(h, float(math.pi / 3), 32, tx=np.float64, polyorder=4, pnfb=pn).filt(x)
    assert y.shape == yo.shape and np.abs(y - yo).max() <= 1e-10 * np.abs(yo).max()
    # stateless form filt(h, x, rate, Nphi, polyorder), Filters.jl:870-873
    assert np.array_equal(pkg.filt(h, x, float(math.pi / 3), 32, 4), y)
    # setphase: FIRFarrow 𝜙Idx = 𝜙*(N𝜙-1)+1 (:226); FIRArbitrary (α, 𝜙Idx) = modf(𝜙*N𝜙) (:219)
    assert f.setphase(0.5) == 0.5 * 31 + 1 and f.state.phiAccumulator == 16.5
    fa = pkg.FIRFilter(h, 0.9, 32).bind(np.float64)
    idx, alpha = fa.setphase(0.3)
    assert (idx, alpha) == math.modf(0.3 * 32)[::-1] and fa.state.phiIdx == 9
    fr = pkg.FIRFilter(h, Fraction(32, 5)).bind(np.float64)
    assert fr.setphase(0.5) == 17 and fr.state.phiIdx == 17
    with pytest.raises(pkg.MultirateHIPError):
        fr.setphase(1.5)


def test_farrow_golden_vectors_on_gpu(pkg, torch_cuda):
    """GPU vs the committed FIRFarrow fixtures (no oracle involved at run time)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_farrow_v1.npz"))
    for k in g["names"]:
        Nphi, order = (int(v) for v in g[k + "_par"])
        x = g[k + "_x"]
        f = pkg.FIRFilter(g[k + "_h"], float(g[k + "_rate"]), Nphi, order, pnfb=g[k + "_pnfb"])
        outs, pos = [], 0
        for s in g[k + "_sizes"]:
            outs.append(f.filt(x[pos:pos + int(s)]))
            pos += int(s)
        assert_bit_equal(np.concatenate(outs), g[k + "_y"], k)
        assert [len(o) for o in outs] == g[k + "_counts"].tolist()
        st = f.state
        assert st.inputDeficit == int(g[k + "_state"][0]) and st.phiAccumulator == float(g[k + "_acc"])
        assert_bit_equal(f.history, g[k + "_hist"], k + " history")
        f.close()


def test_config4_farrow_vs_naive_tolerance(pkg, torch_cuda):
    """BASELINE config 4 names "FIRArbitrary (Farrow)": the Farrow form on the same shape (pi/3, 32 taps x 32
    filters, Float64, 64 channels).  A degree-4 polynomial per tap row approximates the filter bank, so the
    stated tolerance vs NaiveResamplers' algorithm is the fit error, 5e-3 of the peak output, far above the
    FIRArbitrary bound (|h[end]|*max|x|); counts are identical to FIRArbitrary's."""
    from oracle import naive as N
    torch = torch_cuda
    h = pkg.firdes(1024, 0.45 / 32, beta=7.8562) * 32
    rng = np.random.default_rng(11)
    x = rng.random((64, 3000))
    f = pkg.FIRFilter(h, float(math.pi / 3), 32, 4)
    y = f.filt(torch.from_numpy(x).cuda()).cpu().numpy()
    ya = pkg.FIRFilter(h, float(math.pi / 3), 32).filt(torch.from_numpy(x).cuda()).cpu().numpy()
    assert y.shape == ya.shape
    for c in (0, 31, 63):
        ref = N.naive_arbitrary(h, x[c], float(math.pi / 3), 32)
        n = min(len(ref), y.shape[1])
        assert abs(len(ref) - y.shape[1]) <= 1
        assert np.abs(y[c, :n] - ref[:n]).max() <= 5e-3 * np.abs(ref).max()


def test_dynamic_scheduling_paths_match_generic_at_scale(pkg, torch_cuda, monkeypatch):
    """Launches large enough that the loader waves draw their work from the group counters (more than three
    grabs per workgroup; smaller launches are dealt statically): complex pair kernel and the interpolator kernel,
    chunked so that the counters are re-armed between launches, against the universal kernel bit for bit."""
    torch = torch_cuda
    g = torch.Generator(device="cuda").manual_seed(77)
    nch, n = 32, 300_000
    xr = torch.rand((nch, n, 2), generator=g, device="cuda", dtype=torch.float32) - 0.5
    xc = torch.view_as_complex(xr)
    rng = np.random.default_rng(3)
    for (L, M, hl, kname) in [(147, 160, 147 * 24, "rational_opair_kernel"), (4, 1, 128, "rational_opair_kernel"),
                              (13, 16, 13 * 9, "rational_opair_kernel"), (3, 1, 3 * 20, "rational_opair_kernel")]:
        h = rng.standard_normal(hl).astype(np.float32)
        sizes = [120_001, 7, 179_992]
        monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False)
        f = pkg.FIRFilter(h, Fraction(L, M))
        y_t = torch.cat(_run_chunks(f, xc, sizes), dim=-1)
        assert f.last_kernel_name() == kname
        monkeypatch.setenv("MRHIP_FORCE_GENERIC", "1")
        gf = pkg.FIRFilter(h, Fraction(L, M))
        y_g = torch.cat(_run_chunks(gf, xc, sizes), dim=-1)
        monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False)
        assert y_t.shape == y_g.shape
        assert torch.equal(torch.view_as_real(y_t).view(torch.int32), torch.view_as_real(y_g).view(torch.int32)), (L, M)
        assert_bit_equal(f.history, gf.history, "history")


def test_fused_numerics_bit_exact_vs_fused_oracle(pkg, O, torch_cuda):
    """NUMERICS_FUSED (opt-in): same order, one fma per tap.  Against the oracle in its fused mode the results are
    bit-identical for every kernel family (the tuned kernels use explicit fma in the same places)."""
    torch = torch_cuda
    rng = np.random.default_rng(123)
    cases = [(Fraction(147, 160), 147 * 24, np.float32, np.float32), (Fraction(147, 160), 147 * 24, np.float32, np.complex64),
             (Fraction(4, 1), 128, np.float32, np.complex64), (Fraction(1, 4), 128, np.float32, np.complex64),
             (Fraction(1, 1), 100, np.float64, np.float64), (Fraction(7, 5), 90, np.float64, np.float32),
             (float(math.pi / 3), 32 * 32, np.float64, np.float64), (0.37, 32 * 10, np.float32, np.complex64)]
    O.set_fused(True)
    try:
        for (ratio, hl, th, tx) in cases:
            h = (rng.standard_normal(hl) if not isinstance(ratio, float) else pkg.firdes(hl, 0.45 / 32, beta=7.0) * 32).astype(th)
            x = _rand(rng, (2, 30_000), tx) - 0.5
            f = pkg.FIRFilter(h, ratio, 32, numerics=pkg.NUMERICS_FUSED)
            sizes = [10_007, 3, 19_990]
            y = torch.cat(_run_chunks(f, torch.from_numpy(x).cuda(), sizes), dim=-1).cpu().numpy()
            for c in range(2):
                fo = O.FIRFilter(h, ratio, 32, tx=tx)
                yo = np.concatenate(_run_chunks(fo, x[c], sizes))
                assert_bit_equal(y[c], yo, f"fused {ratio} {th} {tx} kernel={f.last_kernel_name()}")
    finally:
        O.set_fused(False)


def test_config5_4096_channels_complex_one_gpu(pkg, O, torch_cuda, monkeypatch):
    """BASELINE config 5's full channel count on ONE GPU (4096 x ComplexF32 147//160; per-channel length reduced):
    channel offsets pass 4 GiB, so every 32-bit offset in the kernels is exercised.  Tuned == universal kernel bit
    for bit on all channels (checksum) and element-wise on a few; channel independence (a channel of the batch ==
    the same signal alone)."""
    torch = torch_cuda
    nch, n = 4096, 160_000
    h = pkg.firdes(24 * 147, 0.5 / 147, beta=7.8562).astype(np.float32)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.view_as_complex(torch.rand((nch, n, 2), generator=g, device="cuda", dtype=torch.float32) - 0.5)
    assert x.numel() * 8 > 2 ** 32
    f = pkg.FIRFilter(h, Fraction(147, 160))
    y = f.filt(x)
    assert f.last_kernel_name() == "rational_opair_kernel" and y.shape == (nch, n * 147 // 160)
    monkeypatch.setenv("MRHIP_FORCE_GENERIC", "1")
    gf = pkg.FIRFilter(h, Fraction(147, 160))
    yg = gf.filt(x)
    monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False)
    assert gf.last_kernel_name() == "poly_generic_kernel"
    a, b = torch.view_as_real(y).view(torch.int32), torch.view_as_real(yg).view(torch.int32)
    assert torch.equal(a, b)
    for c in (0, 2047, 4095):
        y1 = pkg.FIRFilter(h, Fraction(147, 160)).filt(x[c].contiguous())
        assert torch.equal(torch.view_as_real(y1).view(torch.int32), a[c])
    assert_bit_equal(f.history[4095], gf.history[4095], "history")
    # ... and the ORACLE itself on the full length of a few channels of the 4096-channel launch (first, last, one past
    # the 4 GiB offset, one in the middle)
    first_past_4g = (2 ** 32) // (n * 8) + 1
    for c in (0, first_past_4g, 2048, 4095):
        yo = O.FIRFilter(h, Fraction(147, 160), tx=np.complex64).filt(x[c].cpu().numpy())
        assert_bit_equal(y[c].cpu().numpy(), yo, f"C5 channel {c} vs oracle")


def test_farrow_errors_and_edge_cases(pkg, O, torch_cuda):
    """Argument checks of the FIRFarrow entry points (reference: error("rate must be greater than 0"), Filters.jl:193;
    tapsforphase range, :765) and the edge cases the other kernels are tested for: empty and short inputs,
    one-sample pieces, polyorder 0, a bank with fewer columns than coefficients."""
    h = (pkg.firdes(8 * 12, 0.45 / 8, beta=6.0) * 8).astype(np.float32)
    with pytest.raises(pkg.MultirateHIPError):
        pkg.FIRFilter(h, -1.0, 8, 3)
    with pytest.raises(pkg.MultirateHIPError):
        pkg.FIRFilter(h, 1.1, 8, 8).bind(np.float32)           # polyorder + 1 > Nphi: rank deficient fit
    with pytest.raises(pkg.MultirateHIPError):
        pkg.FIRFilter(h, 1.1, 8, -1).bind(np.float32)
    fa = pkg.FIRFilter(h, 1.1, 8).bind(np.float32)
    with pytest.raises(pkg.MultirateHIPError):
        fa.pnfb()                                               # not a Farrow filter
    with pytest.raises(pkg.MultirateHIPError):
        pkg.FIRFilter(h, Fraction(2, 3)).bind(np.float32).tapsforphase(1.0)   # neither Arbitrary nor Farrow
    rng = np.random.default_rng(8)
    x = rng.random(300).astype(np.float32)
    for order in (0, 3):
        pn = O.pfb2pnfb(O.taps2pfb(h, 8), order)
        f = pkg.FIRFilter(h, 0.731, 8, order, pnfb=pn)
        fo = O.FIRFilter(h, 0.731, 8, tx=np.float32, polyorder=order, pnfb=pn)
        assert f.filt(x[:0]).shape == (0,)                     # empty input: nothing happens
        outs, refs = [], []
        for a, b in ((0, 1), (1, 2), (2, 3), (3, 3), (3, 40), (40, 41), (41, 300)):   # one-sample pieces, an empty one
            outs.append(f.filt(x[a:b])); refs.append(fo.filt(x[a:b]))
            assert outs[-1].shape == refs[-1].shape
        assert_bit_equal(np.concatenate(outs), np.concatenate(refs), f"farrow pieces order {order}")
        assert f.state.inputDeficit == fo.state.inputDeficit and f.state.phiAccumulator == fo.state.phiAccumulator
        f.reset(); fo.reset()
        assert_bit_equal(f.filt(x), fo.filt(x), "after reset")
    # buffer too small: error before any work, state unchanged (the reference has no check)
    f = pkg.FIRFilter(h, 2.5, 8, 2).bind(np.float32)
    st0 = f.state.phiAccumulator
    with pytest.raises(pkg.MultirateHIPError) as ei:
        f.filt_into(np.empty(10, dtype=np.float32), x)
    assert ei.value.code == 2 and f.state.phiAccumulator == st0


def test_arbitrary_tapsforphase(pkg, torch_cuda):
    """tapsforphase(kernel::FIRArbitrary, phase), src/Filters.jl:677-690: (alpha, phiIdx) = modf(phase);
    pfb[:, phiIdx] + alpha * dpfb[:, phiIdx] in Float64 (alpha is a Float64), stored in the tap type."""
    rng = np.random.default_rng(3)
    for th in (np.float32, np.float64):
        Nphi = 8
        h = rng.standard_normal(5 * Nphi - 3).astype(th)
        f = pkg.FIRFilter(h, 1.37, Nphi).bind(np.float32)
        pfb, dpfb = f.taps(0), f.taps(1)
        assert pfb.shape == (5, Nphi)
        for phase in (1.0, 1.25, 3.999, float(Nphi), Nphi + 0.5, 2.0 + 2.0 ** -30):
            a, i = math.modf(phase)
            want = (pfb[:, int(i) - 1].astype(np.float64) + a * dpfb[:, int(i) - 1].astype(np.float64)).astype(th)
            got = f.tapsforphase(phase)
            assert got.dtype == th
            assert_bit_equal(got, want, f"{th} phase {phase}")
        for bad in (-0.1, Nphi + 1.01, 0.5, Nphi + 1.0):      # outside [0, Nphi+1] (:678) / column 0 or Nphi+1 (BoundsError)
            with pytest.raises(pkg.MultirateHIPError):
                f.tapsforphase(bad)
        f.close()


def test_chunked_streaming_entry_matches_caller_loop(pkg, torch_cuda):
    """mrhip_filt_device_chunked == the caller's own loop of filt! calls, bit for bit (rational and arbitrary)."""
    torch = torch_cuda
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.rand((3, 250_000), generator=g, device="cuda", dtype=torch.float32)
    h = pkg.firdes(24 * 147, 0.5 / 147, beta=7.8562).astype(np.float32)
    for ratio, nphi in ((Fraction(147, 160), 32), (0.9173, 32), (Fraction(1, 4), 32), (Fraction(3, 1), 32), (Fraction(160, 147), 32), (Fraction(1, 1), 32)):
        hh = h if not isinstance(ratio, float) else (pkg.firdes(32 * 16, 0.45 / 32, beta=7.0) * 32).astype(np.float32)
        if ratio in (Fraction(3, 1), Fraction(1, 1)):
            hh = h[:96]
        f1 = pkg.FIRFilter(hh, ratio, nphi)
        y1 = torch.cat([f1.filt(x[:, a:a + 9973]) for a in range(0, x.shape[1], 9973)], dim=1)
        f2 = pkg.FIRFilter(hh, ratio, nphi).bind(np.float32, 3)
        y2 = torch.empty((3, y1.shape[1] + 8), device="cuda", dtype=torch.float32)
        n = f2.filt_into_chunked(y2, x, 9973)
        assert n == y1.shape[1] and torch.equal(y1.view(torch.int32), y2[:, :n].view(torch.int32))
        assert (f1.state.phiIdx, f1.state.inputDeficit, f1.state.phiAccumulator) == (f2.state.phiIdx, f2.state.inputDeficit, f2.state.phiAccumulator)
        assert_bit_equal(f1.history, f2.history, f"history {ratio}")
        # an undersized buffer fails at the piece the caller's own loop would fail at, with the earlier pieces applied
        f3 = pkg.FIRFilter(hh, ratio, nphi).bind(np.float32, 3)
        small = torch.empty((3, n // 2), device="cuda", dtype=torch.float32)
        with pytest.raises(pkg.MultirateHIPError) as ei:
            f3.filt_into_chunked(small, x, 9973)
        assert ei.value.code == 2


def test_poly_tiled_kernel_long_filters(pkg, O, torch_cuda, monkeypatch):
    """Filters the register-resident kernels do not take (tapsPerPhi > 64, > 48 for Float64 arithmetic, > 32 for complex samples with Float64 arithmetic, L > 4096 phases (512 < L <= 4096: the output-pair kernel in period blocks), a decimation whose step does not fit the streaming kernels' LDS stage) run on poly_tiled_kernel: bit-identical to the one-thread-per-output kernel and to the oracle,
    across chunk seams, for every dtype combination, 1..35 channels (all channels-per-lane variants + ragged group)."""
    torch = torch_cuda
    rng = np.random.default_rng(77)
    cases = [(2, 3, 140, np.float32, np.float32, 35), (3, 2, 200, np.float32, np.complex64, 9), (147, 160, 147 * 70, np.float32, np.float32, 33),
             (2, 3, 100, np.float64, np.float64, 4),
             (7, 1, 7 * 50, np.float64, np.float64, 3), (4099, 4000, 4099 * 3, np.float32, np.float32, 8), (1, 250, 700, np.float32, np.float32, 5),
             (1, 33, 600, np.float64, np.complex128, 2), (5, 64, 5 * 40, np.float64, np.float32, 32), (4, 7, 4 * 49, np.float32, np.float64, 1)]
    for (L, M, hl, th, tx, nch) in cases:
        h = rng.standard_normal(hl).astype(th)
        x = _rand(rng, (nch, 30_000), tx) - 0.5
        xd = torch.from_numpy(x).cuda()
        sizes = [12_000, 1, 17, 17_982]
        monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False)
        f = pkg.FIRFilter(h, Fraction(L, M))
        y_t = torch.cat(_run_chunks(f, xd, sizes), dim=-1).cpu().numpy()
        assert f.last_kernel_name() == "poly_tiled_kernel", (L, M, hl, f.last_kernel_name())
        monkeypatch.setenv("MRHIP_FORCE_GENERIC", "1")
        g = pkg.FIRFilter(h, Fraction(L, M))
        y_g = torch.cat(_run_chunks(g, xd, sizes), dim=-1).cpu().numpy()
        assert g.last_kernel_name() == "poly_generic_kernel"
        monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False)
        assert_bit_equal(y_t, y_g, f"tiled vs generic L={L} M={M} hLen={hl} {th} {tx}")
        assert_bit_equal(f.history, g.history, "history")
        fo = O.FIRFilter(h, Fraction(L, M), tx=tx)
        yo = np.concatenate([fo.filt(p) for p in np.split(x[nch - 1], np.cumsum(sizes)[:-1])])
        assert_bit_equal(y_t[nch - 1], yo, f"tiled vs oracle L={L} M={M} hLen={hl}")


def test_hip_graph_capture_of_fixed_chunk_streaming(pkg, torch_cuda):
    """filt! on a caller stream only enqueues kernels, so a fixed-chunk streaming loop can be captured
    in a HIP graph -- here through the plain entry (mrhip_filt_device on a capturing stream), an ODD number of calls per
    graph: a captured call is planned on the device from the device-resident stream state and writes its history back
    into the slot it read, so every replay continues the stream exactly like the plain loop would.  (Chunk sizes that
    advance the state, FIRArbitrary and FIRFarrow: tests/test_gpu_device_state.py.)  A filter whose schedule buffers do
    not exist yet cannot allocate them inside a capture: refused, capture intact."""
    torch = torch_cuda
    L, M, chunk, ncalls, nch = 147, 160, 16_000, 3, 3
    h = pkg.firdes(24 * L, 0.5 / L, beta=7.8562).astype(np.float32)
    x = torch.rand((nch, chunk * ncalls), dtype=torch.float32, device="cuda") - 0.5
    nout = chunk * L // M
    f = pkg.FIRFilter(h, Fraction(L, M)).bind(np.float32, nch)
    y_g = torch.zeros((nch, nout * ncalls), dtype=torch.float32, device="cuda")

    def loop(flt, y):
        for i in range(ncalls):
            assert flt.filt_into(y[:, i * nout:(i + 1) * nout], x[:, i * chunk:(i + 1) * chunk]) == nout

    harb = (pkg.firdes(32 * 8, 0.45 / 32, beta=7.0) * 32).astype(np.float32)
    arb = pkg.FIRFilter(harb, 0.77, 32).bind(np.float32, nch)
    y_bad = torch.zeros((nch, nout + 64), dtype=torch.float32, device="cuda")
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=s):
        loop(f, y_g)
        with pytest.raises(pkg.MultirateHIPError) as ei:                 # cold FIRArbitrary: refused, capture intact
            arb.filt_into(y_bad, x[:, :chunk])
        assert ei.value.code == 5
    ref = pkg.FIRFilter(h, Fraction(L, M)).bind(np.float32, nch)
    y_ref = torch.empty_like(y_g)
    for replay in range(4):                      # pass 0 starts from the zero history, the others continue the stream
        g.replay()
        loop(ref, y_ref)
        torch.cuda.synchronize()
        assert torch.equal(y_g.view(torch.int32), y_ref.view(torch.int32)), f"replay {replay}"
    assert f.last_kernel_name() == "rational_opair_kernel"
    # the host object follows the replays: its history is the device's, so a plain call continues the stream too
    assert_bit_equal(f.history, ref.history, "history after the replays")
    assert (arb.state.phiAccumulator, arb.state.inputDeficit) == (1.0, 1)


def test_calls_on_different_streams_are_ordered(pkg, O, torch_cuda):
    """Per-filter device state (history ping-pong, counters) is ordered across streams by the library: device calls
    alternating between two torch streams and host-pointer calls (the filter's own stream) in between continue one
    stream bit-exactly; reset() in the middle needs no synchronisation either."""
    torch = torch_cuda
    L, M, nch = 147, 160, 4
    h = pkg.firdes(24 * L, 0.5 / L, beta=7.8562).astype(np.float32)
    rng = np.random.default_rng(77)
    x = rng.random((nch, 600_000), dtype=np.float32) - 0.5
    xd = torch.from_numpy(x).cuda()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    sizes = [50_001, 49_999, 1, 100_003, 7, 199_989, 200_000]
    for rep in range(2):
        f = pkg.FIRFilter(h, Fraction(L, M)).bind(np.float32, nch)
        if rep == 1:                      # dirty the history, then reset without any synchronisation
            with torch.cuda.stream(s1):
                f.filt(xd[:, :123_457])
            f.reset()
        outs, pos = [], 0
        for i, n in enumerate(sizes):
            if i % 3 == 2:                # host-pointer call: runs on the filter's own stream
                outs.append(torch.from_numpy(f.filt(x[:, pos:pos + n])))
            else:
                with torch.cuda.stream(s1 if i % 3 == 0 else s2):
                    outs.append(f.filt(xd[:, pos:pos + n]))
            pos += n
        torch.cuda.synchronize()
        y = torch.cat([o.cpu() for o in outs], dim=1).numpy()
        for c in (0, nch - 1):
            fo = O.FIRFilter(h, Fraction(L, M), tx=np.float32)
            yo = np.concatenate([fo.filt(p) for p in np.split(x[c, :pos], np.cumsum(sizes)[:-1])])
            assert_bit_equal(y[c], yo, f"rep {rep} channel {c}")
        f.close()


def test_host_path_pipelined_pieces(pkg, O, torch_cuda, monkeypatch):
    """mrhip_filt_host cuts long host signals into pieces (double-buffered H2D / kernel / D2H on three streams).  With a
    tiny piece size the pipeline runs through many slots: outputs, counts, end state and history equal the oracle's for
    the rational family, FIRArbitrary and row-strided multi-channel buffers."""
    monkeypatch.setenv("MRHIP_HOST_PIECE_KB", "64")
    rng = np.random.default_rng(31)
    h = pkg.firdes(24 * 147, 0.5 / 147, beta=7.8562).astype(np.float32)
    harb = (pkg.firdes(32 * 8, 0.45 / 32, beta=7.0) * 32)
    cases = [(h, Fraction(147, 160), np.float32, 3, 200_003), (h.astype(np.float64), Fraction(160, 147), np.float32, 1, 150_001),
             (h[:128], Fraction(1, 4), np.complex64, 2, 120_000), (h[:96], Fraction(4, 1), np.float32, 2, 90_001),
             (harb, 1.234567, np.float64, 2, 100_000), (h[:64], Fraction(1, 1), np.float32, 1, 70_000)]
    for hh, ratio, tx, nch, n in cases:
        x = _rand(rng, (nch, n), tx) - 0.5
        f = pkg.FIRFilter(hh, ratio, 32)
        y = np.concatenate([f.filt(x[:, :n // 2]), f.filt(x[:, n // 2:])], axis=1)     # two long host calls
        for c in range(nch):
            fo = O.FIRFilter(hh, ratio, 32, tx=tx) if isinstance(ratio, float) else O.FIRFilter(hh, ratio, tx=tx)
            yo = np.concatenate([fo.filt(x[c, :n // 2]), fo.filt(x[c, n // 2:])])
            assert_bit_equal(y[c], yo, f"{ratio} {tx} channel {c}")
        st, so = f.state, fo.state
        assert (st.phiIdx, st.inputDeficit, st.phiAccumulator) == (so.phiIdx, so.inputDeficit, so.phiAccumulator)
        f.close()


def test_filter_cascade_device_resident(pkg, O, torch_cuda):
    """Decimate 1//4 then resample 147//160 then an arbitrary-rate stage, chunked: equal to the oracle stages chained."""
    torch = torch_cuda
    rng = np.random.default_rng(5)
    h1 = rng.standard_normal(64).astype(np.float32)
    h2 = pkg.firdes(24 * 147, 0.5 / 147, beta=7.8562).astype(np.float32)
    h3 = (pkg.firdes(32 * 12, 0.45 / 32, beta=7.0) * 32).astype(np.float32)
    x = (rng.random((2, 90_000), dtype=np.float32) - 0.5)
    casc = pkg.FilterCascade(pkg.FIRFilter(h1, Fraction(1, 4)), pkg.FIRFilter(h2, Fraction(147, 160)), pkg.FIRFilter(h3, 1.2345, 32))
    xd = torch.from_numpy(x).cuda()
    sizes = [40_001, 3, 49_996]
    y = torch.cat(_run_chunks(casc, xd, sizes), dim=-1).cpu().numpy()
    for c in range(2):
        stages = [O.FIRFilter(h1, Fraction(1, 4), tx=np.float32), O.FIRFilter(h2, Fraction(147, 160), tx=np.float32),
                  O.FIRFilter(h3, 1.2345, 32, tx=np.float32)]
        outs = []
        for p in np.split(x[c], np.cumsum(sizes)[:-1]):
            for st in stages:
                p = st.filt(p)
            outs.append(p)
        assert_bit_equal(y[c], np.concatenate(outs), f"cascade ch {c}")
    # a bound cascade refuses another channel count or sample type (it would otherwise read past x / leave rows of y unwritten)
    with pytest.raises(pkg.MultirateHIPError):
        casc.filt(torch.zeros((3, 1000), dtype=torch.float32, device="cuda"))
    with pytest.raises(pkg.MultirateHIPError):
        casc.filt(torch.zeros((2, 1000), dtype=torch.float64, device="cuda"))
    assert casc.reset().stages[1].state.phiIdx == 1


def test_arbitrary_phase_recurrence_many_rates(pkg, O, torch_cuda):
    """The host evaluates FIRArbitrary's serial Float64 phase recurrence (Filters.jl:663-673) through an algebraically
    shortened dependency chain (host_logic.cpp:ArbStepper); the oracle restates the reference's expressions one by
    one.  Outputs (which depend on every accumulator value) and the end state must agree bit for bit for
    power-of-two and other Nphi, rates from 1/300 (many periods per step: general path) to 40, long runs, chunked."""
    torch = torch_cuda
    rng = np.random.default_rng(123)
    rates = [math.pi / 3, 1.0, 0.5, 2.0, 1 / 3, 0.999999, 1.000001, 7.77, 40.0, 0.26, 0.2499, 0.01, 1 / 300, 0.0333, 3.999]
    for i, rate in enumerate(rates):
        for Nphi in (32, 10, 7, 1, 64):
            T = 3
            h = rng.standard_normal(T * Nphi).astype(np.float32)
            n = 60_000 if rate <= 2.0 else 8_000
            x = rng.standard_normal(n).astype(np.float32)
            sizes = [n // 3, 1, n - n // 3 - 1]
            f = pkg.FIRFilter(h, rate, Nphi)
            y = np.concatenate([o.cpu().numpy() for o in _run_chunks(f, torch.from_numpy(x).cuda(), sizes)])
            fo = O.FIRFilter(h, rate, Nphi, tx=np.float32)
            yo = np.concatenate(_run_chunks(fo, x, sizes))
            assert_bit_equal(y, yo, f"rate={rate} Nphi={Nphi}")
            assert f.state.phiAccumulator == fo.state.phiAccumulator and f.state.inputDeficit == fo.state.inputDeficit, (rate, Nphi)
            f.close()


def test_long_launch_two_stage_tiles_match_generic(pkg, O, torch_cuda, monkeypatch):
    """Long launches (tens of tiles per resident workgroup: dynamic grabs, full-size two-stage tiles,
    plan_rational_opair): Float32 64 ch x 3.3e6 and ComplexF32 96 ch x 1.5e6 in ragged pieces, against the universal
    kernel bit for bit, plus oracle spot checks at the seams."""
    torch = torch_cuda
    g = torch.Generator(device="cuda").manual_seed(5)
    rng = np.random.default_rng(8)
    for (nch, n, cplx, L, M, hl) in ((64, 3_300_000, False, 147, 160, 147 * 24), (96, 1_500_000, True, 147, 160, 147 * 24),
                                     (64, 4_000_000, False, 13, 16, 13 * 9), (64, 4_000_000, False, 31, 32, 31 * 7),
                                     (48, 3_000_000, True, 9, 10, 9 * 20)):
        h = rng.standard_normal(hl).astype(np.float32)
        if cplx:
            x = torch.view_as_complex(torch.rand((nch, n, 2), generator=g, device="cuda", dtype=torch.float32) - 0.5)
        else:
            x = torch.rand((nch, n), generator=g, device="cuda", dtype=torch.float32) - 0.5
        sizes = [n - 200_003, 200_003]
        monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False)
        f = pkg.FIRFilter(h, Fraction(L, M))
        y_t = torch.cat(_run_chunks(f, x, sizes), dim=-1)
        assert f.last_kernel_name() == "rational_opair_kernel"
        monkeypatch.setenv("MRHIP_FORCE_GENERIC", "1")
        gf = pkg.FIRFilter(h, Fraction(L, M))
        y_g = torch.cat(_run_chunks(gf, x, sizes), dim=-1)
        monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False)
        a = torch.view_as_real(y_t) if cplx else y_t
        b = torch.view_as_real(y_g) if cplx else y_g
        assert a.shape == b.shape and torch.equal(a.view(torch.int32), b.view(torch.int32)), (nch, n, cplx)
        assert_bit_equal(f.history, gf.history, "history")
        c = nch - 1
        fo = O.FIRFilter(h, Fraction(L, M), tx=np.complex64 if cplx else np.float32)
        yo = fo.filt(x[c, :50_000].cpu().numpy())
        assert_bit_equal(y_t[c, :len(yo)].cpu().numpy(), yo, "oracle spot check")
        del x, y_t, y_g, a, b
        torch.cuda.empty_cache()


def test_randomised_stress_short(torch_cuda):
    """tests/stress_random.py (random kinds / ratios / tap counts / dtypes / channels / chunkings; every tuned kernel
    against the universal kernel and the oracle, bit for bit) on a fixed seed, as a child process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "stress_random.py"), "--cases", "160", "--seed", "11", "--seconds", "90"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches 0" in r.stdout


def test_arb_pipe_kernel_sweep(pkg, O, torch_cuda, monkeypatch):
    """arb_pipe_kernel (Float32, Float64 and ComplexF32 samples with Float64 or Float32 taps: hand-pipelined LDS reads, -0.0 accumulator start, two sample
    buffers): bit-equal to the oracle on some channels and to arb_generic_kernel on all of them, over odd and even
    tapsPerPhi (the single last tap, T = 1), partial channel groups, rates whose span needs fewer channels per lane or
    falls back to arb_tiled_kernel, seams with history, signed zeros, and the fused numerics."""
    torch = torch_cuda
    rng = np.random.default_rng(4242)
    seen = set()
    cases = []
    for T in (1, 2, 3, 4, 5, 7, 8, 31, 32, 33, 40):
        cases.append((32, T, math.pi / 3, np.float64, int(rng.choice([1, 3, 9, 33])), np.float64))
        cases.append((32, T, math.pi / 3, np.complex64, int(rng.choice([1, 3, 9, 33])), np.float32))   # Float32 arithmetic
        cases.append((32, T, math.pi / 3, np.float32, int(rng.choice([1, 3, 9, 33])), np.float32))     # Float32 samples: pair reads
    for rate in (0.05, 0.11, 0.26, 0.6, 1.0, 1.9, 3.3):          # (rate = outputs per input: a small one stretches a tile's span)
        cases.append((32, 12, rate, np.float64, 34, np.float64))
        cases.append((10, 6, rate, np.complex64, 9, np.float64))
        cases.append((10, 6, rate, np.complex64, 34, np.float32))
        cases.append((10, 6, rate, np.float32, 34, np.float64))
        cases.append((32, 7, rate, np.float32, 9, np.float32))
    cases.append((32, 32, math.pi / 3, np.float64, 64, np.float64))
    cases.append((32, 32, math.pi / 3, np.complex64, 37, np.float64))
    cases.append((32, 32, math.pi / 3, np.complex64, 37, np.float32))
    cases.append((32, 32, math.pi / 3, np.float32, 37, np.float32))
    cases.append((32, 32, math.pi / 3, np.float32, 64, np.float64))
    for (Nphi, T, rate, tx, nch, th) in cases:
        h = (pkg.firdes(T * Nphi, 0.45 / Nphi, beta=7.0) * Nphi).astype(th)
        n = (6000 if nch <= 9 else 2500) * (8 if rate < 0.2 else 1)     # (enough outputs per piece for the span to matter)
        x = _rand(rng, (nch, n), tx) - 0.5
        x[:, 100:140] = 0.0
        x[:, 120:130] *= -1.0                                    # -0.0 (and -0.0 - 0.0j)
        xd = torch.from_numpy(x).cuda()
        sizes = [n // 2 - 1, 1, 2, n - n // 2 - 2]
        for numerics in (pkg.NUMERICS_STRICT, pkg.NUMERICS_FUSED):
            monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False)
            f = pkg.FIRFilter(h, float(rate), Nphi, numerics=numerics)
            y = torch.cat(_run_chunks(f, xd, sizes), dim=-1).cpu().numpy()
            seen.add(f.last_kernel_name())
            if nch >= 33 and numerics == pkg.NUMERICS_STRICT:        # the register-staged form of the same kernel (LDS-DMA staging off)
                monkeypatch.setenv("MRHIP_PIPE_DMA", "0")
                f0 = pkg.FIRFilter(h, float(rate), Nphi, numerics=numerics)
                y0 = torch.cat(_run_chunks(f0, xd, sizes), dim=-1).cpu().numpy()
                monkeypatch.delenv("MRHIP_PIPE_DMA")
                assert f0.last_kernel_name() == f.last_kernel_name()
                assert_bit_equal(y0, y, "DMA staging off vs on")
                f0.close()
            monkeypatch.setenv("MRHIP_FORCE_GENERIC", "1")
            g = pkg.FIRFilter(h, float(rate), Nphi, numerics=numerics)
            yg = torch.cat(_run_chunks(g, xd, sizes), dim=-1).cpu().numpy()
            assert g.last_kernel_name() == "arb_generic_kernel"
            monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False)
            tag = f"Nphi={Nphi} T={T} rate={rate} {np.dtype(tx)} x {np.dtype(th)} taps nch={nch} numerics={numerics} kernel={f.last_kernel_name()}"
            assert_bit_equal(y, yg, "pipe vs generic " + tag)
            assert_bit_equal(f.history, g.history, "history " + tag)
            O.set_fused(numerics == pkg.NUMERICS_FUSED)
            try:
                for c in sorted({0, nch - 1}):
                    fo = O.FIRFilter(h, float(rate), Nphi, tx=tx)
                    yo = np.concatenate(_run_chunks(fo, x[c], sizes))
                    assert_bit_equal(y[c], yo, f"pipe vs oracle ch={c} " + tag)
            finally:
                O.set_fused(False)
            f.close(); g.close()
    assert "arb_pipe_kernel" in seen and "arb_tiled_kernel" in seen, seen


def test_farrow_pipe_kernel_sweep(pkg, O, torch_cuda, monkeypatch):
    """farrow_pipe_kernel (Float32, Float64 and ComplexF32 samples with Float64 or Float32 taps, at most 32 taps): bit-equal to the oracle on some channels
    and to the generic farrow_kernel on all of them -- over tap counts on both sides of the 16-tap instantiation, partial
    channel groups, rates whose span needs fewer channels per lane or falls back, seams, signed zeros, fused numerics."""
    torch = torch_cuda
    rng = np.random.default_rng(777)
    seen = set()
    cases = []
    for T in (1, 2, 3, 5, 8, 15, 16, 17, 31, 32):
        cases.append((32, T, math.pi / 3, np.float64, int(rng.choice([1, 2, 3, 6, 9])), np.float64))
        cases.append((32, T, math.pi / 3, np.complex64, int(rng.choice([1, 2, 3, 6, 9])), np.float32))    # Float32 arithmetic
        cases.append((32, T, math.pi / 3, np.float32, int(rng.choice([1, 2, 3, 6, 9])), np.float32))      # Float32 samples: pair reads
    for rate in (0.05, 0.11, 0.26, 0.6, 1.0, 1.9, 3.3):
        cases.append((32, 12, rate, np.float64, 13, np.float64))
        cases.append((10, 6, rate, np.complex64, 5, np.float64))
        cases.append((10, 6, rate, np.complex64, 13, np.float32))
        cases.append((10, 6, rate, np.float32, 13, np.float64))
        cases.append((32, 7, rate, np.float32, 5, np.float32))
    cases.append((32, 32, math.pi / 3, np.float64, 64, np.float64))
    cases.append((32, 32, math.pi / 3, np.complex64, 37, np.float64))
    cases.append((32, 32, math.pi / 3, np.complex64, 37, np.float32))
    cases.append((32, 32, math.pi / 3, np.float32, 37, np.float32))
    cases.append((32, 32, math.pi / 3, np.float32, 64, np.float64))
    cases.append((32, 40, math.pi / 3, np.float64, 8, np.float64))            # more than 32 taps: farrow_tiled_kernel
    for (Nphi, T, rate, tx, nch, th) in cases:
        h = (pkg.firdes(T * Nphi, 0.45 / Nphi, beta=7.0) * Nphi).astype(th)
        n = (6000 if nch <= 9 else 2500) * (8 if rate < 0.2 else 1)
        x = _rand(rng, (nch, n), tx) - 0.5
        x[:, 100:140] = 0.0
        x[:, 120:130] *= -1.0                                    # -0.0
        xd = torch.from_numpy(x).cuda()
        sizes = [n // 2 - 1, 1, 2, n - n // 2 - 2]
        for numerics in (pkg.NUMERICS_STRICT, pkg.NUMERICS_FUSED):
            monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False)
            pn = O.pfb2pnfb(O.taps2pfb(h, Nphi), 4)
            f = pkg.FIRFilter(h, float(rate), Nphi, 4, pnfb=pn, numerics=numerics)
            y = torch.cat(_run_chunks(f, xd, sizes), dim=-1).cpu().numpy()
            seen.add(f.last_kernel_name())
            if nch >= 13 and numerics == pkg.NUMERICS_STRICT:        # the register-staged form of the same kernel (LDS-DMA staging off)
                monkeypatch.setenv("MRHIP_PIPE_DMA", "0")
                f0 = pkg.FIRFilter(h, float(rate), Nphi, 4, pnfb=pn, numerics=numerics)
                y0 = torch.cat(_run_chunks(f0, xd, sizes), dim=-1).cpu().numpy()
                monkeypatch.delenv("MRHIP_PIPE_DMA")
                assert f0.last_kernel_name() == f.last_kernel_name()
                assert_bit_equal(y0, y, "DMA staging off vs on")
                f0.close()
            monkeypatch.setenv("MRHIP_FORCE_GENERIC", "1")
            g = pkg.FIRFilter(h, float(rate), Nphi, 4, pnfb=pn, numerics=numerics)
            yg = torch.cat(_run_chunks(g, xd, sizes), dim=-1).cpu().numpy()
            assert g.last_kernel_name() == "farrow_kernel"
            monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False)
            tag = f"Nphi={Nphi} T={T} rate={rate} {np.dtype(tx)} x {np.dtype(th)} taps nch={nch} numerics={numerics} kernel={f.last_kernel_name()}"
            assert_bit_equal(y, yg, "pipe vs generic " + tag)
            assert_bit_equal(f.history, g.history, "history " + tag)
            O.set_fused(numerics == pkg.NUMERICS_FUSED)
            try:
                for c in sorted({0, nch - 1}):
                    fo = O.FIRFilter(h, float(rate), Nphi, tx=tx, polyorder=4, pnfb=pn)
                    yo = np.concatenate(_run_chunks(fo, x[c], sizes))
                    assert_bit_equal(y[c], yo, f"pipe vs oracle ch={c} " + tag)
            finally:
                O.set_fused(False)
            f.close(); g.close()
    # (at most 16 taps per phase and 8 channels: farrow_wave_kernel, one lane per output straight from global memory)
    assert {"farrow_pipe_kernel", "farrow_tiled_kernel", "farrow_wave_kernel"} <= seen, seen


def test_advance_state_equals_filtering(pkg, O, torch_cuda):
    """mrhip_advance_state(n): the state a filt! call over n samples leaves, without data -- for every kind, from a fresh and
    from an advanced state; the history is untouched; a stream entered that way (advance + set_history + filt) continues
    bit for bit."""
    torch = torch_cuda
    rng = np.random.default_rng(2024)
    cases = [(Fraction(147, 160), 147 * 8), (Fraction(1, 4), 64), (Fraction(5, 1), 60), (Fraction(1, 1), 33), (Fraction(3, 17), 51),
             (float(math.pi / 3), 32 * 6), (0.2345, 32 * 6), (3.0, 32 * 4)]
    for ratio, hl in cases:
        h = rng.standard_normal(hl).astype(np.float32)
        x = rng.standard_normal(5000).astype(np.float32)
        for farrow in ((False, True) if isinstance(ratio, float) else (False,)):
            mk = (lambda: pkg.FIRFilter(h, ratio, 32, 3)) if farrow else (lambda: pkg.FIRFilter(h, ratio, 32) if isinstance(ratio, float) else pkg.FIRFilter(h, ratio))
            a, b = mk(), mk()
            a.bind(np.float32); b.bind(np.float32)
            if farrow:
                b.close(); b = pkg.FIRFilter(h, ratio, 32, 3, pnfb=a.pnfb()); b.bind(np.float32)
            pos = 0
            for n in (0, 1, 7, 1234, 3000):
                ya = a.filt(x[pos:pos + n])
                hist_before = b.history.copy()
                cnt = b.advance_state(n)
                assert cnt == len(ya), (ratio, farrow, n, cnt, len(ya))
                sa, sb = a.state, b.state
                assert (sa.phiIdx, sa.inputDeficit, sa.phiAccumulator) == (sb.phiIdx, sb.inputDeficit, sb.phiAccumulator), (ratio, farrow, n)
                assert_bit_equal(b.history, hist_before, "advance_state must not touch the history")
                pos += n
            b.set_history(a.history)                       # enter the stream here
            assert_bit_equal(b.filt(x[pos:]), a.filt(x[pos:]), f"entered stream {ratio} farrow={farrow}")
            a.close(); b.close()


def test_pipe_kernels_many_tiles_per_workgroup(pkg, O, torch_cuda, monkeypatch):
    """Calls long enough that every persistent workgroup of arb_pipe_kernel / farrow_pipe_kernel takes several tiles (the short
    sweeps above give each workgroup one): few channels (1, 2, 4 channels per lane), tap counts on both sides of the 32-tap
    instantiations, LDS-DMA staging on and off -- against the tiled kernels, bit for bit.  Regression: round 3's first form
    prefetched a tile's first index with an asynchronous s_load from inline assembly; the compiler re-used that SGPR while the
    load was in flight and whole tiles came out as zeros, for some instantiations only."""
    torch = torch_cuda
    rng = np.random.default_rng(808)
    nphi, n_in = 8, 200_000
    seen = set()
    for farrow in (False, True):
        for rate in (7.7, math.pi / 3):
            for T, tx, th in ((17, np.float64, np.float32), (32, np.float64, np.float64), (25, np.complex64, np.float32), (32, np.float32, np.float32)):
                h = rng.standard_normal(T * nphi).astype(th)
                pn = O.pfb2pnfb(O.taps2pfb(h, nphi), 3) if farrow else None
                for nch in (1, 2, 3, 5):
                    x = _rand(rng, (nch, n_in), tx) - 0.5
                    xd = torch.from_numpy(x).cuda()
                    mk = (lambda: pkg.FIRFilter(h, rate, nphi, 3, pnfb=pn)) if farrow else (lambda: pkg.FIRFilter(h, rate, nphi))
                    outs = {}
                    for name, env in (("pipe", {"MRHIP_FARROW_WAVE": "0"}), ("pipe, register staging", {"MRHIP_PIPE_DMA": "0", "MRHIP_FARROW_WAVE": "0"}),
                                      ("tiled", {"MRHIP_FARROW_PIPE": "0", "MRHIP_ARB_PIPE": "0", "MRHIP_FARROW_WAVE": "0"}), ("default", {}),
                                      ("wave", {"MRHIP_FARROW_WAVE_MAXCH": "64"})):
                        for k, v in env.items():
                            monkeypatch.setenv(k, v)
                        f = mk()
                        y = f.filt(xd).cpu().numpy()          # (on the host before the filter goes away)
                        outs[name] = (y, f.last_kernel_name())
                        f.close()
                        for k in env:
                            monkeypatch.delenv(k)
                    want, kt = outs["tiled"]
                    assert kt in ("arb_tiled_kernel", "farrow_tiled_kernel"), kt
                    assert want.shape[1] > 768 * 256          # more tiles than workgroups
                    for name in ("pipe", "pipe, register staging"):
                        got, kn = outs[name]
                        seen.add(kn)
                        assert kn in ("arb_pipe_kernel", "farrow_pipe_kernel"), kn
                        assert_bit_equal(got, want, f"{name} vs {kt}: farrow={farrow} rate={rate} T={T} {np.dtype(tx)} x {np.dtype(th)} taps nch={nch}")
                    got, kn = outs["default"]                 # what the dispatcher picks (more than 16 taps per phase: the pipe kernels)
                    assert kn == ("farrow_pipe_kernel" if farrow else "arb_pipe_kernel"), kn
                    assert_bit_equal(got, want, f"default ({kn}) vs {kt}: farrow={farrow} rate={rate} T={T} nch={nch}")
                    got, kn = outs["wave"]                    # farrow_wave_kernel's 24- and 32-slot classes, forced
                    assert kn == ("farrow_wave_kernel" if farrow else "arb_pipe_kernel"), kn
                    assert_bit_equal(got, want, f"wave ({kn}) vs {kt}: farrow={farrow} rate={rate} T={T} nch={nch}")
    assert seen == {"arb_pipe_kernel", "farrow_pipe_kernel"}


def test_stream_kernel_decimations_33_to_63_real_float32(pkg, O, torch_cuda, monkeypatch):
    """Decimations 33..63 of real Float32 samples (rounds 2-3: extra instantiations of fir_stream_kernel, 36 and 38 on the retired
    fir_direct_kernel; now all on fir_stream_rt_kernel): against the oracle and the universal kernel, chunked."""
    torch = torch_cuda
    rng = np.random.default_rng(3363)
    for M in (33, 35, 37, 44, 49, 52, 57, 63, 36, 38):
        for hl in (M + 3, 128):
            h = rng.standard_normal(hl).astype(np.float32)
            nch, n = 7, 90_001
            x = rng.standard_normal((nch, n)).astype(np.float32)
            xd = torch.from_numpy(x).cuda()
            sizes = [40_000, 1, n - 40_001]
            monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False)
            f = pkg.FIRFilter(h, Fraction(1, M))
            y = torch.cat(_run_chunks(f, xd, sizes), dim=-1).cpu().numpy()
            assert f.last_kernel_name() == "fir_stream_rt_kernel", (M, f.last_kernel_name())
            monkeypatch.setenv("MRHIP_FORCE_GENERIC", "1")
            g = pkg.FIRFilter(h, Fraction(1, M))
            yg = torch.cat(_run_chunks(g, xd, sizes), dim=-1).cpu().numpy()
            monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False)
            assert_bit_equal(y, yg, f"1//{M} hLen={hl} vs universal kernel")
            fo = O.FIRFilter(h, Fraction(1, M), tx=np.float32)
            yo = np.concatenate(_run_chunks(fo, x[nch - 1], sizes))
            assert_bit_equal(y[nch - 1], yo, f"1//{M} hLen={hl} vs oracle")
            assert_bit_equal(f.history, g.history, "history")
            f.close(); g.close()


def test_arb_pipe_exact_32_taps(pkg, O, torch_cuda, monkeypatch):
    """arb_pipe_kernel's whole-pipeline unrolling for 32 taps per phase (BASELINE config 4's shape; taken with >= 32 channels and
    LDS-DMA staging): bit-identical to the looped form (MRHIP_ARB_EXACT=0), to the universal kernel and to the oracle, for every
    sample / tap type the kernel serves, STRICT and FUSED, chunked (first / last tiles are staged through registers)."""
    torch = torch_cuda
    rng = np.random.default_rng(3232)
    Nphi, T, nch, n = 32, 32, 37, 70_001
    for th, tx in ((np.float64, np.float64), (np.float32, np.float32), (np.float32, np.complex64), (np.float64, np.float32), (np.float64, np.complex64)):
        h = rng.standard_normal(Nphi * T).astype(th)
        x = _rand(rng, (nch, n), tx) - 0.5
        xd = torch.from_numpy(x).cuda()
        sizes = [30_000, 1, n - 30_001]
        for rate in (math.pi / 3, 0.71):
            for numerics in (pkg.NUMERICS_STRICT, pkg.NUMERICS_FUSED):
                ys = {}
                for mode, env in (("exact", {}), ("loop", {"MRHIP_ARB_EXACT": "0"}), ("generic", {"MRHIP_FORCE_GENERIC": "1"})):
                    for k, v in env.items():
                        monkeypatch.setenv(k, v)
                    f = pkg.FIRFilter(h, float(rate), Nphi, numerics=numerics)
                    ys[mode] = (torch.cat(_run_chunks(f, xd, sizes), dim=-1).cpu().numpy(), f.last_kernel_name())
                    f.close()
                    for k in env:
                        monkeypatch.delenv(k)
                tag = f"{np.dtype(th)} taps x {np.dtype(tx)} rate={rate:.3f} numerics={numerics}"
                assert ys["exact"][1] == "arb_pipe_kernel" and ys["loop"][1] == "arb_pipe_kernel" and ys["generic"][1] == "arb_generic_kernel", (tag, ys["exact"][1], ys["generic"][1])
                assert_bit_equal(ys["exact"][0], ys["loop"][0], "unrolled vs looped " + tag)
                assert_bit_equal(ys["exact"][0], ys["generic"][0], "unrolled vs universal " + tag)
                if numerics == pkg.NUMERICS_STRICT:
                    fo = O.FIRFilter(h, float(rate), Nphi, tx=tx)
                    yo = np.concatenate(_run_chunks(fo, x[nch - 1], sizes))
                    assert_bit_equal(ys["exact"][0][nch - 1], yo, "unrolled vs oracle " + tag)


def test_stream_kernel_hand_scheduled_pair_config3b(pkg, O, torch_cuda, monkeypatch):
    """fir_stream_kernel's hand-scheduled pair of outputs for BASELINE config 3b's shape (FIRDecimator 1//4, 128 Float32 taps, ComplexF32
    samples; fir_stream_pair_c64_m4.inc): STRICT and FUSED, several channels, chunkings that put the start-from-zero seam (support.jl:46,
    the C++ path) and the hand-scheduled tiles into one stream, -0.0 / +-Inf / NaN samples -- bit for bit the universal kernel's and
    (STRICT) the oracle's outputs, end state and history."""
    torch = torch_cuda
    rng = np.random.default_rng(3404)
    h = (rng.standard_normal(128) / 8).astype(np.float32)
    nch, n = 5, 300_000
    x = (rng.standard_normal((nch, n)) + 1j * rng.standard_normal((nch, n))).astype(np.complex64)
    flat = x.view(np.float32)
    flat[0, 10] = -0.0; flat[0, 2001] = np.inf; flat[1, 2003] = -np.inf; flat[2, 150_001] = np.nan
    xd = torch.from_numpy(x).cuda()
    sizes = [3, 100_001, 1, 126, 131, n - 100_262]
    for numerics in (pkg.NUMERICS_STRICT, pkg.NUMERICS_FUSED):
        got = {}
        for mode, env in (("hand", {}), ("generic", {"MRHIP_FORCE_GENERIC": "1"})):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            f = pkg.FIRFilter(h, Fraction(1, 4), numerics=numerics)
            y = torch.cat(_run_chunks(f, xd, sizes), dim=-1).cpu().numpy()
            got[mode] = (y, f.last_kernel_name(), (f.state.phiIdx, f.state.inputDeficit), np.array(f.history))
            f.close()
            for k in env:
                monkeypatch.delenv(k)
        assert got["hand"][1] == "fir_stream_kernel" and got["generic"][1] != "fir_stream_kernel", (got["hand"][1], got["generic"][1])
        assert_bit_equal(got["hand"][0], got["generic"][0], f"hand-scheduled pair vs universal kernel, numerics {numerics}")
        assert got["hand"][2] == got["generic"][2]
        assert_bit_equal(got["hand"][3], got["generic"][3], "history")
        if numerics == pkg.NUMERICS_STRICT:
            for c in (0, 2, nch - 1):
                fo = O.FIRFilter(h, Fraction(1, 4), tx=np.complex64)
                assert_bit_equal(got["hand"][0][c], np.concatenate(_run_chunks(fo, x[c], sizes)), f"hand-scheduled pair vs oracle, channel {c}")


def test_interp_lane_kernel_config3a_window_in_registers(pkg, O, torch_cuda, monkeypatch):
    """interp_lane_kernel (kernels_interp_lane.hip; BASELINE config 3a's shape: FIRInterpolator 4//1, 32 taps per phase, ComplexF32 samples x
    Float32 taps): a lane per channel, one wave per stretch, the sliding window in registers, taps as SGPR operands, samples and outputs
    transposed through the wave's own LDS patch.  Outputs and history bit for bit those of rational_opair_kernel, of the universal kernel
    and (STRICT) of the oracle: full, several and partial channel groups, chunkings whose calls are not multiples of 8 inputs (the last
    block of a stretch is ragged), too short for the kernel (< 64 inputs: the output-pair kernel takes them, the history carries over) or
    one sample long, x and y given as VIEWS (odd sample offsets: the 8-byte store path; row strides that are not the length), -0.0 /
    +-Inf / NaN samples.  By default only long calls take the kernel (>= 5e7 channel-samples): checked at config 3a's width."""
    torch = torch_cuda
    rng = np.random.default_rng(707)
    for nch, n in ((64, 30_000), (256, 9_000), (50, 9_000), (113, 9_000)):
        h = (rng.standard_normal(128) / 4).astype(np.float32)
        x = (rng.standard_normal((nch, n)) + 1j * rng.standard_normal((nch, n))).astype(np.complex64)
        flat = x.view(np.float32)
        flat[0, 10] = -0.0; flat[0, 2001] = np.inf; flat[1, 2003] = -np.inf; flat[2, 5001] = np.nan
        xbig = torch.zeros((nch, n + 11), dtype=torch.complex64, device="cuda")
        xbig[:, 3:3 + n] = torch.from_numpy(x).cuda()
        xd = xbig[:, 3:3 + n]                                   # a view: 8-byte aligned rows, stride n + 11
        sizes = [3_001, 1, 17, 64, 65, n - 3_148 - 1_003, 1_003]
        for numerics in (pkg.NUMERICS_STRICT, pkg.NUMERICS_FUSED):
            got = {}
            for mode, env in (("lane", {"MRHIP_INTERP_LANE": "2"}), ("opair", {"MRHIP_INTERP_LANE": "0"}), ("generic", {"MRHIP_FORCE_GENERIC": "1"})):
                for k, v in env.items():
                    monkeypatch.setenv(k, v)
                f = pkg.FIRFilter(h, Fraction(4, 1), numerics=numerics)
                ybig = torch.zeros((nch, 4 * n + 9), dtype=torch.complex64, device="cuda")
                pos, names = 0, set()
                for sz in sizes:
                    cnt = f.filt_into(ybig[:, 1 + 4 * pos:1 + 4 * (pos + sz)], xd[:, pos:pos + sz])
                    assert cnt == 4 * sz
                    names.add(f.last_kernel_name())
                    pos += sz
                got[mode] = (ybig.cpu().numpy(), names, np.array(f.history))
                f.close()
                for k in env:
                    monkeypatch.delenv(k)
            tag = f"nch={nch} numerics={numerics}"
            assert got["lane"][1] == {"interp_lane_kernel", "rational_opair_kernel"} and got["opair"][1] == {"rational_opair_kernel"}, (tag, got["lane"][1], got["opair"][1])
            assert_bit_equal(got["lane"][0], got["generic"][0], "lane vs universal " + tag)      # (the guard columns 0 and 4 n + 1 ... stay zero in both)
            assert_bit_equal(got["lane"][0], got["opair"][0], "lane vs output-pair " + tag)
            assert_bit_equal(got["lane"][2], got["generic"][2], "history " + tag)
            if numerics == pkg.NUMERICS_STRICT:
                for c in (0, 1, 2, nch - 1):
                    fo = O.FIRFilter(h, Fraction(4, 1), tx=np.complex64)
                    assert_bit_equal(got["lane"][0][c, 1:1 + 4 * n], np.concatenate(_run_chunks(fo, x[c], sizes)), f"lane vs oracle ch {c} " + tag)
    # the default choice: long calls only; and on such a call every output against the output-pair kernel
    h = pkg.firdes(128, 0.5 / 4, beta=7.8562).astype(np.float32)
    nch, n = 256, 200_000
    xd = torch.view_as_complex(torch.rand((nch, n, 2), dtype=torch.float32, device="cuda") - 0.5)
    ys = {}
    for mode, env in (("default", {}), ("opair", {"MRHIP_INTERP_LANE": "0"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        f = pkg.FIRFilter(h, Fraction(4, 1))
        f.filt(xd[:, :10_000])
        assert f.last_kernel_name() == "rational_opair_kernel"
        y = f.filt(xd)
        ys[mode] = (y, f.last_kernel_name())
        f.close()
        for k in env:
            monkeypatch.delenv(k)
    assert ys["default"][1] == "interp_lane_kernel" and ys["opair"][1] == "rational_opair_kernel"
    assert torch.equal(torch.view_as_real(ys["default"][0]).view(torch.int32), torch.view_as_real(ys["opair"][0]).view(torch.int32))
    # what the kernel does not serve stays where it was: other ratios, lengths, sample types, few channels
    for (ratio, taps, tx, nchs) in ((Fraction(2, 1), 64, np.complex64, 64), (Fraction(4, 1), 64, np.complex64, 64), (Fraction(4, 1), 128, np.float32, 64), (Fraction(4, 1), 128, np.complex64, 16)):
        monkeypatch.setenv("MRHIP_INTERP_LANE", "2")
        f = pkg.FIRFilter((rng.standard_normal(taps) / 4).astype(np.float32), ratio)
        f.filt(torch.from_numpy(_rand(rng, (nchs, 5_000), tx)).cuda())
        assert f.last_kernel_name() != "interp_lane_kernel", (ratio, taps, tx, nchs)
        f.close()
        monkeypatch.delenv("MRHIP_INTERP_LANE")


def test_phase_stationary_kernel_first_call_after_an_upload(pkg, O, torch_cuda, monkeypatch):
    """poly_phase_stationary_kernel stages its tiles by LDS-DMA.  Rounds 3-5 let __syncthreads() stand for "this wave's DMA has landed"; it does
    not (no vmcnt wait is part of a workgroup barrier, and the kernel's LDS reads are assembly the compiler's DMA bookkeeping cannot see): with
    the signal JUST UPLOADED -- every line still to come from HBM -- a tile was read before its samples had arrived, and the first call after an
    upload returned wrong outputs in later tiles (never the second call: the lines were in the cache by then).  Found by
    tests/stress_random.py --seed 61 (4//33, 19 taps, 33 channels x 132 808 samples) in round 6; the wait is explicit now.  Every run below
    uploads the signal afresh and calls the tuned kernel FIRST."""
    torch = torch_cuda
    rng = np.random.default_rng(1061)
    for (L, M, taps, nch, n) in ((4, 33, 19, 33, 132_808), (4, 33, 19, 32, 265_616), (3, 40, 60, 33, 132_808)):
        h = rng.standard_normal(taps).astype(np.float32)
        x = (rng.random((nch, n), dtype=np.float32) - 0.5)
        xt = torch.from_numpy(x).pin_memory()
        for rep in range(4):
            xd = xt.cuda()                                         # a fresh upload: nothing of it is in any cache
            monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False)
            f = pkg.FIRFilter(h, Fraction(L, M))
            y = f.filt(xd).cpu().numpy()
            name = f.last_kernel_name()
            f.close()
            monkeypatch.setenv("MRHIP_FORCE_GENERIC", "1")
            g = pkg.FIRFilter(h, Fraction(L, M))
            yg = g.filt(xd).cpu().numpy()
            g.close()
            monkeypatch.delenv("MRHIP_FORCE_GENERIC")
            assert name == "poly_phase_stationary_kernel", name
            assert_bit_equal(y, yg, f"{L}//{M} {taps} taps {nch} ch x {n}, upload {rep}")
        fo = O.FIRFilter(h, Fraction(L, M), tx=np.float32)
        assert_bit_equal(y[nch - 1], fo.filt(x[nch - 1]), f"{L}//{M} vs oracle")


def test_decim_lane_kernel_behind_its_switch(pkg, O, torch_cuda, monkeypatch):
    """decim_lane_kernel (kernels_decim_lane.hip: FIRDecimator 1//4 x 128 taps, ComplexF32, a lane per channel in TRANSPOSED form -- the 32
    outputs whose windows contain a sample are in flight in registers, every accumulator started from -0.0 or, where the reference's loop
    does, from +0.0) is NOT the default -- on config 3b it measures slower than fir_stream_kernel (profiles/r06/experiments.md J) -- but
    stays in the library behind MRHIP_DECIM_LANE: outputs, end state and history bit for bit those of fir_stream_kernel, of the universal
    kernel and (STRICT) of the oracle: full, several and partial channel groups, every inputDeficit a call can start with (chunks of
    4 k + 1, + 2, + 3 samples), calls too short for the kernel, -0.0 / +-Inf / NaN samples and an all-negative-zero stretch (the sign of a zero
    sum is where "-0.0 start" and "first product initialises" could differ)."""
    torch = torch_cuda
    rng = np.random.default_rng(909)
    for nch, n in ((64, 30_000), (256, 9_000), (50, 9_000), (113, 9_000)):
        h = (rng.standard_normal(128) / 4).astype(np.float32)
        x = (rng.standard_normal((nch, n)) + 1j * rng.standard_normal((nch, n))).astype(np.complex64)
        flat = x.view(np.float32)
        flat[0, 10] = -0.0; flat[0, 2001] = np.inf; flat[1, 2003] = -np.inf; flat[2, 5001] = np.nan
        flat[3, 2 * 4_000:2 * 4_600] = -0.0                      # 600 samples of (-0.0, -0.0): whole windows of zeros of either sign
        flat[4, :400] = -0.0                                     # ... and at the very start (the +0.0 starts of support.jl:46)
        xd = torch.from_numpy(x).cuda()
        sizes = [3_001, 1, 17, 64, 65, 2, 3, 1_002, n - 5_158 - 1_003, 1_003]
        for numerics in (pkg.NUMERICS_STRICT, pkg.NUMERICS_FUSED):
            got = {}
            for mode, env in (("lane", {"MRHIP_DECIM_LANE": "2"}), ("stream", {}), ("generic", {"MRHIP_FORCE_GENERIC": "1"})):
                for k, v in env.items():
                    monkeypatch.setenv(k, v)
                f = pkg.FIRFilter(h, Fraction(1, 4), numerics=numerics)
                outs, names, pos = [], set(), 0
                for sz in sizes:
                    outs.append(f.filt(xd[:, pos:pos + sz]))
                    names.add(f.last_kernel_name())
                    pos += sz
                got[mode] = (torch.cat(outs, dim=-1).cpu().numpy(), names, (f.state.phiIdx, f.state.inputDeficit), np.array(f.history))
                f.close()
                for k in env:
                    monkeypatch.delenv(k)
            tag = f"nch={nch} numerics={numerics}"
            assert "decim_lane_kernel" in got["lane"][1] and "decim_lane_kernel" not in got["stream"][1], (tag, got["lane"][1], got["stream"][1])
            assert_bit_equal(got["lane"][0], got["generic"][0], "lane vs universal " + tag)
            assert_bit_equal(got["lane"][0], got["stream"][0], "lane vs streaming kernel " + tag)
            assert got["lane"][2] == got["generic"][2], tag
            assert_bit_equal(got["lane"][3], got["generic"][3], "history " + tag)
            if numerics == pkg.NUMERICS_STRICT:
                for c in (0, 3, 4, nch - 1):
                    fo = O.FIRFilter(h, Fraction(1, 4), tx=np.complex64)
                    assert_bit_equal(got["lane"][0][c], np.concatenate(_run_chunks(fo, x[c], sizes)), f"lane vs oracle ch {c} " + tag)


def test_arb_lane_kernel_float64_lane_per_channel(pkg, O, torch_cuda, monkeypatch):
    """arb_lane_kernel (kernels_arb_lane.hip; BASELINE config 4's shape: Float64 samples, 32 taps per phase, 64 channels, rate >= 1): a lane
    per channel, taps by scalar loads into SGPR operands, two outputs per window, samples through an LDS ring.  Outputs, end state and
    history bit for bit those of arb_pipe_kernel, of the universal kernel and of the oracle: STRICT and FUSED, Float64 and Float32 taps,
    16 and 32 taps per phase, full, several and partial channel groups, rates whose consecutive windows coincide (10.3) or never do (1.0),
    chunkings with a one-sample and a 17-sample call (first / last blocks come from the history), -0.0 / +-Inf / NaN samples."""
    torch = torch_cuda
    rng = np.random.default_rng(606)
    monkeypatch.setenv("MRHIP_ARB_SMALL_MAX", "0")           # (small calls go to the universal kernel: not here)
    cases = [(32, 32, 64, math.pi / 3, 40_000), (32, 32, 64, 1.0, 24_000), (32, 32, 128, 1.5, 21_000), (32, 32, 50, 2.7, 21_000),
             (32, 16, 64, 10.3, 12_000), (8, 32, 64, 1.9, 21_000), (10, 16, 113, math.e / 2, 21_000)]
    for (nphi, T, nch, rate, n) in cases:
        for th in (np.float64, np.float32):
            h = rng.standard_normal(nphi * T).astype(th)
            x = rng.random((nch, n)) - 0.5
            x[0, 5] = -0.0; x[0, 1000] = np.inf; x[1, 1001] = -np.inf; x[2, 7_000] = np.nan
            xd = torch.from_numpy(x).cuda()
            sizes = [5_000, 1, 17, n - 5_018 - 3_003, 3_003]
            for numerics in (pkg.NUMERICS_STRICT, pkg.NUMERICS_FUSED):
                got = {}
                for mode, env in (("lane", {}), ("pipe", {"MRHIP_ARB_LANE": "0"}), ("generic", {"MRHIP_FORCE_GENERIC": "1"})):
                    for k, v in env.items():
                        monkeypatch.setenv(k, v)
                    f = pkg.FIRFilter(h, float(rate), nphi, numerics=numerics)
                    y = torch.cat(_run_chunks(f, xd, sizes), dim=-1).cpu().numpy()
                    st = f.state
                    got[mode] = (y, f.last_kernel_name(), (st.phiIdx, st.inputDeficit, st.phiAccumulator, st.alpha), np.array(f.history))
                    f.close()
                    for k in env:
                        monkeypatch.delenv(k)
                tag = f"Nphi={nphi} T={T} nch={nch} rate={rate:.4f} taps={np.dtype(th)} numerics={numerics}"
                assert got["lane"][1] == "arb_lane_kernel" and got["pipe"][1] != "arb_lane_kernel" and got["generic"][1] == "arb_generic_kernel", (tag, got["lane"][1], got["pipe"][1])
                assert_bit_equal(got["lane"][0], got["generic"][0], "lane vs universal " + tag)
                assert_bit_equal(got["lane"][0], got["pipe"][0], "lane vs pipe " + tag)
                assert got["lane"][2] == got["generic"][2], tag
                assert_bit_equal(got["lane"][3], got["generic"][3], "history " + tag)
                if numerics == pkg.NUMERICS_STRICT:
                    for c in (0, 1, 2, nch - 1):
                        fo = O.FIRFilter(h, float(rate), nphi, tx=np.float64)
                        assert_bit_equal(got["lane"][0][c], np.concatenate(_run_chunks(fo, x[c], sizes)), f"lane vs oracle ch {c} " + tag)
    # what the kernel does not serve stays where it was: a decimating rate, few channels, Float32 samples
    for (nch, rate, tx) in ((64, 0.71, np.float64), (16, math.pi / 3, np.float64), (64, math.pi / 3, np.float32)):
        f = pkg.FIRFilter(rng.standard_normal(32 * 32), float(rate), 32)
        f.filt(torch.from_numpy(rng.random((nch, 20_000)).astype(tx)).cuda())
        assert f.last_kernel_name() != "arb_lane_kernel", (nch, rate, tx)
        f.close()


def test_arb_window_kernel_behind_its_switch(pkg, O, torch_cuda, monkeypatch):
    """arb_window_kernel (kernels_arb_window.hip: FIRArbitrary, Float64, 32 taps per phase, one wave per stretch with the window in
    registers) is NOT the default -- on config 4 it measures slower than arb_lane_kernel (profiles/r06/experiments.md G) -- but stays in the
    library behind MRHIP_ARB_WINDOW: outputs, end state and history bit for bit those of arb_lane_kernel, of the universal kernel and
    (STRICT) of the oracle: full, several and partial channel groups, rates whose consecutive windows coincide (10.3) or never do (1.0),
    chunkings with a one-sample and a 17-sample call, -0.0 / +-Inf / NaN samples."""
    torch = torch_cuda
    rng = np.random.default_rng(808)
    monkeypatch.setenv("MRHIP_ARB_SMALL_MAX", "0")
    cases = [(32, 64, math.pi / 3, 40_000), (32, 64, 1.0, 24_000), (32, 128, 1.5, 21_000), (32, 50, 2.7, 21_000), (32, 64, 10.3, 12_000), (10, 113, math.e / 2, 21_000)]
    for (nphi, nch, rate, n) in cases:
        h = rng.standard_normal(nphi * 32)
        x = rng.random((nch, n)) - 0.5
        x[0, 5] = -0.0; x[0, 1000] = np.inf; x[1, 1001] = -np.inf; x[2, 7_000] = np.nan
        xd = torch.from_numpy(x).cuda()
        sizes = [5_000, 1, 17, n - 5_018 - 3_003, 3_003]
        for numerics in (pkg.NUMERICS_STRICT, pkg.NUMERICS_FUSED):
            got = {}
            for mode, env in (("window", {"MRHIP_ARB_WINDOW": "2"}), ("lane", {}), ("generic", {"MRHIP_FORCE_GENERIC": "1"})):
                for k, v in env.items():
                    monkeypatch.setenv(k, v)
                f = pkg.FIRFilter(h, float(rate), nphi, numerics=numerics)
                y = torch.cat(_run_chunks(f, xd, sizes), dim=-1).cpu().numpy()
                st = f.state
                got[mode] = (y, f.last_kernel_name(), (st.phiIdx, st.inputDeficit, st.phiAccumulator, st.alpha), np.array(f.history))
                f.close()
                for k in env:
                    monkeypatch.delenv(k)
            tag = f"Nphi={nphi} nch={nch} rate={rate:.4f} numerics={numerics}"
            assert got["window"][1] == "arb_window_kernel" and got["lane"][1] == "arb_lane_kernel" and got["generic"][1] == "arb_generic_kernel", (tag, got["window"][1], got["lane"][1])
            assert_bit_equal(got["window"][0], got["generic"][0], "window vs universal " + tag)
            assert_bit_equal(got["window"][0], got["lane"][0], "window vs lane " + tag)
            assert got["window"][2] == got["generic"][2], tag
            assert_bit_equal(got["window"][3], got["generic"][3], "history " + tag)
            if numerics == pkg.NUMERICS_STRICT:
                for c in (0, 2, nch - 1):
                    fo = O.FIRFilter(h, float(rate), nphi, tx=np.float64)
                    assert_bit_equal(got["window"][0][c], np.concatenate(_run_chunks(fo, x[c], sizes)), f"window vs oracle ch {c} " + tag)


def test_large_L_runs_on_the_output_pair_kernel_in_period_blocks(pkg, O, torch_cuda, monkeypatch):
    """L > 512 (625//512, 1000//999, 640//441: ordinary clock-trim ratios) used to fall off the tuned kernels onto poly_tiled /
    poly_generic at 1-9 % of the HBM roofline.  The output-pair kernel now cuts the period of 2L outputs into BLOCKS, one workgroup
    per block for its whole life (kernels_rational_opair.hip: plan_rational_opair_blocks): outputs, end state and history must be
    bit for bit those of the universal kernel and of the oracle -- every tap / sample type, a prime L (ragged last block), L just
    above the old limit, the largest L, chunkings that change the phase every call, a chunk shorter than the history, and the
    special values (-0.0, +-Inf, NaN) the exact no-op slots must not disturb."""
    torch = torch_cuda
    rng = np.random.default_rng(77)
    cases = [(625, 512, 24, np.float32, np.float32, 3), (1000, 999, 24, np.float32, np.float32, 2), (640, 441, 24, np.float32, np.complex64, 2),
             (1009, 1000, 24, np.float32, np.float32, 1), (513, 512, 32, np.float32, np.float32, 2), (999, 1000, 24, np.float64, np.float32, 2),
             (1000, 999, 24, np.float64, np.float64, 1), (640, 441, 16, np.float64, np.complex128, 1), (4096, 4095, 24, np.float32, np.float32, 1),
             (2048, 1375, 7, np.float32, np.complex64, 2), (779, 1400, 24, np.float32, np.float32, 2)]
    for (L, M, T, th, tx, nch) in cases:
        h = (rng.standard_normal(T * L) / 8).astype(th)
        n = 120_000
        x = _rand(rng, (nch, n), tx) - 0.5
        flat = x.view(np.float32 if np.dtype(tx).itemsize // (2 if np.dtype(tx).kind == "c" else 1) == 4 else np.float64)
        flat[0, 5] = -0.0; flat[0, 1000] = np.inf; flat[0, 1001] = -np.inf; flat[0, 70_000] = np.nan
        xd = torch.from_numpy(x).cuda()
        sizes = [40_009, 3, 1, 30_011, n - 70_024]
        monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False)
        f = pkg.FIRFilter(h, Fraction(L, M))
        y_t = torch.cat(_run_chunks(f, xd, sizes), dim=-1).cpu().numpy()
        assert f.last_kernel_name() == "rational_opair_kernel", (L, M, T, f.last_kernel_name())
        monkeypatch.setenv("MRHIP_FORCE_GENERIC", "1")
        g = pkg.FIRFilter(h, Fraction(L, M))
        y_g = torch.cat(_run_chunks(g, xd, sizes), dim=-1).cpu().numpy()
        assert g.last_kernel_name() == "poly_generic_kernel"
        monkeypatch.delenv("MRHIP_FORCE_GENERIC", raising=False)
        assert_bit_equal(y_t, y_g, f"blocks vs universal kernel L={L} M={M} T={T} {th} {tx}")
        assert_bit_equal(f.history, g.history, "history")
        assert (f.state.phiIdx, f.state.inputDeficit) == (g.state.phiIdx, g.state.inputDeficit)
        fo = O.FIRFilter(h, Fraction(L, M), tx=tx)
        pos, ref = 0, []
        for s_ in sizes:
            ref.append(fo.filt(x[0, pos:pos + s_])); pos += s_
        # (NaN payloads / signs are the host FPU's on the oracle side: NaNs in the same places, everything else bit for bit)
        want = np.concatenate(ref)
        ft = np.float64 if want.dtype in (np.float64, np.complex128) else np.float32
        got_r, want_r = np.ascontiguousarray(y_t.reshape(nch, -1)[0]).view(ft), want.view(ft)
        assert np.array_equal(np.isnan(got_r), np.isnan(want_r)), f"NaN positions L={L} M={M}"
        assert_bit_equal(got_r[~np.isnan(want_r)], want_r[~np.isnan(want_r)], f"blocks vs oracle L={L} M={M}")
        assert (f.state.phiIdx, f.state.inputDeficit) == (fo.state.phiIdx, fo.state.inputDeficit)
        f.close(); g.close()
    # the opt-in FUSED numerics take the same road (instantiated for tapsPerPhi a multiple of 4) and match the oracle's fused switch
    L, M, T = 1000, 999, 24
    h = (rng.standard_normal(T * L) / 8).astype(np.float32)
    x = (_rand(rng, (2, 60_000), np.float32) - 0.5).astype(np.float32)
    f = pkg.FIRFilter(h, Fraction(L, M), numerics=pkg.NUMERICS_FUSED)
    y = torch.cat(_run_chunks(f, torch.from_numpy(x).cuda(), [25_001, 34_999]), dim=-1).cpu().numpy()
    assert f.last_kernel_name() == "rational_opair_kernel"
    O.set_fused(True)
    try:
        fo = O.FIRFilter(h, Fraction(L, M), tx=np.float32)
        ref = np.concatenate([fo.filt(x[1, :25_001]), fo.filt(x[1, 25_001:])])
    finally:
        O.set_fused(False)
    assert_bit_equal(y[1], ref, "blocks, FUSED numerics vs the oracle's fused switch")
    f.close()
