"""GPU tests of the device-resident stream state (csrc/stream_state.hip, include/multirate_hip.h: mrhip_filt_device_async,
mrhip_sync_state): the reference mutates 𝜙Idx / inputDeficit / 𝜙Accumulator at the end of every filt!
(src/Filters.jl:571-572, 627-628, 734-735, update() :663-673); here a device record carries them, so that

* a streaming loop captured into a HIP graph replays correctly at ANY fixed chunk size -- also chunk sizes
  that advance (𝜙Idx, inputDeficit) every call, and FIRArbitrary / FIRFarrow, whose output count varies from call to call;
* calls can be issued without the host ever waiting in their middle.

Bar: outputs, per-call counts and end state bit for bit equal to the oracle's chunk loop."""
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


def _tdtype(torch, d):
    return {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
            np.dtype(np.complex64): torch.complex64, np.dtype(np.complex128): torch.complex128}[np.dtype(d)]


def _graph_stream(torch, f, x_all, chunk, nrep, ncalls=1):
    """Capture `ncalls` consecutive filt! calls of `chunk` samples each into ONE graph (static input / output / count
    buffers), replay it `nrep` times with fresh input copied in front of every replay, collect what each call wrote."""
    nch = x_all.shape[0]
    bound = f.outputlength_bound(chunk)
    xs = torch.zeros((nch, chunk * ncalls), dtype=x_all.dtype, device="cuda")
    ys = torch.zeros((ncalls, nch, bound), dtype=_tdtype(torch, f.output_dtype), device="cuda")
    cnt = torch.zeros(ncalls, dtype=torch.int64, device="cuda")
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=s):
        for i in range(ncalls):
            f.filt_into_async(ys[i], xs[:, i * chunk:(i + 1) * chunk], cnt[i:i + 1])
    outs, counts = [], []
    pos = 0
    for rep in range(nrep):
        xs.copy_(x_all[:, pos:pos + chunk * ncalls])
        pos += chunk * ncalls
        g.replay()
        torch.cuda.synchronize()
        c = cnt.cpu().tolist()
        for i in range(ncalls):
            outs.append(ys[i, :, :c[i]].cpu().numpy().copy())
            counts.append(c[i])
    return outs, counts


def _oracle_chunks(fo, x, chunk, n):
    return [fo.filt(x[i * chunk:(i + 1) * chunk]) for i in range(n)]


def test_graph_capture_rational_chunk_that_advances_the_state(pkg, O, torch_cuda):
    """147//160 (the headline shape) with chunk 999 983 -- prime, so (𝜙Idx, inputDeficit) differ at every call and the
    count alternates between two values: 50 replays of a captured call == the oracle's chunk loop, end state included."""
    torch = torch_cuda
    L, M, chunk, nrep, nch = 147, 160, 999_983, 50, 2
    h = pkg.firdes(24 * L, 0.5 / L, beta=7.8562).astype(np.float32)
    gen = torch.Generator(device="cuda").manual_seed(7)
    x = torch.rand((nch, chunk * nrep), generator=gen, device="cuda", dtype=torch.float32) - 0.5
    f = pkg.FIRFilter(h, Fraction(L, M)).bind(np.float32, nch)
    outs, counts = _graph_stream(torch, f, x, chunk, nrep)
    assert f.last_kernel_name() == "rational_opair_kernel"
    assert len(set(counts)) == 2, counts                      # the state really moves
    xh = x.cpu().numpy()
    for c in (0, nch - 1):
        fo = O.FIRFilter(h, Fraction(L, M), tx=np.float32)
        yo = _oracle_chunks(fo, xh[c], chunk, nrep)
        assert counts == [len(v) for v in yo]
        assert_bit_equal(np.concatenate([o[c] for o in outs]), np.concatenate(yo), f"channel {c}")
    last = f.sync_state()
    assert last == counts[-1]
    st, so = f.state, fo.state
    assert (st.phiIdx, st.inputDeficit) == (so.phiIdx, so.inputDeficit)
    assert_bit_equal(f.history.reshape(nch, -1)[nch - 1], fo.history, "history after the replays")
    # a plain call continues the stream the replays left
    tail = torch.rand((nch, 12_345), generator=gen, device="cuda", dtype=torch.float32) - 0.5
    assert_bit_equal(f.filt(tail).cpu().numpy()[nch - 1], fo.filt(tail.cpu().numpy()[nch - 1]), "plain call after the replays")
    f.close()


def test_once_captured_filter_does_not_break_another_filters_capture(pkg, O, torch_cuda):
    """ADVICE r4 / VERDICT r5 item 7: a filter whose calls were captured once waits for the whole DEVICE whenever the host needs its state
    (replays run on streams the library never saw) -- and a plain hipDeviceSynchronize issued while ANOTHER stream is being captured in
    the global mode (torch.cuda.graph's default) is an unsafe call that invalidates that capture.  The library's device waits now run in
    the relaxed capture mode of the calling thread where they must happen, and -- measured: hipDeviceSynchronize invalidates a capture
    in that mode too -- are REFUSED while a stream the library has been called on is still capturing: filter B (captured and replayed
    earlier) is asked for its state while filter A's stream is capturing: MRHIP_ERR_UNSUPPORTED, A's capture survives and replays
    correctly, B answers once the capture is over."""
    torch = torch_cuda
    L, M, chunk, nch = 147, 160, 50_003, 2
    h = pkg.firdes(24 * L, 0.5 / L, beta=7.8562).astype(np.float32)
    gen = torch.Generator(device="cuda").manual_seed(11)
    x = torch.rand((nch, chunk * 6), generator=gen, device="cuda", dtype=torch.float32) - 0.5
    xh = x.cpu().numpy()
    # B: captured and replayed first (its `captured` flag is set for good)
    fb = pkg.FIRFilter(h, Fraction(L, M)).bind(np.float32, nch)
    outs_b, counts_b = _graph_stream(torch, fb, x, chunk, 3)
    # A: captured now; in the middle of the capture B is used from the host
    fa = pkg.FIRFilter(h, Fraction(L, M)).bind(np.float32, nch)
    bound = fa.outputlength_bound(chunk)
    xs = torch.zeros((nch, chunk), dtype=torch.float32, device="cuda")
    ys = torch.zeros((nch, bound), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=s):                                  # (capture_error_mode: the default, "global")
        fa.filt_into_async(ys, xs, cnt)
        # -> fresh() -> the device-wide wait of a once-captured filter: refused while a capture is active (nothing unsafe is issued)
        with pytest.raises(pkg.MultirateHIPError, match="being captured"):
            fb.sync_state()
        assert fb.next_output_count(chunk) == -1
    nb = fb.next_output_count(chunk)                                     # (the capture is over)
    fo_b = O.FIRFilter(h, Fraction(L, M), tx=np.float32)
    ref_b = _oracle_chunks(fo_b, xh[0], chunk, 4)
    assert counts_b == [len(v) for v in ref_b[:3]] and nb == len(ref_b[3])
    # A's graph is intact: replays == the oracle's chunk loop
    fo_a = O.FIRFilter(h, Fraction(L, M), tx=np.float32)
    for rep in range(4):
        xs.copy_(x[:, rep * chunk:(rep + 1) * chunk])
        g.replay()
        torch.cuda.synchronize()
        ref = fo_a.filt(xh[nch - 1, rep * chunk:(rep + 1) * chunk])
        c = int(cnt.item())
        assert c == len(ref)
        assert_bit_equal(ys[nch - 1, :c].cpu().numpy(), ref, f"replay {rep} of the graph captured while another filter was used")
    # and B goes on where its replays left it
    yb = fb.filt(x[:, 3 * chunk:4 * chunk]).cpu().numpy()
    assert_bit_equal(yb[0], ref_b[3], "the once-captured filter's next plain call")
    fa.close(); fb.close()


@pytest.mark.parametrize("L,M,tx,chunk,ncalls", [(1, 4, np.complex64, 10_007, 3), (3, 17, np.float64, 4_099, 2), (1, 1, np.float32, 5_001, 2),
                                                 (4, 1, np.float32, 3_001, 2), (2, 13, np.float32, 7_919, 1),
                                                 (1, 20, np.float32, 10_007, 3), (1, 37, np.complex64, 9_001, 2)])
def test_graph_capture_other_rational_kinds(pkg, O, torch_cuda, L, M, tx, chunk, ncalls):
    """FIRDecimator (inputDeficit moves; 1//20 and 1//37: fir_stream_rt_kernel, the decimation a run-time value), FIRRational on other kernels (2//13 has no device-planned tuned kernel: the
    universal kernel serves it), FIRStandard / FIRInterpolator (no state), several calls per graph."""
    torch = torch_cuda
    nrep, nch = 12, 2
    rng = np.random.default_rng(L * 100 + M)
    h = rng.standard_normal(96 if L > 1 else 64).astype(np.float32)
    n = chunk * ncalls * nrep
    xh = (rng.standard_normal((nch, n)) + (1j * rng.standard_normal((nch, n)) if np.issubdtype(tx, np.complexfloating) else 0)).astype(tx)
    x = torch.from_numpy(xh).cuda()
    f = pkg.FIRFilter(h, Fraction(L, M)).bind(tx, nch)
    outs, counts = _graph_stream(torch, f, x, chunk, nrep, ncalls)
    if L == 1 and M >= 16:
        assert f.last_kernel_name() == "fir_stream_rt_kernel", f.last_kernel_name()
    fo = O.FIRFilter(h, Fraction(L, M), tx=tx)
    yo = _oracle_chunks(fo, xh[1], chunk, ncalls * nrep)
    assert counts == [len(v) for v in yo]
    assert_bit_equal(np.concatenate([o[1] for o in outs]), np.concatenate(yo), f"{L}//{M} {tx}")
    f.sync_state()
    assert (f.state.phiIdx, f.state.inputDeficit) == (fo.state.phiIdx, fo.state.inputDeficit)
    f.close()


@pytest.mark.parametrize("kind", ["arbitrary", "farrow"])
@pytest.mark.parametrize("rate", [math.pi / 3, 1 / 2.123456789, 3.0])
def test_graph_capture_arbitrary_and_farrow(pkg, O, torch_cuda, kind, rate):
    """FIRArbitrary / FIRFarrow: the phase schedule, the output count and the accumulator all live on the device, so a
    captured call replays: 50 replays == the oracle's chunk loop (outputs, counts, 𝜙Accumulator, inputDeficit).  Rate 3.0
    cycles: the closed form of its cycle is evaluated on the device too."""
    torch = torch_cuda
    chunk, nrep, nch, Nphi = 100_003, 52, 2, 32
    h = (pkg.firdes(Nphi * 8, 0.45 / Nphi, beta=7.8562) * Nphi)
    rng = np.random.default_rng(17)
    xh = rng.random((nch, chunk * nrep))
    x = torch.from_numpy(xh).cuda()
    po = 4 if kind == "farrow" else None
    f = pkg.FIRFilter(h, rate, Nphi, po).bind(np.float64, nch)
    fo = O.FIRFilter(h, rate, Nphi, tx=np.float64, polyorder=po, pnfb=f.pnfb()) if po else O.FIRFilter(h, rate, Nphi, tx=np.float64)
    # The stream starts with a plain call and an asynchronous one of the captured size: the schedule's work buffers are
    # allocated by the first call of a size (allocations cannot be captured), and the plain call's serial prefix is what
    # finds a cycle of the accumulator (rate 3.0).  The graph then takes the stream over where they left it.
    y0 = f.filt(x[:, :chunk])
    assert_bit_equal(y0.cpu().numpy()[1], fo.filt(xh[1, :chunk]), "first plain call")
    yb = torch.empty((nch, f.outputlength_bound(chunk)), dtype=torch.float64, device="cuda")
    f.filt_into_async(yb, x[:, chunk:2 * chunk])
    n1 = f.sync_state()
    assert_bit_equal(yb[1, :n1].cpu().numpy(), fo.filt(xh[1, chunk:2 * chunk]), "first asynchronous call")
    nrep -= 2
    x, xh = x[:, 2 * chunk:], xh[:, 2 * chunk:]
    outs, counts = _graph_stream(torch, f, x, chunk, nrep)
    if rate == 3.0:
        assert f.schedule_info()["period"] == 3
    yo = _oracle_chunks(fo, xh[1], chunk, nrep)
    assert counts == [len(v) for v in yo], (counts[:5], [len(v) for v in yo][:5])
    assert_bit_equal(np.concatenate([o[1] for o in outs]), np.concatenate(yo), f"{kind} rate={rate}")
    assert f.sync_state() == counts[-1]
    st, so = f.state, fo.state
    assert (st.phiAccumulator, st.inputDeficit) == (so.phiAccumulator, so.inputDeficit)
    # and the stream goes on with plain calls
    tail = rng.random((nch, 5_000))
    assert_bit_equal(f.filt(torch.from_numpy(tail).cuda()).cpu().numpy()[1], fo.filt(tail[1]), "plain call after the replays")
    f.close()


@pytest.mark.parametrize("kind", ["rational", "decimator", "arbitrary", "farrow"])
def test_async_calls_never_wait_and_match_the_plain_loop(pkg, O, torch_cuda, kind):
    """mrhip_filt_device_async outside a graph: a loop of ragged calls that only enqueues (counts land in a device
    array), then ONE sync_state; == the same loop of plain calls; set_state / reset in between follow in stream order."""
    torch = torch_cuda
    rng = np.random.default_rng(5)
    nch = 3
    if kind == "rational":
        h, ratio, tx, po = rng.standard_normal(24 * 7).astype(np.float32), Fraction(7, 9), np.float32, None
    elif kind == "decimator":
        h, ratio, tx, po = rng.standard_normal(40).astype(np.float32), Fraction(1, 5), np.complex64, None
    else:
        h, ratio, tx, po = (pkg.firdes(32 * 6, 0.45 / 32, beta=7.0) * 32).astype(np.float32), 0.8123, np.float32, (3 if kind == "farrow" else None)
    sizes = [40_001, 3, 17_777, 1, 65_536, 29_999, 2, 50_000]
    n = sum(sizes)
    xh = rng.standard_normal((nch, n)).astype(np.float32)
    if np.issubdtype(tx, np.complexfloating):
        xh = (xh + 1j * rng.standard_normal((nch, n))).astype(tx)
    x = torch.from_numpy(xh).cuda()
    f = pkg.FIRFilter(h, ratio, 32, po).bind(tx, nch)
    g = pkg.FIRFilter(h, ratio, 32, po, pnfb=f.pnfb() if po else None).bind(tx, nch)
    for rnd in range(2):
        cnt = torch.zeros(len(sizes), dtype=torch.int64, device="cuda")
        ys, pos = [], 0
        for i, s in enumerate(sizes):
            y = torch.empty((nch, f.outputlength_bound(s)), dtype=_tdtype(torch, f.output_dtype), device="cuda")
            f.filt_into_async(y, x[:, pos:pos + s], cnt[i:i + 1])
            ys.append(y)
            pos += s
        last = f.sync_state()
        c = cnt.cpu().tolist()
        assert last == c[-1]
        ref = [g.filt(x[:, a:a + s]) for a, s in zip(np.cumsum([0] + sizes[:-1]), sizes)]
        assert c == [r.shape[1] for r in ref], (kind, c, [r.shape[1] for r in ref])
        for y, r, k in zip(ys, ref, c):
            assert torch.equal(torch.view_as_real(y[:, :k].contiguous()).view(torch.int32) if y.is_complex() else y[:, :k].contiguous().view(torch.int32),
                               torch.view_as_real(r.contiguous()).view(torch.int32) if r.is_complex() else r.contiguous().view(torch.int32))
        sf, sg = f.state, g.state
        assert (sf.phiIdx, sf.inputDeficit, sf.phiAccumulator) == (sg.phiIdx, sg.inputDeficit, sg.phiAccumulator)
        assert_bit_equal(f.history, g.history, "history")
        if rnd == 0:                      # move both streams, asynchronously for f
            if kind in ("arbitrary", "farrow"):
                f.set_state(1, 2, 7.25); g.set_state(1, 2, 7.25)
            else:
                f.reset(); g.reset()
    f.close(); g.close()


@pytest.mark.parametrize("kind", ["arbitrary", "farrow"])
@pytest.mark.parametrize("nch", [1, 3])
def test_small_and_long_asynchronous_calls_share_the_record(pkg, O, torch_cuda, kind, nch):
    """A call of at most 65 536 outputs runs its schedule on the caller's stream, a longer one on the filter's schedule stream beside the
    filter kernel before it (api.hip: inline_sched); both write the device record, and so do reset / set_state (on the schedule stream):
    any mix of them, never waited for in the middle, is the oracle's loop of plain calls -- outputs, counts, end state, history.  (The
    filter kernel of these calls also writes the next call's history: ShiftFold; inputs shorter than the history included.)"""
    torch = torch_cuda
    rng = np.random.default_rng(23 + nch)
    po = 3 if kind == "farrow" else None
    h = (pkg.firdes(32 * 6, 0.45 / 32, beta=7.0) * 32).astype(np.float32)
    rate = 0.8123
    sizes = [300_000, 5, 70_000, 400_000, 20_000, 1, 250_000, 60_000, 3, 330_000]
    xh = rng.standard_normal((nch, sum(sizes))).astype(np.float32)
    x = torch.from_numpy(xh).cuda()
    f = pkg.FIRFilter(h, rate, 32, po).bind(np.float32, nch)
    fo = [O.FIRFilter(h, rate, 32, tx=np.float32, polyorder=po, pnfb=f.pnfb()) if po else O.FIRFilter(h, rate, 32, tx=np.float32) for _ in range(nch)]
    for rnd in range(3):
        cnt = torch.zeros(len(sizes), dtype=torch.int64, device="cuda")
        ys, pos = [], 0
        for i, s in enumerate(sizes):
            y = torch.empty((nch, f.outputlength_bound(s)), dtype=_tdtype(torch, f.output_dtype), device="cuda")
            if rnd == 2 and i == 4:            # a plain call (the host waits for its count) in the middle of the asynchronous ones
                k = f.filt_into(y, x[:, pos:pos + s])
                cnt[i] = k
            else:
                f.filt_into_async(y, x[:, pos:pos + s], cnt[i:i + 1])
            ys.append(y)
            pos += s
        f.sync_state()
        c = cnt.cpu().tolist()
        pos = 0
        for i, s in enumerate(sizes):
            for ch in range(nch):
                r = fo[ch].filt(xh[ch, pos:pos + s])
                assert c[i] == len(r), (kind, nch, rnd, i, c[i], len(r))
                assert_bit_equal(ys[i][ch, :c[i]].cpu().numpy(), r, f"{kind} nch={nch} round {rnd} call {i} channel {ch}")
            pos += s
        st, so = f.state, fo[0].state
        assert (st.phiAccumulator, st.inputDeficit) == (so.phiAccumulator, so.inputDeficit)
        assert_bit_equal(np.atleast_2d(f.history)[0], fo[0].history, "history")
        if rnd == 0:
            f.reset()
            for o in fo:
                o.reset()
        elif rnd == 1:
            f.set_state(1, 2, 7.25)
            for o in fo:
                o.set_state(1, 2, 7.25)
    f.close()


def test_small_arbitrary_calls_take_the_universal_kernel(pkg, O, torch_cuda, monkeypatch):
    """A FIRArbitrary call of at most MRHIP_ARB_SMALL_MAX outputs x channels runs on arb_generic_kernel (faster there: nothing to set up), which
    then also writes the next call's history (ShiftFold) -- plain, asynchronous and captured calls against the oracle; a larger call of the same
    filter goes back to the tuned kernel."""
    torch = torch_cuda
    monkeypatch.setenv("MRHIP_ARB_SMALL_MAX", "150000")
    rng = np.random.default_rng(41)
    h = (pkg.firdes(32 * 10, 0.45 / 32, beta=7.8562) * 32)
    rate, chunk, nrep = 1 / 2.123456789, 50_021, 8
    xh = rng.standard_normal((1, chunk * (nrep + 3))).astype(np.float32)
    x = torch.from_numpy(xh).cuda()
    f = pkg.FIRFilter(h, rate, 32).bind(np.float32, 1)
    fo = O.FIRFilter(h, rate, 32, tx=np.float32)
    y0 = f.filt(x[:, :chunk])
    assert f.last_kernel_name() == "arb_generic_kernel"
    assert_bit_equal(y0.cpu().numpy()[0], fo.filt(xh[0, :chunk]), "plain small call")
    yb = torch.empty((1, f.outputlength_bound(chunk)), dtype=torch.float64, device="cuda")
    f.filt_into_async(yb, x[:, chunk:2 * chunk])
    n1 = f.sync_state()
    assert_bit_equal(yb[0, :n1].cpu().numpy(), fo.filt(xh[0, chunk:2 * chunk]), "asynchronous small call")
    outs, counts = _graph_stream(torch, f, x[:, 2 * chunk:], chunk, nrep)
    yo = _oracle_chunks(fo, xh[0, 2 * chunk:], chunk, nrep)
    assert counts == [len(v) for v in yo]
    assert_bit_equal(np.concatenate([o[0] for o in outs]), np.concatenate(yo), "captured small calls")
    assert_bit_equal(np.atleast_2d(f.history)[0], fo.history, "history behind the replays")
    big = rng.standard_normal((1, 600_000)).astype(np.float32)
    yb2 = f.filt(torch.from_numpy(big).cuda())
    assert f.last_kernel_name() == "arb_pipe_kernel"
    assert_bit_equal(yb2.cpu().numpy()[0], fo.filt(big[0]), "a larger call on the tuned kernel")
    f.close()


def test_async_call_needs_room_for_the_bound(pkg, torch_cuda):
    torch = torch_cuda
    h = np.ones(24 * 3, dtype=np.float32)
    f = pkg.FIRFilter(h, Fraction(3, 5)).bind(np.float32, 1)
    x = torch.zeros(1000, dtype=torch.float32, device="cuda")
    assert f.outputlength_bound(1000) == 600
    with pytest.raises(pkg.MultirateHIPError) as ei:
        f.filt_into_async(torch.empty(599, dtype=torch.float32, device="cuda"), x)
    assert ei.value.code == 2
    st = f.state
    assert (st.phiIdx, st.inputDeficit) == (1, 1)
    f.filt_into_async(torch.empty(600, dtype=torch.float32, device="cuda"), x)
    assert f.sync_state() == 600
    f.close()


def test_independent_streams_in_one_launch(pkg, O, torch_cuda):
    """mrhip_filt_device_multi: 64 INDEPENDENT single-channel FIRFilter objects (the reference's one-FIRFilter-per-signal
    streaming usage, README.md:87-141) with different phases, deficits, histories and chunk lengths, filtered by ONE
    launch per round of chunks; every stream bit-equal to its own oracle, round after round, state and history included.
    Streams with two channels, a stream too short for an output and the mixed-shape fallback ride along."""
    torch = torch_cuda
    L, M, ns = 147, 160, 64
    h = pkg.firdes(24 * L, 0.5 / L, beta=7.8562).astype(np.float32)
    rng = np.random.default_rng(64)
    nchs = [2 if i % 9 == 4 else 1 for i in range(ns)]
    fs = [pkg.FIRFilter(h, Fraction(L, M)).bind(np.float32, nchs[i]) for i in range(ns)]
    fos = [[O.FIRFilter(h, Fraction(L, M), tx=np.float32) for _ in range(nchs[i])] for i in range(ns)]
    # every stream starts somewhere else: a first plain call of its own length
    for i, f in enumerate(fs):
        x0 = rng.standard_normal((nchs[i], 1000 + 37 * i)).astype(np.float32)
        y0 = f.filt(torch.from_numpy(x0).cuda()).cpu().numpy()
        for c in range(nchs[i]):
            assert_bit_equal(y0[c], fos[i][c].filt(x0[c]), f"first call, stream {i}")
    for rnd in range(3):
        lens = [int(rng.integers(20_000, 60_000)) for _ in range(ns)]
        lens[7] = 3 if rnd == 1 else lens[7]                   # a chunk of three samples: sometimes no output at all
        xs_h = [rng.standard_normal((nchs[i], lens[i])).astype(np.float32) for i in range(ns)]
        xs = [torch.from_numpy(x).cuda() for x in xs_h]
        ys = pkg.filt_multi(fs, xs)
        assert fs[0].last_kernel_name() == "rational_opair_kernel"
        for i in range(ns):
            for c in range(nchs[i]):
                assert_bit_equal(ys[i].cpu().numpy()[c], fos[i][c].filt(xs_h[i][c]), f"round {rnd} stream {i} channel {c}")
            st, so = fs[i].state, fos[i][0].state
            assert (st.phiIdx, st.inputDeficit) == (so.phiIdx, so.inputDeficit), (rnd, i)
            assert_bit_equal(fs[i].history.reshape(nchs[i], -1)[0], fos[i][0].history, f"history of stream {i}")
    # filters that do not agree (another ratio): the plain loop of single calls behind the same entry
    g = pkg.FIRFilter(h[:96], Fraction(3, 2)).bind(np.float32, 1)
    go = O.FIRFilter(h[:96], Fraction(3, 2), tx=np.float32)
    xa, xb = rng.standard_normal(5000).astype(np.float32), rng.standard_normal(4000).astype(np.float32)
    ya, yb = pkg.filt_multi([fs[0], g], [torch.from_numpy(xa[None, :]).cuda(), torch.from_numpy(xb).cuda()])
    assert_bit_equal(ya.cpu().numpy()[0], fos[0][0].filt(xa), "fallback, stream 0")
    assert_bit_equal(yb.cpu().numpy(), go.filt(xb), "fallback, other ratio")
    for f in fs + [g]:
        f.close()


@pytest.mark.parametrize("ratio,tx", [(Fraction(1, 5), np.complex64), (Fraction(1, 1), np.float32), (Fraction(4, 1), np.float32), (Fraction(1, 24), np.float32)])
def test_independent_streams_other_kinds(pkg, O, torch_cuda, ratio, tx):
    """mrhip_filt_device_multi for FIRDecimator / FIRStandard (fir_stream_kernel: the Vector seam's start from zero applies to
    every stream's own call) and FIRInterpolator: 24 streams of unequal lengths, three rounds, each against its own oracle."""
    torch = torch_cuda
    ns = 24
    rng = np.random.default_rng(7)
    h = rng.standard_normal(48).astype(np.float32)
    fs = [pkg.FIRFilter(h, ratio).bind(tx, 1) for _ in range(ns)]
    fos = [O.FIRFilter(h, ratio, tx=tx) for _ in range(ns)]
    for rnd in range(3):
        lens = [int(rng.integers(5_000, 30_000)) + i for i in range(ns)]
        xs_h = []
        for m in lens:
            x = rng.standard_normal(m).astype(np.float32)
            if np.dtype(tx).kind == "c":
                x = (x + 1j * rng.standard_normal(m)).astype(tx)
            x[:40] = -0.0                                   # the seam's start from zero is visible only on signed zeros
            xs_h.append(x.astype(tx))
        ys = pkg.filt_multi(fs, [torch.from_numpy(x).cuda() for x in xs_h])
        assert fs[0].last_kernel_name() == ("rational_opair_kernel" if ratio.numerator != 1 else "fir_stream_rt_kernel" if ratio.denominator == 24 else "fir_stream_kernel")
        for i in range(ns):
            assert_bit_equal(ys[i].cpu().numpy(), fos[i].filt(xs_h[i]), f"{ratio} round {rnd} stream {i}")
            assert (fs[i].state.phiIdx, fs[i].state.inputDeficit) == (fos[i].state.phiIdx, fos[i].state.inputDeficit)
    for f in fs:
        f.close()


def test_cascade_under_graph_capture(pkg, O, torch_cuda):
    """FilterCascade (decimate 1//4, then 3//2) captured into a HIP graph: every stage maps its input length to the same count
    on every replay (8 000 -> 2 000 -> 3 000), so the lengths baked into the launches hold; 20 replays == the oracle's chunk
    loop.  A chunk that does not (8 001) is refused before anything is launched, the capture stays valid; so is a capture
    before the buffers between the stages exist."""
    torch = torch_cuda
    rng = np.random.default_rng(99)
    h1 = rng.standard_normal(64).astype(np.float32)
    h2 = rng.standard_normal(3 * 24).astype(np.float32)
    nch, chunk, nrep = 3, 8_000, 20
    xh = rng.standard_normal((nch, chunk * (nrep + 1))).astype(np.float32)
    x = torch.from_numpy(xh).cuda()
    cas = pkg.FilterCascade(pkg.FIRFilter(h1, Fraction(1, 4)), pkg.FIRFilter(h2, Fraction(3, 2)))
    o1, o2 = O.FIRFilter(h1, Fraction(1, 4), tx=np.float32), O.FIRFilter(h2, Fraction(3, 2), tx=np.float32)
    xs = torch.zeros((nch, chunk), dtype=torch.float32, device="cuda")
    # room for the last stage's bound of a device-planned call
    bound = cas.stages[1].bind(np.float32, nch).outputlength_bound(cas.stages[0].bind(np.float32, nch).outputlength_bound(chunk))
    ys = torch.zeros((nch, bound), dtype=torch.float32, device="cuda")

    s = torch.cuda.Stream()
    g0 = torch.cuda.CUDAGraph()                                 # cold: the buffers between the stages do not exist yet
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g0, stream=s):
        with pytest.raises(pkg.MultirateHIPError) as ei:
            cas.filt_into(ys, xs)
        assert ei.value.code == 5                               # MRHIP_ERR_UNSUPPORTED
        xs.add_(0)                                              # (the capture is intact: it records this)
    # one plain call of the size: allocates the buffers, and is the stream's first chunk
    xs.copy_(x[:, :chunk])
    n0 = cas.filt_into(ys, xs)
    ref0 = o2.filt(o1.filt(xh[1, :chunk]))
    assert n0 == len(ref0) == 3_000
    assert_bit_equal(ys[1, :n0].cpu().numpy(), ref0, "plain call")
    for f in cas.stages:
        f.sync_state()
    xlong = torch.zeros((nch, chunk + 1), dtype=torch.float32, device="cuda")
    g = torch.cuda.CUDAGraph()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=s):
        n1 = cas.filt_into(ys, xs)
        with pytest.raises(pkg.MultirateHIPError):              # 8 001 samples: the decimator's count would differ between replays
            cas.filt_into(ys, xlong)
    assert n1 == 3_000
    for rep in range(1, nrep + 1):
        xs.copy_(x[:, rep * chunk:(rep + 1) * chunk])
        g.replay()
        torch.cuda.synchronize()
        ref = o2.filt(o1.filt(xh[1, rep * chunk:(rep + 1) * chunk]))
        assert_bit_equal(ys[1, :3_000].cpu().numpy(), ref, f"replay {rep}")
    for f, fo in zip(cas.stages, (o1, o2)):
        f.sync_state()
        assert (f.state.phiIdx, f.state.inputDeficit) == (fo.state.phiIdx, fo.state.inputDeficit)
    cas.close()


def test_chained_asynchronous_calls(pkg, O, torch_cuda):
    """mrhip_filt_device_chained: decimate 1//4, then 147//160, both asynchronous -- the second filter's input length is the
    first one's count, which only the device knows (ragged chunks: it changes from call to call).  Outputs, counts and end
    states == the oracle's loop of filt(f2, filt(f1, x))."""
    torch = torch_cuda
    rng = np.random.default_rng(123)
    h1 = rng.standard_normal(64).astype(np.float32)
    h2 = pkg.firdes(24 * 147, 0.5 / 147, beta=7.8562).astype(np.float32)
    nch = 2
    f1 = pkg.FIRFilter(h1, Fraction(1, 4)).bind(np.float32, nch)
    f2 = pkg.FIRFilter(h2, Fraction(147, 160)).bind(np.float32, nch)
    o1, o2 = O.FIRFilter(h1, Fraction(1, 4), tx=np.float32), O.FIRFilter(h2, Fraction(147, 160), tx=np.float32)
    sizes = [40_001, 17, 3, 99_999, 1, 64_000, 2, 77_777]
    xh = rng.standard_normal((nch, sum(sizes))).astype(np.float32)
    x = torch.from_numpy(xh).cuda()
    big = max(sizes)
    mid = torch.zeros((nch, f1.outputlength_bound(big)), dtype=torch.float32, device="cuda")
    outs = [torch.zeros((nch, f2.outputlength_bound(f1.outputlength_bound(s))), dtype=torch.float32, device="cuda") for s in sizes]
    cnt = torch.zeros(len(sizes), dtype=torch.int64, device="cuda")
    pos = 0
    for i, s in enumerate(sizes):                               # only enqueues: no count ever reaches the host
        b1 = f1.outputlength_bound(s)
        f1.filt_into_async(mid[:, :b1], x[:, pos:pos + s])
        f2.filt_into_async(outs[i], mid[:, :b1], cnt[i:i + 1], after=f1)
        pos += s
    torch.cuda.synchronize()
    assert f2.last_kernel_name() == "rational_opair_kernel" and f1.last_kernel_name() == "fir_stream_kernel"
    c = cnt.cpu().tolist()
    pos = 0
    for i, s in enumerate(sizes):
        ref = o2.filt(o1.filt(xh[1, pos:pos + s]))
        assert c[i] == len(ref), (i, s, c[i], len(ref))
        assert_bit_equal(outs[i][1, :c[i]].cpu().numpy(), ref, f"call {i} ({s} samples)")
        pos += s
    for f, fo in ((f1, o1), (f2, o2)):
        f.sync_state()
        assert (f.state.phiIdx, f.state.inputDeficit) == (fo.state.phiIdx, fo.state.inputDeficit)
    # a chained call behind a HOST-planned call of its predecessor: that call's count is not in the call record -- refused
    y1 = f1.filt(x[:, :4_000])
    with pytest.raises(pkg.MultirateHIPError):
        f2.filt_into_async(outs[0], mid[:, :f1.outputlength_bound(4_000)], after=f1)
    # a filter the pair kernels do not serve (2//13: the universal kernel plans on the device too) chains as well
    hg = rng.standard_normal(2 * 24).astype(np.float32)
    g = pkg.FIRFilter(hg, Fraction(2, 13)).bind(np.float32, nch)
    og = O.FIRFilter(hg, Fraction(2, 13), tx=np.float32)
    o1b = O.FIRFilter(h1, Fraction(1, 4), tx=np.float32)
    f1.reset()
    yg = torch.zeros((nch, g.outputlength_bound(f1.outputlength_bound(50_003))), dtype=torch.float32, device="cuda")
    cg = torch.zeros(2, dtype=torch.int64, device="cuda")
    pos = 0
    for i, sz in enumerate((50_003, 7_919)):
        b1 = f1.outputlength_bound(sz)
        f1.filt_into_async(mid[:, :b1], x[:, pos:pos + sz])
        g.filt_into_async(yg, mid[:, :b1], cg[i:i + 1], after=f1)
        torch.cuda.synchronize()
        ref = og.filt(o1b.filt(xh[1, pos:pos + sz]))
        assert int(cg[i].item()) == len(ref), (i, int(cg[i].item()), len(ref))
        assert_bit_equal(yg[1, :len(ref)].cpu().numpy(), ref, f"2//13 chained, call {i}")
        pos += sz
    assert g.last_kernel_name() == "poly_generic_kernel"
    g.sync_state()
    assert (g.state.phiIdx, g.state.inputDeficit) == (og.state.phiIdx, og.state.inputDeficit)
    for f in (f1, f2, g):
        f.close()


def test_cascade_captured_at_a_chunk_size_that_changes_the_counts(pkg, O, torch_cuda):
    """FilterCascade.filt_into_async (mrhip_cascade_filt_device_async) under HIP-graph capture with a PRIME chunk: decimate 1//4,
    then 3//2 -- the decimator's count alternates between replays, the rational stage takes it from the device.  30 replays ==
    the oracle's chunk loop (outputs, counts, end states)."""
    torch = torch_cuda
    rng = np.random.default_rng(321)
    h1 = rng.standard_normal(48).astype(np.float32)
    h2 = rng.standard_normal(3 * 24).astype(np.float32)
    nch, chunk, nrep = 2, 10_007, 30
    xh = rng.standard_normal((nch, chunk * (nrep + 1))).astype(np.float32)
    x = torch.from_numpy(xh).cuda()
    cas = pkg.FilterCascade(pkg.FIRFilter(h1, Fraction(1, 4)), pkg.FIRFilter(h2, Fraction(3, 2)))
    o1, o2 = O.FIRFilter(h1, Fraction(1, 4), tx=np.float32), O.FIRFilter(h2, Fraction(3, 2), tx=np.float32)
    xs = torch.zeros((nch, chunk), dtype=torch.float32, device="cuda")
    xs.copy_(x[:, :chunk])
    y0 = cas.filt(xs)                                            # a plain call of the size: allocates the buffers; the stream's first chunk
    assert_bit_equal(y0[1].cpu().numpy(), o2.filt(o1.filt(xh[1, :chunk])), "plain call")
    ys = torch.zeros((nch, cas.outputlength_bound(chunk)), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    for f in cas.stages:
        f.sync_state()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=s):
        cas.filt_into_async(ys, xs, cnt)
    counts = set()
    for rep in range(1, nrep + 1):
        xs.copy_(x[:, rep * chunk:(rep + 1) * chunk])
        g.replay()
        torch.cuda.synchronize()
        ref = o2.filt(o1.filt(xh[1, rep * chunk:(rep + 1) * chunk]))
        c = int(cnt.item())
        counts.add(c)
        assert c == len(ref), (rep, c, len(ref))
        assert_bit_equal(ys[1, :c].cpu().numpy(), ref, f"replay {rep}")
    assert len(counts) >= 2                                      # the counts did change between replays
    for f, fo in zip(cas.stages, (o1, o2)):
        f.sync_state()
        assert (f.state.phiIdx, f.state.inputDeficit) == (fo.state.phiIdx, fo.state.inputDeficit)
    cas.close()


@pytest.mark.parametrize("kind", ["arbitrary", "farrow"])
def test_cascade_with_an_arbitrary_rate_first_stage_under_capture(pkg, O, torch_cuda, kind):
    """FIRArbitrary / FIRFarrow (rate 0.37: the count differs from call to call and only the device knows it), then a decimator 1//3
    chained on that count: FilterCascade.filt_into_async captured once, 25 replays == the oracle's chunk loop."""
    torch = torch_cuda
    rng = np.random.default_rng(777)
    Nphi, chunk, nrep, nch = 32, 50_021, 25, 2
    h1 = (pkg.firdes(Nphi * 8, 0.45 / Nphi, beta=7.8562) * Nphi)
    h2 = rng.standard_normal(45).astype(np.float64)
    po = 3 if kind == "farrow" else None
    f1 = pkg.FIRFilter(h1, 0.37, Nphi, po)
    cas = pkg.FilterCascade(f1, pkg.FIRFilter(h2, Fraction(1, 3)))
    xh = rng.random((nch, chunk * (nrep + 2)))
    x = torch.from_numpy(xh).cuda()
    xs = torch.zeros((nch, chunk), dtype=torch.float64, device="cuda")
    xs.copy_(x[:, :chunk])
    y0 = cas.filt(xs)                                            # plain call: buffers, schedule work space, the stream's first chunk
    o1 = O.FIRFilter(h1, 0.37, Nphi, tx=np.float64, polyorder=po, pnfb=f1.pnfb()) if po else O.FIRFilter(h1, 0.37, Nphi, tx=np.float64)
    o2 = O.FIRFilter(h2, Fraction(1, 3), tx=np.float64)
    assert_bit_equal(y0[1].cpu().numpy(), o2.filt(o1.filt(xh[1, :chunk])), "plain call")
    ys = torch.zeros((nch, cas.outputlength_bound(chunk)), dtype=torch.float64, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    xs.copy_(x[:, chunk:2 * chunk])
    cas.filt_into_async(ys, xs, cnt)                             # an asynchronous call of the size: the device schedule's buffers exist before the capture
    torch.cuda.synchronize()
    ref = o2.filt(o1.filt(xh[1, chunk:2 * chunk]))
    assert int(cnt.item()) == len(ref)
    assert_bit_equal(ys[1, :len(ref)].cpu().numpy(), ref, "asynchronous call")
    for f in cas.stages:
        f.sync_state()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=s):
        cas.filt_into_async(ys, xs, cnt)
    counts = set()
    for rep in range(2, nrep + 2):
        xs.copy_(x[:, rep * chunk:(rep + 1) * chunk])
        g.replay()
        torch.cuda.synchronize()
        ref = o2.filt(o1.filt(xh[1, rep * chunk:(rep + 1) * chunk]))
        c = int(cnt.item())
        counts.add(c)
        assert c == len(ref), (rep, c, len(ref))
        assert_bit_equal(ys[1, :c].cpu().numpy(), ref, f"replay {rep}")
    assert len(counts) >= 2
    for f, fo in zip(cas.stages, (o1, o2)):
        f.sync_state()
    assert (cas.stages[1].state.phiIdx, cas.stages[1].state.inputDeficit) == (o2.state.phiIdx, o2.state.inputDeficit)
    cas.close()


@pytest.mark.parametrize("kind,rate", [("arbitrary", math.pi / 3), ("farrow", 0.83), ("arbitrary", 3.0)])
def test_cascade_with_an_arbitrary_rate_later_stage_under_capture(pkg, O, torch_cuda, kind, rate):
    """Decimate 1//4 (prime chunk: its count alternates), THEN FIRArbitrary / FIRFarrow: the phase schedule of the second stage is
    laid out on the device for an input length only the device knows (the BEGIN kernel takes it from the decimator's call record).
    Captured once, 25 replays == the oracle's chunk loop; rate 3.0 cycles (closed form on the device)."""
    torch = torch_cuda
    rng = np.random.default_rng(888)
    Nphi, chunk, nrep, nch = 32, 40_009, 25, 2
    h1 = rng.standard_normal(40).astype(np.float64)
    h2 = (pkg.firdes(Nphi * 8, 0.45 / Nphi, beta=7.8562) * Nphi)
    po = 3 if kind == "farrow" else None
    f2 = pkg.FIRFilter(h2, rate, Nphi, po).bind(np.float64, nch)
    cas = pkg.FilterCascade(pkg.FIRFilter(h1, Fraction(1, 4)), f2)
    xh = rng.random((nch, chunk * (nrep + 3)))
    x = torch.from_numpy(xh).cuda()
    xs = torch.zeros((nch, chunk), dtype=torch.float64, device="cuda")
    o1 = O.FIRFilter(h1, Fraction(1, 4), tx=np.float64)
    o2 = O.FIRFilter(h2, rate, Nphi, tx=np.float64, polyorder=po, pnfb=f2.pnfb()) if po else O.FIRFilter(h2, rate, Nphi, tx=np.float64)
    ys = None
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    for rep in range(3):                                         # a plain call, then two asynchronous ones of the size (buffers, schedule work space)
        xs.copy_(x[:, rep * chunk:(rep + 1) * chunk])
        ref = o2.filt(o1.filt(xh[1, rep * chunk:(rep + 1) * chunk]))
        if rep == 0:
            y0 = cas.filt(xs)
            assert_bit_equal(y0[1].cpu().numpy(), ref, "plain call")
            ys = torch.zeros((nch, cas.outputlength_bound(chunk)), dtype=torch.float64, device="cuda")
        else:
            cas.filt_into_async(ys, xs, cnt)
            torch.cuda.synchronize()
            assert int(cnt.item()) == len(ref), (rep, int(cnt.item()), len(ref))
            assert_bit_equal(ys[1, :len(ref)].cpu().numpy(), ref, f"asynchronous call {rep}")
    for f in cas.stages:
        f.sync_state()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=s):
        cas.filt_into_async(ys, xs, cnt)
    for rep in range(3, nrep + 3):
        xs.copy_(x[:, rep * chunk:(rep + 1) * chunk])
        g.replay()
        torch.cuda.synchronize()
        ref = o2.filt(o1.filt(xh[1, rep * chunk:(rep + 1) * chunk]))
        c = int(cnt.item())
        assert c == len(ref), (rep, c, len(ref))
        assert_bit_equal(ys[1, :c].cpu().numpy(), ref, f"replay {rep}")
    cas.stages[0].sync_state()
    assert (cas.stages[0].state.phiIdx, cas.stages[0].state.inputDeficit) == (o1.state.phiIdx, o1.state.inputDeficit)
    cas.stages[1].sync_state()
    assert cas.stages[1].state.inputDeficit == o2.state.inputDeficit
    cas.close()
