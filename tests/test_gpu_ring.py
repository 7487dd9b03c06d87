"""GPU tests of the ring of arriving chunks (include/multirate_hip.h: mrhip_ring_*; csrc/ring_api.inc, pair_loader.h:
pair_ring_loader_wave): the reference's streaming loop  y_i = filt(self, x_i)  over the chunks of a signal (README.md:87-141, state
carried by src/Filters.jl:571-572, history by support.jl:61-80) fed to ONE resident kernel instead of one launch per chunk.

Bar: outputs, per-chunk counts, end state and history bit for bit equal to the oracle's chunk loop -- for chunk sizes that change
the phase every chunk, chunks shorter than the history, chunks without outputs, buffers refilled in place, a ring left idle past
its deadline, and the filter handed back and forth between the ring and plain calls."""
import time
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


def _signal(rng, tx, nch, n):
    x = rng.standard_normal((nch, n)).astype(np.float32)
    if np.dtype(tx).kind == "c":
        x = x + 1j * rng.standard_normal((nch, n)).astype(np.float32)
    return x.astype(tx)


def _oracle_loop(O, h, ratio, tx, x, cuts):
    """per channel: the chunk loop's outputs, chunk by chunk; end state; history"""
    nch = x.shape[0]
    fs = [O.FIRFilter(h, ratio, tx=tx) for _ in range(nch)]
    outs = [[f.filt(x[c, a:b]) for a, b in zip(cuts[:-1], cuts[1:])] for c, f in enumerate(fs)]
    return outs, fs


def _cuts(n, sizes):
    cuts, i = [0], 0
    while cuts[-1] < n:
        cuts.append(min(n, cuts[-1] + sizes[i % len(sizes)]))
        i += 1
    return cuts


CASES = [
    # (ratio, taps per phase, tap dtype, sample dtype, channels, samples, chunk sizes)
    (Fraction(147, 160), 24, np.float32, np.float32, 1, 400_000, [50_000]),                      # the phase returns to 1 every chunk
    (Fraction(147, 160), 24, np.float32, np.float32, 1, 300_007, [9_973, 30_011, 7, 60_013]),     # it does not; a chunk shorter than the history
    (Fraction(147, 160), 24, np.float32, np.float32, 3, 200_000, [33_331, 1, 2, 5, 20_000]),      # chunks without outputs, several channels
    (Fraction(147, 160), 24, np.float32, np.complex64, 2, 150_000, [25_013, 40_000]),
    (Fraction(160, 147), 24, np.float32, np.float32, 2, 120_000, [17_777, 30_000]),               # L > M
    (Fraction(4, 1), 32, np.float32, np.complex64, 2, 60_000, [7_001, 12_345]),                   # FIRInterpolator, C3a's shape
    (Fraction(147, 160), 24, np.float64, np.float32, 2, 120_000, [20_011, 33_000]),               # the README's mixed case
    (Fraction(147, 160), 24, np.float64, np.float64, 1, 120_000, [20_011, 33_000]),
    (Fraction(3, 2), 32, np.float64, np.complex128, 1, 60_000, [9_001, 15_000]),
    # found by scripts/stress_ring.py: the Float64 instantiation's grid (two workgroups per CU by the occupancy query) did not all fit the
    # chip -- 509 of 512 started -- and tickets dealt to the others were never served; ring_launch now checks the launch and shrinks it
    (Fraction(5, 3), 24, np.float64, np.float64, 5, 567_712, [15567, 23797, 17, 37742, 33328, 37889, 26609, 13129, 20436, 37278, 40601, 47597, 5894,
                                                                51521, 15, 1554, 31927, 30770, 58479, 53535, 27]),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0].numerator}_{c[0].denominator}-T{c[1]}-{np.dtype(c[2]).name}x{np.dtype(c[3]).name}-{c[4]}ch")
def test_ring_equals_the_chunk_loop(pkg, O, torch_cuda, case, monkeypatch):
    torch = torch_cuda
    monkeypatch.setenv("MRHIP_RING_IDLE_MS", "500")
    ratio, T, th, tx, nch, n, sizes = case
    rng = np.random.default_rng(hash((ratio.numerator, T, nch, n)) % (1 << 31))
    L = ratio.numerator
    h = (pkg.firdes(T * L, 0.45 / max(L, ratio.denominator), beta=7.0) * L).astype(th)
    x = _signal(rng, tx, nch, n)
    cuts = _cuts(n, sizes)
    ref, fos = _oracle_loop(O, h, ratio, tx, x, cuts)
    f = pkg.FIRFilter(h, ratio, device=0).bind(tx, nch)
    xd = torch.from_numpy(x).cuda()
    bound = max(f.outputlength_bound(b - a) for a, b in zip(cuts[:-1], cuts[1:]))
    ys = torch.zeros((len(cuts) - 1, nch, max(bound, 1)), dtype=_tdtype(torch, f.output_dtype), device="cuda")
    torch.cuda.synchronize()
    with f.open_ring() as ring:
        assert ring.info()["resident"], "this shape is served by the resident kernel"
        counts, seqs = [], []
        for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
            cnt, seq = ring.push(ys[i], xd[:, a:b])
            counts.append(cnt); seqs.append(seq)
        ring.wait(seqs[len(seqs) // 2])                     # one chunk by its own flag
        ring.drain()
        assert ring.info()["pushed"] == len(cuts) - 1
    for i in range(len(cuts) - 1):
        got = ys[i].cpu().numpy()
        for c in range(nch):
            assert counts[i] == len(ref[c][i]), (i, counts[i], len(ref[c][i]))
            assert_bit_equal(got[c, :counts[i]], ref[c][i], f"chunk {i} channel {c}")
    # the stream goes on in the filter object: state, history, and a plain call behind the ring
    st, so = f.state, fos[0].state
    assert (st.phiIdx, st.inputDeficit) == (so.phiIdx, so.inputDeficit)
    hist = f.history
    for c in range(nch):
        assert_bit_equal(np.asarray(hist).reshape(nch, -1)[c], fos[c].history, f"history of channel {c}")
    more = _signal(rng, tx, nch, 5_003)
    y2 = f.filt(torch.from_numpy(more).cuda()).cpu().numpy().reshape(nch, -1)
    for c in range(nch):
        assert_bit_equal(y2[c], fos[c].filt(more[c]), f"plain call behind the ring, channel {c}")
    f.close()


def test_ring_more_chunks_than_slots_and_buffers_refilled_in_place(pkg, O, torch_cuda, monkeypatch):
    """300 chunks through a ring of 64 slots (a push into a full ring waits for the oldest) out of TWO input and TWO output buffers
    that are refilled / read back as soon as their chunk's flag is up: what the kernel reads must be what was written last."""
    torch = torch_cuda
    monkeypatch.setenv("MRHIP_RING_IDLE_MS", "500")
    ratio, L, M = Fraction(147, 160), 147, 160
    h = pkg.firdes(24 * L, 0.5 / L, beta=7.8562).astype(np.float32)
    rng = np.random.default_rng(5)
    nchunks, chunk = 300, 9_973
    x = _signal(rng, np.float32, 1, nchunks * chunk)
    fo = O.FIRFilter(h, ratio, tx=np.float32)
    f = pkg.FIRFilter(h, ratio, device=0).bind(np.float32, 1)
    xb = [torch.zeros((1, chunk), dtype=torch.float32, device="cuda") for _ in range(2)]
    yb = [torch.zeros((1, f.outputlength_bound(chunk)), dtype=torch.float32, device="cuda") for _ in range(2)]
    xh = torch.from_numpy(x).pin_memory()
    pending = [None, None]
    with f.open_ring() as ring:
        def collect(slot):
            cnt, seq, i = pending[slot]
            ring.wait(seq)
            got = yb[slot][0, :cnt].cpu().numpy()
            assert_bit_equal(got, fo.filt(x[0, i * chunk:(i + 1) * chunk]), f"chunk {i}")
            pending[slot] = None
        for i in range(nchunks):
            slot = i & 1
            if pending[slot] is not None:
                collect(slot)
            xb[slot].copy_(xh[:, i * chunk:(i + 1) * chunk])
            # x must be complete when it is pushed: wait for the STREAM that filled it (a device-wide synchronize would wait for
            # the resident kernel to leave on its idle deadline)
            torch.cuda.current_stream().synchronize()
            cnt, seq = ring.push(yb[slot], xb[slot])
            pending[slot] = (cnt, seq, i)
        # (collected in order: the oracle is one stream)
        for slot in ((nchunks & 1), (nchunks & 1) ^ 1):
            if pending[slot] is not None:
                collect(slot)
    f.close()

    # the same with all chunks in flight at once: 300 pushes back to back through the 64 slots
    fo = O.FIRFilter(h, ratio, tx=np.float32)
    f = pkg.FIRFilter(h, ratio, device=0).bind(np.float32, 1)
    xd = torch.from_numpy(x).cuda()
    n_out = f.outputlength(x.shape[1])
    y = torch.zeros((1, n_out), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    with f.open_ring() as ring:
        total, last = ring.push_chunks(y, xd, chunk)
        ring.drain()
        assert total == n_out and last == nchunks - 1
    assert_bit_equal(y.cpu().numpy()[0], fo.filt(x[0]), "300 chunks back to back")
    f.close()


def test_ring_outlives_its_idle_deadline(pkg, O, torch_cuda, monkeypatch):
    """The resident kernel ends by itself when nothing is pushed for MRHIP_RING_IDLE_MS; the next push starts a new one and the
    stream continues (state, history slot and numbering carried over)."""
    torch = torch_cuda
    monkeypatch.setenv("MRHIP_RING_IDLE_MS", "60")
    ratio = Fraction(147, 160)
    h = pkg.firdes(24 * 147, 0.5 / 147, beta=7.8562).astype(np.float32)
    rng = np.random.default_rng(6)
    x = _signal(rng, np.float32, 2, 90_000)
    cuts = [0, 20_011, 40_001, 65_000, 90_000]
    ref, fos = _oracle_loop(O, h, ratio, np.float32, x, cuts)
    f = pkg.FIRFilter(h, ratio, device=0).bind(np.float32, 2)
    xd = torch.from_numpy(x).cuda()
    ys = torch.zeros((4, 2, f.outputlength_bound(30_000)), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ring = f.open_ring()
    counts = []
    for i in range(4):
        cnt, seq = ring.push(ys[i], xd[:, cuts[i]:cuts[i + 1]])
        counts.append(cnt)
        ring.wait(seq)
        if i in (0, 2):
            time.sleep(0.4)                                 # well past the deadline: the kernel has left
    info = ring.info()
    ring.close()
    assert info["restarts"] >= 2, info
    for i in range(4):
        for c in range(2):
            assert_bit_equal(ys[i, c, :counts[i]].cpu().numpy(), ref[c][i], f"chunk {i} channel {c}")
    assert (f.state.phiIdx, f.state.inputDeficit) == (fos[0].state.phiIdx, fos[0].state.inputDeficit)
    f.close()


def test_ring_interface_on_filters_without_a_resident_kernel(pkg, O, torch_cuda):
    """Every other filter takes the same interface as stream-ordered launches, one per chunk: FIRDecimator, FIRArbitrary, a
    rational filter outside the resident kernel's shapes."""
    torch = torch_cuda
    rng = np.random.default_rng(8)
    for ratio, h, tx in ((Fraction(1, 4), pkg.firdes(128, 0.125, beta=7.0).astype(np.float32), np.complex64),
                         (float(np.pi / 3), (pkg.firdes(32 * 12, 0.45 / 32, beta=7.0) * 32).astype(np.float64), np.float64),
                         (Fraction(3, 5), rng.standard_normal(3 * 11).astype(np.float32), np.float32)):
        x = _signal(rng, tx, 2, 40_000)
        cuts = [0, 9_001, 9_003, 25_000, 40_000]
        mk = (lambda: O.FIRFilter(h, ratio, 32, tx=tx)) if isinstance(ratio, float) else (lambda: O.FIRFilter(h, ratio, tx=tx))
        fos = [mk() for _ in range(2)]
        f = pkg.FIRFilter(h, ratio, 32, device=0).bind(tx, 2)
        xd = torch.from_numpy(x).cuda()
        ys = torch.zeros((4, 2, f.outputlength_bound(16_000)), dtype=_tdtype(torch, f.output_dtype), device="cuda")
        torch.cuda.synchronize()
        with f.open_ring() as ring:
            assert not ring.info()["resident"]
            got = [ring.push(ys[i], xd[:, cuts[i]:cuts[i + 1]]) for i in range(4)]
            ring.wait(got[1][1])
            ring.drain()
        for i, (cnt, _) in enumerate(got):
            for c in range(2):
                assert_bit_equal(ys[i, c, :cnt].cpu().numpy(), fos[c].filt(x[c, cuts[i]:cuts[i + 1]]), f"{ratio} chunk {i} channel {c}")
        f.close()


def test_two_rings_at_once_one_resident(pkg, O, torch_cuda, monkeypatch):
    """One resident consumer per device (its workgroups must all be on the chip at once): a ring opened while another is resident runs
    as stream-ordered launches behind the same interface; both streams come out bit-equal to the oracle, interleaved pushes included."""
    torch = torch_cuda
    monkeypatch.setenv("MRHIP_RING_IDLE_MS", "500")
    ratio = Fraction(147, 160)
    h = pkg.firdes(24 * 147, 0.5 / 147, beta=7.8562).astype(np.float32)
    rng = np.random.default_rng(12)
    xs = [_signal(rng, np.float32, 1, 120_000) for _ in range(2)]
    fs = [pkg.FIRFilter(h, ratio, device=0).bind(np.float32, 1) for _ in range(2)]
    fos = [O.FIRFilter(h, ratio, tx=np.float32) for _ in range(2)]
    xd = [torch.from_numpy(x).cuda() for x in xs]
    ys = [torch.zeros((4, 1, fs[0].outputlength_bound(30_000)), dtype=torch.float32, device="cuda") for _ in range(2)]
    torch.cuda.current_stream().synchronize()
    r0 = fs[0].open_ring()
    r1 = fs[1].open_ring()
    assert r0.info()["resident"] and not r1.info()["resident"]
    got = [[], []]
    for i in range(4):
        for k, ring in enumerate((r0, r1)):
            got[k].append(ring.push(ys[k][i], xd[k][:, i * 30_000:(i + 1) * 30_000]))
    r1.drain(); r0.drain()
    r0.close()
    r2 = fs[0].open_ring()                                  # the place is free again
    assert r2.info()["resident"]
    r2.close(); r1.close()
    for k in range(2):
        for i in range(4):
            cnt = got[k][i][0]
            assert_bit_equal(ys[k][i, 0, :cnt].cpu().numpy(), fos[k].filt(xs[k][0, i * 30_000:(i + 1) * 30_000]), f"ring {k} chunk {i}")
        fs[k].close()


def test_ring_errors(pkg, torch_cuda):
    torch = torch_cuda
    h = pkg.firdes(24 * 147, 0.5 / 147, beta=7.8562).astype(np.float32)
    f = pkg.FIRFilter(h, Fraction(147, 160), device=0).bind(np.float32, 1)
    x = torch.rand((1, 10_000), device="cuda")
    y = torch.zeros((1, 100), device="cuda")
    torch.cuda.synchronize()
    ring = f.open_ring()
    with pytest.raises(pkg.MultirateHIPError, match="buffer is too small"):
        ring.push(y, x)                                     # Filters.jl:550
    with pytest.raises(pkg.MultirateHIPError, match="feeds a ring"):
        f.filt(x)
    with pytest.raises(pkg.MultirateHIPError, match="already feeds a ring"):
        f.open_ring()
    with pytest.raises(pkg.MultirateHIPError, match="no such chunk"):
        ring.wait(0)
    ring.close()
    assert f.filt(x).shape[-1] == 9188                      # the refused push left the stream where it was
    f.close()


def test_filter_destroyed_while_it_feeds_a_ring(pkg, O, torch_cuda):
    """FIRFilter.close() before ChunkRing.close() (a garbage collector runs the two finalizers in any order) used to free the
    taps and the history under the resident kernel and leave the ring a dangling filter pointer.  mrhip_destroy now shuts the ring down
    first: what was pushed is complete, the ring's entry points fail, its close only frees the handle."""
    torch = torch_cuda
    h = pkg.firdes(24 * 147, 0.5 / 147, beta=7.8562).astype(np.float32)
    x = torch.rand((1, 200_000), device="cuda")
    torch.cuda.synchronize()
    for resident in (True, False):
        f = (pkg.FIRFilter(h, Fraction(147, 160), device=0) if resident else pkg.FIRFilter(h[:128].copy(), Fraction(1, 4), device=0)).bind(np.float32, 1)
        y = torch.zeros((1, f.outputlength(200_000) + 8), device="cuda")
        ring = f.open_ring()
        info = ring.info()
        assert info["resident"] == resident and info["xcds"] in (0, 8) and info["shrunk"] >= 0 and info["workgroups"] >= 0
        cnt, _ = ring.push(y, x)
        href = (h if resident else h[:128].copy())
        ref = O.FIRFilter(href, Fraction(147, 160) if resident else Fraction(1, 4), tx=np.float32).filt(x[0].cpu().numpy())
        f.close()                                               # the filter goes first
        assert np.array_equal(y[0, :cnt].cpu().numpy().view(np.uint32), ref.view(np.uint32))      # the pushed chunk was completed
        with pytest.raises(pkg.MultirateHIPError, match="has been destroyed"):
            ring.push(y, x)
        with pytest.raises(pkg.MultirateHIPError, match="has been destroyed"):
            ring.drain()
        ring.close()                                            # frees the handle, nothing else


def test_ring_randomised_stress_short():
    """scripts/stress_ring.py (random eligible shapes, types, channel counts, ragged chunkings against the oracle's chunk loop) for a few
    seconds, under both completion protocols -- the script that found the launch that did not fit the chip (profiles/r05/experiments.md S)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ({}, {"MRHIP_RING_FLUSH_MIN_MB": "0"}):
        env = dict(os.environ, MRHIP_RING_IDLE_MS="500", **extra)
        p = subprocess.run([sys.executable, os.path.join(root, "scripts", "stress_ring.py"), "--cases", "30", "--seconds", "40", "--seed", "101"],
                           env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, (extra, p.stdout[-800:], p.stderr[-800:])
        assert "mismatches 0" in p.stdout, p.stdout[-400:]
