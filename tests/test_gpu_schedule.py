"""GPU tests of the device-evaluated phase schedule of FIRArbitrary / FIRFarrow (csrc/kernels_schedule.hip,
csrc/arb_schedule.hip; reference: update(), src/Filters.jl:663-673 and :780-792).

The filter's outputs depend on every schedule entry (input index, phase, alpha of every output), so bit-equal outputs
and end state against the oracle -- or, at sizes the oracle would take minutes for, against the library's own serial
host loop (MRHIP_SCHED_DEVICE=0; itself checked against the oracle in test_gpu_parity.py) -- is equality of schedules."""
import math

import numpy as np
import pytest

from conftest import assert_bit_equal

pytestmark = pytest.mark.gpu

RATES = [math.pi / 3, 1.0, 0.5, 2.0, 1 / 3, 0.999999, 1.000001, 7.77, 40.0, 0.26, 0.2499, 0.01, 1 / 300, 0.0333, 3.999,
         1 / 2.123456789, 48000 / 44100, 3.0, 11 / 7, 56 / 37, 2.5]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


def _small_pieces(monkeypatch, prefix=4096, pmax=8192, device_min=6000):
    monkeypatch.setenv("MRHIP_SCHED_PREFIX", str(prefix))
    monkeypatch.setenv("MRHIP_SCHED_PMAX", str(pmax))
    monkeypatch.setenv("MRHIP_SCHED_DEVICE_MIN", str(device_min))


def _chunks(f, x, sizes):
    outs, pos = [], 0
    for s in sizes:
        outs.append(f.filt(x[..., pos:pos + s]))
        pos += s
    return outs


def test_device_schedule_vs_oracle_many_rates(pkg, O, torch_cuda, monkeypatch):
    """15 + 6 rates x 5 Nphi, three calls per stream (so the second and third start from a carried state), pieces small
    enough that every call spans several of them: outputs and end state bit for bit == oracle."""
    torch = torch_cuda
    _small_pieces(monkeypatch)
    rng = np.random.default_rng(321)
    seen = dict(device=0, periodic=0, host=0, fallback=0)
    for rate in RATES:
        for Nphi in (32, 10, 7, 1, 64):
            T = 3
            h = rng.standard_normal(T * Nphi).astype(np.float32)
            n_out_target = 70_000
            n = max(int(n_out_target / rate), 400)
            x = rng.standard_normal(n).astype(np.float32)
            sizes = [n // 2, 1, n - n // 2 - 1]
            f = pkg.FIRFilter(h, rate, Nphi)
            y = np.concatenate([o.cpu().numpy() for o in _chunks(f, torch.from_numpy(x).cuda(), sizes)])
            fo = O.FIRFilter(h, rate, Nphi, tx=np.float32)
            yo = np.concatenate(_chunks(fo, x, sizes))
            assert_bit_equal(y, yo, f"rate={rate} Nphi={Nphi}")
            assert f.state.phiAccumulator == fo.state.phiAccumulator and f.state.inputDeficit == fo.state.inputDeficit, (rate, Nphi)
            info = f.schedule_info()
            seen["device"] += info["device_pieces"] > 0
            seen["periodic"] += info["periodic_steps"] > 0
            seen["host"] += info["host_steps"] > 0
            seen["fallback"] += info["fallback_pieces"]
            f.close()
    assert seen["device"] >= 30 and seen["periodic"] >= 5, seen      # both device paths really ran
    print("schedule paths over the sweep:", seen)


def test_farrow_uses_the_same_schedule(pkg, O, torch_cuda, monkeypatch):
    torch = torch_cuda
    _small_pieces(monkeypatch)
    rng = np.random.default_rng(9)
    for rate in (math.pi / 3, 1.3, 3.0):
        h = (pkg.firdes(32 * 8, 0.45 / 32, beta=7.8562) * 32).astype(np.float64)
        x = rng.random(int(60_000 / rate))
        f = pkg.FIRFilter(h, rate, 32, 4).bind(np.float64)
        fo = O.FIRFilter(h, rate, 32, tx=np.float64, polyorder=4, pnfb=f.pnfb())
        y = f.filt(torch.from_numpy(x).cuda()).cpu().numpy()
        yo = fo.filt(x)
        assert_bit_equal(y, yo, f"farrow rate={rate}")
        assert f.state.phiAccumulator == fo.state.phiAccumulator
        info = f.schedule_info()
        assert info["device_pieces"] > 0 or info["periodic_steps"] > 0
        f.close()


def test_falsified_table_is_caught_and_the_piece_redone_by_the_host(pkg, O, torch_cuda, monkeypatch):
    """MRHIP_SCHED_CORRUPT=k falsifies the candidate tables of the k-th device piece: its verification must fail, the
    host's serial loop must redo exactly that piece, and the result must still be exact."""
    torch = torch_cuda
    _small_pieces(monkeypatch)
    monkeypatch.setenv("MRHIP_SCHED_CORRUPT", "1")
    rng = np.random.default_rng(3)
    h = rng.standard_normal(3 * 32).astype(np.float32)
    x = rng.standard_normal(60_000).astype(np.float32)
    rate = math.pi / 3
    f = pkg.FIRFilter(h, rate, 32)
    y = f.filt(torch.from_numpy(x).cuda()).cpu().numpy()
    yo = O.FIRFilter(h, rate, 32, tx=np.float32).filt(x)
    assert_bit_equal(y, yo, "after a forced fallback")
    info = f.schedule_info()
    assert info["fallback_pieces"] == 1 and info["device_pieces"] >= 3, info
    f.close()


def test_cycle_shortcut_off_the_tables_still_agree(pkg, O, torch_cuda, monkeypatch):
    """Rates whose accumulator cycles exactly (phase ON the wrap / binade thresholds every period) through the table
    path: exact either way -- by verification, or by the host loop where a piece does not verify."""
    torch = torch_cuda
    _small_pieces(monkeypatch)
    monkeypatch.setenv("MRHIP_SCHED_CYCLE", "0")
    rng = np.random.default_rng(4)
    for rate, Nphi in ((3.0, 32), (1.0, 32), (56 / 37, 16), (0.75, 32), (2.0, 7)):
        h = rng.standard_normal(3 * Nphi).astype(np.float32)
        x = rng.standard_normal(int(50_000 / rate)).astype(np.float32)
        f = pkg.FIRFilter(h, rate, Nphi)
        y = f.filt(torch.from_numpy(x).cuda()).cpu().numpy()
        yo = O.FIRFilter(h, rate, Nphi, tx=np.float32).filt(x)
        assert_bit_equal(y, yo, f"rate={rate} Nphi={Nphi}")
        assert f.schedule_info()["periodic_steps"] == 0
        f.close()


def test_set_state_and_reset_drop_the_cycle_and_the_drift_estimate(pkg, O, torch_cuda, monkeypatch):
    torch = torch_cuda
    _small_pieces(monkeypatch)
    rng = np.random.default_rng(6)
    h = rng.standard_normal(96).astype(np.float32)
    x = rng.standard_normal(30_000).astype(np.float32)
    f = pkg.FIRFilter(h, 3.0, 32)
    fo = O.FIRFilter(h, 3.0, 32, tx=np.float32)
    xd = torch.from_numpy(x).cuda()
    assert_bit_equal(f.filt(xd).cpu().numpy(), fo.filt(x), "first")
    assert f.schedule_info()["period"] == 3
    for acc in (7.25, 1.0 + 2.0 ** -40 + 2.0 ** -52):       # the second is off the grid of values the recurrence produces
        f.set_state(1, 2, acc)
        fo.set_state(1, 2, acc)
        assert_bit_equal(f.filt(xd).cpu().numpy(), fo.filt(x), f"after set_state({acc!r})")
        assert f.state.phiAccumulator == fo.state.phiAccumulator
    f.reset()
    fo.reset()
    assert_bit_equal(f.filt(xd).cpu().numpy(), fo.filt(x), "after reset")
    f.close()


@pytest.mark.parametrize("rate,n_out", [(math.pi / 3, 100_000_000), (48000 / 44100, 100_000_000), (7.77, 30_000_000),
                                        (0.26, 30_000_000), (1 / 2.123456789, 30_000_000), (11 / 7, 30_000_000), (3.0, 30_000_000)])
def test_long_runs_device_schedule_vs_host_loop(pkg, torch_cuda, monkeypatch, rate, n_out):
    """1e8 / 3e7 outputs in one call with the default piece sizes: device-evaluated schedule vs the serial host loop,
    outputs (Float32, 1 channel, 3 taps per phase: every entry matters) and end state bit for bit."""
    torch = torch_cuda
    g = torch.Generator(device="cuda").manual_seed(11)
    n = int(n_out / rate)
    x = torch.rand(n, generator=g, device="cuda", dtype=torch.float32) - 0.5
    h = np.random.default_rng(2).standard_normal(96).astype(np.float32)
    monkeypatch.setenv("MRHIP_SCHED_DEVICE", "0")
    fh = pkg.FIRFilter(h, rate, 32)
    yh = fh.filt(x)
    assert fh.schedule_info()["device_ok"] == 0
    monkeypatch.delenv("MRHIP_SCHED_DEVICE")
    fd = pkg.FIRFilter(h, rate, 32)
    yd = fd.filt(x)
    info = fd.schedule_info()
    assert info["device_pieces"] > 0 or info["periodic_steps"] > 0, info
    assert yd.shape == yh.shape and torch.equal(yd.view(torch.int32), yh.view(torch.int32)), (rate, info)
    sd, sh = fd.state, fh.state
    assert (sd.phiAccumulator, sd.inputDeficit, sd.phiIdx) == (sh.phiAccumulator, sh.inputDeficit, sh.phiIdx)
    # and a second call from the carried state (warm drift estimate: no host prefix any more)
    yd2, yh2 = fd.filt(x[: n // 7]), fh.filt(x[: n // 7])
    assert torch.equal(yd2.view(torch.int32), yh2.view(torch.int32))
    print(f"rate={rate:.6g}: {info}")
    fd.close()
    fh.close()


def test_late_group_of_the_last_piece_still_writes_its_entries(pkg, torch_cuda, monkeypatch):
    """Regression (round 3): the workgroup that holds the call's end sets `done`; a workgroup of an EARLIER group of the
    same piece that starts after that must not take it as a reason to return.  The hook delays group 0 of every piece by
    ~0.5 ms (far longer than the kernel runs)."""
    torch = torch_cuda
    rate, n_out = math.pi / 3, 1_000_000
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.rand(int(n_out / rate), generator=g, device="cuda", dtype=torch.float32) - 0.5
    h = np.random.default_rng(3).standard_normal(96).astype(np.float32)
    monkeypatch.setenv("MRHIP_SCHED_DEVICE", "0")
    fh = pkg.FIRFilter(h, rate, 32)
    yh = fh.filt(x)
    monkeypatch.delenv("MRHIP_SCHED_DEVICE")
    monkeypatch.setenv("MRHIP_SCHED_CORRUPT", "-2")
    fd = pkg.FIRFilter(h, rate, 32)
    for _ in range(3):                                           # (the first call also warms the drift estimate)
        fd.reset(); fh.reset()
        yd, yh = fd.filt(x), fh.filt(x)
        info = fd.schedule_info()
        assert info["device_pieces"] > 0 and info["fallback_pieces"] == 0, info
        assert torch.equal(yd.view(torch.int32), yh.view(torch.int32))
    fd.close(); fh.close()


@pytest.mark.parametrize("rate", [math.pi / 3, 0.26, 3.0])
def test_advance_state_over_long_stretches_uses_the_device_schedule(pkg, torch_cuda, monkeypatch, rate):
    """mrhip_advance_state for FIRArbitrary: 6e7 samples through the device-evaluated schedule (launch-sized pieces, no filter
    kernel) == the host's serial loop (MRHIP_SCHED_DEVICE=0), count and end state; and a filter entered there produces what
    the one that filtered its way there produces."""
    torch = torch_cuda
    h = np.random.default_rng(5).standard_normal(96).astype(np.float32)
    n = 60_000_000
    monkeypatch.setenv("MRHIP_SCHED_DEVICE", "0")
    fh = pkg.FIRFilter(h, rate, 32).bind(np.float32, 1)
    ch = fh.advance_state(n)
    monkeypatch.delenv("MRHIP_SCHED_DEVICE")
    fd = pkg.FIRFilter(h, rate, 32).bind(np.float32, 1)
    cd = fd.advance_state(n)
    assert cd == ch
    sd, sh = fd.state, fh.state
    assert (sd.phiAccumulator, sd.inputDeficit, sd.phiIdx) == (sh.phiAccumulator, sh.inputDeficit, sh.phiIdx)
    info = fd.schedule_info()
    assert info["device_pieces"] > 0 or info["periodic_steps"] > 0, info
    x = torch.rand(100_000, device="cuda", dtype=torch.float32) - 0.5
    assert torch.equal(fd.filt(x).view(torch.int32), fh.filt(x).view(torch.int32))
    fd.close(); fh.close()


def test_stream_of_short_calls_leaves_the_host_loop(pkg, O, torch_cuda):
    """Calls of about 21 000 outputs each -- shorter than the serial prefix that measures the drift baseline (65 536 steps): the
    prefix's measurement is kept when it reaches a call's end, so after the first few calls the stream runs on the device
    (round 4 found such streams re-running the prefix on the host for ever); outputs and end state == the oracle's loop."""
    torch = torch_cuda
    rng = np.random.default_rng(99)
    Nphi, T, rate, chunk, ncalls = 32, 5, 1 / 2.123456789, 44_987, 30
    h = rng.standard_normal(T * Nphi).astype(np.float32)
    x = rng.standard_normal(chunk * ncalls).astype(np.float32)
    f = pkg.FIRFilter(h, rate, Nphi)
    fo = O.FIRFilter(h, rate, Nphi, tx=np.float32)
    xd = torch.from_numpy(x).cuda()
    for i in range(ncalls):
        y = f.filt(xd[i * chunk:(i + 1) * chunk]).cpu().numpy()
        assert_bit_equal(y, fo.filt(x[i * chunk:(i + 1) * chunk]), f"call {i}")
    assert f.state.phiAccumulator == fo.state.phiAccumulator and f.state.inputDeficit == fo.state.inputDeficit
    info = f.schedule_info()
    total = int(chunk * ncalls * rate)
    assert info["host_steps"] < 4 * 65_536 < total // 2, info          # the prefix, not the stream
    assert info["device_pieces"] >= ncalls - 6, info
    f.close()


@pytest.mark.parametrize("fuse,win", [("0", ("4", "16")), ("1", ("4", "16")), ("0", ("1", "4")), ("1", ("2", "8"))])
def test_schedule_switches_change_nothing(pkg, O, torch_cuda, monkeypatch, fuse, win):
    """BEGIN / FINISH as launches of their own or inside the pieces' kernels (MRHIP_SCHED_FUSE), one, two or four residue
    systems of candidates per segment (MRHIP_SCHED_WIN_MULT / _MIN: rounds 3-4 ran 4 / 16): synchronous and asynchronous
    calls, several pieces per call -- outputs, counts and end state == the oracle."""
    torch = torch_cuda
    _small_pieces(monkeypatch)
    monkeypatch.setenv("MRHIP_SCHED_FUSE", fuse)
    monkeypatch.setenv("MRHIP_SCHED_WIN_MULT", win[0])
    monkeypatch.setenv("MRHIP_SCHED_WIN_MIN", win[1])
    rng = np.random.default_rng(17)
    for rate, Nphi in ((math.pi / 3, 32), (1 / 2.123456789, 32), (0.37, 10), (7.77, 7), (2.5, 32)):
        T = 4
        h = rng.standard_normal(T * Nphi).astype(np.float32)
        n = max(int(60_000 / rate), 500)
        x = rng.standard_normal(3 * n).astype(np.float32)
        xd = torch.from_numpy(x).cuda()
        f = pkg.FIRFilter(h, rate, Nphi)
        fo = O.FIRFilter(h, rate, Nphi, tx=np.float32)
        assert f.bind(np.float32, 1).schedule_info()["nwin"] == max(int(win[0]) * f.schedule_info()["ncand"], int(win[1]))
        y0 = f.filt(xd[:n]).cpu().numpy()                                  # waited for
        assert_bit_equal(y0, fo.filt(x[:n]), f"rate={rate} synchronous")
        cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
        ys = torch.zeros((2, f.outputlength_bound(n)), dtype=torch.float32, device="cuda")
        f.filt_into_async(ys[0], xd[n:2 * n], cnt[0:1])                    # nobody waits
        f.filt_into_async(ys[1], xd[2 * n:], cnt[1:2])
        f.sync_state()
        for i in range(2):
            ref = fo.filt(x[(i + 1) * n:(i + 2) * n])
            assert int(cnt[i]) == len(ref), (rate, i)
            assert_bit_equal(ys[i, :len(ref)].cpu().numpy(), ref, f"rate={rate} asynchronous {i}")
        assert f.state.phiAccumulator == fo.state.phiAccumulator and f.state.inputDeficit == fo.state.inputDeficit, rate
        f.close()


@pytest.mark.parametrize("fuse", ["1", "0"])
def test_falsified_table_in_an_asynchronous_call_is_redone_on_the_device(pkg, O, torch_cuda, monkeypatch, fuse):
    """Nobody waits for an asynchronous call, so a piece that does not verify cannot be handed to the host: the FINISH (inside
    the last piece's emit kernel, or a launch of its own) redoes the schedule serially from that piece's verified start to the
    end of the call.  MRHIP_SCHED_CORRUPT=1 falsifies piece 1 of the call: outputs, count and end state == the oracle."""
    torch = torch_cuda
    _small_pieces(monkeypatch)
    monkeypatch.setenv("MRHIP_SCHED_CORRUPT", "1")
    monkeypatch.setenv("MRHIP_SCHED_FUSE", fuse)
    rng = np.random.default_rng(4)
    h = rng.standard_normal(3 * 32).astype(np.float32)
    x = rng.standard_normal(60_000).astype(np.float32)
    rate = math.pi / 3
    f = pkg.FIRFilter(h, rate, 32).bind(np.float32, 1)
    fo = O.FIRFilter(h, rate, 32, tx=np.float32)
    xd = torch.from_numpy(x).cuda()
    y = torch.zeros(f.outputlength_bound(len(x)), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    f.filt_into_async(y, xd, cnt)
    n = f.sync_state()
    yo = fo.filt(x)
    assert n == len(yo) == int(cnt[0])
    assert_bit_equal(y[:n].cpu().numpy(), yo, "after a forced fallback on the device")
    assert f.state.phiAccumulator == fo.state.phiAccumulator and f.state.inputDeficit == fo.state.inputDeficit
    # the stream goes on from there
    x2 = rng.standard_normal(5_000).astype(np.float32)
    monkeypatch.delenv("MRHIP_SCHED_CORRUPT")
    assert_bit_equal(f.filt(torch.from_numpy(x2).cuda()).cpu().numpy(), fo.filt(x2), "the next call")
    f.close()


def test_short_host_loop_call_moves_the_cycle_position_on(pkg, torch_cuda):
    """Rate 2.5 with N𝜙 = 10 cycles (period 5).  A long synchronous call finds the cycle; a SHORT synchronous call is evaluated by the
    host's loop; the asynchronous call behind it plans from the cycle position -- which the short call must have moved on (it was
    left where it was: the plan kernel refused the call, tests/stress_random.py --async-mix seed 111).  Every mix of synchronous and
    asynchronous calls == the host loop's stream."""
    import itertools
    import os
    torch = torch_cuda
    rng = np.random.default_rng(0)
    rate, nphi, nch = 2.5, 10, 3
    sizes = [17696, 104308, 6459, 21814]
    h = rng.standard_normal(36).astype(np.float32)
    x = torch.from_numpy(rng.standard_normal((nch, sum(sizes))).astype(np.float32)).cuda()
    os.environ["MRHIP_SCHED_DEVICE"] = "0"
    try:
        g = pkg.FIRFilter(h, rate, nphi)
        ref, pos = [], 0
        for s in sizes:
            ref.append(g.filt(x[:, pos:pos + s])); pos += s
        g.close()
    finally:
        os.environ.pop("MRHIP_SCHED_DEVICE", None)
    for pat in itertools.product([0, 1], repeat=4):
        f = pkg.FIRFilter(h, rate, nphi).bind(np.float32, nch)
        cnt = torch.full((4,), -1, dtype=torch.int64, device="cuda")
        outs, pos = [], 0
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
        for i, o in enumerate(outs):
            got = o[:, :c[i]] if pat[i] else o
            assert got.shape == ref[i].shape and torch.equal(got, ref[i]), (pat, i)
        f.close()


@pytest.mark.parametrize("Nphi,rate", [(10, float(np.pi / 3)), (7, 1.7), (12, 0.37), (32, float(np.pi / 3))])
def test_mod_form_of_julia_0_3_in_the_product(pkg, O, Nphi, rate):
    """update() wraps the phase accumulator with mod(acc - 1, N𝜙) (src/Filters.jl:668); Julia's Base before 0.4 computed a float
    mod as rem(y + rem(x, y), y).  Oracle AND product run either form (mrhip_set_mod_form / oracle.set_mod_form): bit-equal outputs,
    counts and end state in both; for a power-of-two N𝜙 the two forms are the same stream."""
    import torch
    rng = np.random.default_rng(Nphi)
    h = (pkg.firdes(Nphi * 12, 0.45 / Nphi, beta=7.0) * Nphi).astype(np.float64)
    x = rng.standard_normal(300_000)
    got = {}
    for form in (False, True):
        O.set_mod_form(form)
        try:
            fo = O.FIRFilter(h, rate, Nphi, tx=np.float64)
            ref = [fo.filt(x[:100_003]), fo.filt(x[100_003:])]
            so = fo.state
        finally:
            O.set_mod_form(False)
        f = pkg.FIRFilter(h, rate, Nphi, device=0).bind(np.float64, 1)
        f.set_mod_form(form)
        xd = torch.from_numpy(x).cuda()
        y = [f.filt(xd[:100_003]).cpu().numpy(), f.filt(xd[100_003:]).cpu().numpy()]
        for i in range(2):
            assert_bit_equal(y[i], ref[i], f"N𝜙 {Nphi} rate {rate} form {int(form)} call {i}")
        st = f.state
        assert (st.inputDeficit, st.phiAccumulator) == (so.inputDeficit, so.phiAccumulator)
        info = f.schedule_info()
        if form and Nphi & (Nphi - 1):
            assert info["device_pieces"] == 0, "the older mod() form runs the host's serial schedule"
            with pytest.raises(pkg.MultirateHIPError, match="mod form 1"):
                f.filt_into_async(torch.empty(f.outputlength_bound(1000), dtype=torch.float64, device="cuda"), xd[:1000])
        got[form] = np.concatenate(y)
        got[("acc", form)] = st.phiAccumulator
        f.close()
    if Nphi & (Nphi - 1) == 0:
        assert_bit_equal(got[False], got[True], "power-of-two N𝜙: the two forms are one stream")
    elif (Nphi, rate) != (12, 0.37):
        assert got[("acc", False)] != got[("acc", True)] or got[False].shape != got[True].shape or not np.array_equal(got[False], got[True]), \
            "the forms differ for this N𝜙 (DESIGN.md 3.4)"
