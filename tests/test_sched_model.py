"""CPU tests of the numpy model of the device-side exact phase schedule (scripts/sched_model.py, followed step by step
by csrc/kernels_schedule.hip): the parallel evaluation must reproduce the serial recurrence of
update(::FIRArbitrary) / update(::FIRFarrow) (src/Filters.jl:663-673, :780-792) bit for bit, and a falsified table must
be caught by the verification (never silently used)."""
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import sched_model as sm  # noqa: E402


def _check(delta, Nphi, total, **kw):
    xs, accs, acc, x, st = sm.stream(delta, Nphi, total, **kw)
    xs_s, accs_s, acc_s, x_s = sm.serial(kw.get("acc0", 1.0), kw.get("x0", 1), delta, Nphi, total)
    assert np.array_equal(xs, xs_s), (delta, Nphi)
    assert np.array_equal(accs.view(np.uint64), accs_s.view(np.uint64)), (delta, Nphi)
    assert acc == acc_s and x == x_s, (delta, Nphi)
    return st


def test_vectorised_step_equals_the_reference_expressions():
    """vstep (the device's form: exact subtraction of k*N, overshoot correction) == step_serial (fmod, as written)."""
    rng = np.random.default_rng(0)
    for Nphi in (32, 7, 10, 48, 1, 33):
        N = float(Nphi)
        for delta in (Nphi / (math.pi / 3), 0.3, 1.7 * Nphi, 300.0, Nphi / 3.0):
            acc = 1.0 + rng.random(4000) * N
            # values next to the wrap thresholds, where the rounded quotient can overshoot
            acc[:200] = np.nextafter(N + 1.0 - delta + np.arange(200) // 20 * N, -np.inf) - delta * 0 - rng.integers(0, 3, 200) * math.ulp(N)
            acc = np.clip(acc, 1.0, np.nextafter(N + 1.0, 0))
            a2, dx = sm.vstep(acc, delta, N)
            for i in range(len(acc)):
                r, d = sm.step_serial(float(acc[i]), delta, N)
                assert r == a2[i] and d == dx[i], (Nphi, delta, acc[i])


@pytest.mark.parametrize("rate,Nphi", [(math.pi / 3, 32), (48000 / 44100, 32), (1 / 2.123456789, 32), (1.0, 32), (3.0, 32),
                                       (11 / 7, 32), (56 / 37, 16), (7.77, 7), (2.5, 32), (10.0, 7), (0.3, 48), (1 / 300, 10),
                                       (math.e, 33), (5.0, 48)])
def test_parallel_schedule_equals_serial_recurrence(rate, Nphi):
    st = _check(Nphi / rate, Nphi, 4096 + 3 * 16384 + 640, prefix=4096, pmax=16384)
    assert st["failed"] == 0      # (a failed verification would still be exact -- the serial loop redoes the piece)


def test_tables_path_without_cycle_shortcut():
    """Rates whose accumulator falls into an exact cycle normally take the closed form; the table path must handle
    them too (exact hits of the wrap / binade thresholds at every period)."""
    for rate, Nphi in ((3.0, 32), (1.0, 32), (2.0, 7), (56 / 37, 16), (0.75, 32)):
        _check(Nphi / rate, Nphi, 4096 + 2 * 8192, prefix=4096, pmax=8192, use_cycle=False)


def test_random_rates_and_start_states():
    rng = np.random.default_rng(5)
    periodic = 0
    for i in range(60):
        kind = i % 4
        rate = (float(rng.uniform(0.05, 40)), float(rng.integers(1, 64) / rng.integers(1, 64)),
                float(round(rng.uniform(0.1, 20), int(rng.integers(1, 4)))), float(rng.integers(1, 48)) + float(rng.random()))[kind]
        Nphi = int(rng.choice([32, 32, 64, 16, 10, 7, 48, 3, 33, 1]))
        # a start state as a stream leaves it: some steps into the recurrence
        _, _, acc0, x0 = sm.serial(1.0, 1, Nphi / rate, Nphi, int(rng.integers(0, 500)))
        st = _check(Nphi / rate, Nphi, 2048 + 2 * 8192 + 100, acc0=acc0, x0=int(rng.integers(1, 9)), prefix=2048, pmax=8192)
        periodic += st["period"] > 0
    assert periodic > 0           # the sweep meets both evaluation paths


def test_falsified_table_is_caught_and_redone_serially():
    delta, Nphi = 32 / (math.pi / 3), 32
    st = _check(delta, Nphi, 4096 + 2 * 8192, prefix=4096, pmax=8192, corrupt=lambda piece: 5 if piece == 1 else -1)
    assert st["failed"] == 1


def test_periodic_closed_form():
    xs, accs, acc, x = sm.serial(1.0, 1, 32 / 3.0, 32, 300)
    Q = sm.find_cycle(accs, acc)
    assert Q == 3
    xs2, accs2 = sm.periodic(accs, xs, x, Q, 1000)
    xs_s, accs_s, _, _ = sm.serial(1.0, 1, 32 / 3.0, 32, 1300)
    assert np.array_equal(xs2, xs_s[300:]) and np.array_equal(accs2, accs_s[300:])
