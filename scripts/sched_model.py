"""Numpy model of the DEVICE-side exact phase schedule for FIRArbitrary / FIRFarrow.

`csrc/kernels_schedule.hip` follows this file step by step; `tests/test_sched_model.py` runs it on the CPU against the
plain serial recurrence.  (Test infrastructure and design record: the product never imports it.)

The reference's update() (src/Filters.jl:663-673, :780-792) is a serial Float64 recurrence

    a1 = fl(acc + delta);  if a1 > N: xIdx += floor(fl((a1-1)/N)); acc = mod(a1-1, N) + 1   else acc = a1

whose roundings the outputs depend on, so there is no closed form.  It can still be evaluated in parallel, EXACTLY:

1. Candidates.  Every value the recurrence produces is a multiple of Umin = ulp(fl(1 + delta)): a1 >= fl(1 + delta),
   and the wrap subtracts an integer exactly.  A piece of the schedule is cut into segments of LSEG steps.  For every
   segment an *anchor* predicts where it starts (closed form of the un-rounded recurrence from the piece's true start
   state, double-double, plus the drift per step measured so far: the roundings make the true position drift away from
   the un-rounded one linearly).  The segment is then run, in real Float64 arithmetic, from each of the NWIN values of
   the Umin grid around the anchor -- one of them is the true start unless the prediction is off by more than NWIN/2
   grid steps.  That gives a table  candidate -> (end value, xIdx advance).
2. Equivariance, for starts outside the window.  Every rounding is to a multiple of ulp(a1) <= Utop = ulp(fl(N+1+delta)),
   ties to even, so for a shift d that is a multiple of G = 2*Utop:  step(acc + d) == step(acc) + d  as long as both
   sums fall in the same binade and on the same side of the wrap thresholds.  A start T outside the window is therefore
   looked up as the window's candidate congruent to T modulo G (NWIN >= G/Umin) plus the shift.  NWIN = G/Umin (at least
   4) is therefore enough -- one residue system; rounds 3-4 ran four of them (at least 16 values) so that most starts
   needed no shift, which quadrupled the tables' work for no fewer failed pieces.
3. Chain.  The end value of candidate c of segment s is itself (candidate c', shift) of segment s+1, so a segment is a
   map  c -> (c', shift, advance)  on a finite set with additive shift/advance: maps compose, and the true start of
   every segment follows from the piece's true start by composing tables (hierarchically on the device).
4. Verify.  Every segment is re-run from the start the chain gave it, emitting the schedule, and must end exactly where
   the next segment was told to start.  If all checks hold, the emitted schedule IS the serial recurrence (induction
   from the piece's true start state); if one fails (a threshold decision inside the equivariance argument flipped, or
   a start was off the grid) the piece is recomputed by the host's serial loop.  Nothing is ever accepted unverified.
"""
from __future__ import annotations

import math

import numpy as np

LSEG = 64


def step_serial(acc: float, delta: float, N: float):
    """One update() as the reference writes it (Filters.jl:663-669): returns (acc', xIdx advance)."""
    a1 = acc + delta
    if a1 > N:
        am1 = a1 - 1.0
        dx = math.floor(am1 / N)          # quotient rounded BEFORE the floor, as in the reference
        return math.fmod(am1, N) + 1.0, dx
    return a1, 0


def serial(acc0: float, x0: int, delta: float, Nphi: int, nsteps: int):
    """(xIdx_k, acc_k) for k < nsteps and the state after nsteps updates -- the checker."""
    N = float(Nphi)
    xs = np.empty(nsteps, np.int64)
    accs = np.empty(nsteps, np.float64)
    acc, x = acc0, x0
    for k in range(nsteps):
        xs[k] = x
        accs[k] = acc
        acc, dx = step_serial(acc, delta, N)
        x += dx
    return xs, accs, acc, x


def vstep(acc, delta, N):
    """update() on float64 arrays, in the form the device uses: mod(a1-1, N) == (a1-1) - k*N exactly, with
    k = floor(fl((a1-1)/N)) corrected by one when the rounded quotient overshot (non-power-of-two N only)."""
    a1 = acc + delta
    am1 = a1 - 1.0
    dx = np.floor(am1 / N)
    r = am1 - dx * N                      # exact: a multiple of ulp(a1) no larger than a1
    r = np.where(r < 0.0, r + N, r)
    wrap = a1 > N
    return np.where(wrap, r + 1.0, a1), np.where(wrap, dx, 0.0).astype(np.int64)


def _split(a):
    c = 134217729.0 * a
    hi = c - (c - a)
    return hi, a - hi


def _two_prod_err(a, b, p):               # fma(a, b, -p) without an fma
    ah, al = _split(a)
    bh, bl = _split(np.float64(b))
    return ((ah * bh - p) + ah * bl + al * bh) + al * bl


def wrapd(d, N):
    """Signed distance on the phase circle of circumference N."""
    return np.where(d > N / 2, d - N, np.where(d < -N / 2, d + N, d))


class Plan:
    """Constants of the parallel evaluation for one (delta, Nphi)."""

    def __init__(self, delta: float, Nphi: int, max_win: int = 64, win_mult: int = 1):
        self.delta = float(delta)
        self.N = float(Nphi)
        self.umin = math.ulp(1.0 + delta)
        self.utop = math.ulp(self.N + 1.0 + delta)
        self.G = 2.0 * self.utop
        self.ncand = int(round(self.G / self.umin))
        self.nwin = max(win_mult * self.ncand, 4)     # one residue system modulo G (rounds 3-4: four, at least 16 values)
        self.ok = self.nwin <= max_win and delta < 2.0 ** 40 and self.umin <= 2.0 ** -20

    def anchor(self, acc_p: float, k, slope: float = 0.0):
        """Predicted phase after k steps from acc_p, as a legal state in [1, N+1)."""
        kf = np.asarray(k, dtype=np.float64)
        hi = kf * self.delta
        lo = _two_prod_err(kf, self.delta, hi)
        w = np.floor(((acc_p - 1.0) + hi) / self.N)
        r = ((acc_p - 1.0) + (hi - w * self.N)) + (lo + slope * kf)
        r = np.where(r < 0.0, r + self.N, np.where(r >= self.N, r - self.N, r))
        return r + 1.0

    def candidates(self, anc):
        """[..., nwin] start values: the Umin grid around the anchor, folded onto legal states."""
        base = np.floor((anc - (self.nwin // 2) * self.umin) / self.umin) * self.umin
        off = np.arange(self.nwin, dtype=np.float64) * self.umin
        C = base[..., None] + off
        # a window that straddles the wrap: fold FIRST, then add the offset, so that the part near 1.0 keeps the
        # fine grid (base - N is exact; base + offset near N + 1 is not representable for every grid point)
        lo = (base - self.N)[..., None] + off
        hi = (base + self.N)[..., None] + off
        return base, np.where(C >= self.N + 1.0, lo, np.where(C < 1.0, hi, C))

    def locate(self, T, base, C):
        """(candidate, shift, ok) with  C[candidate] + shift == T."""
        d = T - base
        if d > 0.5 * self.N:                           # the same point of the phase circle, every step exact
            d = (T - self.N) - base
        elif d < -0.5 * self.N:
            d = T - (base - self.N)
        cu = d / self.umin
        if cu != math.floor(cu):
            return 0, 0.0, False                       # off the Umin grid (a state the recurrence cannot produce)
        cu = int(cu)
        if 0 <= cu < self.nwin:
            c, shift = cu, 0.0
        else:
            m = math.floor(d / self.G)
            c, shift = cu - m * self.ncand, m * self.G
        return c, shift, bool(C[c] + shift == T)


def piece(plan: Plan, acc_p: float, x_p: int, nsteps: int, slope: float = 0.0, corrupt_segment: int = -1):
    """One piece of nsteps (a multiple of LSEG) steps from the true state (acc_p, x_p).
    Returns dict(xs, accs, acc_end, x_end, ok, shifted)."""
    assert nsteps % LSEG == 0
    nseg = nsteps // LSEG
    anc = plan.anchor(acc_p, np.arange(nseg + 1, dtype=np.int64) * LSEG, slope)
    base, C = plan.candidates(anc)
    # K1: candidate runs
    acc = C[:nseg].copy()
    W = np.zeros((nseg, plan.nwin), np.int64)
    for _ in range(LSEG):
        acc, dx = vstep(acc, plan.delta, plan.N)
        W += dx
    R = acc
    if corrupt_segment >= 0:
        R[corrupt_segment] += plan.G                  # test hook: a wrong table must be caught by the verification
    # K1 (cont.) + K2: chain.  (The device composes the tables hierarchically; the result is the same walk.)
    T = np.empty(nseg + 1, np.float64)
    X = np.empty(nseg + 1, np.int64)
    c, S, ok = plan.locate(acc_p, base[0], C[0])
    T[0], X[0] = acc_p, x_p
    shifted = 0
    for s in range(nseg):
        if not ok:
            break
        T[s + 1] = R[s, c] + S
        X[s + 1] = X[s] + W[s, c]
        c2, sh, ok = plan.locate(float(R[s, c]), base[s + 1], C[s + 1])
        c, S = c2, S + sh
        shifted += S != 0.0
        ok = ok and bool(C[s + 1, c] + S == T[s + 1])
    if not ok:
        return dict(ok=False, shifted=shifted)
    # K3: run from the true starts, emit, verify
    acc = T[:nseg].copy()
    x = X[:nseg].copy()
    xs = np.empty((nseg, LSEG), np.int64)
    accs = np.empty((nseg, LSEG), np.float64)
    for i in range(LSEG):
        xs[:, i] = x
        accs[:, i] = acc
        acc, dx = vstep(acc, plan.delta, plan.N)
        x = x + dx
    ok = bool(np.all(acc == T[1:]) and np.all(x == X[1:]))
    return dict(ok=ok, shifted=shifted, xs=xs.reshape(-1), accs=accs.reshape(-1), acc_end=float(T[nseg]), x_end=int(X[nseg]))


def find_cycle(accs: np.ndarray, acc_end: float):
    """The recurrence is a deterministic map of acc alone: if the state after the prefix equals (bit for bit) a state
    inside it, the schedule is periodic from there on.  Returns the period (0: none found)."""
    hit = np.flatnonzero(accs == acc_end)
    return int(len(accs) - hit[-1]) if len(hit) else 0


def periodic(accs: np.ndarray, xs: np.ndarray, x_end: int, Q: int, nsteps: int):
    """Closed form of nsteps further steps after a prefix that ends on a cycle of period Q."""
    n = len(accs)
    j = np.arange(nsteps, dtype=np.int64)
    r, cyc = j % Q, j // Q
    XQ = x_end - int(xs[n - Q])
    return xs[n - Q + r] + (cyc + 1) * XQ, accs[n - Q + r]


def stream(delta: float, Nphi: int, total_steps: int, acc0: float = 1.0, x0: int = 1, prefix: int = 1 << 16,
           pmax: int = 1 << 18, corrupt=None, use_cycle: bool = True):
    """The host's orchestration: a serial prefix (which also measures the drift per step and looks for a cycle), then
    either the closed form of a periodic schedule, or pieces whose size doubles up to pmax; a piece that fails
    verification is redone serially.  Returns (xs, accs, acc, x, stats)."""
    plan = Plan(delta, Nphi)
    N = float(Nphi)
    xs_all, accs_all = [], []
    n0 = min(prefix, total_steps)
    xs, accs, acc, x = serial(acc0, x0, delta, Nphi, n0)
    xs_all.append(xs)
    accs_all.append(accs)
    done = n0
    drift = float(wrapd(acc - plan.anchor(acc0, done), N))
    stats = dict(pieces=0, failed=0, shifted=0, plan_ok=plan.ok, period=0)
    Q = find_cycle(accs, acc) if use_cycle and done < total_steps else 0
    if Q:
        stats["period"] = Q
        m = total_steps - done
        xs2, accs2 = periodic(accs, xs, x, Q, m + 1)
        return np.concatenate([xs, xs2[:m]]), np.concatenate([accs, accs2[:m]]), float(accs2[m]), int(xs2[m]), stats
    while done < total_steps:
        P = min(pmax, 16 * done, total_steps - done) // LSEG * LSEG      # a piece up to 16x the drift baseline behind it
        if P == 0 or not plan.ok:
            xs, accs, acc, x = serial(acc, x, delta, Nphi, total_steps - done)
            xs_all.append(xs)
            accs_all.append(accs)
            break
        r = piece(plan, acc, x, P, drift / done, corrupt_segment=corrupt(stats["pieces"]) if corrupt else -1)
        stats["pieces"] += 1
        stats["shifted"] += r["shifted"]
        if r["ok"]:
            xs, accs, acc2, x2 = r["xs"], r["accs"], r["acc_end"], r["x_end"]
        else:
            stats["failed"] += 1
            xs, accs, acc2, x2 = serial(acc, x, delta, Nphi, P)
        drift += float(wrapd(acc2 - plan.anchor(acc, P), N))
        acc, x = acc2, x2
        done += P
        xs_all.append(xs)
        accs_all.append(accs)
    return np.concatenate(xs_all), np.concatenate(accs_all), acc, x, stats
