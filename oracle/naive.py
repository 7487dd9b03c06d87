"""Independent naive models (TEST INFRASTRUCTURE ONLY).

These restate the *reference's own test oracles*, not its polyphase code, so they
cross-check oracle/multirate_oracle.c from a different direction:

* ``naive_rational``  -- zero-stuff by L, FIR with h, keep every M-th sample starting at the
  first: src/NaiveResamplers.jl:5-18 and the recipes in test/runtests.jl:60 (single-rate),
  :123-124 (decimation), :190-194 (interpolation), :270-278 (rational).
* ``naive_arbitrary`` -- interpolate by Nphi with the naive model, then two-neighbour linear
  interpolation with a fractional stride: src/NaiveResamplers.jl:24-49.

Everything is evaluated in extended precision (numpy longdouble) so the result can also
serve as the "exact" value for ULP histograms.
"""
from __future__ import annotations

import math

import numpy as np


def _conv_full_ld(h: np.ndarray, x: np.ndarray) -> np.ndarray:
    """Base.filt(h, 1.0, x): causal FIR, output length == len(x), long double accumulate."""
    h = np.asarray(h, dtype=np.longdouble)
    if np.iscomplexobj(x):
        re = np.convolve(np.asarray(x.real, dtype=np.longdouble), h)[: len(x)]
        im = np.convolve(np.asarray(x.imag, dtype=np.longdouble), h)[: len(x)]
        return re + 1j * im.astype(np.clongdouble)
    return np.convolve(np.asarray(x, dtype=np.longdouble), h)[: len(x)]


def naive_rational(h, x, L: int = 1, M: int = 1) -> np.ndarray:
    """src/NaiveResamplers.jl:5-18 (long double)."""
    x = np.asarray(x)
    cplx = np.iscomplexobj(x)
    stuffed = np.zeros(len(x) * L, dtype=np.clongdouble if cplx else np.longdouble)
    stuffed[::L] = x
    y = _conv_full_ld(h, stuffed)
    return y[::M]


def naive_arbitrary(h, x, rate: float, numfilters: int = 32) -> np.ndarray:
    """src/NaiveResamplers.jl:24-49.  Returned in long double; length as the reference's."""
    xi = naive_rational(h, x, numfilters, 1)
    xlen = len(xi)
    ylen = int(math.ceil(xlen * rate))
    y = np.zeros(ylen + 1, dtype=xi.dtype)
    yidx, xidx, alpha = 0, 1, 0.0
    frac, stride = math.modf(numfilters / rate)
    stride = int(stride)
    while xidx < xlen:
        lo = xi[xidx - 1]
        up = xi[xidx]
        y[yidx] = lo + np.longdouble(alpha) * (up - lo)
        yidx += 1
        alpha += frac
        xidx += int(math.floor(alpha)) + stride
        alpha = math.fmod(alpha, 1.0)
    return y[:yidx]
