"""Windowed-sinc FIR design used to generate taps for benchmarks, tests and examples.

Host-only, O(taps), off the hot path.  Follows the formulas of the reference's
src/FIRDesign.jl:18-95 (``kaiserlength`` :18-33, ``firprototype`` LOWPASS :52, ``firdes`` :76-95).
The Kaiser window takes beta directly (as the reference's in-tree src/Window.jl:53-58 does); the
window function the reference actually calls lives in the un-vendored DSP.jl, so tap values are
NOT claimed to be identical to a historical Multirate.jl run -- the same taps are always fed to
both the engine and the oracle, which is what parity needs.
"""
from __future__ import annotations

import math

import numpy as np


def kaiserlength(transition: float, attenuation: float = 60.0, samplerate: float = 1.0):
    """(numtaps, beta) for a Kaiser-window design, src/FIRDesign.jl:18-33."""
    transition = transition / samplerate
    numtaps = int(math.ceil((attenuation - 7.95) / (2 * math.pi * 2.285 * transition)))
    if attenuation > 50:
        beta = 0.1102 * (attenuation - 8.7)
    elif attenuation >= 21:
        beta = 0.5842 * (attenuation - 21) ** 0.4 + 0.07886 * (attenuation - 21)
    else:
        beta = 0.0
    return numtaps, beta


def firdes(numtaps: int, cutoff: float, window="kaiser", *, samplerate: float = 1.0, beta: float = 6.75,
           dtype=np.float64) -> np.ndarray:
    """Low-pass windowed-sinc taps: 2F sinc(2F (n - M/2)) * window, src/FIRDesign.jl:52,76-88."""
    F = cutoff / samplerate
    M = numtaps - 1
    n = np.arange(numtaps, dtype=np.float64)
    proto = 2.0 * F * np.sinc(2.0 * F * (n - M / 2.0))
    if window == "kaiser":
        w = np.kaiser(numtaps, beta)
    elif callable(window):
        w = window(numtaps)
    else:
        raise ValueError("window must be 'kaiser' or a callable")
    return (proto * w).astype(dtype)
