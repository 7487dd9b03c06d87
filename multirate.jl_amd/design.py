"""Windowed-sinc FIR design: host-side mirror of the reference's src/FIRDesign.jl:7-95
(``FIRResponse`` enum :7, ``kaiserlength`` :18-33, ``firprototype`` :47-66, both ``firdes`` methods :76-95).

Host-only, O(taps), off the hot path: it lets a ``FIRFilter``-level user stay inside this
package.  The prototype formulas are the reference's, evaluated in Float64.  The Kaiser window takes
beta directly (as the reference's in-tree src/Window.jl:53-58 does); the window function the reference
actually calls lives in the un-vendored DSP.jl of 2014, so tap VALUES are not claimed to be bit-identical
to a historical Multirate.jl run ("parity unpinned") -- the same taps are always fed to both
the engine and the oracle, which is what parity of the hot path needs.
"""
from __future__ import annotations

import math
from typing import Callable, Sequence, Union

import numpy as np

# @enum( FIRResponse, LOWPASS, BANDPASS, HIGHPASS, BANDSTOP ), src/FIRDesign.jl:7 (values 0..3, src/enum.jl:11-13)
LOWPASS, BANDPASS, HIGHPASS, BANDSTOP = 0, 1, 2, 3
_RESPONSE_NAMES = ("LOWPASS", "BANDPASS", "HIGHPASS", "BANDSTOP")


def kaiser(n: int, beta: float) -> np.ndarray:
    """Kaiser window of length n with shape parameter beta (src/Window.jl:53-58: beta taken as is)."""
    return np.kaiser(n, beta)


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


def firprototype(numtaps: int, F: Union[float, Sequence[float]], response: int = LOWPASS) -> np.ndarray:
    """Ideal (unwindowed) impulse response, src/FIRDesign.jl:47-66.  ``F`` is the cutoff in cycles/sample
    (a pair for BANDPASS/BANDSTOP).  HIGHPASS returns numtaps+1 samples when numtaps is even (:55: M is made
    even so that the filter is type 1)."""
    M = numtaps - 1
    if response == LOWPASS:
        n = np.arange(M + 1, dtype=np.float64)
        return 2.0 * F * np.sinc(2.0 * F * (n - M / 2.0))
    if response == BANDPASS:
        n = np.arange(M + 1, dtype=np.float64)
        return 2.0 * (F[0] * np.sinc(2.0 * F[0] * (n - M / 2.0)) - F[1] * np.sinc(2.0 * F[1] * (n - M / 2.0)))
    if response == HIGHPASS:
        M = M + 1 if M % 2 else M
        n = np.arange(M + 1, dtype=np.float64)
        return np.sinc(n - M / 2.0) - 2.0 * F * np.sinc(2.0 * F * (n - M / 2.0))
    if response == BANDSTOP:
        n = np.arange(M + 1, dtype=np.float64)
        return 2.0 * (F[1] * np.sinc(2.0 * F[1] * (n - M / 2.0)) - F[0] * np.sinc(2.0 * F[0] * (n - M / 2.0)))
    raise ValueError("Not a valid FIR_TYPE")          # src/FIRDesign.jl:62


def firdes(a, b, window: Union[str, Callable, float, None] = "kaiser", *, response: int = LOWPASS,
           samplerate: float = 1.0, beta: float = 6.75, dtype=np.float64) -> np.ndarray:
    """Both ``firdes`` methods of the reference:

    * ``firdes(numtaps::Integer, cutoff, windowfunction; response, samplerate, beta)`` (src/FIRDesign.jl:76-88):
      ``firdes(numtaps, cutoff, "kaiser" | callable, ...)`` -- the first argument is an int.
    * ``firdes(cutoff, transitionwidth, stopbandAttenuation = 60; response, samplerate)`` (:90-95):
      ``firdes(cutoff, transitionwidth[, attenuation], ...)`` -- the first argument is a float or a pair;
      the length and beta come from ``kaiserlength``.
    """
    if isinstance(a, (int, np.integer)) and not isinstance(a, bool):
        numtaps, cutoff = int(a), b
        cutoff = (np.asarray(cutoff, dtype=np.float64) / samplerate).tolist() if np.ndim(cutoff) else cutoff / samplerate
        proto = firprototype(numtaps, cutoff, response)
        numtaps = len(proto)
        if window == "kaiser" or window is kaiser:
            w = kaiser(numtaps, beta)
        elif callable(window):
            w = np.asarray(window(numtaps), dtype=np.float64)
        else:
            raise ValueError("window must be 'kaiser' or a callable")
        return (proto * w).astype(dtype)
    cutoff, transitionwidth = a, b
    attenuation = 60.0 if window in ("kaiser", None) else float(window)     # third positional = stopbandAttenuation
    numtaps, kbeta = kaiserlength(transitionwidth, attenuation, samplerate=samplerate)
    return firdes(numtaps, cutoff, "kaiser", response=response, samplerate=samplerate, beta=kbeta, dtype=dtype)
