"""Windowed-sinc FIR design: host-side mirror of the reference's src/FIRDesign.jl:7-95
(``FIRResponse`` enum :7, ``kaiserlength`` :18-33, ``firprototype`` :47-66, both ``firdes`` methods :76-95).

Every function here is a ctypes call into libmultirate_hip.so (``mrhip_kaiserlength``, ``mrhip_kaiser``,
``mrhip_firprototype``, ``mrhip_firdes``, ``mrhip_firdes_kaiser`` -- csrc/design.cpp), the same entry points the Julia
shim binds, so a ``FIRFilter``-level user of either language stays inside the package.  Host only, O(taps), Float64,
off the hot path.  The Kaiser window takes beta directly (as the reference's in-tree
src/Window.jl:53-58 does); the window function the reference actually calls lives in the un-vendored DSP.jl of 2014,
so tap VALUES are not claimed to be bit-identical to a historical Multirate.jl run ("parity unpinned")
-- the same taps are always fed to both the engine and the oracle, which is what parity of the hot path needs.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Sequence, Union

import numpy as np

from .host import MultirateHIPError, load_library

# @enum( FIRResponse, LOWPASS, BANDPASS, HIGHPASS, BANDSTOP ), src/FIRDesign.jl:7 (values 0..3, src/enum.jl:11-13)
LOWPASS, BANDPASS, HIGHPASS, BANDSTOP = 0, 1, 2, 3


def _err(what: str):
    return MultirateHIPError(1, f"{what}: {load_library().mrhip_last_error().decode('utf-8', 'replace')}")


def _cutoffs(F) -> np.ndarray:
    return np.ascontiguousarray(np.atleast_1d(np.asarray(F, dtype=np.float64)))


def kaiser(n: int, beta: float) -> np.ndarray:
    """Kaiser window of length n with shape parameter beta (src/Window.jl:53-58: beta taken as is)."""
    out = np.empty(int(n), dtype=np.float64)
    if load_library().mrhip_kaiser(int(n), float(beta), out.ctypes.data_as(C.c_void_p)) != 0:
        raise _err("kaiser")
    return out


def kaiserlength(transition: float, attenuation: float = 60.0, samplerate: float = 1.0):
    """(numtaps, beta) for a Kaiser-window design, src/FIRDesign.jl:18-33."""
    n, b = C.c_int64(0), C.c_double(0.0)
    if load_library().mrhip_kaiserlength(float(transition), float(attenuation), float(samplerate), C.byref(n), C.byref(b)) != 0:
        raise _err("kaiserlength")
    return int(n.value), float(b.value)


def firprototype(numtaps: int, F: Union[float, Sequence[float]], response: int = LOWPASS) -> np.ndarray:
    """Ideal (unwindowed) impulse response, src/FIRDesign.jl:47-66.  ``F`` is the cutoff in cycles/sample
    (a pair for BANDPASS/BANDSTOP).  HIGHPASS returns numtaps+1 samples when numtaps is even (:55: M is made
    even so that the filter is type 1)."""
    lib = load_library()
    f = _cutoffs(F)
    n = lib.mrhip_firprototype(int(numtaps), f.ctypes.data_as(C.c_void_p), len(f), int(response), None)
    if n < 0:
        raise ValueError(lib.mrhip_last_error().decode("utf-8", "replace"))          # "Not a valid FIR_TYPE", :62
    out = np.empty(n, dtype=np.float64)
    lib.mrhip_firprototype(int(numtaps), f.ctypes.data_as(C.c_void_p), len(f), int(response), out.ctypes.data_as(C.c_void_p))
    return out


def firdes(a, b, window: Union[str, Callable, float, None] = "kaiser", *, response: int = LOWPASS,
           samplerate: float = 1.0, beta: float = 6.75, dtype=np.float64) -> np.ndarray:
    """Both ``firdes`` methods of the reference:

    * ``firdes(numtaps::Integer, cutoff, windowfunction; response, samplerate, beta)`` (src/FIRDesign.jl:76-88):
      ``firdes(numtaps, cutoff, "kaiser" | callable, ...)`` -- the first argument is an int.
    * ``firdes(cutoff, transitionwidth, stopbandAttenuation = 60; response, samplerate)`` (:90-95):
      ``firdes(cutoff, transitionwidth[, attenuation], ...)`` -- the first argument is a float or a pair;
      the length and beta come from ``kaiserlength``.
    """
    lib = load_library()
    if isinstance(a, (int, np.integer)) and not isinstance(a, bool):
        numtaps, cut = int(a), _cutoffs(b)
        cp = cut.ctypes.data_as(C.c_void_p)
        n = lib.mrhip_firdes(numtaps, cp, len(cut), int(response), float(samplerate), float(beta), None, None)
        if n < 0:
            raise ValueError(lib.mrhip_last_error().decode("utf-8", "replace"))
        wp = None
        if window == "kaiser" or window is kaiser:
            pass                                            # windowfunction == kaiser: the library's window with `beta`
        elif callable(window):
            w = np.ascontiguousarray(window(n), dtype=np.float64)     # :85 prototype .* windowfunction(numtaps)
            if w.shape != (n,):
                raise ValueError("the window function must return numtaps samples")
            wp = w.ctypes.data_as(C.c_void_p)
        else:
            raise ValueError("window must be 'kaiser' or a callable")
        out = np.empty(n, dtype=np.float64)
        if lib.mrhip_firdes(numtaps, cp, len(cut), int(response), float(samplerate), float(beta), wp, out.ctypes.data_as(C.c_void_p)) != n:
            raise _err("firdes")
        return out.astype(dtype)
    cut = _cutoffs(a)
    cp = cut.ctypes.data_as(C.c_void_p)
    attenuation = 60.0 if window in ("kaiser", None) else float(window)     # third positional = stopbandAttenuation
    n = lib.mrhip_firdes_kaiser(cp, len(cut), float(b), attenuation, int(response), float(samplerate), None)
    if n < 0:
        raise ValueError(lib.mrhip_last_error().decode("utf-8", "replace"))
    out = np.empty(n, dtype=np.float64)
    if lib.mrhip_firdes_kaiser(cp, len(cut), float(b), attenuation, int(response), float(samplerate), out.ctypes.data_as(C.c_void_p)) != n:
        raise _err("firdes")
    return out.astype(dtype)
