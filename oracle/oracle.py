"""ctypes face of the CPU oracle (oracle/multirate_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from the product package.  See the header of
multirate_oracle.c for what is restated (reference file:line) and for the
pinning status of the oracle.

The class mirrors the reference call shape: ``FIRFilter(h, ratio)`` /
``FIRFilter(h, rate, Nphi)`` (src/Filters.jl:158,183) and ``filt(self, x)``
(src/Filters.jl:475,519,577,633,744).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from fractions import Fraction

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmultirate_oracle.so")

F32, F64, C64, C128 = 0, 1, 2, 3
STANDARD, DECIMATOR, INTERPOLATOR, RATIONAL, ARBITRARY = 0, 1, 2, 3, 4
KIND_NAMES = {0: "FIRStandard", 1: "FIRDecimator", 2: "FIRInterpolator", 3: "FIRRational", 4: "FIRArbitrary"}

_NP2DT = {np.dtype(np.float32): F32, np.dtype(np.float64): F64,
          np.dtype(np.complex64): C64, np.dtype(np.complex128): C128}
_DT2NP = {v: k for k, v in _NP2DT.items()}


class _State(C.Structure):
    _fields_ = [("kind", C.c_int), ("phiIdx", C.c_long), ("inputDeficit", C.c_long),
                ("phiAccumulator", C.c_double), ("alpha", C.c_double), ("delta", C.c_double),
                ("xIdx", C.c_long), ("tapsPerPhi", C.c_long), ("Nphi", C.c_long),
                ("historyLen", C.c_long), ("L", C.c_long), ("M", C.c_long), ("hLen", C.c_long)]


class _Sched(C.Structure):
    _fields_ = [("xIdx", C.c_long), ("phiIdx", C.c_long), ("alpha", C.c_double)]


def build(force: bool = False) -> str:
    """Compile the oracle with oracle/Makefile (gcc).  Building the checker is not using it."""
    src = [os.path.join(_HERE, n) for n in ("multirate_oracle.c", "oracle_typed.inc", "multirate_oracle.h")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, cl, ci, cd = C.c_void_p, C.c_long, C.c_int, C.c_double
        L.mro_create_rational.restype = vp
        L.mro_create_rational.argtypes = [vp, cl, ci, cl, cl, ci]
        L.mro_create_arbitrary.restype = vp
        L.mro_create_arbitrary.argtypes = [vp, cl, ci, cd, cl, ci]
        L.mro_create_farrow.restype = vp
        L.mro_create_farrow.argtypes = [cl, ci, cd, cl, cl, vp, ci]
        L.mro_polyval.restype = cd
        L.mro_polyval.argtypes = [vp, cl, cd]
        L.mro_get_current_taps.argtypes = [vp, vp]
        L.mro_destroy.argtypes = [vp]
        L.mro_set_fused.argtypes = [ci]
        L.mro_set_mod_form.argtypes = [ci]
        L.mro_outputlength.restype = cl
        L.mro_outputlength.argtypes = [vp, cl]
        L.mro_inputlength.restype = cl
        L.mro_inputlength.argtypes = [vp, cl]
        L.mro_filt.restype = cl
        L.mro_filt.argtypes = [vp, vp, cl, vp, cl]
        L.mro_filt_sched.restype = cl
        L.mro_filt_sched.argtypes = [vp, vp, cl, vp, cl, vp]
        L.mro_get_state.argtypes = [vp, C.POINTER(_State)]
        L.mro_set_state.argtypes = [vp, cl, cl, cd]
        L.mro_get_history.argtypes = [vp, vp]
        L.mro_set_history.argtypes = [vp, vp]
        L.mro_get_taps.argtypes = [vp, ci, vp]
        L.mro_reset.argtypes = [vp]
        L.mro_taps2pfb.restype = cl
        L.mro_taps2pfb.argtypes = [vp, cl, ci, cl, vp]
        L.mro_nextphase.restype = cl
        L.mro_nextphase.argtypes = [cl, cl, cl]
        L.mro_outputlength_ratio.restype = cl
        L.mro_outputlength_ratio.argtypes = [cl, cl, cl, cl]
        L.mro_inputlength_ratio.restype = cl
        L.mro_inputlength_ratio.argtypes = [cl, cl, cl, cl]
        L.mro_output_dtype.restype = ci
        L.mro_output_dtype.argtypes = [ci, ci]
        L.mro_shiftin.argtypes = [vp, cl, vp, cl, C.c_size_t]
        _lib = L
    return _lib


def set_fused(fused: bool):
    """Process-wide: True makes every dot product use one fma per tap (checker for the library's opt-in
    NUMERICS_FUSED mode); False (default) is the reference's separately rounded multiply and add."""
    lib().mro_set_fused(1 if fused else 0)


def set_mod_form(julia03: bool):
    """Process-wide: the float mod() of update() (src/Filters.jl:668) as rem(y + rem(x, y), y) (True: the form older Julia
    Base versions used) instead of the exact remainder (False, default: Julia >= 0.4)."""
    lib().mro_set_mod_form(1 if julia03 else 0)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def taps2pfb(h, Nphi: int) -> np.ndarray:
    """src/Filters.jl:284-298.  Returns the tapsPerPhi x Nphi matrix (numpy, row index first)."""
    h = np.ascontiguousarray(h)
    if h.dtype not in (np.float32, np.float64):
        h = h.astype(np.float64)
    T = lib().mro_taps2pfb(_ptr(h), len(h), _NP2DT[h.dtype], Nphi, None)
    out = np.empty(T * Nphi, dtype=h.dtype)
    lib().mro_taps2pfb(_ptr(h), len(h), _NP2DT[h.dtype], Nphi, _ptr(out))
    return out.reshape(Nphi, T).T.copy()  # column-major T x Nphi


def nextphase(phase: int, L: int, M: int) -> int:
    return lib().mro_nextphase(phase, L, M)


def shiftin(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a).copy()
    b = np.ascontiguousarray(b, dtype=a.dtype)
    lib().mro_shiftin(_ptr(a), len(a), _ptr(b), len(b), a.dtype.itemsize)
    return a


def polyfit(y, polyorder: int) -> np.ndarray:
    """src/support.jl:85-88: A = [x^p for x in 1:length(y), p = 0:polyorder]; Poly(A \\ y).
    Julia's `\\` on a tall matrix is a QR least-squares solve in Float64; restated with LAPACK's
    Householder QR (numpy.linalg.qr) and a triangular solve.  Different QR implementations agree to
    rounding (amplified by the conditioning of the Vandermonde matrix), not bit for bit -- the reference
    pins nothing here -- so the fit is done ONCE on the caller's side and the same
    coefficients are handed to oracle and GPU.  Returns coefficients in ascending powers, Float64."""
    y = np.asarray(y, dtype=np.float64)
    xs = np.arange(1, len(y) + 1, dtype=np.float64)
    A = np.vander(xs, polyorder + 1, increasing=True)
    Q, R = np.linalg.qr(A)
    import scipy.linalg
    return scipy.linalg.solve_triangular(R, Q.T @ y)


def pfb2pnfb(pfb: np.ndarray, polyorder: int) -> np.ndarray:
    """src/Filters.jl:311-321: one polynomial per ROW of the tapsPerPhi x Nphi filter bank.  The result
    is stored as Poly{T} (T = tap type), i.e. the Float64 fit rounded to T.  Shape (tapsPerPhi, polyorder+1),
    ascending powers, dtype float64 holding T-representable values."""
    T = pfb.shape[0]
    out = np.empty((T, polyorder + 1), dtype=np.float64)
    for i in range(T):
        out[i] = polyfit(pfb[i, :], polyorder).astype(pfb.dtype).astype(np.float64)
    return out


def polyval(coeffs, x: float) -> float:
    c = np.ascontiguousarray(coeffs, dtype=np.float64)
    return lib().mro_polyval(_ptr(c), len(c) - 1, float(x))


class FIRFilter:
    """Oracle-side FIRFilter.  ``ratio`` may be a Fraction / (num, den) tuple / int (rational
    family, Filters.jl:158) or a float (FIRArbitrary, Filters.jl:183, with ``Nphi``)."""

    def __init__(self, h, ratio=Fraction(1, 1), Nphi: int = 32, tx=np.float32, polyorder=None, pnfb=None):
        """``polyorder`` (with a float ``ratio``) selects FIRFarrow, src/Filters.jl:192-198; ``pnfb`` overrides
        the fitted polynomial bank (tapsPerPhi x (polyorder+1), ascending powers)."""
        h = np.ascontiguousarray(h)
        if h.dtype not in (np.float32, np.float64):
            raise TypeError("taps must be float32 or float64")
        self.th = h.dtype
        self.tx = np.dtype(tx)
        L = lib()
        self.pnfb = None
        if isinstance(ratio, float) and polyorder is not None:
            if pnfb is None:
                pnfb = pfb2pnfb(taps2pfb(h, Nphi), polyorder)
            self.pnfb = np.ascontiguousarray(pnfb, dtype=np.float64)
            assert self.pnfb.shape == (-(-len(h) // Nphi), polyorder + 1), self.pnfb.shape
            self._h = L.mro_create_farrow(len(h), _NP2DT[h.dtype], ratio, Nphi, polyorder, _ptr(self.pnfb), _NP2DT[self.tx])
            if not self._h:
                raise ValueError("rate must be greater than 0")
        elif isinstance(ratio, float):
            self._h = L.mro_create_arbitrary(_ptr(h), len(h), _NP2DT[h.dtype], ratio, Nphi, _NP2DT[self.tx])
            if not self._h:
                raise ValueError("rate must be greater than 0")
        else:
            if isinstance(ratio, tuple):
                ratio = Fraction(*ratio)
            ratio = Fraction(ratio)
            self._h = L.mro_create_rational(_ptr(h), len(h), _NP2DT[h.dtype], ratio.numerator,
                                            ratio.denominator, _NP2DT[self.tx])
            if not self._h:
                raise ValueError("bad arguments")
        self.ty = _DT2NP[L.mro_output_dtype(_NP2DT[self.th], _NP2DT[self.tx])]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().mro_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def state(self) -> _State:
        s = _State()
        lib().mro_get_state(self._h, C.byref(s))
        return s

    @property
    def kind(self) -> int:
        return self.state.kind

    @property
    def history(self) -> np.ndarray:
        out = np.zeros(self.state.historyLen, dtype=self.tx)
        lib().mro_get_history(self._h, _ptr(out))
        return out

    def set_history(self, hist):
        hist = np.ascontiguousarray(hist, dtype=self.tx)
        assert len(hist) == self.state.historyLen
        lib().mro_set_history(self._h, _ptr(hist))

    def set_state(self, phiIdx=1, inputDeficit=1, phiAccumulator=1.0):
        lib().mro_set_state(self._h, phiIdx, inputDeficit, phiAccumulator)

    def taps(self, which: int = 0) -> np.ndarray:
        s = self.state
        out = np.zeros(s.tapsPerPhi * s.Nphi, dtype=self.th)
        lib().mro_get_taps(self._h, which, _ptr(out))
        return out.reshape(s.Nphi, s.tapsPerPhi).T.copy()

    def current_taps(self) -> np.ndarray:
        """FIRFarrow.currentTaps (src/Filters.jl:128)"""
        out = np.zeros(self.state.tapsPerPhi, dtype=self.th)
        lib().mro_get_current_taps(self._h, _ptr(out))
        return out

    def outputlength(self, xlen: int) -> int:
        return lib().mro_outputlength(self._h, xlen)

    def inputlength(self, ylen: int) -> int:
        return lib().mro_inputlength(self._h, ylen)

    def reset(self):
        lib().mro_reset(self._h)

    def filt(self, x, return_schedule: bool = False):
        x = np.ascontiguousarray(x, dtype=self.tx)
        n = len(x)
        cap = max(self.outputlength(n), 0) + 2  # +2: FIRArbitrary's outputlength is only a guess
        y = np.empty(cap, dtype=self.ty)
        sched = (_Sched * cap)() if return_schedule else None
        cnt = lib().mro_filt_sched(self._h, _ptr(x), n, _ptr(y), cap,
                                   C.cast(sched, C.c_void_p) if sched is not None else None)
        if cnt < 0:
            raise RuntimeError({-1: "buffer is too small", -2: "unsafedot guard"}.get(cnt, "oracle error"))
        y = y[:cnt].copy()
        if return_schedule:
            sc = np.array([(sched[i].xIdx, sched[i].phiIdx, sched[i].alpha) for i in range(cnt)],
                          dtype=[("xIdx", np.int64), ("phiIdx", np.int64), ("alpha", np.float64)])
            return y, sc
        return y


def filt(h, x, ratio=Fraction(1, 1), Nphi: int = 32, polyorder=None):
    """Stateless filt(h, x, ratio) / filt(h, x, rate, Nphi) / filt(h, x, rate, Nphi, polyorder):
    src/Filters.jl:858-873."""
    x = np.ascontiguousarray(x)
    return FIRFilter(h, ratio, Nphi, tx=x.dtype, polyorder=polyorder).filt(x)
