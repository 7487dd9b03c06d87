"""Host-side mirror of the reference operator interface over the C ABI (ctypes).

Reference interface mirrored (paths relative to /root/reference):
  FIRFilter(h, ratio) / FIRFilter(h, rate, Nphi)        src/Filters.jl:158-189
  filt(self, x) / filt(h, x, ratio) / filt(h, x, rate, Nphi)   src/Filters.jl:475-873
  filt!(buffer, self, x)  (here ``filt_``)               src/Filters.jl:450,489,536,598,693
  taps2pfb, outputlength, inputlength, nextphase, reset  src/Filters.jl:284,352,396,433,244

Signals are numpy arrays (host path, ``mrhip_filt_host``) or torch tensors on a ROCm device
(device path, ``mrhip_filt_device`` on torch's current stream).  Shape ``(n,)`` is one channel;
shape ``(nchannels, n)`` (C-contiguous: one channel per row == one channel per column of a Julia
Matrix) is a batch of independent streams sharing taps and phase state.

No arithmetic is done in Python and nothing here imports ``oracle/``: if ``libmultirate_hip.so``
is missing, or no gfx950 device is visible, the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from fractions import Fraction

import numpy as np

try:  # torch first: its bundled libamdhip64 must be the one HIP runtime of the process
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional for pure-numpy use
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))

F32, F64, C64, C128 = 0, 1, 2, 3
STANDARD, DECIMATOR, INTERPOLATOR, RATIONAL, ARBITRARY, FARROW = 0, 1, 2, 3, 4, 5
KIND_NAMES = {STANDARD: "FIRStandard", DECIMATOR: "FIRDecimator", INTERPOLATOR: "FIRInterpolator",
              RATIONAL: "FIRRational", ARBITRARY: "FIRArbitrary", FARROW: "FIRFarrow"}
NUMERICS_STRICT, NUMERICS_FUSED = 0, 1

_NP2DT = {np.dtype(np.float32): F32, np.dtype(np.float64): F64,
          np.dtype(np.complex64): C64, np.dtype(np.complex128): C128}
_DT2NP = {v: k for k, v in _NP2DT.items()}


class MultirateHIPError(RuntimeError):
    """Raised where the reference raises Julia ``error(...)`` (and for HIP/runtime failures)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"[mrhip status {code}] {msg}")
        self.code = code


class _State(C.Structure):
    _fields_ = [("kind", C.c_int32), ("tap_dtype", C.c_int32), ("sample_dtype", C.c_int32),
                ("output_dtype", C.c_int32), ("nchannels", C.c_int64), ("hLen", C.c_int64),
                ("interpolation", C.c_int64), ("decimation", C.c_int64), ("Nphi", C.c_int64),
                ("tapsPerPhi", C.c_int64), ("historyLen", C.c_int64), ("phiIdx", C.c_int64),
                ("inputDeficit", C.c_int64), ("xIdx", C.c_int64), ("rate", C.c_double),
                ("phiAccumulator", C.c_double), ("alpha", C.c_double), ("delta", C.c_double)]


# every symbol include/multirate_hip.h declares: (name, restype, argtypes)
_vp, _i64, _i, _d = C.c_void_p, C.c_int64, C.c_int, C.c_double
_pi64 = C.POINTER(C.c_int64)
ABI = [
    ("mrhip_abi_version", _i, []),
    ("mrhip_last_error", C.c_char_p, []),
    ("mrhip_device_count", _i, []),
    ("mrhip_taps2pfb", _i64, [_vp, _i64, _i, _i64, _vp]),
    ("mrhip_nextphase", _i64, [_i64, _i64, _i64]),
    ("mrhip_outputlength_ratio", _i64, [_i64, _i64, _i64, _i64]),
    ("mrhip_inputlength_ratio", _i64, [_i64, _i64, _i64, _i64]),
    ("mrhip_polyfit", _i, [_vp, _i64, _i64, _vp]),
    ("mrhip_output_dtype", _i, [_i, _i]),
    ("mrhip_kaiserlength", _i, [_d, _d, _d, _pi64, C.POINTER(C.c_double)]),
    ("mrhip_kaiser", _i, [_i64, _d, _vp]),
    ("mrhip_firprototype", _i64, [_i64, _vp, _i, _i, _vp]),
    ("mrhip_firdes", _i64, [_i64, _vp, _i, _i, _d, _d, _vp, _vp]),
    ("mrhip_firdes_kaiser", _i64, [_vp, _i, _d, _d, _i, _d, _vp]),
    ("mrhip_create_rational", _i, [_vp, _i64, _i, _i64, _i64, _i, _i64, _i, C.POINTER(_vp)]),
    ("mrhip_create_arbitrary", _i, [_vp, _i64, _i, _d, _i64, _i, _i64, _i, C.POINTER(_vp)]),
    ("mrhip_create_farrow", _i, [_vp, _i64, _i, _d, _i64, _i64, _i, _i64, _i, C.POINTER(_vp)]),
    ("mrhip_create_farrow_pnfb", _i, [_vp, _i64, _i, _d, _i64, _i64, _i, _i64, _i, C.POINTER(_vp)]),
    ("mrhip_get_pnfb", _i, [_vp, _vp]),
    ("mrhip_farrow_tapsforphase", _i, [_vp, _d, _vp]),
    ("mrhip_arbitrary_tapsforphase", _i, [_vp, _d, _vp]),
    ("mrhip_destroy", None, [_vp]),
    ("mrhip_outputlength", _i64, [_vp, _i64]),
    ("mrhip_next_output_count", _i64, [_vp, _i64]),
    ("mrhip_advance_state", _i64, [_vp, _i64]),
    ("mrhip_inputlength", _i64, [_vp, _i64]),
    ("mrhip_get_state", _i, [_vp, C.POINTER(_State)]),
    ("mrhip_set_state", _i, [_vp, _i64, _i64, _d]),
    ("mrhip_get_history", _i, [_vp, _vp]),
    ("mrhip_set_history", _i, [_vp, _vp]),
    ("mrhip_reset", _i, [_vp]),
    ("mrhip_set_numerics", _i, [_vp, _i]),
    ("mrhip_set_mod_form", _i, [_vp, _i]),
    ("mrhip_get_taps", _i, [_vp, _i, _vp]),
    ("mrhip_filt_device", _i, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _pi64, _vp]),
    ("mrhip_filt_device_async", _i, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp]),
    ("mrhip_filt_device_chained", _i, [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp]),
    ("mrhip_filt_device_multi", _i, [C.POINTER(_vp), _i, C.POINTER(_vp), _pi64, C.POINTER(_vp), _pi64, _pi64, _vp]),
    ("mrhip_outputlength_bound", _i64, [_vp, _i64]),
    ("mrhip_sync_state", _i, [_vp, _pi64]),
    ("mrhip_set_history_device", _i, [_vp, _vp, _vp]),
    ("mrhip_filt_device_chunked", _i, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _pi64, _vp]),
    ("mrhip_filt_host", _i, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _pi64]),
    ("mrhip_synchronize", _i, [_vp, _vp]),
    ("mrhip_cascade_create", _i, [C.POINTER(_vp), _i, C.POINTER(_vp)]),
    ("mrhip_cascade_destroy", None, [_vp]),
    ("mrhip_cascade_outputlength", _i64, [_vp, _i64]),
    ("mrhip_cascade_next_output_count", _i64, [_vp, _i64]),
    ("mrhip_cascade_filt_device", _i, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _pi64, _vp]),
    ("mrhip_cascade_filt_device_async", _i, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp]),
    ("mrhip_cascade_reset", _i, [_vp]),
    ("mrhip_filt_once", _i, [_vp, _i64, _i, _i64, _i64, _d, _i64, _vp, _i64, _i, _vp, _i64, _pi64, _i]),
    ("mrhip_set_timing", _i, [_vp, _i]),
    ("mrhip_timing_read", _i, [_vp, _pi64, C.POINTER(C.c_double)]),
    ("mrhip_last_kernel_name", C.c_char_p, [_vp]),
    ("mrhip_schedule_info", _i, [_vp, _pi64, _i]),
    ("mrhip_sharded_create", _i, [_i, _vp, _i64, _i, _i64, _i64, _d, _i64, _i64, _i, _i64, C.POINTER(_i), _i, C.POINTER(_vp)]),
    ("mrhip_sharded_destroy", None, [_vp]),
    ("mrhip_sharded_nshards", _i, [_vp]),
    ("mrhip_sharded_shard", _i, [_vp, _i, _pi64, _pi64, C.POINTER(_i), C.POINTER(_vp)]),
    ("mrhip_sharded_outputlength", _i64, [_vp, _i64]),
    ("mrhip_sharded_next_output_count", _i64, [_vp, _i64]),
    ("mrhip_sharded_reset", _i, [_vp]),
    ("mrhip_sharded_filt_device", _i, [_vp, C.POINTER(_vp), _i64, _pi64, C.POINTER(_vp), _i64, _pi64, _pi64]),
    ("mrhip_sharded_filt_host", _i, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _pi64]),
    ("mrhip_sharded_gather", _i, [_vp, C.POINTER(_vp), _i64, _pi64, _vp, _i64, _i]),
    ("mrhip_sharded_wait_stream", _i, [_vp, _i, _vp]),
    ("mrhip_sharded_signal_stream", _i, [_vp, _i, _vp]),
    ("mrhip_sharded_synchronize", _i, [_vp]),
    ("mrhip_ring_open", _i, [_vp, C.POINTER(_vp)]),
    ("mrhip_ring_push", _i, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _pi64, C.POINTER(C.c_uint64)]),
    ("mrhip_ring_push_chunks", _i, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _pi64, C.POINTER(C.c_uint64)]),
    ("mrhip_ring_wait", _i, [_vp, C.c_uint64]),
    ("mrhip_ring_drain", _i, [_vp]),
    ("mrhip_ring_close", _i, [_vp]),
    ("mrhip_ring_info", _i, [_vp, _pi64, _i]),
]

_lib = None


def library_path() -> str:
    # MRHIP_LIB_PATH: developer override used to A/B-test alternative builds of the same ABI
    return os.environ.get("MRHIP_LIB_PATH") or os.path.join(_HERE, "libmultirate_hip.so")


def load_library():
    """dlopen libmultirate_hip.so and bind every ABI symbol.  Raises if the library is missing --
    there is no fallback implementation."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise MultirateHIPError(-1, f"{path} not found: build it with `make -C multirate.jl_amd/csrc` "
                                        "(or __graft_entry__.build()); there is no CPU fallback")
        lib = C.CDLL(path)
        for name, res, args in ABI:
            fn = getattr(lib, name)  # AttributeError if the header and the .so ever disagree
            fn.restype = res
            fn.argtypes = args
        if lib.mrhip_abi_version() != 1:
            raise MultirateHIPError(-1, "ABI version mismatch")
        _lib = lib
    return _lib


def _check(rc: int):
    if rc != 0:
        raise MultirateHIPError(rc, load_library().mrhip_last_error().decode("utf-8", "replace"))


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _as_taps(h) -> np.ndarray:
    h = np.ascontiguousarray(h)
    if h.dtype == np.float32 or h.dtype == np.float64:
        return h
    if np.issubdtype(h.dtype, np.integer) or h.dtype == np.float16:
        return h.astype(np.float64)
    raise MultirateHIPError(5, f"unsupported tap dtype {h.dtype} (taps must be Float32/Float64)")


def _is_torch(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor)


def _torch_np_dtype(t):
    return {torch.float32: np.dtype(np.float32), torch.float64: np.dtype(np.float64),
            torch.complex64: np.dtype(np.complex64), torch.complex128: np.dtype(np.complex128)}[t]


def _np_torch_dtype(d):
    return {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
            np.dtype(np.complex64): torch.complex64, np.dtype(np.complex128): torch.complex128}[np.dtype(d)]


# ---- host-only helpers ------------------------------------------------------------------------
def taps2pfb(h, Nphi: int) -> np.ndarray:
    """taps2pfb(h, N𝜙), src/Filters.jl:284-298.  Returns the tapsPer𝜙 x N𝜙 matrix."""
    h = _as_taps(h)
    lib = load_library()
    T = lib.mrhip_taps2pfb(_ptr(h), len(h), _NP2DT[h.dtype], Nphi, None)
    if T < 0:
        raise MultirateHIPError(1, "bad taps2pfb arguments")
    out = np.empty(T * Nphi, dtype=h.dtype)
    lib.mrhip_taps2pfb(_ptr(h), len(h), _NP2DT[h.dtype], Nphi, _ptr(out))
    return out.reshape(Nphi, T).T.copy()


def polyfit(y, polyorder: int) -> np.ndarray:
    """polyfit(y, polyorder), src/support.jl:85-88: coefficients in ascending powers (Poly.a)."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.zeros(polyorder + 1, dtype=np.float64)
    _check(load_library().mrhip_polyfit(_ptr(y), len(y), polyorder, _ptr(out)))
    return out


def nextphase(currentphase: int, ratio) -> int:
    """nextphase(currentphase, ratio), src/Filters.jl:433-439."""
    r = Fraction(ratio)
    return load_library().mrhip_nextphase(currentphase, r.numerator, r.denominator)


# ---- the filter object ------------------------------------------------------------------------
class FIRFilter:
    """FIRFilter(h, ratio::Rational = 1//1)  or  FIRFilter(h, rate::Float, N𝜙 = 32)
    or  FIRFilter(h, rate::Float, N𝜙, polyorder)  (FIRFarrow, src/Filters.jl:192-198).

    Mirrors src/Filters.jl:158-189: a ``Fraction``/int/(num, den) ratio selects FIRStandard,
    FIRDecimator, FIRInterpolator or FIRRational exactly as the reference does; a ``float`` selects
    FIRArbitrary.  Like the reference, whose ``history`` vector takes the sample type of the first
    ``filt`` call (src/Filters.jl:452), the device-side object is created on the first call, when
    the sample dtype and the number of channels are known.
    """

    def __init__(self, h, ratio=Fraction(1, 1), Nphi: int = 32, polyorder=None, *, device: int = 0,
                 numerics: int = NUMERICS_STRICT, pnfb=None):
        self._lib = load_library()
        self.h = _as_taps(h).copy()
        if len(self.h) < 1:
            raise MultirateHIPError(1, "h must hold at least one tap")
        self.device = int(device)
        self.numerics = int(numerics)
        self._handle = None
        self._tx = None
        self._nch = None
        self.polyorder = None
        self._pnfb_in = None
        if isinstance(ratio, (float, np.floating)):
            if not ratio > 0.0:
                raise MultirateHIPError(1, "rate must be greater than 0")  # Filters.jl:184
            self.rate = float(ratio)
            self.ratio = None
            self.Nphi = int(Nphi)
            self.kind = ARBITRARY if polyorder is None else FARROW
            self.polyorder = None if polyorder is None else int(polyorder)
            # optional caller-fitted polynomial bank (tapsPerPhi x (polyorder+1), ascending powers)
            self._pnfb_in = None if pnfb is None else np.ascontiguousarray(pnfb, dtype=np.float64)
            self.interpolation, self.decimation = self.Nphi, 1
            self.tapsPerPhi = -(-len(self.h) // self.Nphi)
            self.historyLen = self.tapsPerPhi - 1
        else:
            if isinstance(ratio, tuple):
                ratio = Fraction(*ratio)
            self.ratio = Fraction(ratio)
            if self.ratio <= 0:
                raise MultirateHIPError(1, "ratio must be positive")
            self.rate = None
            L, M = self.ratio.numerator, self.ratio.denominator
            self.interpolation, self.decimation = L, M
            if L == 1 and M == 1:
                self.kind = STANDARD
            elif L == 1:
                self.kind = DECIMATOR
            elif M == 1:
                self.kind = INTERPOLATOR
            else:
                self.kind = RATIONAL
            self.Nphi = L
            self.tapsPerPhi = len(self.h) if L == 1 else -(-len(self.h) // L)
            self.historyLen = self.tapsPerPhi - 1

    # -- lifetime
    def _ensure(self, tx: np.dtype, nch: int):
        tx = np.dtype(tx)
        if self._handle is not None:
            if tx != self._tx or nch != self._nch:
                raise MultirateHIPError(1, f"filter was bound to {self._nch} channel(s) of {self._tx}; "
                                           f"got {nch} of {tx}")
            return
        if tx not in _NP2DT:
            raise MultirateHIPError(1, f"unsupported sample dtype {tx}")
        out = C.c_void_p()
        if self.kind == FARROW and self._pnfb_in is not None:
            rc = self._lib.mrhip_create_farrow_pnfb(_ptr(self._pnfb_in), len(self.h), _NP2DT[self.h.dtype], self.rate,
                                                    self.Nphi, self.polyorder, _NP2DT[tx], nch, self.device, C.byref(out))
        elif self.kind == FARROW:
            rc = self._lib.mrhip_create_farrow(_ptr(self.h), len(self.h), _NP2DT[self.h.dtype], self.rate, self.Nphi,
                                               self.polyorder, _NP2DT[tx], nch, self.device, C.byref(out))
        elif self.kind == ARBITRARY:
            rc = self._lib.mrhip_create_arbitrary(_ptr(self.h), len(self.h), _NP2DT[self.h.dtype], self.rate,
                                                  self.Nphi, _NP2DT[tx], nch, self.device, C.byref(out))
        else:
            rc = self._lib.mrhip_create_rational(_ptr(self.h), len(self.h), _NP2DT[self.h.dtype],
                                                 self.ratio.numerator, self.ratio.denominator, _NP2DT[tx], nch,
                                                 self.device, C.byref(out))
        _check(rc)
        self._handle, self._tx, self._nch = out, tx, nch
        if self.numerics != NUMERICS_STRICT:
            _check(self._lib.mrhip_set_numerics(self._handle, self.numerics))

    def close(self):
        if getattr(self, "_handle", None):
            self._lib.mrhip_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- introspection
    @property
    def kernel_name(self) -> str:
        return KIND_NAMES[self.kind]

    @property
    def output_dtype(self):
        if self._tx is None:
            return None
        return _DT2NP[self._lib.mrhip_output_dtype(_NP2DT[self.h.dtype], _NP2DT[self._tx])]

    @property
    def state(self) -> _State:
        st = _State()
        if self._handle is None:  # constructor state, src/Filters.jl:77-78,110-115
            st.kind = self.kind
            st.phiIdx, st.inputDeficit, st.xIdx = 1, 1, 1
            st.phiAccumulator, st.alpha = 1.0, 0.0
            st.Nphi, st.tapsPerPhi, st.historyLen = self.Nphi, self.tapsPerPhi, self.historyLen
            st.interpolation, st.decimation, st.hLen = self.interpolation, self.decimation, len(self.h)
            st.rate = self.rate or 0.0
            st.delta = self.Nphi / self.rate if self.rate else 0.0
            return st
        _check(self._lib.mrhip_get_state(self._handle, C.byref(st)))
        return st

    def set_state(self, phiIdx: int = 1, inputDeficit: int = 1, phiAccumulator: float = 1.0):
        if self._handle is None:
            raise MultirateHIPError(1, "set_state needs a bound filter (call filt once, or bind())")
        _check(self._lib.mrhip_set_state(self._handle, phiIdx, inputDeficit, phiAccumulator))

    def bind(self, dtype, nchannels: int = 1):
        """Create the device object now (normally done by the first filt call)."""
        self._ensure(np.dtype(dtype), int(nchannels))
        return self

    @property
    def history(self) -> np.ndarray:
        """FIRFilter.history (src/Filters.jl:153); shape (historyLen,) or (nchannels, historyLen)."""
        if self._handle is None:
            return np.zeros(self.historyLen)
        out = np.zeros((self._nch, self.historyLen), dtype=self._tx)
        _check(self._lib.mrhip_get_history(self._handle, _ptr(out)))
        return out[0] if self._nch == 1 else out

    def set_history(self, hist):
        hist = np.ascontiguousarray(hist, dtype=self._tx).reshape(self._nch, self.historyLen)
        _check(self._lib.mrhip_set_history(self._handle, _ptr(hist)))

    def taps(self, which: int = 0) -> np.ndarray:
        """kernel.h (flipped) / kernel.pfb (which=0) or kernel.dpfb (which=1) as stored."""
        if self._handle is None:
            raise MultirateHIPError(1, "taps() needs a bound filter")
        out = np.zeros(self.tapsPerPhi * self.Nphi, dtype=self.h.dtype)
        _check(self._lib.mrhip_get_taps(self._handle, which, _ptr(out)))
        return out.reshape(self.Nphi, self.tapsPerPhi).T.copy()

    def pnfb(self) -> np.ndarray:
        """FIRFarrow.pnfb (src/Filters.jl:126): tapsPer𝜙 polynomials, ascending powers, shape (tapsPer𝜙, polyorder+1)."""
        if self._handle is None or self.kind != FARROW:
            raise MultirateHIPError(1, "pnfb() needs a bound FIRFarrow filter")
        out = np.zeros((self.tapsPerPhi, self.polyorder + 1), dtype=np.float64)
        _check(self._lib.mrhip_get_pnfb(self._handle, _ptr(out)))
        return out

    def tapsforphase(self, phase: float) -> np.ndarray:
        """tapsforphase(kernel::FIRArbitrary, phase), src/Filters.jl:677-690, and
        tapsforphase(kernel::FIRFarrow, phase), src/Filters.jl:764-775."""
        if self._handle is None or self.kind not in (ARBITRARY, FARROW):
            raise MultirateHIPError(1, "tapsforphase() needs a bound FIRArbitrary or FIRFarrow filter")
        out = np.zeros(self.tapsPerPhi, dtype=self.h.dtype)
        fn = self._lib.mrhip_farrow_tapsforphase if self.kind == FARROW else self._lib.mrhip_arbitrary_tapsforphase
        _check(fn(self._handle, float(phase), _ptr(out)))
        return out

    def setphase(self, phi: float):
        """setphase(self::FIRFilter, 𝜙), src/Filters.jl:210-235, 𝜙 in [0, 1].  The reference's methods for
        FIRInterpolator/FIRRational use an undefined variable (:212); the evident intent -- the phase index that
        corresponds to the fraction 𝜙 of one input sample -- is implemented: 𝜙Idx = floor(𝜙*N𝜙) + 1, clipped to
        N𝜙.  FIRArbitrary: (α, 𝜙Idx) = modf(𝜙*N𝜙) as written (:217-222), stored as 𝜙Accumulator = 𝜙Idx + α clipped
        to [1, N𝜙+1).  FIRFarrow: 𝜙Idx = 𝜙*(N𝜙-1)+1 (:226)."""
        if not 0.0 <= phi <= 1.0:
            raise MultirateHIPError(1, "phase must be in [0, 1]")            # @assert, :211
        if self._handle is None:
            raise MultirateHIPError(1, "setphase needs a bound filter (call filt once, or bind())")
        st = self.state
        if self.kind in (INTERPOLATOR, RATIONAL):
            idx = min(int(phi * self.Nphi) + 1, self.Nphi)
            self.set_state(idx, st.inputDeficit, 1.0)
            return idx
        if self.kind == ARBITRARY:
            import math
            alpha, idx = math.modf(phi * self.Nphi)
            acc = min(max(idx + alpha, 1.0), math.nextafter(self.Nphi + 1.0, 0.0))
            self.set_state(1, st.inputDeficit, acc)
            return idx, alpha
        if self.kind == FARROW:
            acc = phi * (self.Nphi - 1) + 1
            self.set_state(1, st.inputDeficit, acc)
            return acc
        raise MultirateHIPError(5, f"setphase is not defined for {self.kernel_name}")

    def set_mod_form(self, julia03: bool):
        """FIRArbitrary / FIRFarrow: update()'s mod() (src/Filters.jl:668) as rem(y + rem(x, y), y) -- Julia Base before 0.4 -- instead
        of the exact remainder (``mrhip_set_mod_form``); identical for a power-of-two N𝜙."""
        if self._handle is None:
            raise MultirateHIPError(1, "set_mod_form needs a bound filter (call filt once, or bind())")
        _check(self._lib.mrhip_set_mod_form(self._handle, 1 if julia03 else 0))

    def set_timing(self, enabled=True):
        """True/False, or an int n > 1 to bracket every n-th compute launch only."""
        _check(self._lib.mrhip_set_timing(self._handle, int(enabled)))

    def timing_read(self):
        """(number of compute-kernel launches since the last read, sum of their durations in ms)."""
        n, ms = C.c_int64(0), C.c_double(0.0)
        _check(self._lib.mrhip_timing_read(self._handle, C.byref(n), C.byref(ms)))
        return int(n.value), float(ms.value)

    def last_kernel_name(self) -> str:
        return self._lib.mrhip_last_kernel_name(self._handle).decode()

    def schedule_info(self) -> dict:
        """How the phase schedule of this FIRArbitrary / FIRFarrow filter (update(), src/Filters.jl:663-673) has been
        evaluated so far: on the device, by the closed form of a detected cycle, or by the host's serial loop."""
        v = (C.c_int64 * 8)()
        _check(self._lib.mrhip_schedule_info(self._handle, v, 8))
        keys = ("device_ok", "ncand", "nwin", "period", "host_steps", "periodic_steps", "device_pieces", "fallback_pieces")
        return dict(zip(keys, (int(a) for a in v)))

    # -- bookkeeping (src/Filters.jl:352-422)
    def outputlength(self, inputlength: int) -> int:
        if self._handle is None:
            self_state = self.state
            if self.kind == STANDARD:
                return inputlength
            if self.kind == INTERPOLATOR:
                return self.interpolation * inputlength
            if self.kind in (ARBITRARY, FARROW):
                import math
                return int(math.ceil((inputlength - self_state.inputDeficit + 1) * self.rate))
            return self._lib.mrhip_outputlength_ratio(inputlength, self.interpolation, self.decimation, 1)
        return self._lib.mrhip_outputlength(self._handle, inputlength)

    def inputlength(self, outputlength: int) -> int:
        if self._handle is None:
            if self.kind in (ARBITRARY, FARROW):
                raise MultirateHIPError(5, "inputlength is not defined for FIRArbitrary in the reference")
            return self._lib.mrhip_inputlength_ratio(outputlength, self.interpolation, self.decimation, 1)
        n = self._lib.mrhip_inputlength(self._handle, outputlength)
        if n < 0:
            raise MultirateHIPError(5, "inputlength is not defined for FIRArbitrary in the reference")
        return n

    def advance_state(self, inputlength: int) -> int:
        """Advance the stream state as a ``filt`` call over ``inputlength`` samples would, without data and without
        touching the history; returns the outputs that call would have written (``mrhip_advance_state``)."""
        if self._handle is None:
            raise MultirateHIPError(1, "advance_state needs a bound filter (call filt once, or bind())")
        n = self._lib.mrhip_advance_state(self._handle, int(inputlength))
        if n < 0:
            raise MultirateHIPError(1, self._lib.mrhip_last_error().decode("utf-8", "replace"))
        return n

    def next_output_count(self, inputlength: int) -> int:
        if self._handle is None:
            raise MultirateHIPError(1, "next_output_count needs a bound filter")
        return self._lib.mrhip_next_output_count(self._handle, inputlength)

    def reset(self):
        """reset(self::FIRFilter), src/Filters.jl:256-260."""
        if self._handle is not None:
            _check(self._lib.mrhip_reset(self._handle))
        return self

    # -- the hot path
    def _shape(self, x):
        if x.ndim == 1:
            return 1, x.shape[0], True
        if x.ndim == 2:
            return x.shape[0], x.shape[1], False
        raise MultirateHIPError(1, "x must be (n,) or (nchannels, n)")

    def filt_into(self, buffer, x) -> int:
        """filt!(buffer, self, x): writes per-channel outputs into ``buffer`` (same container kind as
        ``x``), returns the per-channel output count."""
        nw = C.c_int64(0)
        if _is_torch(x):
            if not x.is_cuda:
                raise MultirateHIPError(1, "torch input must live on the GPU (use numpy for host data)")
            if x.stride(-1) != 1 or buffer.stride(-1) != 1:
                raise MultirateHIPError(1, "x and buffer must be contiguous along time (planar channels)")
            nch, n, _ = self._shape(x)
            self._ensure(_torch_np_dtype(x.dtype), nch)
            if _torch_np_dtype(buffer.dtype) != self.output_dtype:
                raise MultirateHIPError(1, f"buffer dtype must be {self.output_dtype}")
            if x.device.index != self.device or buffer.device != x.device:
                raise MultirateHIPError(1, "x / buffer are not on the filter's device")
            cap = buffer.shape[-1]
            # channel strides in samples: views such as big[:, a:b] are filtered in place, no copy
            xs = x.stride(0) if x.ndim == 2 and nch > 1 else n
            ys = buffer.stride(0) if buffer.ndim == 2 and nch > 1 else cap
            if buffer.ndim != x.ndim or (x.ndim == 2 and buffer.shape[0] != nch):
                raise MultirateHIPError(1, "buffer must have one row per channel")
            stream = torch.cuda.current_stream(x.device).cuda_stream
            _check(self._lib.mrhip_filt_device(self._handle, C.c_void_p(x.data_ptr()), n, xs,
                                               C.c_void_p(buffer.data_ptr()), cap, ys, C.byref(nw),
                                               C.c_void_p(stream)))
            return nw.value
        x = np.asarray(x)
        if not x.flags.c_contiguous:
            x = np.ascontiguousarray(x)
        nch, n, _ = self._shape(x)
        self._ensure(x.dtype, nch)
        if not isinstance(buffer, np.ndarray) or buffer.dtype != self.output_dtype or not buffer.flags.c_contiguous:
            raise MultirateHIPError(1, f"buffer must be a C-contiguous numpy array of {self.output_dtype}")
        cap = buffer.shape[-1]
        _check(self._lib.mrhip_filt_host(self._handle, _ptr(x), n, n, _ptr(buffer), cap, cap, C.byref(nw)))
        return nw.value

    def outputlength_bound(self, inputlength: int) -> int:
        """The largest per-channel output count a call of ``inputlength`` samples can have whatever the stream state
        (``mrhip_outputlength_bound``): the room ``filt_into_async`` -- and every call captured into a HIP graph -- needs."""
        if self._handle is None:
            raise MultirateHIPError(1, "outputlength_bound needs a bound filter (call filt once, or bind())")
        return self._lib.mrhip_outputlength_bound(self._handle, int(inputlength))

    def filt_into_async(self, buffer, x, count=None, after: "FIRFilter" = None) -> None:
        """filt!(buffer, self, x) planned ON THE DEVICE from the device-resident stream state (``mrhip_filt_device_async``):
        nothing is returned and the host never waits, so a loop of such calls only enqueues -- and captures into a HIP
        graph at any fixed chunk size, for every kind.  ``buffer`` must hold ``outputlength_bound(n)`` samples per channel;
        ``count`` (optional) is a one-element int64 CUDA tensor that receives the per-channel output count in stream order.
        ``sync_state()`` returns the last call's count and brings the host-side view of the state up to date.
        ``after=prev`` makes this a CHAINED call (``mrhip_filt_device_chained``): ``x`` is the buffer ``prev``'s latest
        asynchronous or captured call wrote, its length is that call's count (known on the device only) and ``x.shape[-1]`` is the
        bound the launch is sized for (``prev.outputlength_bound(...)``)."""
        if not _is_torch(x) or not x.is_cuda:
            raise MultirateHIPError(1, "filt_into_async takes torch device tensors")
        if x.stride(-1) != 1 or buffer.stride(-1) != 1:
            raise MultirateHIPError(1, "x and buffer must be contiguous along time (planar channels)")
        nch, n, _ = self._shape(x)
        self._ensure(_torch_np_dtype(x.dtype), nch)
        if _torch_np_dtype(buffer.dtype) != self.output_dtype:
            raise MultirateHIPError(1, f"buffer dtype must be {self.output_dtype}")
        if buffer.ndim != x.ndim or (x.ndim == 2 and buffer.shape[0] != nch):
            raise MultirateHIPError(1, "buffer must have one row per channel")
        cap = buffer.shape[-1]
        xs = x.stride(0) if x.ndim == 2 and nch > 1 else n
        ys = buffer.stride(0) if buffer.ndim == 2 and nch > 1 else cap
        cptr = None
        if count is not None:
            if not (_is_torch(count) and count.is_cuda and count.dtype == torch.int64 and count.numel() >= 1):
                raise MultirateHIPError(1, "count must be an int64 CUDA tensor")
            cptr = C.c_void_p(count.data_ptr())
        stream = torch.cuda.current_stream(x.device).cuda_stream
        if after is not None:
            if after._handle is None:
                raise MultirateHIPError(1, "after= needs a bound filter")
            _check(self._lib.mrhip_filt_device_chained(self._handle, after._handle, C.c_void_p(x.data_ptr()), n, xs, C.c_void_p(buffer.data_ptr()),
                                                       cap, ys, cptr, C.c_void_p(stream)))
            return
        _check(self._lib.mrhip_filt_device_async(self._handle, C.c_void_p(x.data_ptr()), n, xs, C.c_void_p(buffer.data_ptr()),
                                                 cap, ys, cptr, C.c_void_p(stream)))

    def sync_state(self) -> int:
        """Wait for the filter's enqueued calls (after a graph capture: for the device) and take the device-resident
        stream state over into the host object; returns the per-channel output count of the last call
        (``mrhip_sync_state``).  Raises if a device-planned call failed since the last ``sync_state``."""
        nw = C.c_int64(0)
        _check(self._lib.mrhip_sync_state(self._handle, C.byref(nw)))
        return nw.value

    def set_history_device(self, hist) -> None:
        """FIRFilter.history from a torch CUDA tensor, asynchronously on the current stream (``mrhip_set_history_device``)."""
        if not _is_torch(hist) or not hist.is_cuda or not hist.is_contiguous():
            raise MultirateHIPError(1, "set_history_device takes a contiguous torch CUDA tensor")
        if _torch_np_dtype(hist.dtype) != self._tx or hist.numel() != self._nch * self.historyLen:
            raise MultirateHIPError(1, f"history must hold {self._nch} x {self.historyLen} samples of {self._tx}")
        stream = torch.cuda.current_stream(hist.device).cuda_stream
        _check(self._lib.mrhip_set_history_device(self._handle, C.c_void_p(hist.data_ptr()), C.c_void_p(stream)))

    def filt_into_chunked(self, buffer, x, chunk: int) -> int:
        """The loop ``for a in range(0, n, chunk): filt!(buffer[k:], self, x[a:a+chunk])`` issued by the library in one
        call (torch device tensors only): streaming with the state carried on the device, bit-identical to the
        caller's own loop, without its per-call host overhead.  Returns the total per-channel output count."""
        if not _is_torch(x) or not x.is_cuda:
            raise MultirateHIPError(1, "filt_into_chunked takes torch device tensors")
        if x.stride(-1) != 1 or buffer.stride(-1) != 1:
            raise MultirateHIPError(1, "x and buffer must be contiguous along time (planar channels)")
        nch, n, _ = self._shape(x)
        self._ensure(_torch_np_dtype(x.dtype), nch)
        if _torch_np_dtype(buffer.dtype) != self.output_dtype:
            raise MultirateHIPError(1, f"buffer dtype must be {self.output_dtype}")
        cap = buffer.shape[-1]
        xs = x.stride(0) if x.ndim == 2 and nch > 1 else n
        ys = buffer.stride(0) if buffer.ndim == 2 and nch > 1 else cap
        nw = C.c_int64(0)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _check(self._lib.mrhip_filt_device_chunked(self._handle, C.c_void_p(x.data_ptr()), n, xs, int(chunk),
                                                   C.c_void_p(buffer.data_ptr()), cap, ys, C.byref(nw), C.c_void_p(stream)))
        return nw.value

    def open_ring(self, dtype=None, nchannels: int = None) -> "ChunkRing":
        """A ring of arriving chunks on this filter (``mrhip_ring_open``): one resident kernel instead of a launch per chunk.  The
        filter must be bound (``bind()`` / a first ``filt``), or ``dtype`` and ``nchannels`` given.  Use as a context manager."""
        if dtype is not None:
            self._ensure(np.dtype(dtype), int(nchannels or 1))
        if self._handle is None:
            raise MultirateHIPError(1, "open_ring needs a bound filter (bind(), or pass dtype and nchannels)")
        return ChunkRing(self)

    def filt(self, x):
        """filt(self, x): allocate the output, run filt!, trim to the samples written
        (src/Filters.jl:475,519,577,633,744)."""
        # FIRArbitrary / FIRFarrow: like the reference (Filters.jl:744-752, 841-849) allocate the outputlength
        # estimate (+2: it is only a guess there) and trim to the count filt! returns -- the exact count would need
        # the whole serial phase recurrence up front, which the library instead pipelines with the kernels.
        est_mode = self.kind in (ARBITRARY, FARROW)
        if _is_torch(x):
            nch, n, one = self._shape(x)
            self._ensure(_torch_np_dtype(x.dtype), nch)
            cnt = max(self.outputlength(n), 0) + 2 if est_mode else max(self.next_output_count(n), 0)
            y = torch.empty((nch, cnt), dtype=_np_torch_dtype(self.output_dtype), device=x.device)
            got = 0
            if n > 0:
                got = self.filt_into(y[0] if one else y, x if x.stride(-1) == 1 else x.contiguous())
                assert est_mode or got == cnt
            if est_mode:
                y = y[:, :got]
            return y[0] if one else y
        x = np.ascontiguousarray(x)
        nch, n, one = self._shape(x)
        self._ensure(x.dtype, nch)
        cnt = max(self.outputlength(n), 0) + 2 if est_mode else max(self.next_output_count(n), 0)
        y = np.empty((nch, cnt), dtype=self.output_dtype)
        got = 0
        if n > 0:
            got = self.filt_into(y, x)
            assert est_mode or got == cnt
        if est_mode:
            y = np.ascontiguousarray(y[:, :got])
        return y[0] if one else y


class ChunkRing:
    """The reference's streaming loop ``for x_i in chunks: y_i = filt(self, x_i)`` (README.md:87-141) fed through a ring into ONE
    resident kernel (``mrhip_ring_*``): ``push(buffer, x)`` is ``filt!(buffer, self, x)`` for the next arriving chunk -- it returns
    the per-channel output count at once and the chunk's number; ``wait(seq)`` / ``drain()`` block until outputs are complete;
    ``close()`` hands the stream back to the filter.  ``x`` must be complete on the device when pushed (the resident kernel is not
    ordered behind any stream) and ``x`` / ``buffer`` must stay untouched until the chunk is complete."""

    def __init__(self, f: "FIRFilter"):
        self.filter = f
        self._lib = f._lib
        h = C.c_void_p()
        _check(self._lib.mrhip_ring_open(f._handle, C.byref(h)))
        self._h = h
        self._push = self._lib.mrhip_ring_push                   # (push_raw: no attribute look-ups, no byref objects per call)
        self._nw, self._seq = C.byref(C.c_int64(0)), C.byref(C.c_uint64(0))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> dict:
        v = (C.c_int64 * 9)()
        _check(self._lib.mrhip_ring_info(self._h, v, 9))
        return {"resident": bool(v[0]), "depth": v[1], "pushed": v[2], "restarts": v[3], "steps_per_grab": v[4], "outputs_per_step": v[5],
                "shrunk": v[6], "workgroups": v[7], "xcds": v[8]}

    def _args(self, buffer, x):
        f = self.filter
        if not _is_torch(x) or not x.is_cuda or not _is_torch(buffer) or not buffer.is_cuda:
            raise MultirateHIPError(1, "a ring takes torch device tensors")
        if x.stride(-1) != 1 or buffer.stride(-1) != 1:
            raise MultirateHIPError(1, "x and buffer must be contiguous along time (planar channels)")
        nch, n, _ = f._shape(x)
        if _torch_np_dtype(x.dtype) != f._tx or nch != f._nch:
            raise MultirateHIPError(1, "x does not match the sample type / channel count the filter is bound to")
        if _torch_np_dtype(buffer.dtype) != f.output_dtype:
            raise MultirateHIPError(1, f"buffer dtype must be {f.output_dtype}")
        if x.device.index != f.device or buffer.device != x.device:
            raise MultirateHIPError(1, "x / buffer are not on the filter's device")
        if buffer.ndim != x.ndim or (x.ndim == 2 and buffer.shape[0] != nch):
            raise MultirateHIPError(1, "buffer must have one row per channel")
        cap = buffer.shape[-1]
        xs = x.stride(0) if x.ndim == 2 and nch > 1 else n
        ys = buffer.stride(0) if buffer.ndim == 2 and nch > 1 else cap
        return n, xs, cap, ys

    def push(self, buffer, x):
        """filt!(buffer, self, x) for the next chunk: (per-channel output count, chunk number)"""
        n, xs, cap, ys = self._args(buffer, x)
        nw, seq = C.c_int64(0), C.c_uint64(0)
        _check(self._lib.mrhip_ring_push(self._h, C.c_void_p(x.data_ptr()), n, xs, C.c_void_p(buffer.data_ptr()), cap, ys, C.byref(nw), C.byref(seq)))
        return nw.value, seq.value

    def push_raw(self, x_ptr: int, n: int, x_stride: int, y_ptr: int, y_capacity: int, y_stride: int):
        """``push`` for a caller that already holds device addresses (a loop in another language, a preplanned stream): one
        ``mrhip_ring_push`` and nothing else -- no tensor checks; the same contract, unchecked: (count, chunk number)"""
        nw, seq = self._nw, self._seq
        _check(self._push(self._h, x_ptr, n, x_stride, y_ptr, y_capacity, y_stride, nw, seq))
        return nw._obj.value, seq._obj.value

    def push_chunks(self, buffer, x, chunk: int):
        """the library's loop of ``push`` over consecutive ``chunk``-sample pieces of a resident signal, outputs back to back in
        ``buffer``: (total per-channel output count, number of the last chunk)"""
        n, xs, cap, ys = self._args(buffer, x)
        nw, seq = C.c_int64(0), C.c_uint64(0)
        _check(self._lib.mrhip_ring_push_chunks(self._h, C.c_void_p(x.data_ptr()), n, xs, int(chunk), C.c_void_p(buffer.data_ptr()), cap, ys,
                                                C.byref(nw), C.byref(seq)))
        return nw.value, seq.value

    def wait(self, seq: int) -> None:
        _check(self._lib.mrhip_ring_wait(self._h, C.c_uint64(seq)))

    def drain(self) -> None:
        _check(self._lib.mrhip_ring_drain(self._h))

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            _check(self._lib.mrhip_ring_close(h))


class ShardedFIRFilter:
    """One FIRFilter whose channels are split over several GPUs of THIS process (``mrhip_sharded_*``; the one-process counterpart of
    sharding.py's one-rank-per-GPU ``ChannelShardedFilter``): shard i holds channels ``[start_i, start_i + count_i)`` on
    ``devices[i]``, no exchange during compute, one gather of the outputs at the end.

    ``filt(X)`` with a host (numpy) matrix ``(nchannels, n)`` filters every column split on all devices at once and returns the host
    result; ``filt_shards(xs)`` takes one CUDA tensor per shard (on that shard's device) and returns the per-shard outputs, which
    ``gather(ys, device)`` brings together on one device (peer copies)."""

    def __init__(self, h, ratio, nchannels: int, devices, *, Nphi: int = 32, polyorder=None, dtype=np.float32):
        self._lib = load_library()
        self.h = _as_taps(h)
        self._tx = np.dtype(dtype)
        self.nchannels = int(nchannels)
        self.devices = [int(d) for d in devices]
        ctor, num, den, rate = 0, 1, 1, 0.0
        if isinstance(ratio, float):
            ctor, rate = (2 if polyorder is not None else 1), float(ratio)
        else:
            r = Fraction(*ratio) if isinstance(ratio, tuple) else Fraction(ratio)
            num, den = r.numerator, r.denominator
        devs = (C.c_int * len(self.devices))(*self.devices)
        out = C.c_void_p()
        _check(self._lib.mrhip_sharded_create(ctor, _ptr(self.h), len(self.h), _NP2DT[self.h.dtype], num, den, rate, int(Nphi),
                                              int(polyorder or 0), _NP2DT[self._tx], self.nchannels, devs, len(self.devices), C.byref(out)))
        self._h = out
        self.output_dtype = _DT2NP[self._lib.mrhip_output_dtype(_NP2DT[self.h.dtype], _NP2DT[self._tx])]
        self.shards = []
        for i in range(self._lib.mrhip_sharded_nshards(self._h)):
            st, cnt, dev = C.c_int64(0), C.c_int64(0), C.c_int(0)
            _check(self._lib.mrhip_sharded_shard(self._h, i, C.byref(st), C.byref(cnt), C.byref(dev), None))
            self.shards.append((st.value, cnt.value, dev.value))

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.mrhip_sharded_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        _check(self._lib.mrhip_sharded_reset(self._h))
        return self

    def outputlength(self, n: int) -> int:
        return self._lib.mrhip_sharded_outputlength(self._h, int(n))

    def next_output_count(self, n: int) -> int:
        return self._lib.mrhip_sharded_next_output_count(self._h, int(n))

    def filt(self, X):
        """filt(self, X) for a host matrix with one channel per row: (nchannels, n) -> (nchannels, n_out)"""
        X = np.ascontiguousarray(X, dtype=self._tx)
        if X.ndim != 2 or X.shape[0] != self.nchannels:
            raise MultirateHIPError(1, f"X must be ({self.nchannels}, n)")
        n = X.shape[1]
        cap = max(self.outputlength(n), 0) + 2
        Y = np.empty((self.nchannels, cap), dtype=self.output_dtype)
        nw = C.c_int64(0)
        _check(self._lib.mrhip_sharded_filt_host(self._h, _ptr(X), n, n, _ptr(Y), cap, cap, C.byref(nw)))
        return np.ascontiguousarray(Y[:, :nw.value])

    def filt_shards(self, xs):
        """filt on device-resident data: xs[i] a (count_i, n) CUDA tensor on shard i's device; returns the per-shard outputs
        (asynchronous on the shards' streams: ``synchronize()`` or ``gather`` + ``synchronize()`` before reading)"""
        n = None
        for (st, cnt, dev), x in zip(self.shards, xs):
            if cnt and (not _is_torch(x) or not x.is_cuda or x.device.index != dev or x.shape != (cnt, x.shape[-1]) or x.stride(-1) != 1):
                raise MultirateHIPError(1, "xs[i] must be a (count_i, n) CUDA tensor on shard i's device")
            if cnt:
                n = x.shape[-1] if n is None else n
                if x.shape[-1] != n:
                    raise MultirateHIPError(1, "every shard takes the same number of samples per channel")
        cap = max(self.outputlength(n), 0) + 2
        ys = [torch.empty((cnt, cap), dtype=_np_torch_dtype(self.output_dtype), device=f"cuda:{dev}") if cnt else None for (st, cnt, dev) in self.shards]
        k = len(self.shards)
        xp = (C.c_void_p * k)(*[C.c_void_p(x.data_ptr()) if cnt else None for (st, cnt, dev), x in zip(self.shards, xs)])
        yp = (C.c_void_p * k)(*[C.c_void_p(y.data_ptr()) if y is not None else None for y in ys])
        xst = (C.c_int64 * k)(*[x.stride(0) if cnt else 0 for (st, cnt, dev), x in zip(self.shards, xs)])
        yst = (C.c_int64 * k)(*[cap] * k)
        nw = C.c_int64(0)
        # the shards run on private streams: each behind torch's current stream on its device (xs[i] may still be written there, ys[i]
        # came from the caching allocator in that stream's order) ...
        self._order(self._lib.mrhip_sharded_wait_stream)
        _check(self._lib.mrhip_sharded_filt_device(self._h, xp, n, xst, yp, cap, yst, C.byref(nw)))
        # ... and torch's stream behind them: later torch kernels see the outputs, the allocator may re-use ys[i] in stream order
        self._order(self._lib.mrhip_sharded_signal_stream)
        return [y[:, :nw.value] if y is not None else None for y in ys]

    def _order(self, fn):
        for i, (st, cnt, dev) in enumerate(self.shards):
            if cnt:
                _check(fn(self._h, i, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))

    def gather(self, ys, device: int):
        """(nchannels, n_out) on ``device`` from the per-shard outputs (views as ``filt_shards`` returns them)"""
        n_out = next(y.shape[-1] for y in ys if y is not None)
        out = torch.empty((self.nchannels, n_out), dtype=_np_torch_dtype(self.output_dtype), device=f"cuda:{device}")
        k = len(self.shards)
        yp = (C.c_void_p * k)(*[C.c_void_p(y.data_ptr()) if y is not None else None for y in ys])
        yst = (C.c_int64 * k)(*[y.stride(0) if y is not None else 0 for y in ys])
        self._order(self._lib.mrhip_sharded_wait_stream)         # (ys may come from torch kernels; `out` from the allocator of its device:)
        for i, (st, cnt, dev) in enumerate(self.shards):
            if cnt and dev != int(device):
                # a shard on another device writes into `out`: behind the destination device's current stream too (an event recorded
                # there, waited for by the shard's stream -- HIP events order streams across devices)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(int(device)))
                with torch.cuda.device(dev):
                    torch.cuda.current_stream(dev).wait_event(ev)
                _check(self._lib.mrhip_sharded_wait_stream(self._h, i, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        _check(self._lib.mrhip_sharded_gather(self._h, yp, n_out, yst, C.c_void_p(out.data_ptr()), n_out, int(device)))
        self._order(self._lib.mrhip_sharded_signal_stream)
        for i, (st, cnt, dev) in enumerate(self.shards):         # the destination's stream behind every shard's copy
            if cnt and dev != int(device):
                ev = torch.cuda.Event()
                with torch.cuda.device(dev):
                    ev.record(torch.cuda.current_stream(dev))
                torch.cuda.current_stream(int(device)).wait_event(ev)
        return out

    def synchronize(self):
        _check(self._lib.mrhip_sharded_synchronize(self._h))


def filt_multi(filters, xs):
    """``[filt(f, x) for f, x in zip(filters, xs)]`` for INDEPENDENT FIRFilter objects (each its own phase, deficit, history
    and call length: the reference's one-FIRFilter-per-signal streaming usage, README.md:87-141) issued as ONE launch
    (``mrhip_filt_device_multi``).  ``xs`` are torch CUDA tensors, ``(n_i,)`` or ``(nchannels_i, n_i)``, contiguous."""
    if not filters or len(filters) != len(xs):
        raise MultirateHIPError(1, "filt_multi takes as many signals as filters")
    lib = load_library()
    n = len(filters)
    ys, shapes = [], []
    for f, x in zip(filters, xs):
        if not _is_torch(x) or not x.is_cuda or not x.is_contiguous():
            raise MultirateHIPError(1, "filt_multi takes contiguous torch CUDA tensors")
        nch, m, one = f._shape(x)
        f._ensure(_torch_np_dtype(x.dtype), nch)
        cnt = max(f.next_output_count(m), 0)
        ys.append(torch.empty((nch, cnt), dtype=_np_torch_dtype(f.output_dtype), device=x.device))
        shapes.append((m, cnt, one))
    hs = (C.c_void_p * n)(*[f._handle.value for f in filters])
    xp = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
    yp = (C.c_void_p * n)(*[y.data_ptr() for y in ys])
    xl = (C.c_int64 * n)(*[s[0] for s in shapes])
    yc = (C.c_int64 * n)(*[s[1] for s in shapes])
    nw = (C.c_int64 * n)()
    stream = torch.cuda.current_stream(xs[0].device).cuda_stream
    _check(lib.mrhip_filt_device_multi(hs, n, xp, xl, yp, yc, nw, C.c_void_p(stream)))
    out = []
    for y, (m, cnt, one), got in zip(ys, shapes, nw):
        assert got == cnt
        out.append(y[0] if one else y)
    return out


class MultiStream:
    """A fixed set of independent FIRFilter streams and their (fixed) device buffers, prepared once so that every
    ``run()`` is ONE library call = one launch (``mrhip_filt_device_multi``): the per-round form of ``filt_multi`` for a
    loop over arriving chunks that land in the same buffers.  ``ys[i]`` must hold ``filters[i].next_output_count`` of every
    round (e.g. ``outputlength_bound``)."""

    def __init__(self, filters, ys, xs):
        self._lib = load_library()
        self.filters, self.ys, self.xs = list(filters), list(ys), list(xs)
        n = self.n = len(self.filters)
        for f, x in zip(self.filters, self.xs):
            nch, m, _ = f._shape(x)
            f._ensure(_torch_np_dtype(x.dtype), nch)
        self._hs = (C.c_void_p * n)(*[f._handle.value for f in self.filters])
        self._xp = (C.c_void_p * n)(*[x.data_ptr() for x in self.xs])
        self._yp = (C.c_void_p * n)(*[y.data_ptr() for y in self.ys])
        self._xl = (C.c_int64 * n)(*[x.shape[-1] for x in self.xs])
        self._yc = (C.c_int64 * n)(*[y.shape[-1] for y in self.ys])
        self.counts = (C.c_int64 * n)()

    def run(self):
        stream = torch.cuda.current_stream(self.xs[0].device).cuda_stream
        _check(self._lib.mrhip_filt_device_multi(self._hs, self.n, self._xp, self._xl, self._yp, self._yc, self.counts, C.c_void_p(stream)))
        return self.counts


# ---- free functions with the reference's names -------------------------------------------------
def filt(a, x, ratio=Fraction(1, 1), Nphi: int = 32, polyorder=None, **kw):
    """filt(self::FIRFilter, x)                                  src/Filters.jl:475,519,577,633,744,841
       filt(h::Vector, x::Vector, ratio::Rational)               src/Filters.jl:858-861
       filt(h::Vector, x::Vector, rate::Float, N𝜙)               src/Filters.jl:864-867
       filt(h::Vector, x::Vector, rate::Float, N𝜙, polyorder)    src/Filters.jl:870-873"""
    if isinstance(a, FIRFilter):
        return a.filt(x)
    f = FIRFilter(a, ratio, Nphi, polyorder, **kw)
    try:
        return f.filt(x)
    finally:
        f.close()


class FilterCascade:
    """Device-resident cascade (the reference chains ``filt`` calls by hand): ``filt`` runs the stages
    back to back on the current stream through the library's cascade object (``mrhip_cascade_*``) -- the intermediate
    signals live in device buffers the cascade owns and never leave HBM and, because every output count is
    closed-form on the host, nothing is read back between stages.  Each stage is an ordinary stateful FIRFilter, so
    chunked calls continue the stream exactly like calling the stages by hand (e.g. decimate 1//4, then 147//160).
    numpy inputs take the stages' host path one after the other."""

    def __init__(self, *stages: "FIRFilter"):
        if not stages or not all(isinstance(f, FIRFilter) for f in stages):
            raise MultirateHIPError(1, "FilterCascade takes one or more FIRFilter stages")
        self.stages = tuple(stages)
        self._lib = load_library()
        self._handle = None

    def _ensure(self, tx, nch: int):
        if self._handle is not None:
            # the bound cascade only knows the sample type and channel count of its first call: a later call with
            # another shape must raise like a lone FIRFilter does, not read or write past the caller's buffers
            self.stages[0]._ensure(np.dtype(tx), nch)
            return
        for f in self.stages:                      # stage i+1's sample type is stage i's output type
            f._ensure(np.dtype(tx), nch)
            tx = f.output_dtype
        arr = (C.c_void_p * len(self.stages))(*[f._handle for f in self.stages])
        out = C.c_void_p()
        _check(self._lib.mrhip_cascade_create(arr, len(self.stages), C.byref(out)))
        self._handle = out

    @property
    def output_dtype(self):
        return self.stages[-1].output_dtype

    def filt(self, x):
        if not _is_torch(x):
            for f in self.stages:
                x = f.filt(x)
            return x
        if not x.is_cuda:
            raise MultirateHIPError(1, "torch input must live on the GPU (use numpy for host data)")
        one = x.ndim == 1
        nch, n = (1, x.shape[0]) if one else (x.shape[0], x.shape[1])
        self._ensure(_torch_np_dtype(x.dtype), nch)
        if x.stride(-1) != 1:
            x = x.contiguous()
        cnt = max(self._lib.mrhip_cascade_next_output_count(self._handle, n), 0)
        y = torch.empty((nch, cnt), dtype=_np_torch_dtype(self.output_dtype), device=x.device)
        nw = C.c_int64(0)
        xs = x.stride(0) if (not one and nch > 1) else n
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _check(self._lib.mrhip_cascade_filt_device(self._handle, C.c_void_p(x.data_ptr()), n, xs, C.c_void_p(y.data_ptr()),
                                                   cnt, max(cnt, 1), C.byref(nw), C.c_void_p(stream)))
        assert nw.value == cnt
        return y[0] if one else y

    def filt_into(self, buffer, x) -> int:
        """The chain into a caller-owned device buffer (one row per channel); returns the per-channel output count.  This is
        the form a HIP graph can capture: every stage must then map its input length to the same count on every replay
        (inputlength * L a multiple of M), the buffer needs room for the last stage's ``outputlength_bound`` and one plain
        call of the same size must have run before (it allocates the buffers between the stages)."""
        if not (_is_torch(x) and x.is_cuda and _is_torch(buffer) and buffer.is_cuda):
            raise MultirateHIPError(1, "FilterCascade.filt_into takes device tensors")
        one = x.ndim == 1
        nch, n = (1, x.shape[0]) if one else (x.shape[0], x.shape[1])
        self._ensure(_torch_np_dtype(x.dtype), nch)
        if x.stride(-1) != 1 or buffer.stride(-1) != 1:
            raise MultirateHIPError(1, "x and buffer must be contiguous along time (planar channels)")
        if _torch_np_dtype(buffer.dtype) != self.output_dtype:
            raise MultirateHIPError(1, f"buffer dtype must be {self.output_dtype}")
        if buffer.ndim != x.ndim or (not one and buffer.shape[0] != nch):
            raise MultirateHIPError(1, "buffer must have one row per channel")
        cap = buffer.shape[-1]
        xs = x.stride(0) if (not one and nch > 1) else n
        ys = buffer.stride(0) if (not one and nch > 1) else cap
        nw = C.c_int64(0)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _check(self._lib.mrhip_cascade_filt_device(self._handle, C.c_void_p(x.data_ptr()), n, xs, C.c_void_p(buffer.data_ptr()),
                                                   cap, ys, C.byref(nw), C.c_void_p(stream)))
        return nw.value

    def outputlength_bound(self, inputlength: int) -> int:
        """Room per channel an asynchronous (or captured) call of ``inputlength`` samples needs: the stages' bounds in turn."""
        n = int(inputlength)
        for f in self.stages:
            n = f.outputlength_bound(n)
        return n

    def filt_into_async(self, buffer, x, count=None) -> None:
        """The chain with nothing returned to the host (``mrhip_cascade_filt_device_async``): the first stage is planned on the
        device, every later one takes its input length from the previous stage's count ON THE DEVICE -- so the chain captures into a
        HIP graph at any chunk size.  ``buffer`` holds ``outputlength_bound(n)`` samples per channel; ``count``: one-element int64 CUDA
        tensor for the chain's per-channel output count.  Before a capture run one plain call of the same size (buffers)."""
        if not (_is_torch(x) and x.is_cuda and _is_torch(buffer) and buffer.is_cuda):
            raise MultirateHIPError(1, "FilterCascade.filt_into_async takes device tensors")
        one = x.ndim == 1
        nch, n = (1, x.shape[0]) if one else (x.shape[0], x.shape[1])
        self._ensure(_torch_np_dtype(x.dtype), nch)
        if x.stride(-1) != 1 or buffer.stride(-1) != 1:
            raise MultirateHIPError(1, "x and buffer must be contiguous along time (planar channels)")
        if _torch_np_dtype(buffer.dtype) != self.output_dtype:
            raise MultirateHIPError(1, f"buffer dtype must be {self.output_dtype}")
        if buffer.ndim != x.ndim or (not one and buffer.shape[0] != nch):
            raise MultirateHIPError(1, "buffer must have one row per channel")
        cap = buffer.shape[-1]
        xs = x.stride(0) if (not one and nch > 1) else n
        ys = buffer.stride(0) if (not one and nch > 1) else cap
        cptr = None
        if count is not None:
            if not (_is_torch(count) and count.is_cuda and count.dtype == torch.int64 and count.numel() >= 1):
                raise MultirateHIPError(1, "count must be an int64 CUDA tensor")
            cptr = C.c_void_p(count.data_ptr())
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _check(self._lib.mrhip_cascade_filt_device_async(self._handle, C.c_void_p(x.data_ptr()), n, xs, C.c_void_p(buffer.data_ptr()),
                                                         cap, ys, cptr, C.c_void_p(stream)))

    def outputlength(self, inputlength: int) -> int:
        n = int(inputlength)
        for f in self.stages:
            n = f.outputlength(n)
        return n

    def reset(self):
        for f in self.stages:
            f.reset()
        return self

    def close(self):
        if getattr(self, "_handle", None):
            self._lib.mrhip_cascade_destroy(self._handle)
            self._handle = None
        for f in self.stages:
            f.close()

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                self._lib.mrhip_cascade_destroy(self._handle)
                self._handle = None
        except Exception:
            pass


def filt_(buffer, self: FIRFilter, x):
    """filt!(buffer, self, x).  Returns what the reference returns: ``buffer`` for FIRStandard /
    FIRInterpolator (src/Filters.jl:472,516), the number of samples written for FIRRational /
    FIRDecimator / FIRArbitrary (:574,:630,:741)."""
    n = self.filt_into(buffer, x)
    return buffer if self.kind in (STANDARD, INTERPOLATOR) else n


def outputlength(self: FIRFilter, inputlength: int) -> int:
    """outputlength(self::FIRFilter, inputlength), src/Filters.jl:383-385."""
    return self.outputlength(inputlength)


def inputlength(self: FIRFilter, outputlength: int) -> int:
    """inputlength(self::FIRFilter, outputlength), src/Filters.jl:403-422."""
    return self.inputlength(outputlength)


def reset(self: FIRFilter) -> FIRFilter:
    """reset(self::FIRFilter), src/Filters.jl:256-260."""
    return self.reset()


def setphase(self: FIRFilter, phi: float):
    """setphase(self::FIRFilter, 𝜙), src/Filters.jl:235."""
    return self.setphase(phi)


def tapsforphase(self: FIRFilter, phase: float) -> np.ndarray:
    """tapsforphase(kernel::FIRArbitrary, phase), src/Filters.jl:690; tapsforphase(kernel::FIRFarrow, phase), :775."""
    return self.tapsforphase(phase)
