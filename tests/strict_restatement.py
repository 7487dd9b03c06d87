"""A SECOND, independent restatement of the reference's hot path -- test infrastructure, pure NumPy scalar arithmetic.

Written straight from the Julia source, loop by loop and index by index (1-based indices kept; the C oracle under
oracle/ instead uses the closed-form output -> (phase, input index) map and 0-based windows), so that the bit-level
order of operations is pinned between two restatements that share no code and no formulation:

    unsafedot x4, shiftin!      /root/reference/src/support.jl:5-80
    taps2pfb                    src/Filters.jl:284-298
    constructors, kernel choice src/Filters.jl:20-24, 35-41, 52-58, 72-80, 105-117, 158-189
    outputlength, nextphase     src/Filters.jl:352-385, 433-439
    filt! x5 + update           src/Filters.jl:450-473, 489-517, 536-575, 598-631, 663-673, 693-742

Arithmetic: every product and every sum is ONE NumPy scalar operation in promote_type(Th, Tx)'s real type (np.float32
scalars round after each operation, like Julia's Float32), the first product initialises the accumulator (the Vector seam
variant starts from zero(...), support.jl:46), real taps x complex samples are two independent real chains (Julia's
Real*Complex multiplies the two parts, and complex + adds them part by part).  Nothing here is vectorised and nothing
calls the oracle: tests/test_second_restatement.py compares the two bit for bit.
"""
from __future__ import annotations

import math
from fractions import Fraction

import numpy as np


def _real_type(th, tx):
    f64 = np.dtype(th) == np.float64 or np.dtype(tx) in (np.dtype(np.float64), np.dtype(np.complex128))
    return np.float64 if f64 else np.float32


class _Num:
    """a sample or an accumulator of promote_type(Th, Tx): (re,) or (re, im), parts of the real type R"""
    __slots__ = ("p",)

    def __init__(self, parts):
        self.p = parts


def _widen(v, R, cplx):
    return _Num((R(v.real), R(v.imag))) if cplx else _Num((R(v),))


def _mul(t, x: _Num, R):           # a[i] * b[j]: tap (real) times sample, every part one rounded product
    tt = R(t)
    return _Num(tuple(tt * c for c in x.p))


def _add(a: _Num, b: _Num):        # dotprod += product: part by part, one rounded sum each
    return _Num(tuple(u + v for u, v in zip(a.p, b.p)))


def taps2pfb(h, Nphi):
    """src/Filters.jl:284-298, as written: rows filled from the LAST one up, column by column, zero(T) past the end"""
    hLen = len(h)
    T = int(math.ceil(hLen / Nphi))                      # iceil(hLen/N𝜙)
    pfb = np.zeros((T, Nphi), dtype=h.dtype)
    hIdx = 1
    for rowIdx in range(T, 0, -1):
        for colIdx in range(1, Nphi + 1):
            pfb[rowIdx - 1, colIdx - 1] = h.dtype.type(0) if hIdx > hLen else h[hIdx - 1]
            hIdx += 1
    return pfb


class Restated:
    """FIRFilter(h, ratio) / FIRFilter(h, rate, N𝜙) and filt(self, x), one channel, restated literally."""

    def __init__(self, h, ratio, Nphi=32, tx=np.float32):
        h = np.ascontiguousarray(h)
        self.th, self.tx = h.dtype, np.dtype(tx)
        self.cplx = self.tx.kind == "c"
        self.R = _real_type(self.th, self.tx)
        self.out_dtype = np.dtype({(np.float32, False): np.float32, (np.float64, False): np.float64,
                                   (np.float32, True): np.complex64, (np.float64, True): np.complex128}[(self.R, self.cplx)])
        if isinstance(ratio, float):                     # src/Filters.jl:183-189 -> FIRArbitrary, :105-117
            assert ratio > 0.0
            self.kind = "arbitrary"
            dh = np.concatenate([np.diff(h), np.zeros(1, dtype=h.dtype)])      # [diff(h), 0] in the tap type
            self.pfb, self.dpfb = taps2pfb(h, Nphi), taps2pfb(dh, Nphi)
            self.Nphi, self.T = Nphi, self.pfb.shape[0]
            self.acc, self.phiIdx, self.alpha = 1.0, 1, 0.0
            self.delta = Nphi / ratio
            self.rate = ratio
            self.inputDeficit, self.xIdx = 1, 1
            self.historyLen = self.T - 1
        else:                                            # src/Filters.jl:158-180
            r = Fraction(ratio)
            self.L, self.M = r.numerator, r.denominator
            if r == 1:
                self.kind, self.h, self.hLen = "standard", h[::-1].copy(), len(h)          # flipud, :21
                self.historyLen = self.hLen - 1
            elif self.L == 1:
                self.kind, self.h, self.hLen = "decimator", h[::-1].copy(), len(h)
                self.inputDeficit = 1
                self.historyLen = self.hLen - 1
            elif self.M == 1:
                self.kind = "interpolator"
                self.pfb = taps2pfb(h, self.L)
                self.T, self.Nphi = self.pfb.shape
                self.historyLen = self.T - 1
            else:
                self.kind = "rational"
                self.pfb = taps2pfb(h, self.L)
                self.T, self.Nphi = self.pfb.shape
                self.phiIdx, self.inputDeficit = 1, 1
                self.historyLen = self.T - 1
        self.history = [_widen(self.tx.type(0), self.R, self.cplx) for _ in range(self.historyLen)]      # zeros(historyLen), :177

    # ---- support.jl ----------------------------------------------------------------------------------
    def _dot_matrix(self, a, col, b, bLast):             # support.jl:5-14   (b: list of _Num, 1-based bLast)
        aLen = a.shape[0]
        base = bLast - aLen
        d = _mul(a[0, col - 1], b[base + 1 - 1], self.R)
        for i in range(2, aLen + 1):
            d = _add(d, _mul(a[i - 1, col - 1], b[base + i - 1], self.R))
        return d

    def _dot_matrix_seam(self, a, col, b, c, cLast):     # support.jl:16-31
        aLen = a.shape[0]
        assert len(b) == aLen - 1 and cLast < aLen
        d = _mul(a[0, col - 1], b[cLast - 1], self.R)
        for i in range(2, aLen - cLast + 1):
            d = _add(d, _mul(a[i - 1, col - 1], b[i + cLast - 1 - 1], self.R))
        for i in range(1, cLast + 1):
            d = _add(d, _mul(a[aLen - cLast + i - 1, col - 1], c[i - 1], self.R))
        return d

    def _dot_vector(self, a, b, bLast):                  # support.jl:33-42
        aLen = len(a)
        base = bLast - aLen
        d = _mul(a[0], b[base + 1 - 1], self.R)
        for i in range(2, aLen + 1):
            d = _add(d, _mul(a[i - 1], b[base + i - 1], self.R))
        return d

    def _dot_vector_seam(self, a, b, c, cLast):          # support.jl:44-55: starts from zero(a[1]*b[1])
        aLen = len(a)
        d = _Num(tuple(self.R(0) for _ in range(2 if self.cplx else 1)))
        for i in range(1, aLen - cLast + 1):
            d = _add(d, _mul(a[i - 1], b[i + cLast - 1 - 1], self.R))
        for i in range(1, cLast + 1):
            d = _add(d, _mul(a[aLen - cLast + i - 1], c[i - 1], self.R))
        return d

    def _shiftin(self, b):                               # support.jl:61-80
        a = self.history
        aLen, bLen = len(a), len(b)
        if bLen >= aLen:
            a[:] = b[bLen - aLen:]
        else:
            for i in range(1, aLen - bLen + 1):
                a[i - 1] = a[i + bLen - 1]
            bIdx = 1
            for i in range(aLen - bLen + 1, aLen + 1):
                a[i - 1] = b[bIdx - 1]
                bIdx += 1

    # ---- Filters.jl ----------------------------------------------------------------------------------
    def _nextphase(self, p):                             # :433-439
        n = p + self.M % self.L
        return n - self.L if n > self.L else n

    def filt(self, x):
        x = np.ascontiguousarray(x, dtype=self.tx)
        xs = [_widen(v, self.R, self.cplx) for v in x]
        xLen = len(xs)
        out = []
        if self.kind == "standard":                      # :450-473
            crit = min(self.hLen, xLen)
            for yIdx in range(1, crit + 1):
                out.append(self._dot_vector_seam(self.h, self.history, xs, yIdx))
            for yIdx in range(crit + 1, xLen + 1):
                out.append(self._dot_vector(self.h, xs, yIdx))
            self._shiftin(xs)
        elif self.kind == "interpolator":                # :489-517
            outLen = self.L * xLen
            crit = min(self.historyLen * self.L, outLen)
            inputIdx, phi = 1, 1
            for yIdx in range(1, outLen + 1):
                if yIdx <= crit:
                    out.append(self._dot_matrix_seam(self.pfb, phi, self.history, xs, inputIdx))
                else:
                    out.append(self._dot_matrix(self.pfb, phi, xs, inputIdx))
                phi, inputIdx = (1, inputIdx + 1) if phi == self.Nphi else (phi + 1, inputIdx)
            self._shiftin(xs)
        elif self.kind == "rational":                    # :536-575
            if xLen < self.inputDeficit:
                self._shiftin(xs)
                self.inputDeficit -= xLen
            else:
                inputIdx = self.inputDeficit
                while inputIdx <= xLen:
                    if inputIdx < self.T:
                        out.append(self._dot_matrix_seam(self.pfb, self.phiIdx, self.history, xs, inputIdx))
                    else:
                        out.append(self._dot_matrix(self.pfb, self.phiIdx, xs, inputIdx))
                    inputIdx += int(math.floor((self.phiIdx + self.M - 1) / self.L))
                    self.phiIdx = self._nextphase(self.phiIdx)
                self.inputDeficit = inputIdx - xLen
                self._shiftin(xs)
        elif self.kind == "decimator":                   # :598-650 (short input handled by filt, :638-643)
            if xLen < self.inputDeficit:
                self._shiftin(xs)
                self.inputDeficit -= xLen
            else:
                inputIdx = self.inputDeficit
                while inputIdx <= xLen:
                    if inputIdx < self.hLen:
                        out.append(self._dot_vector_seam(self.h, self.history, xs, inputIdx))
                    else:
                        out.append(self._dot_vector(self.h, xs, inputIdx))
                    inputIdx += self.M
                self.inputDeficit = inputIdx - xLen
                self._shiftin(xs)
        else:                                            # arbitrary, :693-742
            if xLen < self.inputDeficit:
                self._shiftin(xs)
                self.inputDeficit -= xLen
            else:
                self.xIdx = self.inputDeficit
                while self.xIdx <= xLen:
                    if self.xIdx < self.T:
                        yL = self._dot_matrix_seam(self.pfb, self.phiIdx, self.history, xs, self.xIdx)
                        yU = self._dot_matrix_seam(self.dpfb, self.phiIdx, self.history, xs, self.xIdx)
                    else:
                        yL = self._dot_matrix(self.pfb, self.phiIdx, xs, self.xIdx)
                        yU = self._dot_matrix(self.dpfb, self.phiIdx, xs, self.xIdx)
                    # buffer[bufIdx] = yLower + yUpper * kernel.α : α is a Float64, so the combine is in Float64 and the
                    # store into Vector{Tb} rounds once (:730)
                    a = np.float64(self.alpha)
                    out.append(_Num(tuple(self.R(np.float64(l) + np.float64(u) * a) for l, u in zip(yL.p, yU.p))))
                    self._update()
                self.inputDeficit = self.xIdx - xLen
                self._shiftin(xs)
        if self.cplx:
            return np.array([complex(float(o.p[0]), float(o.p[1])) for o in out]).astype(self.out_dtype) if out else np.zeros(0, self.out_dtype)
        return np.array([o.p[0] for o in out], dtype=self.out_dtype) if out else np.zeros(0, self.out_dtype)

    def _update(self):                                   # :663-673
        self.acc = float(np.float64(self.acc) + np.float64(self.delta))
        if self.acc > self.Nphi:
            self.xIdx += int(math.floor((self.acc - 1.0) / self.Nphi))
            self.acc = math.fmod(self.acc - 1.0, float(self.Nphi)) + 1.0
        self.phiIdx = int(math.floor(self.acc))
        self.alpha = self.acc - self.phiIdx

    def history_array(self):
        if self.cplx:
            return np.array([complex(float(v.p[0]), float(v.p[1])) for v in self.history]).astype(self.tx)
        return np.array([v.p[0] for v in self.history]).astype(self.tx)
