"""Host-side mirror of the reference's `Impulse` (src/dsp/Impulse.h:12-93) for the stages that
run on the device -- SURVEY.md 8(f) row f-1.

Same member names as the reference: rawBufferLL/RR/LR/RL in, bufferLL/RR/LR/RL out, the
parameters attack / decay / trimLeft / trimRight / gain / reverse / isQuad, the by-products peak /
trimLeftSamples / trimRightSamples, and recalcImpulse(). File decoding, resampling / stretch
(juce::ResamplingAudioSource, Impulse.cpp:362-434) and the serial IIR paramEQ (:501-533) stay on
the host: pass the decoded channels to setRaw(), and give `paramEQ` a callable (array -> array)
if bands are active -- it is applied between the gain and decay-EQ stages like the reference does.
The decay EQ takes the filters' combined magnitude per bin (`decayMagnitude`, 2049 values,
SVF::getMagnitude is filter design) or a ready table (`decayLUT`).

All array arithmetic happens in libreevr_amd.so on the GPU; the prepared IR stays in HBM and
StereoConvolver.loadImpulse(imp) initialises the convolvers from it without a host round trip
(rvc_set_init_impulse). bufferXX are fetched lazily for callers that want to look at them
(the plug-in draws them, src/ui/...; not needed for convolution).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .convolver import RvcError, _f32

_NAMES = ("LL", "RR", "LR", "RL")


class Impulse:
    FFT_SIZE = L.RVC_IMPULSE_FFT_SIZE          # Impulse.h:22
    HOP_SIZE = FFT_SIZE // 4                   # Impulse.h:23

    def __init__(self, device: int = 0):
        self._lib = L.lib()
        self._h = self._lib.rvc_impulse_create(int(device))
        if not self._h:
            raise RvcError("rvc_impulse_create failed")
        self.device = int(device)
        self.srate = 44100.0                   # Impulse.h:58
        self.attack = 0.0                      # Impulse.h:64-71
        self.decay = 1.0
        self.trimLeft = 0.0
        self.trimRight = 0.0
        self.decayRate = 1.0
        self.gain = 1.0
        self.reverse = False
        self.isQuad = False
        self.numChans = 1
        self.peak = 0.0
        self.trimLeftSamples = 0
        self.trimRightSamples = 0
        self.duration = 0.0
        self.version = 1
        self.decayMagnitude = None             # 2049 combined filter magnitudes, or None (no decay EQ bands)
        self.decayLUT = None                   # ... or the table itself
        self.paramEQ = None                    # callable(np.ndarray) -> np.ndarray, run on the host between stages
        self._cache = {}

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rvc_impulse_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def prepare(self, srate: float):           # Impulse.cpp:73-76
        self.srate = float(srate)

    @staticmethod
    def tail_start(channels) -> int:
        """Impulse::load's silence trim (Impulse.cpp:151-156, getTailStart :696-703): one past the last
        sample whose magnitude reaches 1e-3 in any channel (file-loading step, host side)."""
        end = 0
        for c in channels:
            nz = np.flatnonzero(np.abs(_f32(c)) >= np.float32(1e-3))
            if nz.size:
                end = max(end, int(nz[-1]) + 1)
        return end

    def setRaw(self, *channels, trim_tail: bool = False):
        """The product of Impulse::load (Impulse.cpp:160-196): (LL, RR) or (LL, RR, LR, RL).
        trim_tail=True applies load's trailing-silence trim first (decoded file channels in)."""
        if len(channels) not in (2, 4):
            raise ValueError("2 (LL, RR) or 4 (LL, RR, LR, RL) channels")
        raw = [_f32(c) for c in channels]
        n = raw[0].size
        if any(r.size != n for r in raw):
            raise ValueError("channels differ in length")
        if trim_tail:
            n = self.tail_start(raw)
            raw = [np.ascontiguousarray(r[:n]) for r in raw]
        self.isQuad = len(raw) == 4
        self.numChans = len(raw)
        ptrs = (L.F32P * len(raw))(*[r.ctypes.data_as(L.F32P) for r in raw])
        if not self._lib.rvc_impulse_set_raw(self._h, len(raw), ptrs, n):
            raise RvcError(self._error())
        self._raw = raw
        self._cache = {}

    def _error(self) -> str:
        return (self._lib.rvc_impulse_last_error_string(self._h) or b"").decode()

    def _params(self):
        lut = self.decayLUT
        if lut is None and self.decayMagnitude is not None:
            lut = self.decay_lut(self.decayMagnitude, self.srate, self.decayRate)
        self._lut = None if lut is None else np.ascontiguousarray(lut, np.float64)
        if self._lut is not None and self._lut.size != L.RVC_IMPULSE_LUT_SIZE:
            raise ValueError("decay table must have %d entries" % L.RVC_IMPULSE_LUT_SIZE)
        return L.ImpulseParams(int(bool(self.reverse)), self.trimLeft, self.trimRight, self.gain, self.attack, self.decay,
                               self.srate,
                               self._lut.ctypes.data_as(C.POINTER(C.c_double)) if self._lut is not None else None)

    @staticmethod
    def decay_lut(mag, srate: float, decay_rate: float) -> np.ndarray:
        """Impulse.cpp:561-590: combined filter magnitude per bin -> per-frame decay factor."""
        mag = _f32(mag)
        if mag.size != L.RVC_IMPULSE_LUT_SIZE:
            raise ValueError("need %d magnitudes" % L.RVC_IMPULSE_LUT_SIZE)
        lut = np.empty(L.RVC_IMPULSE_LUT_SIZE, np.float64)
        L.lib().rvc_impulse_decay_lut(mag.ctypes.data_as(L.F32P), srate, decay_rate,
                                      lut.ctypes.data_as(C.POINTER(C.c_double)))
        return lut

    def recalcImpulse(self):                   # Impulse.cpp:307-360
        p = self._params()
        self._cache = {}
        if not self._lib.rvc_impulse_stage_a(self._h, C.byref(p)):
            raise RvcError(self._error())
        if self.paramEQ is not None and self.size:           # applyParamEQ, :501-533 (host IIR)
            for c in range(self.numChans):
                y = _f32(self.paramEQ(self._read(c)))
                if not self._lib.rvc_impulse_write(self._h, c, y.ctypes.data_as(L.F32P), y.size):
                    raise RvcError(self._error())
        if not self._lib.rvc_impulse_stage_b(self._h, C.byref(p)):
            raise RvcError(self._error())
        self.peak = float(self._lib.rvc_impulse_peak(self._h))
        self.trimLeftSamples = int(self._lib.rvc_impulse_trim_left_samples(self._h))
        self.trimRightSamples = int(self._lib.rvc_impulse_trim_right_samples(self._h))
        self.duration = (self.size + self.trimLeftSamples + self.trimRightSamples) / self.srate   # :358
        self.version += 1

    @property
    def size(self) -> int:
        return int(self._lib.rvc_impulse_size(self._h))

    def _read(self, c: int) -> np.ndarray:
        out = np.empty(self.size, np.float32)
        if out.size and not self._lib.rvc_impulse_read(self._h, c, out.ctypes.data_as(L.F32P), out.size):
            raise RvcError(self._error())
        return out

    def _buffer(self, c: int) -> np.ndarray:
        if c >= self.numChans:
            return np.zeros(0, np.float32)
        if c not in self._cache:
            self._cache[c] = self._read(c)
        return self._cache[c]

    bufferLL = property(lambda self: self._buffer(0))
    bufferRR = property(lambda self: self._buffer(1))
    bufferLR = property(lambda self: self._buffer(2))
    bufferRL = property(lambda self: self._buffer(3))

    def device_ptr(self, c: int) -> int:
        return int(self._lib.rvc_impulse_device_ptr(self._h, c) or 0)
