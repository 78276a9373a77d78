"""ctypes bindings for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Two back-ends with the same function set:
  * "orc": oracle/liboracle.so, this repo's C restatement (rvc_oracle.c)
  * "ref": oracle/_ref/libref_fftconvolver.so, the untouched reference sources compiled
           where they lie (oracle/Makefile target `ref`), when it has been built.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package (reevr_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_F32P = C.POINTER(C.c_float)
_F64P = C.POINTER(C.c_double)


def build(ref: bool = True) -> None:
    """Compile liboracle.so and, when the reference sources are present, oracle/_ref."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    if ref:
        subprocess.run(["make", "-s", "-C", _HERE, "ref"], check=True)


def _fp(a: np.ndarray):
    return a.ctypes.data_as(_F32P)


class _Backend:
    def __init__(self, path: str, prefix: str):
        self.path = path
        self.prefix = prefix
        self.lib = C.CDLL(path)
        for kind in ("fftconv", "twostage"):
            f = getattr(self.lib, f"{prefix}_{kind}_create")
            f.restype = C.c_void_p
            f.argtypes = []
            for name in ("destroy", "clear", "reset"):
                g = getattr(self.lib, f"{prefix}_{kind}_{name}")
                g.restype = None
                g.argtypes = [C.c_void_p]
            p = getattr(self.lib, f"{prefix}_{kind}_process")
            p.restype = None
            p.argtypes = [C.c_void_p, _F32P, _F32P, C.c_size_t]
        self.lib.__getattr__(f"{prefix}_fftconv_init").restype = C.c_int
        self.lib.__getattr__(f"{prefix}_fftconv_init").argtypes = [C.c_void_p, C.c_size_t, _F32P, C.c_size_t]
        self.lib.__getattr__(f"{prefix}_twostage_init").restype = C.c_int
        self.lib.__getattr__(f"{prefix}_twostage_init").argtypes = [
            C.c_void_p, C.c_size_t, C.c_size_t, _F32P, C.c_size_t]
        for name in ("rfft", "irfft"):
            g = getattr(self.lib, f"{prefix}_{name}")
            g.restype = None
            g.argtypes = [C.c_size_t, _F32P, _F32P, _F32P]
        if prefix == "orc":
            self.lib.orc_direct_convolve.restype = None
            self.lib.orc_direct_convolve.argtypes = [_F32P, C.c_size_t, _F32P, C.c_size_t, _F64P]

    def fn(self, name: str):
        return getattr(self.lib, f"{self.prefix}_{name}")


_backends: dict = {}


def backend(which: str = "orc") -> _Backend:
    if which not in _backends:
        if which == "orc":
            path = os.path.join(_HERE, "liboracle.so")
            if not os.path.exists(path):
                build(ref=False)
        elif which == "ref":
            path = os.path.join(_HERE, "_ref", "libref_fftconvolver.so")
            if not os.path.exists(path):
                raise FileNotFoundError(path)
        else:
            raise ValueError(which)
        _backends[which] = _Backend(path, which)
    return _backends[which]


def have_ref() -> bool:
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_fftconvolver.so"))


class _Conv:
    _kind = ""

    def __init__(self, which: str = "orc"):
        self._b = backend(which)
        self._h = self._b.fn(f"{self._kind}_create")()

    def __del__(self):
        try:
            if self._h:
                self._b.fn(f"{self._kind}_destroy")(self._h)
                self._h = None
        except Exception:
            pass

    def process(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        if x.size:
            self._b.fn(f"{self._kind}_process")(self._h, _fp(x), _fp(out), x.size)
        return out

    def clear(self):
        self._b.fn(f"{self._kind}_clear")(self._h)

    def reset(self):
        self._b.fn(f"{self._kind}_reset")(self._h)


class FFTConvolver(_Conv):
    """fftconvolver::FFTConvolver (FFTConvolver.h:52-80) on the chosen CPU back-end."""
    _kind = "fftconv"

    def init(self, blockSize: int, ir: np.ndarray) -> bool:
        ir = np.ascontiguousarray(ir, dtype=np.float32)
        return bool(self._b.fn("fftconv_init")(self._h, blockSize, _fp(ir), ir.size))


class TwoStageFFTConvolver(_Conv):
    """fftconvolver::TwoStageFFTConvolver (TwoStageFFTConvolver.h:54-83), tail inline."""
    _kind = "twostage"

    def init(self, head: int, tail: int, ir: np.ndarray) -> bool:
        ir = np.ascontiguousarray(ir, dtype=np.float32)
        return bool(self._b.fn("twostage_init")(self._h, head, tail, _fp(ir), ir.size))


def rfft(x: np.ndarray, which: str = "orc"):
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = x.size
    re = np.empty(n // 2 + 1, np.float32)
    im = np.empty(n // 2 + 1, np.float32)
    backend(which).fn("rfft")(n, _fp(x), _fp(re), _fp(im))
    return re, im


def irfft(re: np.ndarray, im: np.ndarray, which: str = "orc"):
    re = np.ascontiguousarray(re, dtype=np.float32)
    im = np.ascontiguousarray(im, dtype=np.float32)
    n = 2 * (re.size - 1)
    out = np.empty(n, np.float32)
    backend(which).fn("irfft")(n, _fp(out), _fp(re), _fp(im))
    return out


def direct_convolve(x: np.ndarray, ir: np.ndarray) -> np.ndarray:
    """Test.cpp:33-66 SimpleConvolve with a double accumulator."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    ir = np.ascontiguousarray(ir, dtype=np.float32)
    out = np.zeros(x.size + ir.size - 1, np.float64)
    backend("orc").lib.orc_direct_convolve(_fp(x), x.size, _fp(ir), ir.size,
                                           out.ctypes.data_as(_F64P))
    return out


def run_schedule(conv, x: np.ndarray, schedule) -> np.ndarray:
    """Feed x to conv.process in calls of the given sizes (sum(schedule) == len(x))."""
    out = np.empty(len(x), np.float32)
    pos = 0
    for n in schedule:
        out[pos:pos + n] = conv.process(x[pos:pos + n])
        pos += n
    assert pos == len(x)
    return out
