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
        try:
            g = getattr(self.lib, f"{prefix}_cmac")       # (a reference library built before the shim had it: rebuild)
            g.restype = None
            g.argtypes = [_F32P] * 6 + [C.c_size_t]
        except AttributeError:
            pass
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


def cmac(re: np.ndarray, im: np.ndarray, reA, imA, reB, imB, which: str = "orc") -> None:
    """re/im += A * B in place, the reference's ComplexMultiplyAccumulate (Utilities.cpp:62-111) / its restatement"""
    arrs = [np.ascontiguousarray(v, dtype=np.float32) for v in (reA, imA, reB, imB)]
    assert re.dtype == np.float32 and im.dtype == np.float32 and re.flags.c_contiguous and im.flags.c_contiguous
    backend(which).fn("cmac")(_fp(re), _fp(im), *[_fp(v) for v in arrs], re.size)


def direct_convolve(x: np.ndarray, ir: np.ndarray) -> np.ndarray:
    """Test.cpp:33-66 SimpleConvolve with a double accumulator."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    ir = np.ascontiguousarray(ir, dtype=np.float32)
    out = np.zeros(x.size + ir.size - 1, np.float64)
    backend("orc").lib.orc_direct_convolve(_fp(x), x.size, _fp(ir), ir.size,
                                           out.ctypes.data_as(_F64P))
    return out


def cpu_bench(which: str, n_threads: int, head: int, tail: int, block: int, irs, ins, seconds: float):
    """bench.py's CPU baseline (cpu_bench.c): n_threads independent TwoStageFFTConvolver instances (tail == 0:
    FFTConvolver instances of block `head`) of back-end `which` ("ref" = the untouched reference, "orc" = the
    restatement), one per thread, `block`-frame process() calls back to back for ~`seconds`, no Python in the loop.
    Returns (channel-samples convolved, wall seconds, per-thread samples)."""
    b = backend(which)
    lib = backend("orc").lib
    lib.orc_cpu_bench.restype = C.c_double
    lib.orc_cpu_bench.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(_F32P), C.c_int,
                                                     C.c_size_t, C.POINTER(_F32P), C.c_int, C.c_size_t, C.c_double,
                                                     C.POINTER(C.c_ulonglong)]
    irs = [np.ascontiguousarray(a, np.float32) for a in irs]
    ins = [np.ascontiguousarray(a, np.float32) for a in ins]
    assert len({a.size for a in irs}) == 1 and len({a.size for a in ins}) == 1
    irp = (_F32P * len(irs))(*[_fp(a) for a in irs])
    inp = (_F32P * len(ins))(*[_fp(a) for a in ins])
    cnt = (C.c_ulonglong * n_threads)()
    addr = lambda name: C.cast(b.fn(name), C.c_void_p)
    kind = "twostage" if tail else "fftconv"
    wall = lib.orc_cpu_bench(addr(kind + "_create"), addr(kind + "_destroy"), addr(kind + "_init"),
                             addr(kind + "_process"), n_threads, head, tail, block, irp, len(irs), irs[0].size,
                             inp, len(ins), ins[0].size, float(seconds), cnt)
    if wall <= 0.0:
        raise RuntimeError("orc_cpu_bench failed")
    per = [int(v) for v in cnt]
    return sum(per), float(wall), per


def run_schedule(conv, x: np.ndarray, schedule) -> np.ndarray:
    """Feed x to conv.process in calls of the given sizes (sum(schedule) == len(x))."""
    out = np.empty(len(x), np.float32)
    pos = 0
    for n in schedule:
        out[pos:pos + n] = conv.process(x[pos:pos + n])
        pos += n
    assert pos == len(x)
    return out


# ---- impulse preparation stages (impulse_oracle.c) ---------------------

IMP_FFT_SIZE = 4096
IMP_LUT_SIZE = IMP_FFT_SIZE // 2 + 1


class _ImpParams(C.Structure):
    _fields_ = [("n_channels", C.c_int), ("reverse", C.c_int), ("trim_left", C.c_float),
                ("trim_right", C.c_float), ("gain", C.c_float), ("attack", C.c_float),
                ("decay", C.c_float), ("srate", C.c_double), ("decay_lut", _F64P)]


def _imp_lib(fft: str = "orc"):
    """liboracle.so with the STFT stage running on this repo's FFT ("orc") or on the reference's
    AudioFFT from oracle/_ref ("ref")."""
    lib = backend("orc").lib
    if not getattr(lib, "_imp_ready", False):
        lib.orc_impulse_auto_gain.restype = C.c_float
        lib.orc_impulse_auto_gain.argtypes = [_F32P, _F32P, C.c_size_t]
        lib.orc_impulse_decay_lut.restype = None
        lib.orc_impulse_decay_lut.argtypes = [_F32P, C.c_double, C.c_float, _F64P]
        lib.orc_impulse_apply_decay.restype = None
        lib.orc_impulse_apply_decay.argtypes = [_F32P, C.c_size_t, _F64P, C.c_double]
        lib.orc_impulse_stage_a.restype = C.c_size_t
        lib.orc_impulse_stage_a.argtypes = [C.POINTER(_ImpParams), C.POINTER(_F32P), C.c_size_t, C.POINTER(_F32P),
                                            _F32P, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.orc_impulse_stage_b.restype = None
        lib.orc_impulse_stage_b.argtypes = [C.POINTER(_ImpParams), C.POINTER(_F32P), C.c_size_t]
        lib.orc_impulse_set_fft.restype = None
        lib.orc_impulse_set_fft.argtypes = [C.c_void_p, C.c_void_p]
        lib._imp_ready = True
    if fft == "ref":
        r = backend("ref").lib
        lib.orc_impulse_set_fft(C.cast(r.ref_rfft, C.c_void_p), C.cast(r.ref_irfft, C.c_void_p))
    else:
        lib.orc_impulse_set_fft(None, None)
    return lib


def impulse_decay_lut(mag: np.ndarray, srate: float, decay_rate: float) -> np.ndarray:
    mag = np.ascontiguousarray(mag, np.float32)
    assert mag.size == IMP_LUT_SIZE
    lut = np.empty(IMP_LUT_SIZE, np.float64)
    _imp_lib().orc_impulse_decay_lut(_fp(mag), srate, decay_rate, lut.ctypes.data_as(_F64P))
    return lut


def impulse_recalc(raw, *, reverse=False, trim_left=0.0, trim_right=0.0, gain=1.0, attack=0.0, decay=0.0,
                   srate=48000.0, decay_lut=None, fft: str = "orc", stage: str = "ab", param_eq=None):
    """raw: list of 2 (LL, RR) or 4 (LL, RR, LR, RL) equal-length float32 arrays.
    Returns dict(buffers=[...], peak, trim_left_samples, trim_right_samples)."""
    lib = _imp_lib(fft)
    raw = [np.ascontiguousarray(r, np.float32) for r in raw]
    nc, n = len(raw), raw[0].size
    lut = None if decay_lut is None else np.ascontiguousarray(decay_lut, np.float64)
    p = _ImpParams(nc, int(bool(reverse)), trim_left, trim_right, gain, attack, decay, srate,
                   lut.ctypes.data_as(_F64P) if lut is not None else None)
    out = [np.zeros(max(n, 1), np.float32) for _ in range(nc)]
    rawp = (_F32P * nc)(*[_fp(r) for r in raw])
    outp = (_F32P * nc)(*[_fp(o) for o in out])
    peak = C.c_float(0)
    tl, tr = C.c_int(0), C.c_int(0)
    m = lib.orc_impulse_stage_a(C.byref(p), rawp, n, outp, _fp(np.ctypeslib.as_array(C.pointer(peak), (1,))),
                                C.byref(tl), C.byref(tr))
    if param_eq is not None and m:     # applyParamEQ sits between applyGain and applyDecayEQ (Impulse.cpp:355-357)
        for o in out:
            o[:m] = np.asarray(param_eq(o[:m].copy()), np.float32)
    if "b" in stage and m:
        lib.orc_impulse_stage_b(C.byref(p), outp, m)
    lib.orc_impulse_set_fft(None, None)
    return {"buffers": [o[:m].copy() for o in out], "peak": float(peak.value),
            "trim_left_samples": tl.value, "trim_right_samples": tr.value}
