"""Host-side mirror of the reference's convolver classes on top of the C ABI.

Same names, argument meaning and error behaviour as the reference (paths relative to the
reference tree) so parity tests read like the reference's own:

  FFTConvolver            libs/FFTConvolver/FFTConvolver.h:52-80      init/process/clear/reset
  TwoStageFFTConvolver    libs/FFTConvolver/TwoStageFFTConvolver.h:54-83
  Convolver               src/dsp/Convolver.h:28-45                   (+ isFinished)
  StereoConvolver         src/dsp/StereoConvolver.h:7-46              prepare/loadImpulse/process/...

plus ConvolverSet, the batched form the GPU wants (n channels in lock-step, host or
device-resident buffers). All arithmetic happens in libreevr_amd.so on the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from . import _lib as L


class RvcError(RuntimeError):
    pass


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


class ConvolverSet:
    """n independent mono convolvers sharing one block geometry (one launch per stage)."""

    def __init__(self, n_channels: int, device: int = 0, bg_stream: bool = False, timing: bool = False,
                 fft_f64: bool = False, fixed_partitions: bool = False, time_tiling=True,
                 fft_f32: bool = False, child_sets=None, fft_f64_long: bool = False, tune=None):
        """time_tiling: True (by size) / False / "force" (every stage, one level unless long) / "force2" (two levels).
        fft_f64 / fft_f32: every transform in double / in float (default: rvc.h, RVC_FLAG_FFT_F64); fft_f64_long: the small
        sets' default rule (double for partitions of 2048 .. 8192 samples) whatever the channel count (RVC_FLAG_FFT_F64_LONG).
        child_sets: None / True = the engine's default (sets of thousands of block-synchronous channels are served by child sets on
        their own streams, fenced internally), False = RVC_FLAG_NO_SUBSETS (one set on one queue: per-launch profiling),
        "unfenced" = RVC_FLAG_CHILD_SETS (no fences inside the calls: rvc_set_fork / rvc_set_join around each ordered call here).
        tune: dict of measurement knobs for THIS set only (rvc_set_create_tuned; keys: reevr_amd.TUNING_DEFAULTS)."""
        self._lib = L.lib()
        flags = ((L.RVC_FLAG_BG_STREAM if bg_stream else 0) | (L.RVC_FLAG_TIMING if timing else 0)
                 | (L.RVC_FLAG_FFT_F64 if fft_f64 else 0) | (L.RVC_FLAG_FFT_F32 if fft_f32 else 0)
                 | (L.RVC_FLAG_FFT_F64_LONG if fft_f64_long else 0)
                 | (L.RVC_FLAG_FIXED_PARTITIONS if fixed_partitions else 0)
                 | (0 if time_tiling else L.RVC_FLAG_NO_TIME_TILING)
                 | (L.RVC_FLAG_FORCE_TIME_TILING if time_tiling == "force" else 0)
                 | (L.RVC_FLAG_FORCE_TWO_LEVEL if time_tiling == "force2" else 0)
                 | (L.RVC_FLAG_NO_SUBSETS if child_sets is False else 0) | (L.RVC_FLAG_CHILD_SETS if child_sets == "unfenced" else 0))
        self.unfenced = child_sets == "unfenced"
        self.n_channels = int(n_channels)
        self.device = int(device)
        if tune:
            knobs = ",".join("%s=%d" % (k, int(v)) for k, v in tune.items())
            self._h = self._lib.rvc_set_create_tuned(self.n_channels, self.device, flags, knobs.encode())
            if not self._h:
                raise KeyError(f"rvc_set_create_tuned refused {knobs!r}")
        else:
            self._h = self._lib.rvc_set_create(self.n_channels, self.device, flags)
        if not self._h:
            raise RvcError("rvc_set_create failed")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rvc_set_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- init ---------------------------------------------------------------------------
    def _ir_args(self, irs: Sequence[np.ndarray]):
        if len(irs) != self.n_channels:
            raise ValueError(f"need {self.n_channels} impulse responses, got {len(irs)}")
        keep = [_f32(ir).reshape(-1) for ir in irs]   # keep alive during the call
        ptrs = (L.F32P * self.n_channels)(*[ir.ctypes.data_as(L.F32P) for ir in keep])
        lens = (C.c_size_t * self.n_channels)(*[ir.size for ir in keep])
        return keep, ptrs, lens

    def init(self, headBlockSize: int, tailBlockSize: int, irs: Sequence[np.ndarray], max_len: int = 0) -> bool:
        keep, ptrs, lens = self._ir_args(irs)
        ok = bool(self._lib.rvc_set_init(self._h, headBlockSize, tailBlockSize, ptrs, lens, max_len))
        del keep
        return ok

    def init_uniform(self, blockSize: int, irs: Sequence[np.ndarray], max_len: int = 0) -> bool:
        keep, ptrs, lens = self._ir_args(irs)
        ok = bool(self._lib.rvc_set_init_uniform(self._h, blockSize, ptrs, lens, max_len))
        del keep
        return ok

    def init_impulse(self, headBlockSize: int, tailBlockSize: int, imp, channels: Sequence[int], max_len: int = 0) -> bool:
        """init() from a device-resident reevr_amd.Impulse: set channel c <- prepared channel channels[c]
        (0 LL, 1 RR, 2 LR, 3 RL). No host round trip (rvc_set_init_impulse)."""
        if len(channels) != self.n_channels:
            raise ValueError("one impulse channel index per set channel")
        idx = (C.c_int * self.n_channels)(*[int(c) for c in channels])
        return bool(self._lib.rvc_set_init_impulse(self._h, headBlockSize, tailBlockSize, imp._h, idx, max_len))

    # -- process ------------------------------------------------------------------------
    def process(self, x: np.ndarray) -> np.ndarray:
        """x: (n_channels, len) host array -> (n_channels, len) float32."""
        x = _f32(x)
        if x.ndim == 1:
            x = x[None, :]
        assert x.shape[0] == self.n_channels
        out = np.empty_like(x)
        n = x.shape[1]
        if n == 0:
            return out
        ins = (L.F32P * self.n_channels)(*[x[c].ctypes.data_as(L.F32P) for c in range(self.n_channels)])
        outs = (L.F32P * self.n_channels)(*[out[c].ctypes.data_as(L.F32P) for c in range(self.n_channels)])
        self._lib.rvc_set_process(self._h, ins, outs, n)
        return out

    def host_buffers(self):
        """(inputs, outputs): float32 views (n_channels rows of max_len) of the set's own pinned staging rows
        (rvc_set_host_buffers). process_in_place(n) runs a call on the first n frames of `inputs` and leaves the result in the
        first n frames of `outputs` -- no copy into or out of pinned memory."""
        n, ml = self.n_channels, self.max_len
        ins, outs = (C.c_void_p * n)(), (C.c_void_p * n)()
        if not self._lib.rvc_set_host_buffers(self._h, ins, outs):
            raise RvcError("rvc_set_host_buffers: the set is not initialised")
        self._stage_ptrs = (ins, outs)
        view = lambda p: np.ctypeslib.as_array(C.cast(p, L.F32P), shape=(ml,))
        return [view(p) for p in ins], [view(p) for p in outs]

    def process_in_place(self, n: int):
        """One call of n <= max_len frames on the staging rows of host_buffers() (call that first)."""
        ins, outs = self._stage_ptrs
        self._lib.rvc_set_process(self._h, C.cast(ins, L.F32PP), C.cast(outs, L.F32PP), n)

    def process_in_place_timed(self, n: int, calls: int) -> np.ndarray:
        """`calls` back-to-back calls of n frames on the staging rows of host_buffers(), in C with a stopwatch around every call
        (rvc_set_process_host_blocks_timed on the staging pointers themselves: block = len = n, no staging copy). Returns the
        per-call durations in microseconds."""
        ins, outs = self._stage_ptrs
        us = np.zeros(calls, np.float64)
        one = np.zeros(1, np.float64)
        for i in range(calls):
            self._lib.rvc_set_process_host_blocks_timed(self._h, C.cast(ins, L.F32PP), C.cast(outs, L.F32PP), n, n,
                                                        one.ctypes.data_as(C.POINTER(C.c_double)))
            us[i] = one[0]
        return us

    def process_host_blocks_timed(self, x: np.ndarray, block: int):
        """x: (n_channels, len) host array fed through process() in calls of `block` frames, all in C.
        Returns (output, per-call durations in microseconds)."""
        x = _f32(x)
        assert x.ndim == 2 and x.shape[0] == self.n_channels
        out = np.empty_like(x)
        ncalls = -(-x.shape[1] // block)
        us = np.zeros(ncalls, np.float64)
        ins = (L.F32P * self.n_channels)(*[x[c].ctypes.data_as(L.F32P) for c in range(self.n_channels)])
        outs = (L.F32P * self.n_channels)(*[out[c].ctypes.data_as(L.F32P) for c in range(self.n_channels)])
        self._lib.rvc_set_process_host_blocks_timed(self._h, ins, outs, x.shape[1], block,
                                                    us.ctypes.data_as(C.POINTER(C.c_double)))
        return out, us

    def process_begin(self, x: np.ndarray):
        """Non-blocking half of process(): stage x (n_channels, len <= max_len) and enqueue the work."""
        x = _f32(x)
        assert x.ndim == 2 and x.shape[0] == self.n_channels
        self._pending = x
        ins = (L.F32P * self.n_channels)(*[x[c].ctypes.data_as(L.F32P) for c in range(self.n_channels)])
        self._lib.rvc_set_process_begin(self._h, ins, x.shape[1])

    def process_end(self) -> np.ndarray:
        """Wait for the work enqueued by process_begin and return its output."""
        out = np.empty_like(self._pending)
        outs = (L.F32P * self.n_channels)(*[out[c].ctypes.data_as(L.F32P) for c in range(self.n_channels)])
        self._lib.rvc_set_process_end(self._h, outs)
        self._pending = None
        return out

    def _order_after_torch(self):
        """The engine runs on its own (non-blocking) HIP streams: make them wait for whatever torch's
        current stream has queued (the kernels still producing d_in). Returns the wrapped stream: stream 0 of the set is the
        one callers order against also when child sets serve it (the library fences them internally, rvc.h rvc_set_stream)."""
        import torch
        ptrs = [p for p in [self._lib.rvc_set_stream(self._h, 0)] if p]
        if not ptrs:
            return None
        if getattr(self, "_ext_ptrs", None) != ptrs:
            self._ext = [torch.cuda.ExternalStream(p, device=torch.device("cuda", self.device)) for p in ptrs]
            self._ext_ptrs = ptrs
        cur = torch.cuda.current_stream(self.device)
        for e in self._ext:
            e.wait_stream(cur)
        if self.unfenced:         # RVC_FLAG_CHILD_SETS: no fences inside the calls -- fork here, join in _order_torch_after
            self._lib.rvc_set_fork(self._h)
        return self._ext

    def _order_torch_after(self, ext):
        import torch
        if ext is not None:
            if self.unfenced:
                self._lib.rvc_set_join(self._h)
            cur = torch.cuda.current_stream(self.device)
            for e in ext:
                cur.wait_stream(e)

    def process_device(self, d_in, d_out=None, sync: bool = True, order: bool = True):
        """d_in / d_out: torch float32 CUDA tensors (n_channels, len), last dim contiguous.
        Runs on the set's own HIP stream, ordered after torch's current stream (d_in may still be
        in production there); with sync=True waits for completion, else torch's current stream is
        made to wait for the result. order=False skips both stream dependencies: for callers whose
        buffers are already complete and who synchronise themselves (bench.py's timed loop)."""
        import torch
        assert d_in.is_cuda and d_in.dtype == torch.float32 and d_in.dim() == 2 and d_in.stride(1) == 1
        assert d_in.shape[0] == self.n_channels
        if d_out is None:
            d_out = torch.empty_like(d_in)
        assert d_out.is_cuda and d_out.dtype == torch.float32 and d_out.shape == d_in.shape and d_out.stride(1) == 1
        ext = self._order_after_torch() if order else None
        self._lib.rvc_set_process_device(self._h, d_in.data_ptr(), d_in.stride(0), d_out.data_ptr(),
                                         d_out.stride(0), d_in.shape[1])
        if sync:
            self.sync()
            self.check()
        else:
            self._order_torch_after(ext)
        return d_out

    def process_device_blocks(self, d_in, block: int, d_out=None, sync: bool = True, order: bool = True):
        """The host's per-block loop in C: d_in (n_channels, len) is fed in consecutive calls of
        `block` frames (strict streaming; every call takes the latency path). order=False: see
        process_device."""
        import torch
        assert d_in.is_cuda and d_in.dtype == torch.float32 and d_in.dim() == 2 and d_in.stride(1) == 1
        assert d_in.shape[0] == self.n_channels
        if d_out is None:
            d_out = torch.empty_like(d_in)
        assert d_out.is_cuda and d_out.dtype == torch.float32 and d_out.shape == d_in.shape and d_out.stride(1) == 1
        ext = self._order_after_torch() if order else None
        self._lib.rvc_set_process_device_blocks(self._h, d_in.data_ptr(), d_in.stride(0), d_out.data_ptr(),
                                                d_out.stride(0), d_in.shape[1], block)
        if sync:
            self.sync()
            self.check()
        else:
            self._order_torch_after(ext)
        return d_out

    def process_device_blocks_stamped(self, d_in, block: int, d_out=None):
        """process_device_blocks with a completion stamp behind every call (rvc_set_process_device_blocks_stamped): returns
        (d_out, done_ms) -- done_ms[i] = milliseconds after the loop's start at which everything call i enqueued had completed on
        the device; np.diff(done_ms) is the cost of each call in the back-to-back loop. Synchronous."""
        import torch
        assert d_in.is_cuda and d_in.dtype == torch.float32 and d_in.dim() == 2 and d_in.stride(1) == 1
        assert d_in.shape[0] == self.n_channels
        if d_out is None:
            d_out = torch.empty_like(d_in)
        torch.cuda.current_stream().synchronize()
        calls = -(-d_in.shape[1] // block)
        done = np.zeros(calls, np.float64)
        n = self._lib.rvc_set_process_device_blocks_stamped(self._h, d_in.data_ptr(), d_in.stride(0), d_out.data_ptr(), d_out.stride(0),
                                                            d_in.shape[1], block, done.ctypes.data_as(C.POINTER(C.c_double)))
        if n != calls:
            raise RvcError("rvc_set_process_device_blocks_stamped failed")
        self.check()
        return d_out, done

    # -- state --------------------------------------------------------------------------
    def clear(self):
        self._lib.rvc_set_clear(self._h)

    def reset(self):
        self._lib.rvc_set_reset(self._h)

    def isFinished(self) -> bool:
        return bool(self._lib.rvc_set_is_finished(self._h))

    def sync(self):
        self._lib.rvc_set_sync(self._h)

    # -- introspection ------------------------------------------------------------------
    @property
    def head_block(self) -> int:
        return int(self._lib.rvc_set_head_block(self._h))

    @property
    def tail_block(self) -> int:
        """the block the tail stage runs: the requested one (rounded up to a power of two), or twice that for the widened
        delay-1 tail of lock-step sets of many channels (rvc.h, RVC_MAX_BLOCK); 0 for single-stage sets"""
        return int(self._lib.rvc_set_tail_block(self._h))

    @property
    def max_len(self) -> int:
        return int(self._lib.rvc_set_max_len(self._h))

    def partitions(self, stage: int) -> int:
        return int(self._lib.rvc_set_partitions(self._h, stage))

    def tile_rows(self, stage: int) -> int:
        """blocks per first-level sweep tile of a stage's time-tiled delay line (0: not tiled)"""
        return int(self._lib.rvc_set_tile_rows(self._h, stage))

    @property
    def subsets(self) -> int:
        return int(self._lib.rvc_set_subsets(self._h))

    def plan(self) -> dict:
        """What the set runs (rvc_set_plan): stage blocks, partitions, tail delay, precision, tiles, child sets."""
        p = L.Plan()
        if not self._lib.rvc_set_plan(self._h, C.byref(p), C.sizeof(p)):
            raise RvcError("rvc_set_plan failed")
        return {name: getattr(p, name) for name, _ in L.Plan._fields_}

    def stream(self, which: int = 0) -> int:
        return int(self._lib.rvc_set_stream(self._h, which) or 0)

    @property
    def last_error(self) -> int:
        return int(self._lib.rvc_last_error(self._h))

    @property
    def last_error_string(self) -> str:
        return (self._lib.rvc_last_error_string(self._h) or b"").decode()

    def check(self):
        if self.last_error != L.RVC_OK:
            raise RvcError(f"rvc error {self.last_error}: {self.last_error_string}")

    def set_timing(self, on: bool):
        self._lib.rvc_set_timing(self._h, int(on))

    def kernel_time(self, kernel: int):
        """(launches, total_ms) of one kernel family since the last reset."""
        ms = C.c_double(0.0)
        n = self._lib.rvc_set_kernel_time(self._h, kernel, C.byref(ms))
        return int(n), float(ms.value)

    def kernel_time_reset(self):
        self._lib.rvc_set_kernel_time_reset(self._h)

    def kernel_intervals(self, kernel: int):
        """(n, 2) array of (start, end) in ms since the last reset of every timed launch of one kernel family
        (all child sets on one clock)."""
        n = int(self._lib.rvc_set_kernel_intervals(self._h, kernel, None, None, 0))
        a = np.zeros(n, np.float64)
        b = np.zeros(n, np.float64)
        if n:
            dp = C.POINTER(C.c_double)
            self._lib.rvc_set_kernel_intervals(self._h, kernel, a.ctypes.data_as(dp), b.ctypes.data_as(dp), n)
        return np.stack([a, b], axis=1)

    def guard_check(self) -> int:
        """Changed guard bytes around the set's device allocations (rvc_debug_guard_check; -1: no guards)."""
        return int(self._lib.rvc_debug_guard_check(self._h))


KERNEL_NAMES = ["ingest", "fft_fwd_head", "fir_head", "fft_inv_head", "fft_fwd_tail", "fir_tail", "fft_inv_tail",
                "fused_block", "premultiply", "sweep_head", "sweep_tail", "sweep2_head", "sweep2_tail", "sweep3_tail", "sweep3_head"]


def set_tuning(key: str, value: int) -> bool:
    """Default of a schedule knob for the sets created afterwards (rvc_debug_set_tuning; measurement hook)."""
    return bool(L.lib().rvc_debug_set_tuning(key.encode(), int(value)))


def stage_plan(n_channels: int, headBlockSize: int, tailBlockSize: int, longest_ir: int, flags: int = 0) -> dict:
    """The stage plan rvc_set_init would choose (rvc_debug_plan: a pure function of the request, no device): block sizes that
    run, impulse samples the zero-latency stage covers, delay of the tail stage in tail blocks (2: the reference's structure,
    1: the widened / shrunk forms of lock-step sets of many channels)."""
    hb, tb, zl = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    d = L.lib().rvc_debug_plan(int(n_channels), int(flags), int(headBlockSize), int(tailBlockSize), int(longest_ir),
                               C.byref(hb), C.byref(tb), C.byref(zl))
    if d == 0:
        raise ValueError("bad arguments")
    return {"head_block": hb.value, "tail_block": tb.value, "zero_latency_samples": zl.value, "tail_delay": d,
            "partitions": (-(-min(longest_ir, zl.value) // hb.value), -(-max(longest_ir - zl.value, 0) // tb.value))}


def _tuning_defaults() -> dict:
    """The knobs and the values the engine ships with, from the library itself (rvc_debug_tuning_keys / _default)."""
    lib = L.lib()
    out = {}
    for key in lib.rvc_debug_tuning_keys().decode().split(","):
        v = C.c_int(0)
        assert lib.rvc_debug_tuning_default(key.encode(), C.byref(v))
        out[key] = v.value
    return out


class _LazyDefaults(dict):
    """TUNING_DEFAULTS: filled from the library on first use (importing the package must not need the built library)."""

    def _fill(self):
        if not dict.__len__(self):
            dict.update(self, _tuning_defaults())

    def __getitem__(self, k):
        self._fill()
        return dict.__getitem__(self, k)

    def __contains__(self, k):
        self._fill()
        return dict.__contains__(self, k)

    def __iter__(self):
        self._fill()
        return dict.__iter__(self)

    def __len__(self):
        self._fill()
        return dict.__len__(self)

    def keys(self):
        self._fill()
        return dict.keys(self)

    def items(self):
        self._fill()
        return dict.items(self)


# the knobs rvc_debug_set_tuning / ConvolverSet(tune=...) know, with the values the engine ships with (what `tuning` restores)
TUNING_DEFAULTS = _LazyDefaults()


class tuning:
    """`with reevr_amd.tuning(sweep_lds=0, k1=32): ...` -- change the knob DEFAULTS for the sets CREATED inside the block (a
    set copies them when it is created and keeps them) and put the engine's defaults back on the way out, also when the block
    raises. Process-wide while the block runs: concurrent code uses ConvolverSet(tune={...}) instead (the set's own knobs)."""

    def __init__(self, **knobs):
        unknown = [k for k in knobs if k not in TUNING_DEFAULTS]
        if unknown:
            raise KeyError(f"unknown tuning knob(s) {unknown}")
        self.knobs = knobs

    def __enter__(self):
        for k, v in self.knobs.items():
            if not set_tuning(k, v):
                raise KeyError(k)
        return self

    def __exit__(self, *exc):
        for k in self.knobs:
            set_tuning(k, TUNING_DEFAULTS[k])
        return False


class _Mono:
    """One mono convolver: the reference's per-object surface on a 1-channel set."""

    def __init__(self, device: int = 0, bg_stream: bool = True, fft_f64: bool = False, fft_f32: bool = False):
        self._set = ConvolverSet(1, device, bg_stream=bg_stream, fft_f64=fft_f64, fft_f32=fft_f32)

    def process(self, input: np.ndarray) -> np.ndarray:
        x = _f32(input).reshape(-1)
        return self._set.process(x[None, :])[0]

    def clear(self):
        self._set.clear()

    def reset(self):
        self._set.reset()

    @property
    def last_error(self) -> int:
        return self._set.last_error

    @property
    def head_block(self) -> int:      # partition sizes in use (after rounding / clamping to RVC_MAX_BLOCK)
        return self._set.head_block

    @property
    def tail_block(self) -> int:
        return self._set.tail_block


class FFTConvolver(_Mono):
    """fftconvolver::FFTConvolver (FFTConvolver.h:52-80)."""

    def __init__(self, device: int = 0, fft_f64: bool = False, fft_f32: bool = False):
        super().__init__(device, bg_stream=False, fft_f64=fft_f64, fft_f32=fft_f32)

    def init(self, blockSize: int, ir: np.ndarray, max_len: int = 0) -> bool:
        return self._set.init_uniform(blockSize, [ir], max_len)


class TwoStageFFTConvolver(_Mono):
    """fftconvolver::TwoStageFFTConvolver (TwoStageFFTConvolver.h:54-83); tail inline on one stream."""

    def __init__(self, device: int = 0, bg_stream: bool = False, fft_f64: bool = False, fft_f32: bool = False):
        super().__init__(device, bg_stream=bg_stream, fft_f64=fft_f64, fft_f32=fft_f32)

    def init(self, headBlockSize: int, tailBlockSize: int, ir: np.ndarray, max_len: int = 0) -> bool:
        return self._set.init(headBlockSize, tailBlockSize, [ir], max_len)


class Convolver(TwoStageFFTConvolver):
    """src/dsp/Convolver.h:28-45: the two-stage convolver whose tail runs in the background
    (there a juce::Thread, here a second HIP stream with events at the same hook points)."""

    def __init__(self, device: int = 0, fft_f64: bool = False):
        super().__init__(device, bg_stream=True, fft_f64=fft_f64)

    def isFinished(self) -> bool:
        return self._set.isFinished()


class StereoConvolver:
    """src/dsp/StereoConvolver.{h,cpp}: LL/RR (+LR/RL when the impulse is quad) convolvers and
    their output buffers. `imp` is any object with bufferLL/bufferRR[/bufferLR/bufferRL] arrays
    and an isQuad flag (the reference's Impulse, src/dsp/Impulse.h)."""

    def __init__(self, device: int = 0):
        self._device = device
        self._main = ConvolverSet(2, device, bg_stream=True)    # LL, RR
        self._cross = ConvolverSet(2, device, bg_stream=True)   # LR, RL
        self.bufferLL = np.zeros(0, np.float32)
        self.bufferRR = np.zeros(0, np.float32)
        self.bufferLR = np.zeros(0, np.float32)
        self.bufferRL = np.zeros(0, np.float32)
        self.size = 0
        self.isQuad = False
        self.headBlockSize = 0
        self.tailBlockSize = 0

    def finishedLoading(self) -> bool:           # StereoConvolver.cpp:3-6
        return self._main.isFinished()

    def prepare(self, samplesPerBlock: int):     # StereoConvolver.cpp:8-20
        self.size = int(samplesPerBlock)
        self.headBlockSize = 1
        while self.headBlockSize < samplesPerBlock:
            self.headBlockSize *= 2
        self.tailBlockSize = max(8192, 2 * self.headBlockSize)
        for name in ("bufferLL", "bufferRR", "bufferLR", "bufferRL"):
            setattr(self, name, np.zeros(self.size, np.float32))

    def loadImpulse(self, imp):                  # StereoConvolver.cpp:22-31
        if getattr(imp, "_h", None) and hasattr(imp, "device_ptr"):   # device-resident reevr_amd.Impulse
            self._main.init_impulse(self.headBlockSize, self.tailBlockSize, imp, [0, 1], self.size)
            self.isQuad = bool(imp.isQuad)
            if self.isQuad:
                self._cross.init_impulse(self.headBlockSize, self.tailBlockSize, imp, [2, 3], self.size)
            return
        self._main.init(self.headBlockSize, self.tailBlockSize, [imp.bufferLL, imp.bufferRR], self.size)
        self.isQuad = bool(imp.isQuad)
        if self.isQuad:
            self._cross.init(self.headBlockSize, self.tailBlockSize, [imp.bufferLR, imp.bufferRL], self.size)

    def process(self, dataL: np.ndarray, dataR: np.ndarray, nsamples: int, force2Chans: bool = False):
        # StereoConvolver.cpp:33-42
        x = np.stack([_f32(dataL)[:nsamples], _f32(dataR)[:nsamples]])
        cross = self.isQuad and not force2Chans
        if nsamples > len(self.bufferLL):        # the reference would overrun its buffers: grow them
            for name in ("bufferLL", "bufferRR", "bufferLR", "bufferRL"):
                setattr(self, name, np.zeros(nsamples, np.float32))
        if nsamples > self.size:                 # longer than prepare() announced: blocking, split inside
            y = self._main.process(x)
            z = self._cross.process(x) if cross else None
        else:                                    # both pairs in flight together, then collect
            self._main.process_begin(x)
            if cross:
                self._cross.process_begin(x)     # LR is fed L, RL is fed R
            y = self._main.process_end()
            z = self._cross.process_end() if cross else None
        self.bufferLL[:nsamples] = y[0]
        self.bufferRR[:nsamples] = y[1]
        if cross:
            self.bufferLR[:nsamples] = z[0]
            self.bufferRL[:nsamples] = z[1]

    def warm(self, dataL: np.ndarray, dataR: np.ndarray, nsamples: int, force2Chans: bool = False):
        """Feed nsamples (any length, typically many blocks) and discard the output: what the
        reference's warm-up loop of block-sized process() calls amounts to
        (src/PluginProcessor.cpp:1716-1750). One multi-block call per pair."""
        x = np.stack([_f32(dataL)[:nsamples], _f32(dataR)[:nsamples]])
        self._main.process(x)
        if self.isQuad and not force2Chans:
            self._cross.process(x)

    def reset(self):                             # StereoConvolver.cpp:44-54
        self._main.reset()
        self._cross.reset()
        for name in ("bufferLL", "bufferRR", "bufferLR", "bufferRL"):
            setattr(self, name, np.zeros(0, np.float32))

    def clear(self):                             # StereoConvolver.cpp:56-62
        self._main.clear()
        self._cross.clear()
