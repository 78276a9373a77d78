"""Deterministic synthetic inputs for parity tests and bench.py.

All generators are counter-based (value i depends only on (seed, i)), so they are
bit-identical however the stream is chunked, vectorise in numpy, and need no state.

  u32 = murmur3_fmix32(seed + 0x9E3779B9 * (i + 1));   x = (u32 >> 8) * 2**-23 - 1   in [-1, 1)

* input, channel c:            seed 0x9E3779B9 + c, white noise (RMS ~ 0.577)
* IR, (instance, channel c):   seed 0x85EBCA6B + 131*inst + c, white noise * exp(-6.9078 i / irLen)
                               (-60 dB at the end, computed in double), then the energy
                               auto-gain of Impulse::calculateAutoGain
                               (reference src/dsp/Impulse.cpp:691-708, applied :314-324):
                               g = min(1, 1/sqrt(sum over L and R of ir^2)).
"""
from __future__ import annotations

import numpy as np

INPUT_SEED = 0x9E3779B9
IR_SEED = 0x85EBCA6B
_M32 = np.uint64(0xFFFFFFFF)


def _fmix32(x: np.ndarray) -> np.ndarray:
    x = x & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & _M32
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & _M32
    x ^= x >> np.uint64(16)
    return x


def white_noise(n: int, seed: int, offset: int = 0) -> np.ndarray:
    """n float32 samples uniform in [-1, 1), starting at stream position `offset`."""
    i = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
    x = _fmix32((np.uint64(seed & 0xFFFFFFFF) + i * np.uint64(0x9E3779B9)) & _M32)
    return ((x >> np.uint64(8)).astype(np.float64) * (2.0 ** -23) - 1.0).astype(np.float32)


def synth_input(n_frames: int, channel: int, offset: int = 0) -> np.ndarray:
    return white_noise(n_frames, INPUT_SEED + channel, offset)


def synth_ir(ir_len: int, n_channels: int = 2, inst: int = 0) -> np.ndarray:
    """(n_channels, ir_len) float32 decaying-noise IR set with the reference's auto-gain."""
    t = np.arange(ir_len, dtype=np.float64)
    env = np.exp(-6.9078 * t / float(ir_len))
    irs = np.stack([
        (white_noise(ir_len, IR_SEED + 131 * inst + c).astype(np.float64) * env).astype(np.float32)
        for c in range(n_channels)
    ])
    # calculateAutoGain uses the LL and RR buffers only (channels 0 and 1).
    e = float(np.sum(irs[:2].astype(np.float64) ** 2))
    g = np.float32(min(1.0 / np.sqrt(e), 1.0)) if e > 0 else np.float32(1.0)
    return (irs * g).astype(np.float32)


def ramp(n: int) -> np.ndarray:
    """The reference test signal 0.1*(i+1) (libs/FFTConvolver/test/Test.cpp:78-87)."""
    return (np.float32(0.1) * np.arange(1, n + 1, dtype=np.float32)).astype(np.float32)
