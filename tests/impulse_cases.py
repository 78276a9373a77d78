"""Impulse-preparation parity cases: raw IR channels + the parameters of
reference src/dsp/Impulse.h:58-72. Shared by the oracle tests, the GPU parity tests and
oracle/gen_golden_impulse.py."""
from __future__ import annotations

import numpy as np

from reevr_amd import synth

LUT_SIZE = 2049


def raw_channels(n: int, nc: int, seed: int = 0) -> list:
    """Decoded IR channels as Impulse::load leaves them (no auto gain yet): decaying noise, peak ~0.5."""
    t = np.arange(n, dtype=np.float64)
    env = np.exp(-6.9078 * t / max(n, 1))
    return [(0.5 * synth.white_noise(n, 0xA5A50000 + 977 * seed + c).astype(np.float64) * env).astype(np.float32)
            for c in range(nc)]


def tilt_magnitude(db_dc: float = 6.0, db_nyq: float = -30.0) -> np.ndarray:
    """Combined decay-EQ magnitude per bin: a straight tilt in dB (what a high shelf cut looks like)."""
    db = np.linspace(db_dc, db_nyq, LUT_SIZE)
    return (10.0 ** (db / 20.0)).astype(np.float32)


# name -> (n, channels, seed, params); decay_mag is a key into MAGS or None
MAGS = {"tilt": tilt_magnitude(), "boost": tilt_magnitude(-12.0, 12.0), "flat": np.ones(LUT_SIZE, np.float32)}

CASES = {
    "plain2":      (9000, 2, 1, dict()),
    "env2":        (9000, 2, 2, dict(attack=0.1, decay=0.6, gain=0.7)),
    "trim_rev4":   (9000, 4, 3, dict(reverse=True, trim_left=0.1, trim_right=0.25, gain=1.5, attack=0.05, decay=0.3)),
    "decay2":      (9000, 2, 4, dict(decay_mag="tilt", srate=48000.0)),
    "decay4_all":  (10001, 4, 5, dict(reverse=True, trim_left=0.03, trim_right=0.11, gain=2.0, attack=0.02, decay=0.5,
                                      decay_mag="tilt", decay_rate=1.5, srate=44100.0)),
    "boost_clip2": (12000, 2, 6, dict(gain=40.0, decay_mag="boost", srate=96000.0)),
    "short2":      (700, 2, 7, dict(decay_mag="tilt", attack=0.5, decay=0.5)),
    "one_frame2":  (4096, 2, 8, dict(decay_mag="tilt")),
    "hop_edge2":   (5121, 2, 9, dict(decay_mag="flat", trim_right=0.2)),
    "all_trimmed": (1000, 2, 10, dict(trim_left=0.6, trim_right=0.5)),
    "tiny":        (1, 2, 11, dict(decay_mag="tilt")),
}

BIG_CASES = {   # GPU vs oracle only (no fixture): the BASELINE impulse lengths
    "cfg2_10s":  (480000, 2, 20, dict(attack=0.01, decay=0.8, decay_mag="tilt", srate=48000.0)),
    "cfg3_30s4": (2880000, 4, 21, dict(reverse=True, trim_right=0.05, decay=0.9, decay_mag="tilt", srate=96000.0)),
}


def params_of(case):
    n, nc, seed, kw = case
    kw = dict(kw)
    mag = kw.pop("decay_mag", None)
    rate = kw.pop("decay_rate", 1.0)
    out = dict(reverse=False, trim_left=0.0, trim_right=0.0, gain=1.0, attack=0.0, decay=0.0, srate=48000.0)
    out.update(kw)
    return n, nc, seed, out, (None if mag is None else MAGS[mag]), rate
