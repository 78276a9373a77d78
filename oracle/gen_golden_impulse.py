#!/usr/bin/env python3
"""Generate tests/golden/impulse.npz: outputs of the impulse-preparation restatement
(oracle/impulse_oracle.c) with the REFERENCE's AudioFFT (oracle/_ref) plugged into its STFT stage.

Run only in the build container:  make -C oracle ref && python oracle/gen_golden_impulse.py
The reference's Impulse.cpp itself cannot be compiled (JUCE headers absent), so these vectors pin
the transform and freeze the restatement; the stage logic stays "parity unpinned"
(impulse_oracle.h). TEST INFRASTRUCTURE ONLY.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle_py as O  # noqa: E402
from tests import impulse_cases as IC  # noqa: E402


def main():
    O.build(ref=True)
    assert O.have_ref()
    out = {}
    for name, case in IC.CASES.items():
        n, nc, seed, kw, mag, rate = IC.params_of(case)
        lut = None if mag is None else O.impulse_decay_lut(mag, kw["srate"], rate)
        r = O.impulse_recalc(IC.raw_channels(n, nc, seed), decay_lut=lut, fft="ref", **kw)
        for c, b in enumerate(r["buffers"]):
            out[f"{name}/ch{c}"] = b
        out[f"{name}/meta"] = np.array([r["peak"], r["trim_left_samples"], r["trim_right_samples"]], np.float64)
        if lut is not None:
            out[f"{name}/lut"] = lut
    path = os.path.join(ROOT, "tests", "golden", "impulse.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB", len(IC.CASES), "cases")


if __name__ == "__main__":
    main()
