#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the UNTOUCHED reference (oracle/_ref).

Run only in the build container, where /root/reference exists:
    make -C oracle ref && python oracle/gen_golden.py
The fixtures are data (inputs are re-derivable from tests/cases.py + reevr_amd/synth.py;
outputs are decimated samples, first/last 512 samples, RMS and sum of the reference's
output) -- no reference source text is stored. TEST INFRASTRUCTURE ONLY.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle_py as O  # noqa: E402
from tests import cases  # noqa: E402


def factory(kind):
    return O.FFTConvolver("ref") if kind == "fftconv" else O.TwoStageFFTConvolver("ref")


CMAC_LENS = (1, 3, 4, 7, 64, 513, 1027)


def cmac_inputs(n):
    """the six operand arrays of one ComplexMultiplyAccumulate known-answer case (re, im, reA, imA, reB, imB)"""
    from reevr_amd import synth
    return [synth.white_noise(n, 0xC0AC + 16 * n + j) for j in range(6)]


def gen_cmac(gold):
    """ComplexMultiplyAccumulate (Utilities.cpp:62-111) known answers: the accumulators after ONE call on seeded noise"""
    out = {}
    for n in CMAC_LENS:
        a = cmac_inputs(n)
        re, im = a[0].copy(), a[1].copy()
        O.cmac(re, im, *a[2:], which="ref")
        out[f"n{n}/re"] = re
        out[f"n{n}/im"] = im
    np.savez_compressed(os.path.join(gold, "cmac.npz"), **out)
    print("cmac.npz:", len(CMAC_LENS), "cases")


def main():
    O.build(ref=True)
    assert O.have_ref(), "oracle/_ref missing: reference sources not available here"
    gold = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    gen_cmac(gold)
    if sys.argv[1:] == ["cmac"]:        # (only this fixture: the others take minutes)
        return

    kat = {}
    for kind, tups in (("fftconv", cases.KAT_FFTCONV), ("twostage", cases.KAT_TWOSTAGE)):
        for tup in tups:
            out = cases.run_kat(factory, kind, tup)
            for k, v in cases.summarize(out).items():
                kat[cases.kat_name(kind, tup) + "/" + k] = v
    np.savez_compressed(os.path.join(gold, "kat.npz"), **kat)
    print("kat.npz:", len(kat) // 6, "cases")

    syn = {}
    for name, case in cases.SYNTH_CASES.items():
        t = time.time()
        out = cases.run_synth_case(factory, case)
        for c in range(out.shape[0]):
            for k, v in cases.summarize(out[c]).items():
                syn[f"{name}/ch{c}/{k}"] = v
        print(f"{name}: {out.shape} rms={np.sqrt(np.mean(out.astype(np.float64)**2)):.4f} ({time.time()-t:.1f}s)")
    np.savez_compressed(os.path.join(gold, "synth.npz"), **syn)

    # AudioFFT known-answer vectors: forward spectrum of seeded noise at a few sizes.
    fft = {}
    from reevr_amd import synth
    for n in (2, 4, 8, 16, 64, 1024, 16384, 32768):       # (32768: round 6, the 16384-bin transforms of a widened tail)
        x = synth.white_noise(n, 0xF00D + n)
        re, im = O.rfft(x, "ref")
        fft[f"n{n}/re"] = re
        fft[f"n{n}/im"] = im
        fft[f"n{n}/rt"] = O.irfft(re, im, "ref")
    np.savez_compressed(os.path.join(gold, "audiofft.npz"), **fft)
    for f in ("kat.npz", "synth.npz", "audiofft.npz"):
        print(f, os.path.getsize(os.path.join(gold, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
