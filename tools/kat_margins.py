#!/usr/bin/env python3
"""Margin against the reference's own pass rule (Test.cpp:129-145; < 1 passes) of its large known-answer cases, run as channel 0
of a 12-channel set (the lock-step sets' precision policy) under measurement knobs.   python tools/kat_margins.py mix64=0 mix64=1 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reevr_amd
from oracle import oracle_py as O
from reevr_amd import synth
from tests import cases

NCH = 12


class Lane:
    def __init__(self, kind, tune):
        self.kind = kind
        self._set = reevr_amd.ConvolverSet(NCH, tune=tune)

    def init(self, *a):
        ir = np.asarray(a[-1], np.float32)
        irs = [ir * np.float32(1.0 - 0.05 * c) for c in range(NCH)]
        return self._set.init_uniform(a[0], irs) if self.kind == "fftconv" else self._set.init(a[0], a[1], irs)

    def process(self, x):
        x = np.asarray(x, np.float32).reshape(-1)
        return self._set.process(np.stack([x * np.float32(1.0 - 0.03 * c) for c in range(NCH)]))[0]


def main():
    big = [("fftconv", t) for t in cases.KAT_FFTCONV if t[0] >= 100000 and t[4] >= 1024] + \
          [("twostage", t) for t in cases.KAT_TWOSTAGE if t[0] >= 100000 and t[4] >= 1024]
    exact = {(k, t): O.direct_convolve(synth.ramp(t[0]), synth.ramp(t[1])) for k, t in big}
    for spec in sys.argv[1:] or ["mix64=0"]:
        tune = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in spec.split(",")}
        row = []
        for k, t in big:
            out = cases.run_kat(lambda kind: Lane(kind, tune), k, t)
            row.append("%s %.3f" % (cases.kat_name(k, t).replace("100000_", ""), cases.kat_margin(out, exact[(k, t)], t[1])))
        print(spec, "|", " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
