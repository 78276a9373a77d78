"""Diagnostic: margin of each implementation against the reference's own pass rule
(Test.cpp:129-145) on the large known-answer cases. factor < 1 passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle_py as O
from reevr_amd import synth
from tests import cases
import reevr_amd

def factor(out, exact, ir_len):
    a = out.astype(np.float64); b = exact
    m = (np.abs(a) > 1) & (np.abs(b) > 1)
    ae = np.abs(a - b)[m]; re = ae / b[m]
    f = np.minimum(ae / (1e-3 * ir_len), re / (1e-4 * np.log(ir_len)))
    i = int(np.argmax(f))
    return float(f.max()), int(np.flatnonzero(m)[i])

facs = {"gpu": lambda k: reevr_amd.FFTConvolver() if k == "fftconv" else reevr_amd.TwoStageFFTConvolver(),
        "gpu64": lambda k: reevr_amd.FFTConvolver(fft_f64=True) if k == "fftconv" else reevr_amd.TwoStageFFTConvolver(fft_f64=True),
        "orc": lambda k: O.FFTConvolver("orc") if k == "fftconv" else O.TwoStageFFTConvolver("orc")}
if O.have_ref():
    facs["ref"] = lambda k: O.FFTConvolver("ref") if k == "fftconv" else O.TwoStageFFTConvolver("ref")
for kind, tups in (("fftconv", cases.KAT_FFTCONV), ("twostage", cases.KAT_TWOSTAGE)):
    for tup in tups:
        if tup[0] < 100000 or tup[4] < 1024: continue
        exact = O.direct_convolve(synth.ramp(tup[0]), synth.ramp(tup[1]))
        row = []
        for name, f in facs.items():
            out = cases.run_kat(f, kind, tup)
            fa, idx = factor(out, exact, tup[1])
            row.append(f"{name} {fa:.3f}@{idx}")
        print(cases.kat_name(kind, tup), " | ".join(row), flush=True)
