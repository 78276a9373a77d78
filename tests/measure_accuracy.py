"""Relative RMS error of the engine against the CPU oracle (the reference restatement) for the
BASELINE geometries, per call pattern and precision mode. For DESIGN.md; the tests assert 1e-5."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reevr_amd
from oracle import oracle_py as O
from reevr_amd import synth

def rel(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2))

for name, head, tail, irlen, frames in (("cfg2 10 s IR, 512/8192", 512, 8192, 480000, 512 * 600),
                                         ("cfg3 30 s IR @96k, 256/8192", 256, 8192, 2880000, 256 * 1536),
                                         ("cfg5 5 s IR, 4096/8192", 4096, 8192, 240000, 4096 * 64)):
    ir = synth.synth_ir(irlen, 1, 0)[0]
    x = synth.synth_input(frames, 0)
    o = O.TwoStageFFTConvolver("ref" if O.have_ref() else "orc")
    assert o.init(head, tail, ir)
    want = np.concatenate([o.process(x[i:i + head]) for i in range(0, frames, head)])
    row = [name]
    for label, kw, big in (("f32 block calls", {}, False), ("f32 one call (adaptive)", {}, True),
                           ("f32 one call (fixed partitions)", {"fixed_partitions": True}, True),
                           ("f64 block calls", {"fft_f64": True}, False), ("f64 one call", {"fft_f64": True}, True)):
        s = reevr_amd.ConvolverSet(1, **kw)
        assert s.init(head, tail, [ir], max_len=frames)
        if big:
            got = s.process(x[None, :])[0]
        else:
            got = np.concatenate([s.process(x[None, i:i + head])[0] for i in range(0, frames, head)])
        row.append(f"{label}: {rel(got, want):.2e}")
        s.close()
    print(" | ".join(row), flush=True)
