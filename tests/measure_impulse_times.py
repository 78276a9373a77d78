"""IR hot-swap cost, device path vs CPU restatement: recalcImpulse (all stages incl. decay EQ) and
loadImpulse for the BASELINE impulse lengths. Run on the GPU box:  python tests/measure_impulse_times.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import reevr_amd  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from tests import impulse_cases as IC  # noqa: E402


def med(f, n=7):
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t)
    return float(np.median(ts)) * 1e3


for name, (n, nc, block, srate) in {"cfg2 10 s stereo": (480000, 2, 512, 48000.0),
                                    "cfg2 10 s quad": (480000, 4, 512, 48000.0),
                                    "cfg3 30 s @96k stereo": (2880000, 2, 256, 96000.0)}.items():
    raw = IC.raw_channels(n, nc, 50)
    imp = reevr_amd.Impulse()
    imp.prepare(srate)
    imp.attack, imp.decay, imp.decayMagnitude = 0.01, 0.8, IC.MAGS["tilt"]
    t_raw = med(lambda: imp.setRaw(*raw), 3)
    imp.recalcImpulse()
    t_recalc = med(imp.recalcImpulse)
    sc = reevr_amd.StereoConvolver()
    sc.prepare(block)
    sc.loadImpulse(imp)
    t_load_dev = med(lambda: sc.loadImpulse(imp))

    class Host:
        pass
    h = Host()
    h.isQuad = nc == 4
    h.bufferLL, h.bufferRR, h.bufferLR, h.bufferRL = imp.bufferLL, imp.bufferRR, imp.bufferLR, imp.bufferRL
    sc.loadImpulse(h)
    t_load_host = med(lambda: sc.loadImpulse(h))
    lut = O.impulse_decay_lut(IC.MAGS["tilt"], srate, 1.0)
    t_cpu = med(lambda: O.impulse_recalc(raw, attack=0.01, decay=0.8, srate=srate, decay_lut=lut), 3)
    ref = O.TwoStageFFTConvolver("ref" if O.have_ref() else "orc")
    t_cpu_init = med(lambda: ref.init(block, 8192, h.bufferLL), 3) * nc
    print(f"{name}: upload raw {t_raw:.2f} ms | recalcImpulse device {t_recalc:.2f} ms (CPU restatement {t_cpu:.0f} ms) | "
          f"loadImpulse device-resident {t_load_dev:.2f} ms, from host buffers {t_load_host:.2f} ms "
          f"(CPU init x{nc}: {t_cpu_init:.0f} ms)", flush=True)
