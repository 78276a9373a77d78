import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reevr_amd
from reevr_amd import synth
from oracle import oracle_py as O
sys.argv = sys.argv[:1]
exec(open(os.path.join(os.path.dirname(__file__), "guard_diag.py")).read().split("for seed in")[0])
for seed in (226, 17):
    head, tail, nch, irs, sched, x = case(seed)
    sched = [head] * 40
    for guard in (0, 1):
        for kw in (dict(), dict(fft_f32=True), dict(fft_f64=True)):
            reevr_amd.set_tuning("guard", guard)
            s = reevr_amd.ConvolverSet(nch, time_tiling=False, **kw)
            assert s.init(head, tail, irs, max_len=head)
            reevr_amd.set_tuning("guard", 0)
            y = np.concatenate([s.process(x[:, i * head:(i + 1) * head]) for i in range(40)], axis=1)
            errs = []
            for c in range(nch):
                o = O.TwoStageFFTConvolver("orc"); o.init(head, tail, irs[c])
                w = o.process(x[c, :40 * head])
                bad = ~np.isfinite(y[c])
                d = np.where(bad, 0, y[c] - w)
                errs.append((int(bad.sum()), int(np.argmax(bad)) if bad.any() else -1, float(np.sqrt(np.mean(d ** 2)) / np.sqrt(np.mean(w ** 2)))))
            print(f"seed {seed} head {head} tail {tail} guard {guard} {kw}: per channel (nan count, first nan, rel err of the rest) {errs}", flush=True)
            s.close()
