import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reevr_amd
from reevr_amd import synth
sys.argv = sys.argv[:1]
exec(open(os.path.join(os.path.dirname(__file__), "guard_diag.py")).read().split("for seed in")[0])

def runs(bad):
    idx = np.flatnonzero(bad)
    if idx.size == 0:
        return []
    cuts = np.flatnonzero(np.diff(idx) > 1)
    starts = np.r_[idx[0], idx[cuts + 1]]
    ends = np.r_[idx[cuts], idx[-1]]
    return [(int(a), int(b - a + 1)) for a, b in zip(starts, ends)][:6]

for seed in (226, 17):
    head, tail, nch, irs, sched, x = case(seed)
    nb = 40
    variants = {"all": list(range(nch)), "ch1 alone": [1], "ch0,ch1": [0, 1], "ch1,ch0": [1, 0], "ch2,ch1,ch0": [2, 1, 0]}
    for name, sel in variants.items():
        for eq in (False, True):
            ii = [irs[c] for c in sel]
            if eq:
                m = min(len(v) for v in ii)
                ii = [v[:m] for v in ii]
            xx = x[sel]
            reevr_amd.set_tuning("guard", 1)
            s = reevr_amd.ConvolverSet(len(sel), time_tiling=False, fft_f32=True)
            assert s.init(head, tail, ii, max_len=head)
            reevr_amd.set_tuning("guard", 0)
            y = np.concatenate([s.process(xx[:, i * head:(i + 1) * head]) for i in range(nb)], axis=1)
            print(f"seed {seed} head {head} tail {tail} {name} equal_len {eq} irs {[len(v) for v in ii]} P {s.partitions(0)}/{s.partitions(1)}: nan runs per channel "
                  f"{[runs(~np.isfinite(y[c])) for c in range(len(sel))]}", flush=True)
            s.close()
