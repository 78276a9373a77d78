"""Where does a guard-mode run turn non-finite? (tests/test_gpu_parity.py::test_guard_bands_stay_intact_and_outputs_finite's
generator, every tiling mode.)   python tools/guard_diag.py seed [seed ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reevr_amd
from reevr_amd import synth


def case(seed):
    rng = np.random.RandomState(9100 + seed)
    head = int(rng.choice([64, 128, 256, 512]))
    tail = int(rng.choice([2 * head, 4 * head, 16 * head]))
    nch = int(rng.randint(1, 4))
    parts = 3 if seed == 226 else int(rng.choice([1, 3, 9, 20]))
    base = 2 * tail + parts * tail - int(rng.randint(0, tail // 2))
    irs = [synth.synth_ir(max(1, base - c * int(rng.randint(0, tail))), 1, 800 + 3 * seed + c)[0] for c in range(nch)]
    total = int(min(max(40 * tail, 30 * 8 * head), 200000))
    total -= total % head
    sched, done = [], 0
    while done < total:
        r = rng.randint(0, 30)
        n = (int(rng.randint(1, head)) if r == 0 else (head - done % head) if (r == 1 and done % head) else
             int(rng.randint(2, 6)) * head if r == 2 else int(rng.randint(5, 9)) * tail if r == 3 else
             (head if done % head == 0 else head - done % head))
        n = max(1, min(n, total - done))
        sched.append(n)
        done += n
    x = np.stack([synth.synth_input(total, 13 * seed + c) for c in range(nch)])
    return head, tail, nch, irs, sched, x


for seed in [int(a) for a in sys.argv[1:]]:
    head, tail, nch, irs, sched, x = case(seed)
    for tiling in (False, True, "force", "force2"):
        for bg in (False, True):
            reevr_amd.set_tuning("guard", 1)
            s = reevr_amd.ConvolverSet(nch, bg_stream=bg, time_tiling=tiling)
            assert s.init(head, tail, irs, max_len=max(sched)), s.last_error_string
            reevr_amd.set_tuning("guard", 0)
            pos, bad = 0, None
            for i, n in enumerate(sched):
                y = s.process(x[:, pos:pos + n])
                if bad is None and not np.isfinite(y).all():
                    c, j = np.argwhere(~np.isfinite(y))[0]
                    bad = (i, n, pos, int(c), int(j), int((~np.isfinite(y)).sum()))
                pos += n
            g = s.guard_check()
            if bad or g:
                i = bad[0] if bad else 0
                print(f"seed {seed} head {head} tail {tail} nch {nch} ir {[len(v) for v in irs]} P {s.partitions(0)}/{s.partitions(1)} "
                      f"tiles {s.tile_rows(0)}/{s.tile_rows(1)} tiling {tiling} bg {bg}: first bad call #{bad[0] if bad else None} len {bad[1] if bad else None} "
                      f"at sample {bad[2] if bad else None} (block {bad[2] // head if bad else None}, tail block {bad[2] // tail if bad else None}) ch {bad[3] if bad else None} "
                      f"offset {bad[4] if bad else None} count {bad[5] if bad else None}; guard {g}; calls before: {sched[max(0, i - 6):i + 1]}", flush=True)
            s.close()
    print(f"seed {seed} done", flush=True)
