"""Out-of-bounds net, fence mode (rvc_debug_set_tuning("guard", 2)): every device allocation of a set ends exactly at the
end of its mapping, with unmapped address space behind (and in front of) it -- an out-of-bounds access of a kernel, also
a read whose value is dropped afterwards, is a GPU memory fault and the process dies. Runs the block-synchronous fuzz
geometries (all tiling modes, both stream modes) and a lock-step set at the bench's geometry; prints one line per case.
NOT part of the test-suite: a failure here is an abort.   python tools/fence_fuzz.py [n_seeds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reevr_amd
from reevr_amd import synth
sys.argv, nseeds = sys.argv[:1], int(sys.argv[1]) if len(sys.argv) > 1 else 12


def case(seed):
    """the generator of tests/test_gpu_parity.py::test_guard_bands_stay_intact_and_outputs_finite"""
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


for seed in [226] + list(range(nseeds)):
    head, tail, nch, irs, sched, x = case(seed)
    # force2_k32: first-level tiles of 32 blocks = the LDS-fed sweeps; widen / shrink: the delay-1 tail stage of many-channel sets
    # round 6: phases = the tail tiles in channel groups out of phase (launches on channel sub-ranges); spread3 = sweeps in channel slices
    # third*: third-level sweeps in both stages (the four-row sweep form, rows written in place); f64*: every transform in double = the general
    # per-block path with its time-tiled zero-latency stage
    for tiling in (False, True, "force", "force2", "force2_k32", "widen", "shrink", "phases", "phases_shrink", "spread3", "phases_spread",
                   "third", "third_phases_shrink", "f64", "f64_third"):
        slack = {"widen": 1, "shrink": 2, "phases_shrink": 2, "phases_spread": 2, "third_phases_shrink": 2}.get(tiling, -1)
        extra = (dict(tail_phases=8, tail_spread=1) if tiling == "phases_spread" else dict(tail_phases=8) if "phases" in str(tiling)
                 else (dict(tail_spread=3) if tiling == "spread3" else {}))
        if "third" in str(tiling):
            extra.update(tail_third=1, head_third=1)
        f64 = str(tiling).startswith("f64")
        with reevr_amd.tuning(guard=2, k1=32 if tiling in ("force2_k32", "third", "f64_third") else 0, tail_slack=slack, **extra):
            s = reevr_amd.ConvolverSet(nch, bg_stream=bool(seed & 1) and slack < 0 and not extra, fft_f32=slack > 0, fft_f64=f64,
                                       time_tiling={"force2_k32": "force2", "widen": "force", "shrink": "force2", "phases": "force2",
                                                    "phases_shrink": "force2", "spread3": "force2", "phases_spread": "force2",
                                                    "third": "force2", "third_phases_shrink": "force2", "f64": "force", "f64_third": "force2"}.get(tiling, tiling))
            ok = s.init(head, tail, irs, max_len=max(sched))
        assert ok, s.last_error_string
        pos = 0
        fin = True
        for n in sched:
            fin = fin and bool(np.isfinite(s.process(x[:, pos:pos + n])).all())
            pos += n
        assert s.last_error == 0, s.last_error_string
        print(f"fence ok: seed {seed} head {head} tail {tail} nch {nch} tiling {tiling} finite {fin}", flush=True)
        s.close()
import torch
for nch, head, tail, ir_len, nblk in ((64, 512, 8192, 480000, 16 * 20), (64, 256, 8192, 700000, 32 * 40), (40, 4096, 8192, 100000, 24)):
    irs = [synth.synth_ir(ir_len - 997 * (c % 5), 1, 600 + c)[0] for c in range(nch)]
    xx = torch.from_numpy(np.stack([synth.synth_input(head * nblk, 20 + c % 7) for c in range(nch)])).cuda()
    # kids: two child sets (forced), default tiling; phases / kids_phases: 8 phase groups (of 8 / 4 channels) with the shrunk tail
    for tiling in (True, "force", "force2", "force2_k32", "kids", "widen", "shrink", "phases", "kids_phases", "spread3"):
        extra = dict(tail_phases=8) if "phases" in str(tiling) else (dict(tail_spread=3) if tiling == "spread3" else {})
        with reevr_amd.tuning(guard=2, k1=32 if tiling == "force2_k32" else 0, subsets=2 if str(tiling).startswith("kids") else -1,
                              tail_slack={"widen": 1, "shrink": 2, "phases": 2, "kids_phases": 2, "spread3": 2}.get(tiling, -1), **extra):
            s = reevr_amd.ConvolverSet(nch, time_tiling={"force2_k32": "force2", "kids": True, "widen": True, "shrink": True, "phases": True,
                                                         "kids_phases": True, "spread3": True}.get(tiling, tiling))
            ok = s.init(head, tail, irs, max_len=head)
        assert ok, s.last_error_string
        y = s.process_device_blocks(xx, head)
        assert s._lib.rvc_debug_fence_probe(s._h) == 1, "the range behind an allocation is readable: no fence"
        print(f"fence ok: lock-step {nch} x head {head} tail {tail}->{s.tail_block} P {s.partitions(0)}+{s.partitions(1)} ir {ir_len} tiling {tiling} tiles {s.tile_rows(0)}/{s.tile_rows(1)} "
              f"finite {bool(torch.isfinite(y).all())}", flush=True)
        s.close()
print("fence fuzz done")
