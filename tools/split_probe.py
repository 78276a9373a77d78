#!/usr/bin/env python3
"""Does splitting the lock-step channels into independent groups on their own streams pay? N sets of C/N channels each,
driven from one host thread (every set's 256-block step is enqueued in turn; their launches overlap on the GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import reevr_amd
from reevr_amd import synth

SR, IR_LEN, BLK, TAIL = 48000, 480000, 512, 8192
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
base = [synth.synth_ir(IR_LEN, 2, inst=i) for i in range(4)]
irs = [base[(c // 2) % 4][c % 2] for c in range(C)]
nblk = 256
g = torch.Generator(device="cuda").manual_seed(1)
x = (torch.rand((C, nblk * BLK), device="cuda", generator=g) * 2 - 1)
y = torch.empty_like(x)
for groups in (1, 2, 4):
    per = C // groups
    sets = []
    for gi in range(groups):
        s = reevr_amd.ConvolverSet(per)
        assert s.init(BLK, TAIL, irs[gi * per:(gi + 1) * per], max_len=BLK)
        sets.append(s)
    def step():
        for gi, s in enumerate(sets):
            s.process_device_blocks(x[gi * per:(gi + 1) * per], BLK, y[gi * per:(gi + 1) * per], sync=False, order=False)
    for _ in range(5):
        step()
    for s in sets:
        s.sync()
    t0 = time.perf_counter()
    K = 10
    for _ in range(K):
        step()
    for s in sets:
        s.sync()
    dt = (time.perf_counter() - t0) / K
    print(f"{groups} group(s) of {per} channels: {dt * 1e3:.3f} ms per 256-block step, {C * nblk * BLK / dt / 1e6:.1f} Msamples/s", flush=True)
    for s in sets:
        s.close()
    torch.cuda.empty_cache()
