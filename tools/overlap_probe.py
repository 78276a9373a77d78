"""How much could overlapping independent long calls gain? N sets (own streams), each processing the bench
workload back to back from its own host thread; aggregate rate vs one set. An upper bound for any
cross-call pipelining inside one set (which has dependencies on top)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import reevr_amd
from reevr_amd import synth

frames = 1920000
irs = synth.synth_ir(480000, 2)
x = torch.from_numpy(np.stack([synth.synth_input(frames, c) for c in range(2)])).cuda()
for nsets in (1, 2, 3, 4):
    sets, outs = [], []
    for _ in range(nsets):
        s = reevr_amd.ConvolverSet(2)
        assert s.init(512, 8192, list(irs), max_len=frames)
        sets.append(s); outs.append(torch.empty_like(x))
    torch.cuda.synchronize()
    reps = 300

    def work(i):
        s, y = sets[i], outs[i]
        for _ in range(reps):
            s.process_device(x, y, sync=False, order=False)
        s.sync()
    for i in range(nsets):
        work(i)                      # warm-up (also clocks)
    t = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(nsets)]
    [a.start() for a in th]; [a.join() for a in th]
    dt = time.perf_counter() - t
    print(f"{nsets} independent sets: {nsets * reps * 2 * frames / dt / 1e9:.1f} Gsamples/s aggregate, "
          f"{dt / reps * 1e6:.1f} us per round of {nsets} calls")
    for s in sets:
        s.close()
