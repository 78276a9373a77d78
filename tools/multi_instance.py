"""Many plug-in instances on ONE GPU, each on its own host thread with its own handle (the
reference's threading contract): aggregate block-synchronous throughput of N stereo instances,
each calling rvc_set_process_device_blocks (512-frame calls) concurrently."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import reevr_amd
from reevr_amd import synth

nblk = 2000
for ninst in (1, 2, 4, 8, 16):
    sets, xs, ys = [], [], []
    for i in range(ninst):
        irs = synth.synth_ir(480000, 2, i)
        s = reevr_amd.ConvolverSet(2, bg_stream=True)
        assert s.init(512, 8192, list(irs), max_len=512)
        x = torch.from_numpy(np.stack([synth.synth_input(512 * nblk, c + 2 * i) for c in range(2)])).cuda()
        sets.append(s); xs.append(x); ys.append(torch.empty_like(x))
        s.process_device_blocks(x[:, :512 * 100].contiguous(), 512)
    torch.cuda.synchronize()
    def run(i):
        sets[i].process_device_blocks(xs[i], 512, ys[i])
    th = [threading.Thread(target=run, args=(i,)) for i in range(ninst)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(f"{ninst:2d} stereo instances: {ninst * nblk / dt / 1e3:.1f} kblocks/s aggregate, "
          f"{2 * ninst * 512 * nblk / dt / 1e6:.1f} Msamples/s, {dt / nblk * 1e6:.1f} us per block round "
          f"= {ninst * 10666.7 / (dt / nblk * 1e6):.0f} real-time instances' worth at 48 kHz / 512", flush=True)
    for s in sets: s.close()
