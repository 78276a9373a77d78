"""What would a uniform delay line at block B cost for the bench workload (stereo, 10 s IR, 40 s call)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import reevr_amd
from reevr_amd import synth, KERNEL_NAMES
frames = 1920000
irs = synth.synth_ir(480000, 2, 0)
x = torch.from_numpy(np.stack([synth.synth_input(frames, c) for c in range(2)])).cuda()
y = torch.empty_like(x)
for B in (4096, 8192, 16384):
    s = reevr_amd.ConvolverSet(2)
    assert s.init_uniform(B, list(irs), max_len=frames), s.last_error_string
    for _ in range(3): s.process_device(x, y)
    t0 = time.perf_counter()
    for _ in range(20): s.process_device(x, y, sync=False, order=False)
    s.sync(); dt = (time.perf_counter() - t0) / 20
    s.set_timing(True); s.kernel_time_reset()
    for _ in range(5): s.process_device(x, y)
    res = {}
    for i, n in enumerate(KERNEL_NAMES):
        c, ms = s.kernel_time(i)
        if c: res[n] = round(ms / c * 1e3, 1)
    print(B, s.partitions(0), f"{2*frames/dt/1e9:.2f} Gs/s {dt*1e6:.1f} us", res)
    s.close()
