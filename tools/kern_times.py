import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import reevr_amd
from reevr_amd import synth, KERNEL_NAMES
nch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1920000
irs = np.concatenate([synth.synth_ir(480000, 2, inst=i) for i in range((nch + 1) // 2)])[:nch]
x = torch.from_numpy(np.stack([synth.synth_input(frames, c % 8) for c in range(nch)])).cuda()
y = torch.empty_like(x)
s = reevr_amd.ConvolverSet(nch)
assert s.init(512, 8192, list(irs), max_len=frames)
for _ in range(2): s.process_device(x, y)
t0 = time.perf_counter()
for _ in range(5): s.process_device(x, y, sync=False, order=False)
s.sync(); dt = (time.perf_counter() - t0) / 5
s.set_timing(True); s.kernel_time_reset()
for _ in range(5): s.process_device(x, y)
res = {}
for i, n in enumerate(KERNEL_NAMES):
    c, ms = s.kernel_time(i)
    if c: res[n] = round(ms / c * 1e3, 1)
print(nch, frames, f"{nch*frames/dt/1e9:.2f} Gs/s", f"{dt*1e6:.0f} us", res)
