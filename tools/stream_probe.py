"""Block-synchronous streaming probe (for rocprofv3 timelines): stereo, 10 s IR, 512-frame calls."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import reevr_amd
from reevr_amd import synth
bg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
irs = synth.synth_ir(480000, 2, 0)
x = torch.from_numpy(np.stack([synth.synth_input(512 * nblk, c) for c in range(2)])).cuda()
s = reevr_amd.ConvolverSet(2, bg_stream=bool(bg))
assert s.init(512, 8192, list(irs), max_len=512)
s.process_device_blocks(x[:, :512 * 200].contiguous(), 512)
t = time.perf_counter(); s.process_device_blocks(x, 512); dt = time.perf_counter() - t
print(f"bg={bg} {dt / nblk * 1e6:.2f} us per block, {2 * 512 * nblk / dt / 1e6:.1f} Msamples/s")
