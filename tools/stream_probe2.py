"""Block-synchronous streaming with different IR lengths (which part of the one-launch block kernel sets its duration?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import reevr_amd
from reevr_amd import synth
nblk = 3000
x = torch.from_numpy(np.stack([synth.synth_input(512 * nblk, c) for c in range(2)])).cuda()
for ir_len in (400, 1000, 4000, 16000, 480000):
    irs = synth.synth_ir(ir_len, 2, 0)
    s = reevr_amd.ConvolverSet(2, bg_stream=False, timing=True)
    assert s.init(512, 8192, list(irs), max_len=512)
    s.process_device_blocks(x[:, :512 * 200].contiguous(), 512)
    s.kernel_time_reset()
    s.process_device_blocks(x[:, :512 * 500].contiguous(), 512)
    kt = {reevr_amd.KERNEL_NAMES[i]: s.kernel_time(i) for i in range(9)}
    s.set_timing(False)
    t = time.perf_counter(); s.process_device_blocks(x, 512); dt = time.perf_counter() - t
    print(f"ir {ir_len:7d}: P_A={s.partitions(0):3d} P_T={s.partitions(1):3d}  {dt / nblk * 1e6:6.2f} us/block   "
          + "  ".join(f"{k} {v[1] / v[0] * 1e3:.2f}us x{v[0]}" for k, v in kt.items() if v[0]))
    s.close()
