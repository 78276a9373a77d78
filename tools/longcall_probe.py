"""Offline long-call rate of ONE stereo pair (bench side number `stereo_offline_long_call`) on its own, optionally after a
persistent set has lived in the process:  python tools/longcall_probe.py [persistent_first]"""
import sys
import time

import numpy as np
import torch

import reevr_amd
from reevr_amd import synth

SR = 48000
dev = torch.device("cuda:0")
irs2 = list(synth.synth_ir(10 * SR, 2, 0))
frames = 40 * SR
if len(sys.argv) > 1 and int(sys.argv[1]):
    p = reevr_amd.ConvolverSet(2, persistent=True, bg_stream=True)
    assert p.init(512, 8192, irs2, max_len=512)
    xb = torch.zeros(2, 512 * 50, device=dev)
    p.process_device_blocks(xb, 512)
    p.close()
xl = torch.from_numpy(np.stack([synth.synth_input(frames, c) for c in range(2)])).to(dev)
yl = torch.empty_like(xl)
for fixed in (False, True):
    s = reevr_amd.ConvolverSet(2, fixed_partitions=fixed, timing=False)
    assert s.init(512, 8192, irs2, max_len=frames)
    for _ in range(20):
        s.process_device(xl, yl, sync=False, order=False)
    s.sync()
    reps = 200
    ts = time.perf_counter()
    for _ in range(reps):
        s.process_device(xl, yl, sync=False, order=False)
    s.sync()
    te = time.perf_counter() - ts
    print("fixed" if fixed else "adaptive", round(2 * frames * reps / te / 1e6, 1), "Msamples/s", round(te / reps * 1e6, 1), "us per call")
    s.close()
    st = reevr_amd.ConvolverSet(2, fixed_partitions=fixed, timing=True)
    assert st.init(512, 8192, irs2, max_len=frames)
    for _ in range(30):
        st.process_device(xl, yl, sync=False, order=False)
    st.sync()
    print("   ", {reevr_amd.KERNEL_NAMES[i]: round(st.kernel_time(i)[1] / max(st.kernel_time(i)[0], 1) * 1e3, 1)
                  for i in range(len(reevr_amd.KERNEL_NAMES)) if st.kernel_time(i)[0]})
    st.close()
